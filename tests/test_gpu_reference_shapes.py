"""The reference's own hot-path tests, re-stated AT THEIR OWN SHAPES: same constructor arguments, same input
shapes, the same assertions -- plus values against the oracle on a sampled slice (the reference's tests assert shapes only).

  /root/reference/tests/model/nerf/test_nerf_mlp.py:6-45    NeRFMLP(8, 256, 4) on two [20000, 32, 99] tensors (99 = 2*3*16 + 3 for BOTH
                                                            inputs: the widths are Keras Dense's to discover at the first call)
  /root/reference/tests/model/nerf/test_nerf_utils.py:8-124 NeRFUtils(2, 128, 128, 1024, 10, 4, white_background=True) on
                                                            [2, 128, 128, 32, .] tensors, n_coarse 32, n_fine 64

tf.random.uniform -> numpy's generator (TF's streams cannot be reproduced); tensors are torch CUDA tensors instead of tf ones, so
`.shape` is a torch.Size (compares equal to the reference's tuples).  The smaller / friendlier restatements of rounds 1-4 stay in
tests/test_gpu_api.py."""
import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu


def test_nerf_mlp_at_the_reference_tests_shapes():
    """test_nerf_mlp.py:6-45, line for line"""
    from keras_nerf_amd.model.nerf.mlp import NeRFMLP
    nerf_mlp = NeRFMLP(n_layers=8, dense_units=256, skip_layer=4)
    POS_ENCODE_DIMS, SAMPLE_POINTS = 16, 32
    FINAL_POS_ENCODE_DIMS = 2 * 3 * POS_ENCODE_DIMS + 3
    assert FINAL_POS_ENCODE_DIMS == 99
    assert not nerf_mlp.built and nerf_mlp.get_weights() == []                       # Keras: no variables before the first call
    gen = torch.Generator(device="cuda").manual_seed(0)
    ray_coordinate_inputs = torch.rand((2, 100, 100, SAMPLE_POINTS, FINAL_POS_ENCODE_DIMS), device="cuda", generator=gen)
    direction_inputs = torch.rand((2, 100, 100, SAMPLE_POINTS, FINAL_POS_ENCODE_DIMS), device="cuda", generator=gen)
    ray_coordinate_inputs = ray_coordinate_inputs.reshape(-1, SAMPLE_POINTS, FINAL_POS_ENCODE_DIMS)
    direction_inputs = direction_inputs.reshape(-1, SAMPLE_POINTS, FINAL_POS_ENCODE_DIMS)

    rgb_out, sigma_out = nerf_mlp((ray_coordinate_inputs, direction_inputs))

    model_params = nerf_mlp.get_config()
    assert model_params['n_layers'] == 8
    assert model_params['dense_units'] == 256
    assert model_params['skip_layer'] == 4

    assert rgb_out.shape == (2 * 100 * 100, SAMPLE_POINTS, 3)
    assert sigma_out.shape == (2 * 100 * 100, SAMPLE_POINTS, 1)

    rgb_out = rgb_out.reshape(2, 100, 100, SAMPLE_POINTS, 3)
    sigma_out = sigma_out.reshape(2, 100, 100, SAMPLE_POINTS, 1)
    assert rgb_out.shape == (2, 100, 100, SAMPLE_POINTS, 3)
    assert sigma_out.shape == (2, 100, 100, SAMPLE_POINTS, 1)

    assert torch.cat(nerf_mlp((ray_coordinate_inputs, direction_inputs)), dim=-1).shape == (2 * 100 * 100, SAMPLE_POINTS, 4)

    # --- beyond the reference's assertions: what was built, and the values
    assert nerf_mlp.built and (nerf_mlp.xyz_dim, nerf_mlp.dir_dim) == (99, 99)
    ws = nerf_mlp.get_weights()
    assert len(ws) == 24 and ws[0].shape == (99, 256) and ws[10].shape == (256 + 99, 256) and ws[20].shape == (256 + 99, 128)
    assert nerf_mlp.count_params() == sum(w.size for w in ws)
    rows = np.sort(np.random.default_rng(1).choice(2 * 100 * 100 * SAMPLE_POINTS, 256, replace=False))      # a slice from all over the batch
    x = ray_coordinate_inputs.reshape(-1, 99)[rows].cpu().numpy(); dd = direction_inputs.reshape(-1, 99)[rows].cpu().numpy()
    cfg = O.NerfConfig(pos_emb_xyz=16, pos_emb_dir=16)
    er, es = O.mlp_forward(ws, x, dd, cfg, emulate_bf16=O.FUSED)
    gr = rgb_out.reshape(-1, 3)[rows].cpu().numpy(); gs = sigma_out.reshape(-1, 1)[rows].cpu().numpy()
    np.testing.assert_allclose(gr, er, atol=2e-3); np.testing.assert_allclose(gs, es, atol=4e-3)
    er32, es32 = O.mlp_forward(ws, x, dd, cfg)
    np.testing.assert_allclose(gr, er32, atol=2e-2)                                  # against the reference's fp32 arithmetic
    assert gr.std() > 1e-3 and float(rgb_out.min()) >= 0 and float(rgb_out.max()) <= 1 and float(sigma_out.min()) >= 0
    # a built Dense refuses another width, naming both (Keras' input-compatibility error) -- never a silent reshape
    with pytest.raises(ValueError, match=r"99.*63|63.*99"):
        nerf_mlp((torch.rand(7, 63, device="cuda"), torch.rand(7, 99, device="cuda")))
    with pytest.raises(ValueError, match="27"):
        nerf_mlp((torch.rand(7, 99, device="cuda"), torch.rand(7, 27, device="cuda")))
    # 7 * 99 elements would re-cut into 11 rows of 63: the round-4 shim did that silently
    with pytest.raises(ValueError):
        nerf_mlp((torch.rand(11, 63, device="cuda"), torch.rand(11, 63, device="cuda")))


def test_nerf_mlp_takes_any_two_widths_from_its_first_call():
    """Keras Dense: the input size is the last dimension of the first call -- no 3 + 6 L structure, and the two inputs are
    independent (mlp.py:11-27 names no input size anywhere)"""
    from keras_nerf_amd.model.nerf.mlp import NeRFMLP
    rng = np.random.default_rng(3)
    for (nl, units, skip), (wx, wd) in (((4, 64, 2), (50, 7)), ((8, 256, 4), (63, 27)), ((3, 128, 1), (5, 130))):
        m = NeRFMLP(nl, units, skip, seed=11)
        x = (rng.random((3, 9, wx), dtype=np.float32) * 2 - 1); dd = (rng.random((3, 9, wd), dtype=np.float32) * 2 - 1)
        rgb, sigma = m((x, dd))
        assert rgb.shape == (3, 9, 3) and sigma.shape == (3, 9, 1) and (m.xyz_dim, m.dir_dim) == (wx, wd)
        ws = m.get_weights()
        assert ws[0].shape == (wx, units) and ws[2 * (nl + 2)].shape == (units + wd, units // 2)
        cfg = O.NerfConfig(n_layers=nl, dense_units=units, skip_layer=skip)
        er, es = O.mlp_forward(ws, x, dd, cfg, emulate_bf16=O.FUSED)
        np.testing.assert_allclose(rgb.cpu().numpy(), er, atol=2e-3); np.testing.assert_allclose(sigma.cpu().numpy(), es, atol=4e-3)
        with pytest.raises(ValueError):
            m((x[..., :-1], dd))
    # more rows than one library call takes (NeRFMLP.CALL_ROWS): the same values as row-by-row slices
    m3 = NeRFMLP(4, 64, 2, seed=2)
    big = torch.rand((3 * 2 ** 19 + 5, 9), device="cuda"); bigd = torch.rand((3 * 2 ** 19 + 5, 4), device="cuda")
    m3.CALL_ROWS = 1 << 19
    r_all, s_all = m3((big, bigd))
    assert r_all.shape == (3 * 2 ** 19 + 5, 3)
    for lo in (0, 2 ** 19 - 3, 3 * 2 ** 19 - 2):
        r_part, s_part = m3((big[lo:lo + 7], bigd[lo:lo + 7]))
        assert torch.equal(r_part, r_all[lo:lo + 7]) and torch.equal(s_part, s_all[lo:lo + 7])
    # an MLP whose widths came from its weights (set_weights / load_weights on an unbuilt model) serves calls of those widths
    m2 = NeRFMLP(4, 64, 2)
    m2.set_weights(NeRFMLP(4, 64, 2, xyz_dim=50, dir_dim=7, seed=5).get_weights())
    assert (m2.xyz_dim, m2.dir_dim) == (50, 7)
    assert m2((np.zeros((2, 50), np.float32), np.zeros((2, 7), np.float32)))[0].shape == (2, 3)
    with pytest.raises(ValueError):
        m2.build(((None, 63), (None, 27)))


@pytest.fixture(scope="module")
def nerf_utils():
    from keras_nerf_amd.model.nerf.utils import NeRFUtils
    return NeRFUtils(batch_size=2, image_height=128, image_width=128, ray_chunks=1024, pos_emb_xyz=10, pos_emb_dir=4, white_background=True)


N_COARSE, N_FINE, POS_XYZ, POS_DIR = 32, 64, 10, 4          # the fixtures of test_nerf_utils.py:20-37


def _uniform(seed, *shape):
    return torch.rand(shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(seed))


def _rays(idx, *arrays):
    """the sampled rays of [2, 128, 128, ...] tensors as numpy [n, ...]"""
    return [a.reshape((2 * 128 * 128,) + tuple(a.shape[3:]))[torch.as_tensor(idx, device="cuda")].cpu().numpy() for a in arrays]


IDX = np.sort(np.random.default_rng(7).choice(2 * 128 * 128, 512, replace=False))


def test_render_image_depth(nerf_utils):
    """test_nerf_utils.py:40-51; values: utils.py:99-134 (the non-chunk twin: no white background, no clip)"""
    rgb, sigma, points = _uniform(1, 2, 128, 128, N_COARSE, 3), _uniform(2, 2, 128, 128, N_COARSE, 1), _uniform(3, 2, 128, 128, N_COARSE)
    image, depth, weights = nerf_utils.render_image_depth(rgb, sigma, points)
    assert image.shape == (2, 128, 128, 3)
    assert depth.shape == (2, 128, 128)
    assert weights.shape == (2, 128, 128, N_COARSE)
    r, s, t, gi, gd, gw = _rays(IDX, rgb, sigma, points, image, depth, weights)
    _, ed, ew = O.render_image_depth_chunk(r, s, t, False)      # the reference feeds UNSORTED points: negative deltas and all
    np.testing.assert_allclose(gw, ew, rtol=2e-5, atol=2e-6); np.testing.assert_allclose(gd, ed, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(gi, np.sum(ew[..., None] * r, -2), rtol=2e-5, atol=1e-5)


def test_positional_encoding(nerf_utils):
    """test_nerf_utils.py:54-62"""
    rays = _uniform(4, 2, 128, 128, N_COARSE, 3)
    pos_encoded_rays = nerf_utils.positional_encoding(rays, POS_XYZ)
    assert pos_encoded_rays.shape == (2, 128, 128, N_COARSE, 3 * 2 * POS_XYZ + 3)
    x, g = _rays(IDX, rays, pos_encoded_rays)
    np.testing.assert_allclose(g, O.positional_encoding(x, POS_XYZ), atol=3e-4)


def test_fine_hierarchical_sampling(nerf_utils):
    """test_nerf_utils.py:65-76 (u drawn inside, as tf.random.uniform is there); then the same call with u injected, bit for bit"""
    coarse_points = _uniform(5, 2, 128, 128, N_COARSE)
    mid_points = 0.5 * (coarse_points[..., 1:] + coarse_points[..., :-1])
    weights = _uniform(6, 2, 128, 128, N_COARSE)
    fine_points = nerf_utils.fine_hierarchical_sampling(mid_points, weights, N_FINE)
    assert fine_points.shape == (2, 128, 128, N_FINE)
    assert bool(torch.isfinite(fine_points).all())
    u = _uniform(7, 2 * 128 * 128, N_FINE)
    fp = nerf_utils.fine_hierarchical_sampling(mid_points, weights, N_FINE, u=u)
    m, w, g = _rays(IDX, mid_points, weights, fp)
    np.testing.assert_array_equal(g, O.fine_hierarchical_sampling_chunk(m, w, u[torch.as_tensor(IDX, device="cuda")].cpu().numpy(), "zero"))


def test_encode_position_and_directions(nerf_utils):
    """test_nerf_utils.py:79-93"""
    ray_origin, ray_direction, sample_points = _uniform(8, 2, 128, 128, 3), _uniform(9, 2, 128, 128, 3), _uniform(10, 2, 128, 128, N_COARSE)
    pos_encoded_rays, pos_encoded_directions = nerf_utils.encode_position_and_directions(ray_origin, ray_direction, sample_points)
    assert pos_encoded_rays.shape == (2, 128, 128, N_COARSE, 3 * 2 * POS_XYZ + 3)
    assert pos_encoded_directions.shape == (2, 128, 128, N_COARSE, 3 * 2 * POS_DIR + 3)
    o, d, t, gx, gd = _rays(IDX, ray_origin, ray_direction, sample_points, pos_encoded_rays, pos_encoded_directions)
    ex, ed = O.encode_position_and_directions(o, d, t, POS_XYZ, POS_DIR)
    np.testing.assert_allclose(gx, ex, atol=3e-4); np.testing.assert_allclose(gd, ed, atol=1e-5)


def test_encode_position_and_directions_chunk(nerf_utils):
    """test_nerf_utils.py:96-114"""
    ray_origin, ray_direction, sample_points = _uniform(11, 2, 128, 128, 3), _uniform(12, 2, 128, 128, 3), _uniform(13, 2, 128, 128, N_COARSE)
    ray_origin, ray_direction, sample_points = ray_origin.reshape(-1, 3), ray_direction.reshape(-1, 3), sample_points.reshape(-1, N_COARSE)
    pos_encoded_rays, pos_encoded_directions = nerf_utils.encode_position_and_directions(ray_origin, ray_direction, sample_points)
    assert pos_encoded_rays.shape == (2 * 128 * 128, N_COARSE, 3 * 2 * POS_XYZ + 3)
    assert pos_encoded_directions.shape == (2 * 128 * 128, N_COARSE, 3 * 2 * POS_DIR + 3)
    ti = torch.as_tensor(IDX, device="cuda")
    ex, ed = O.encode_position_and_directions(ray_origin[ti].cpu().numpy(), ray_direction[ti].cpu().numpy(), sample_points[ti].cpu().numpy(), POS_XYZ, POS_DIR)
    np.testing.assert_allclose(pos_encoded_rays[ti].cpu().numpy(), ex, atol=3e-4)
    np.testing.assert_allclose(pos_encoded_directions[ti].cpu().numpy(), ed, atol=1e-5)


def test_render_image_depth_chunk(nerf_utils):
    """test_nerf_utils.py:117-129 (`depth.shape == (1024)` there compares with the INT 1024, which a TensorShape accepts; a torch.Size
    compares with the tuple)"""
    rgb, sigma, points = _uniform(14, 1024, N_COARSE, 3), _uniform(15, 1024, N_COARSE, 1), _uniform(16, 1024, N_COARSE)
    image, depth, weights = nerf_utils.render_image_depth_chunk(rgb, sigma, points)
    assert image.shape == (1024, 3)
    assert depth.shape == (1024,)
    assert weights.shape == (1024, N_COARSE)
    ei, ed, ew = O.render_image_depth_chunk(rgb.cpu().numpy(), sigma.cpu().numpy(), points.cpu().numpy(), True)
    np.testing.assert_allclose(weights.cpu().numpy(), ew, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(image.cpu().numpy(), ei, rtol=2e-5, atol=1e-5); np.testing.assert_allclose(depth.cpu().numpy(), ed, rtol=2e-5, atol=2e-5)
