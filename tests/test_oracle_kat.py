"""Analytic known-answer tests that pin the NumPy oracle (the reference holds no numeric
fixtures for the hot path, so these + the autograd cross-check are what pin it)."""
import numpy as np
import pytest

from oracle import nerf_oracle as O


def test_focal_known_answer():
    # the only known-answer test in the reference: tests/data/test_utils.py:5-10
    assert O.get_focal_from_fov(0.6911112070083618, 100) == pytest.approx(138.88887889922103, rel=1e-6)


def test_layer_shapes_and_param_count():
    cfg = O.NerfConfig()
    shapes = O.layer_shapes(cfg)
    assert [s[0] for s in shapes] == [f"layer_{i}" for i in range(8)] + ["sigma", "features", "rgb_features", "rgb"]
    assert shapes[0][1:] == (63, 256) and shapes[5][1:] == (319, 256) and shapes[4][1:] == (256, 256)
    assert shapes[8][1:] == (256, 1) and shapes[10][1:] == (283, 128) and shapes[11][1:] == (128, 3)
    assert O.param_count(cfg) == 595844


def test_positional_encoding_known_angles():
    x = np.array([[0.0, np.pi / 2, 1.0]], np.float64)
    pe = O.positional_encoding(x, 2)
    assert pe.shape == (1, 15)
    exp = np.concatenate([x[0], np.sin(x[0]), np.cos(x[0]), np.sin(2 * x[0]), np.cos(2 * x[0])])
    np.testing.assert_allclose(pe[0], exp, atol=1e-15)
    # layout: [x, sin(2^0 x), cos(2^0 x), ...] blocks of 3; no pi factor
    assert pe[0, 3] == 0.0 and pe[0, 6] == 1.0 and pe[0, 4] == pytest.approx(1.0)
    assert O.positional_encoding(np.zeros((2, 3)), 10).shape == (2, 63)


def test_constant_sigma_slab():
    # constant sigma, uniform spacing: w_i = (1-e^{-s d}) e^{-s d i} up to the +1e-10 in the cumprod
    S, s, dlt = 16, 0.7, 0.25
    t = (2.0 + dlt * np.arange(S))[None, :]
    sigma = np.full((1, S, 1), s)
    rgb = np.full((1, S, 3), 0.5)
    img, depth, w = O.render_image_depth_chunk(rgb, sigma, t, False)
    a = 1 - np.exp(-s * dlt)
    exp_w = a * np.exp(-s * dlt * np.arange(S))
    exp_w[-1] = (1 - np.exp(-s * 1e-10)) * np.exp(-s * dlt * (S - 1))  # last delta is 1e-10
    np.testing.assert_allclose(w[0], exp_w, rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(img[0], 0.5 * exp_w.sum(), rtol=1e-8)
    np.testing.assert_allclose(depth[0], (exp_w * t[0]).sum(), rtol=1e-8)
    imgw, _, _ = O.render_image_depth_chunk(rgb, sigma, t, True)
    np.testing.assert_allclose(imgw[0], 0.5 * exp_w.sum() + 1 - exp_w.sum(), rtol=1e-8)


def test_last_delta_is_tiny_not_huge():
    t = np.linspace(2, 6, 8)[None]
    sigma = np.full((1, 8, 1), 50.0)
    _, _, w = O.render_image_depth_chunk(np.ones((1, 8, 3)), sigma, t, False)
    assert w[0, -1] < 1e-8  # alpha_last = 1-exp(-50e-10) ~ 0


def test_clip():
    t = np.linspace(2, 6, 4)[None]
    img, _, _ = O.render_image_depth_chunk(np.full((1, 4, 3), 5.0), np.full((1, 4, 1), 100.0), t, False)
    assert np.all(img == 1.0)


def test_uniform_pdf_inverse_cdf_is_linear_in_u():
    S = 8
    t = np.linspace(2.0, 6.0, S)[None]
    mids = 0.5 * (t[:, 1:] + t[:, :-1])
    w = np.ones((1, S))
    # choose u strictly inside bins whose mid-point indices are in range (idx <= S-2)
    u = np.array([[0.05, 0.13, 0.3, 0.45, 0.6]])
    s = O.fine_hierarchical_sampling_chunk(mids, w, u, "zero")
    cdf = np.arange(S + 1) / S
    idx = np.searchsorted(cdf, u[0], side="right")
    b, a = idx - 1, idx
    exp = mids[0, b] + (u[0] - cdf[b]) / (1 / S) * (mids[0, a] - mids[0, b])
    np.testing.assert_allclose(s[0], exp, rtol=1e-12)


def test_oob_modes_on_hand_built_case():
    # all mass in the last coarse bin -> idx = S -> below = S-1 (one past the mid-points), above = S (two past)
    S = 8
    t = np.linspace(2.0, 6.0, S)[None].astype(np.float32)
    mids = (0.5 * (t[:, 1:] + t[:, :-1])).astype(np.float32)
    w = np.zeros((1, S), np.float32); w[0, -1] = 1.0
    u = np.array([[0.5]], np.float32)
    z = O.fine_hierarchical_sampling_chunk(mids, w, u, "zero")
    c = O.fine_hierarchical_sampling_chunk(mids, w, u, "clamp")
    assert z[0, 0] == 0.0                       # 0 + t*(0-0)
    assert c[0, 0] == mids[0, -1]               # m_b = m_a = last mid
    # a draw whose 'above' is S-1 (past the end) but 'below' S-2 (valid): zero mode interpolates towards 0
    w2 = np.zeros((1, S), np.float32); w2[0, -2] = 1.0
    z2 = O.fine_hierarchical_sampling_chunk(mids, w2, u, "zero")
    c2 = O.fine_hierarchical_sampling_chunk(mids, w2, u, "clamp")
    assert 0.0 < z2[0, 0] < mids[0, -1] and c2[0, 0] == mids[0, -1]


def test_searchsorted_right_semantics():
    cdf = O.cdf_from_weights(np.ones((1, 4)))
    np.testing.assert_allclose(cdf[0], [0, .25, .5, .75, 1.0], atol=1e-6)
    mids = np.array([[1.0, 2.0, 3.0]])
    # u exactly on a cdf knot goes to the right bin (side='right')
    s = O.fine_hierarchical_sampling_chunk(mids, np.ones((1, 4)), np.array([[0.25]]), "zero")
    cdfv = cdf[0]
    exp = 2.0 + (0.25 - cdfv[1]) / (cdfv[2] - cdfv[1]) * (3.0 - 2.0)
    assert s[0, 0] == pytest.approx(exp, abs=1e-6)


def test_pose_spherical_by_hand():
    c2w = O.pose_spherical(0.0, -30.0, 4.0)
    # theta=0: rot_theta = I.  rot_phi(-30deg) @ trans(4): translation column = (0, -sin(phi)*4, cos(phi)*4)
    ph = np.deg2rad(-30.0)
    tcol = np.array([0.0, -np.sin(ph) * 4, np.cos(ph) * 4])
    flip = np.array([[-1, 0, 0], [0, 0, 1], [0, 1, 0]], float)
    np.testing.assert_allclose(c2w[:3, 3], flip @ tcol, atol=1e-6)
    assert np.linalg.norm(c2w[:3, 3]) == pytest.approx(4.0, abs=1e-5)
    np.testing.assert_allclose(c2w[:3, :3] @ c2w[:3, :3].T, np.eye(3), atol=1e-6)


def test_rays_match_reference_test_bounds():
    # tests/data/test_rays.py:50-87: shapes, unit directions, t within [near, far], o == translation
    rng = np.random.default_rng(0)
    c2w = O.pose_spherical(30.0, -30.0, 4.0)
    o, d, t = O.generate_rays(c2w, 138.88887889922103, 128, 128, 2.0, 6.0, 32, rng.random((128, 128, 32)))
    assert o.shape == (128, 128, 3) and d.shape == (128, 128, 3) and t.shape == (128, 128, 32)
    np.testing.assert_allclose(np.linalg.norm(d, axis=-1), 1.0, atol=1e-6)
    assert t.min() >= 2.0 and t.max() <= 6.0 and np.all(np.diff(t, axis=-1) >= 0)
    np.testing.assert_array_equal(o[5, 7], c2w[:3, 3])
    # centre pixel looks along -z of the camera
    np.testing.assert_allclose(d[64, 64], -c2w[:3, 2], atol=1e-6)


def test_keras_adam_first_step():
    p = [np.array([1.0, -2.0], np.float64)]
    g = [np.array([0.5, -0.25], np.float64)]
    opt = O.KerasAdam(p)
    opt.apply(p, g)
    # t=1: m=(1-b1)g, v=(1-b2)g^2, lr_t = lr*sqrt(1-b2)/(1-b1) -> step = lr * g/(|g| + eps*sqrt(1-b2)... ) ~ lr*sign(g)
    lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    exp = np.array([1.0, -2.0]) - lr_t * (0.1 * g[0]) / (np.sqrt(0.001 * g[0] ** 2) + 1e-7)
    np.testing.assert_allclose(p[0], exp, rtol=1e-12)


def test_philox_known_answer():
    # Random123 KAT: philox4x32-10, counter=0, key=0 -> 6627e8d5 e169c58d bc57ac4c 9b00dbd8
    out = O.philox4x32(np.zeros((1, 4), np.uint32), np.zeros((1, 2), np.uint32))[0]
    assert [hex(int(x)) for x in out] == ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']
    out = O.philox4x32(np.full((1, 4), 0xFFFFFFFF, np.uint32), np.full((1, 2), 0xFFFFFFFF, np.uint32))[0]
    assert [hex(int(x)) for x in out] == ['0x408f276d', '0x41c83b0e', '0xa20bc7c6', '0x6d5451fd']
    u = O.philox_uniform_u(7, 1, np.arange(5), 128)
    assert u.shape == (5, 128) and u.min() >= 0 and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.05


def test_bf16_rounding():
    x = np.array([1.0, 1.00390625, 1.0 + 2 ** -9, 3.14159], np.float32)
    r = O.round_bf16(x)
    assert r[0] == 1.0 and r[1] == 1.0 and r[3] == pytest.approx(3.140625)
    assert O.round_bf16(np.array([1.0 + 3 * 2 ** -9], np.float32))[0] == np.float32(1.0 + 2 ** -7)  # ties-to-even up


def test_lego_camera_fixture_of_the_reference_ray_test():
    """tests/golden/lego_c2w.json holds the camera matrix and arguments of the reference's tests/data/test_rays.py:9-47; the
    oracle's generate_rays (rays.py:69-130) under that test's own assertions (:50-87): finite, o/d deterministic, the
    jittered t within [near - 4/32, far + 4/32] and within 4/32 of the previous draw."""
    import json
    import os
    F = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "lego_c2w.json")))
    c2w = np.asarray(F["camera_to_world"], np.float32)
    W, H, S = F["image_width"], F["image_height"], F["n_sample"]
    rng = np.random.default_rng(0)
    last = None
    for _ in range(3):
        o, d, t = O.generate_rays(c2w, F["focal_length"], W, H, F["near"], F["far"], S, rng.random((H, W, S)))
        assert o.shape == (H, W, 3) and d.shape == (H, W, 3) and t.shape == (H, W, S)
        assert o.dtype == np.float32 and d.dtype == np.float32 and t.dtype == np.float32
        assert np.isfinite(o).all() and np.isfinite(d).all() and np.isfinite(t).all()
        assert t.min() >= 2.0 - 4.0 / 32.0 and t.max() <= 6.0 + 4.0 / 32.0
        if last is not None:
            assert np.array_equal(last[0], o) and np.array_equal(last[1], d) and np.allclose(last[2], t, atol=4.0 / 32.0)
        last = (o, d, t)
    np.testing.assert_allclose(np.linalg.norm(d, axis=-1), 1.0, atol=1e-6)
    np.testing.assert_array_equal(o[5, 7], c2w[:3, 3])
    # the centre pixel looks along -z of the camera frame (rays.py:82-113: x - W/2, -(y - H/2), -1)
    np.testing.assert_allclose(d[H // 2, W // 2], -c2w[:3, 2] / np.linalg.norm(c2w[:3, 2]), atol=1e-6)


def test_round_bf16_is_round_to_nearest_even_on_the_bit_pattern():
    """the oracle's bf16 emulation (the kernels' operand rounding: v_cvt_pk_bf16_f32, RNE): the fast uint32 form against the
    definition written out in uint64, on random values of every magnitude, ties to even both ways, signed zeros, the largest finite
    value that still rounds to a finite bf16 and denormals"""
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(200000).astype(np.float32) * np.float32(10.0) ** rng.integers(-30, 30, 200000).astype(np.float32),
                        rng.integers(0, 0x7F7F0000, 100000, dtype=np.uint32).view(np.float32),
                        np.array([0.0, -0.0, 1.0, 1.00390625, 1.0 + 2.0 ** -8, 1.0 + 2.0 ** -8 + 2.0 ** -20, 1.0 + 3 * 2.0 ** -8, 3.3895314e38, -3.3895314e38, 1e-40, 2.0 ** -126],
                                 np.float32)])
    u = x.view(np.uint32).astype(np.uint64)
    want = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    got = O.round_bf16(x)
    np.testing.assert_array_equal(got.view(np.uint32), want)
    assert got.dtype == np.float32 and got.shape == x.shape and not np.shares_memory(got, x)
    # ties: 1 + 2^-8 is halfway between bf16 neighbours 1 and 1 + 2^-7 -> even mantissa (1); 1 + 3 * 2^-8 -> 1 + 2^-6 (even)
    np.testing.assert_array_equal(O.round_bf16(np.array([1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -8], np.float32)), np.array([1.0, 1.0 + 2.0 ** -6], np.float32))
    assert O.round_bf16(x.reshape(3, -1)).shape == (3, x.size // 3) if x.size % 3 == 0 else True
