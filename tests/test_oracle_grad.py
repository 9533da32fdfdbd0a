"""Pins the oracle's hand-written backward and train step: fp64 oracle == torch autograd (fp64) and fp32 ~ fp64."""
import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from oracle import torch_ref as T


def small_problem(seed=0, R=16, cfg=None, dtype=np.float64):
    cfg = cfg or O.NerfConfig(n_coarse=8, n_fine=16, pos_emb_xyz=4, pos_emb_dir=2, n_layers=8, dense_units=32,
                              skip_layer=4)
    rng = np.random.default_rng(seed)
    c2w = O.pose_spherical(40.0, -30.0, 4.0)
    o, d, t = O.generate_rays(c2w, 8.0, 4, 4, 2.0, 6.0, cfg.n_coarse, rng.random((4, 4, cfg.n_coarse)))
    o, d, t = [a.reshape(R, -1).astype(dtype) for a in (o, d, t)]
    u = rng.random((R, cfg.n_fine)).astype(dtype)
    img = rng.random((R, 3)).astype(dtype)
    cp = [p * 3 for p in O.init_params(cfg, 1, dtype)]   # x3: make sigma/rgb non-trivial
    fp = [p * 3 for p in O.init_params(cfg, 2, dtype)]
    for p in cp[1::2] + fp[1::2]:
        p += rng.normal(0, 0.1, p.shape)                 # non-zero biases
    return cfg, o, d, t, u, img, cp, fp


@pytest.mark.parametrize("white", [False, True])
def test_chunk_grads_match_autograd_fp64(white):
    cfg, o, d, t, u, img, cp, fp = small_problem()
    res, loss, grads = O.chunk_loss_and_grads(cp, o, d, t, img, cfg, white)
    tp = [torch.tensor(p, requires_grad=True) for p in cp]
    timg, _, tw = T.chunk_forward(tp, torch.tensor(o), torch.tensor(d), torch.tensor(t), cfg, white)
    tl = torch.mean((torch.tensor(img) - timg) ** 2)
    tg = torch.autograd.grad(tl, tp)
    assert float(tl.detach()) == pytest.approx(float(loss), rel=1e-12)
    np.testing.assert_allclose(res["weights"], tw.detach().numpy(), rtol=1e-10, atol=1e-14)
    for g, t_ in zip(grads, tg):
        np.testing.assert_allclose(g, t_.numpy(), rtol=1e-8, atol=1e-13)
    assert sum(float(np.abs(g).sum()) for g in grads) > 0


@pytest.mark.parametrize("nl,sk,units,lx,ld", [(8, 2, 16, 4, 2), (6, 3, 32, 3, 1), (4, 2, 16, 5, 2), (12, 4, 16, 2, 3), (10, 5, 16, 6, 4)])
def test_chunk_grads_match_autograd_for_other_trunk_shapes(nl, sk, units, lx, ld):
    """the trunk shapes the fused GPU kernels are tested on against THIS oracle (tests/test_gpu_fused_shapes.py, test_gpu_variants.py:
    several concat layers, a concat layer in last position, other encoding depths), at small width: hand-written backward == autograd"""
    cfg = O.NerfConfig(n_coarse=8, n_fine=16, pos_emb_xyz=lx, pos_emb_dir=ld, n_layers=nl, dense_units=units, skip_layer=sk)
    cfg, o, d, t, u, img, cp, fp = small_problem(seed=nl, cfg=cfg)
    res, loss, grads = O.chunk_loss_and_grads(cp, o, d, t, img, cfg, True)
    tp = [torch.tensor(p, requires_grad=True) for p in cp]
    timg, _, tw = T.chunk_forward(tp, torch.tensor(o), torch.tensor(d), torch.tensor(t), cfg, True)
    tl = torch.mean((torch.tensor(img) - timg) ** 2)
    tg = torch.autograd.grad(tl, tp)
    assert float(tl.detach()) == pytest.approx(float(loss), rel=1e-12)
    assert len(grads) == 2 * (nl + 4)
    for g, t_ in zip(grads, tg):
        np.testing.assert_allclose(g, t_.numpy(), rtol=1e-8, atol=1e-13)


def test_fine_sampling_matches_torch_both_modes():
    cfg, o, d, t, u, img, cp, fp = small_problem()
    rng = np.random.default_rng(5)
    w = rng.random(t.shape) ** 6                      # peaky weights -> out-of-range gathers occur
    mids = 0.5 * (t[:, 1:] + t[:, :-1])
    hit = False
    for oob in ("zero", "clamp"):
        s = O.fine_hierarchical_sampling_chunk(mids, w, u, oob)
        ts = T.fine_sampling(torch.tensor(mids), torch.tensor(w), torch.tensor(u), oob).numpy()
        np.testing.assert_allclose(s, ts, rtol=1e-12, atol=1e-14)
    z = O.fine_hierarchical_sampling_chunk(mids, w, u, "zero")
    c = O.fine_hierarchical_sampling_chunk(mids, w, u, "clamp")
    assert np.any(z != c)   # the hazard of SURVEY.md section 8a-6 is exercised


def test_train_step_matches_torch_fp64():
    cfg, o, d, t, u, img, cp, fp = small_problem()
    cp2 = [torch.tensor(p.copy()).requires_grad_() for p in cp]
    fp2 = [torch.tensor(p.copy()).requires_grad_() for p in fp]
    oc, of_ = O.KerasAdam(cp), O.KerasAdam(fp)
    toc, tof = T.TorchKerasAdam(cp2), T.TorchKerasAdam(fp2)
    for step in range(3):
        m, ci, fi, _ = O.train_step(cp, fp, oc, of_, img.reshape(1, 4, 4, 3), o.reshape(1, 4, 4, 3),
                                    d.reshape(1, 4, 4, 3), t.reshape(1, 4, 4, -1), u.reshape(1, 4, 4, -1), cfg, 4, True)
        lc, lf, _, _ = T.train_step(cp2, fp2, toc, tof, torch.tensor(img), torch.tensor(o), torch.tensor(d),
                                    torch.tensor(t), torch.tensor(u), cfg, 4, True)
        assert float(m["coarse_loss"]) == pytest.approx(lc, rel=1e-10)
        assert float(m["fine_loss"]) == pytest.approx(lf, rel=1e-10)
    for a, b in zip(cp + fp, cp2 + fp2):
        np.testing.assert_allclose(a, b.detach().numpy(), rtol=1e-7, atol=1e-10)


def test_chunk_accumulation_equals_full_batch_gradient():
    # accumulating g/C over equal chunks == gradient of the mean MSE over all rays
    cfg, o, d, t, u, img, cp, fp = small_problem()
    _, _, g_full = O.chunk_loss_and_grads(cp, o, d, t, img, cfg, False)
    _, _, _, (acc_c, _) = O.train_step(cp, fp, None, None, img.reshape(1, 4, 4, 3), o.reshape(1, 4, 4, 3),
                                       d.reshape(1, 4, 4, 3), t.reshape(1, 4, 4, -1), u.reshape(1, 4, 4, -1), cfg, 4, False)
    for a, b in zip(acc_c, g_full):
        np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-14)


def test_ray_chunks_divisibility_assert():
    cfg, o, d, t, u, img, cp, fp = small_problem()
    with pytest.raises(AssertionError):
        O.predict_and_render_images(cp, fp, o, d, t, u, cfg, 5, False)


def test_fp32_close_to_fp64():
    cfg, o, d, t, u, img, cp, fp = small_problem()
    c64, f64 = O.predict_and_render_images(cp, fp, o, d, t, u, cfg, 8, True)
    to32 = lambda xs: [x.astype(np.float32) for x in xs]
    c32, f32 = O.predict_and_render_images(to32(cp), to32(fp), *to32([o, d, t, u]), cfg, 8, True)
    np.testing.assert_allclose(c32["image"], c64["image"], atol=1e-4)
    np.testing.assert_allclose(c32["weights"], c64["weights"], atol=1e-4)
    # fine pass: a searchsorted flip moves one sample continuously, so images stay close
    np.testing.assert_allclose(f32["image"], f64["image"], atol=1e-3)
    assert f32["weights"].shape == (16, cfg.n_coarse + cfg.n_fine)


def test_collapsed_head_is_the_same_function_and_the_same_gradients(monkeypatch):
    """The fused kernels evaluate features -> rgb_features -> rgb (all linear in the reference, mlp.py:21-24,44-48) and the
    sigma head as one affine map of (h7, dir_enc) and recover the six head gradients from sums over samples (oracle FUSED
    mode).  With the bf16 rounding switched off that must be the layer-by-layer computation to fp64 rounding: outputs, all
    24 gradients, every tensor."""
    monkeypatch.setattr(O, "_rb", lambda x: np.asarray(x))
    for cfg in (O.NerfConfig(n_coarse=8, n_fine=8, pos_emb_xyz=4, pos_emb_dir=2, n_layers=4, dense_units=32, skip_layer=2),
                O.NerfConfig(n_coarse=8, n_fine=8, pos_emb_xyz=3, pos_emb_dir=1, n_layers=3, dense_units=16, skip_layer=1),    # concat after the LAST layer
                O.NerfConfig(n_coarse=8, n_fine=8, pos_emb_xyz=5, pos_emb_dir=3, n_layers=8, dense_units=16, skip_layer=2),    # three concat layers
                O.NerfConfig(n_coarse=8, n_fine=8, pos_emb_xyz=6, pos_emb_dir=4, n_layers=12, dense_units=16, skip_layer=4)):
        _check_collapsed_head(cfg)


def _check_collapsed_head(cfg):
    rng = np.random.default_rng(5)
    params = [(p * 2).astype(np.float64) for p in O.init_params(cfg, 3)]
    for b in params[1::2]:
        b += rng.normal(0, 0.1, b.shape)
    xyz = rng.normal(0, 1, (5, 8, cfg.xyz_dim)); dire = rng.normal(0, 1, (5, 8, cfg.dir_dim))
    drgb = rng.normal(0, 1, (5, 8, 3)); dsig = rng.normal(0, 1, (5, 8, 1))
    out = {}
    for mode in (False, O.FUSED):
        rgb, sigma, cache = O.mlp_forward(params, xyz, dire, cfg, mode, want_cache=True)
        out[mode] = (rgb, sigma, O.mlp_backward(params, cache, drgb, dsig, cfg))
    np.testing.assert_allclose(out[O.FUSED][0], out[False][0], rtol=0, atol=1e-13)
    np.testing.assert_allclose(out[O.FUSED][1], out[False][1], rtol=0, atol=1e-13)
    names = [f"{n}/{k}" for n, _, _ in O.layer_shapes(cfg) for k in ("kernel", "bias")]
    for name, a, b in zip(names, out[O.FUSED][2], out[False][2]):
        np.testing.assert_allclose(np.asarray(a).reshape(b.shape), b, rtol=1e-10, atol=1e-12, err_msg=name)
        assert np.abs(b).max() > 0, name
