"""Size-independent properties at the bench size (BASELINE cfg2: chunks of 4096 rays, 64 + 128 samples), where the CPU
oracle would take minutes: permutation equivariance over rays, loss/images consistency, sampler invariants, partition
of unity of the compositing weights, accumulation linearity, and the analytic first Adam step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

R, NC, NF = 4096, 64, 128


@pytest.fixture(scope="module")
def setup():
    from keras_nerf_amd.data.utils import get_focal_from_fov, pose_spherical
    from keras_nerf_amd.model.nerf.mlp import NeRFMLP
    from keras_nerf_amd.runtime import KnerfContext
    ctx = KnerfContext(white_background=True)
    for net in (0, 1):
        m = NeRFMLP(seed=10 + net); m.build(); ctx.set_weights(net, m.get_flat_weights() * 1.5)
    o, d, t = ctx.generate_rays(pose_spherical(47.0, -30.0, 4.0)[None], get_focal_from_fov(0.6911112070083618, 64), 64, 64, 2.0, 6.0,
                                NC, None, seed=5)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    data = dict(o=o.reshape(-1, 3).contiguous(), d=d.reshape(-1, 3).contiguous(), t=t.reshape(-1, NC).contiguous(),
                tgt=torch.rand((R, 3), device="cuda", generator=g), u=torch.rand((R, NF), device="cuda", generator=g))
    yield ctx, data
    ctx.close()


def test_render_invariants(setup):
    ctx, D = setup
    out = ctx.render_chunk(D["o"], D["d"], D["t"], D["u"])
    tf = out["t_fine"]
    assert tf.shape == (R, NC + NF)
    assert bool((tf[:, 1:] >= tf[:, :-1]).all())                                   # sorted (nerf.py:190-191)
    assert float(tf.min()) >= 0.0 and float(tf.max()) <= 6.0 + 1e-4                # oob='zero' may gather 0, never beyond far
    # every coarse t survives the merge (multiset inclusion): merging and re-sorting with the coarse set changes nothing
    merged = torch.sort(torch.cat([tf, D["t"]], -1), -1).values
    assert bool((merged[:, ::1].shape[1] == NC + NF + NC))
    for key, S in (("c_weights", NC), ("f_weights", NC + NF)):
        w = out[key]
        assert w.shape == (R, S) and float(w.min()) >= 0.0
        assert float(w.sum(-1).max()) <= 1.0 + 1e-4                                # alpha compositing: partition of at most one
    for key in ("c_image", "f_image"):
        img = out[key]
        assert float(img.min()) >= 0.0 and float(img.max()) <= 1.0                 # clip (utils.py:55-56)
    # white background: image = sum w rgb + (1 - sum w) >= 1 - sum w
    assert bool((out["f_image"].min(-1).values + 1e-4 >= 1.0 - out["f_weights"].sum(-1)).all())


def test_permuting_rays_permutes_outputs_and_keeps_gradients(setup):
    ctx, D = setup
    perm = torch.randperm(R, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    res = []
    for p in (None, perm):
        sel = (lambda x: x) if p is None else (lambda x: x[p].contiguous())
        loss = torch.zeros(2, device="cuda")
        ci = torch.empty((R, 3), device="cuda"); fi = torch.empty((R, 3), device="cuda")
        ctx.zero_grads()
        ctx.train_chunk(sel(D["o"]), sel(D["d"]), sel(D["t"]), sel(D["tgt"]), sel(D["u"]), loss=loss, c_image=ci, f_image=fi)
        torch.cuda.synchronize()
        res.append((ci.clone(), fi.clone(), loss.clone(), ctx.grads_view().clone()))
    (c0, f0, l0, g0), (c1, f1, l1, g1) = res
    # each ray is processed independently of its neighbours: images move with the rays, bit for bit
    assert torch.equal(c0[perm], c1) and torch.equal(f0[perm], f1)
    assert float((l0 - l1).abs().max()) < 1e-6
    assert float((g0 - g1).abs().max()) <= 2e-5 * float(g0.abs().max())           # fp32 summation order only
    # the loss the kernels report is the MSE of the images they return (train_single.py:127)
    for k, img in ((0, c0), (1, f0)):
        mse = float(((img - D["tgt"]) ** 2).mean())
        assert abs(mse - float(l0[k])) < 1e-5


def test_accumulation_is_linear_in_the_chunk_weight(setup):
    ctx, D = setup
    a = [D[k] for k in ("o", "d", "t", "tgt", "u")]
    ctx.zero_grads(); ctx.train_chunk(*a, inv_chunks=1.0); g1 = ctx.grads_view().clone()
    ctx.zero_grads(); ctx.train_chunk(*a, inv_chunks=0.25); ctx.train_chunk(*a, inv_chunks=0.75); g2 = ctx.grads_view().clone()
    assert float((g1 - g2).abs().max()) <= 2e-3 * float(g1.abs().max())            # bf16 dZ of the scaled passes: 2^-9 relative
    ctx.zero_grads()
    assert float(ctx.grads_view().abs().max()) == 0.0


def test_first_adam_step_moves_every_touched_weight_by_the_learning_rate(setup):
    """Keras Adam, step 1, zero slots: m = (1-b1) g, v = (1-b2) g^2, lr_t = lr sqrt(1-b2)/(1-b1) =>
    dw = -lr g / (|g| + eps / sqrt(1-b2)), eps / sqrt(1-b2) = 3.16e-6: -lr sign(g) for large gradients, proportional to g
    for tiny ones -- the closed form pins the placement of epsilon OUTSIDE the root (nerf.py:163-165, 455-458)."""
    from keras_nerf_amd.model.nerf.mlp import NeRFMLP
    from keras_nerf_amd.runtime import KnerfContext
    ctx2 = KnerfContext(white_background=True)
    w0 = []
    for net in (0, 1):
        m = NeRFMLP(seed=20 + net); m.build(); ctx2.set_weights(net, m.get_flat_weights()); w0.append(m.get_flat_weights())
    _, D = setup
    ctx2.train_chunk(D["o"], D["d"], D["t"], D["tgt"], D["u"])
    g = ctx2.grads_view().cpu().numpy().copy()
    ctx2.apply_adam()
    n = ctx2.param_count
    for net in (0, 1):
        gn = g[net * n:(net + 1) * n]
        dw = ctx2.get_weights(net) - w0[net]
        expect = -1e-3 * gn.astype(np.float64) / (np.abs(gn.astype(np.float64)) + 1e-7 / np.sqrt(1.0 - 0.999))
        assert (np.abs(gn) > 1e-6).mean() > 0.5
        np.testing.assert_allclose(dw, expect, rtol=2e-3, atol=2e-7)     # atol: fp32 spacing of the weights themselves
        assert np.all(dw[gn == 0] == 0)
        # with epsilon inside the root (the PyTorch form) small gradients would move by the full lr: rule that out
        small = (np.abs(gn) > 0) & (np.abs(gn) < 1e-6)
        if small.any():
            assert np.abs(dw[small]).max() < 0.5e-3
    ctx2.close()


def test_one_large_chunk_equals_eight_small_ones():
    """ray_chunks = 32768 (6.3 M fine samples, 60 GB of saved activations and dZ: every index is 64-bit) against the same
    rays in 8 chunks of 4096: images bit-identical, gradients equal up to fp32 summation order"""
    from keras_nerf_amd.data.utils import get_focal_from_fov, pose_spherical
    from keras_nerf_amd.model.nerf.mlp import NeRFMLP
    from keras_nerf_amd.runtime import KnerfContext
    ctx = KnerfContext(white_background=True)
    for net in (0, 1):
        m = NeRFMLP(seed=30 + net); m.build(); ctx.set_weights(net, m.get_flat_weights() * 1.5)
    poses = np.stack([pose_spherical(15.0 + 100.0 * i, -30.0, 4.0) for i in range(2)])
    o, d, t = ctx.generate_rays(poses, get_focal_from_fov(0.6911112070083618, 128), 128, 128, 2.0, 6.0, NC, None, seed=11)
    N = 2 * 128 * 128
    o, d, t = o.reshape(N, 3).contiguous(), d.reshape(N, 3).contiguous(), t.reshape(N, NC).contiguous()
    g = torch.Generator(device="cuda"); g.manual_seed(4)
    tgt = torch.rand((N, 3), device="cuda", generator=g); u = torch.rand((N, NF), device="cuda", generator=g)
    res = []
    for chunk in (4096, N):
        loss = torch.zeros(2, device="cuda")
        ci = torch.empty((N, 3), device="cuda"); fi = torch.empty((N, 3), device="cuda")
        ctx.zero_grads()
        ctx.train_batch(o, d, t, tgt, u, seed=0, ray_chunks=chunk, loss=loss, c_image=ci, f_image=fi)
        torch.cuda.synchronize()
        res.append((ci.clone(), fi.clone(), loss.clone(), ctx.grads_view().clone()))
    (c0, f0, l0, g0), (c1, f1, l1, g1) = res
    assert torch.equal(c0, c1) and torch.equal(f0, f1)
    assert float((l0 - l1).abs().max()) < 1e-5
    # the per-chunk factor 1/C enters dL/dimage before the bf16 rounding of dZ: relative 2^-9 per element, averaged out
    assert float((g0 - g1).abs().max()) <= 2e-3 * float(g0.abs().max())
    assert float(g0.abs().max()) > 0
    ctx.close()
