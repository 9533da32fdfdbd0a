"""Deterministic reduction mode and exact dead-tile skipping (knerf_set_option, include/knerf.h).

deterministic: the weight-gradient kernel writes per-workgroup partial sums and an ordered second pass adds them (no fp32
atomics; the chunk losses likewise), so two runs of the same train step are BIT-identical -- gradients, losses, weights after
Adam -- and agree with the default atomic mode up to the order of the sums.

skip_dead_tiles: a sample whose dL/d(rgb, sigma) is exactly zero (sigma's ReLU gate closed so that alpha = w = 0 and no
gradient reaches the pre-activation -- reference utils.py:36-45, mlp.py:40 -- or a pixel error of exactly 0) adds exactly
nothing to any gradient, so the backward may drop whole 32-sample tiles of such samples.  Checked on a scene TRAINED here for
a few hundred steps through the HIP path (the procedural scene of tests/procedural_scene.py; the dead fraction is logged and
must be substantial): in deterministic mode, where both launches form the same sums in the same order, the gradients with and
without skipping are bit-identical; in the default mode they agree up to the atomics' order."""
import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from tests.problem import make_problem
from tests.test_gpu_forward import log_stats

pytestmark = pytest.mark.gpu


def _ctx(P, **options):
    from keras_nerf_amd.runtime import KnerfContext
    options.setdefault("merge_chunk_rays", 0)       # these tests are about the caller's chunks as launches (grouping, odd sizes); merging: test_gpu_merge.py
    ctx = KnerfContext(white_background=True, options=options)
    ctx.set_weights(0, O.flatten_params(P["cp"])); ctx.set_weights(1, O.flatten_params(P["fp"]))
    return ctx


def _batch(P, n):
    N = P["N"]
    f = lambda k: torch.as_tensor(P[k].reshape(N, -1)[:n].copy(), device="cuda")
    return f("o"), f("d"), f("t"), f("img"), f("u")


def _step(ctx, data, ray_chunks):
    o, d, t, img, u = data
    loss = torch.zeros(2, device="cuda")
    ctx.zero_grads()
    ctx.train_batch(o, d, t, img, u, ray_chunks=ray_chunks, loss=loss)
    torch.cuda.synchronize()
    return ctx.grads_view().clone(), loss.clone()


@pytest.mark.parametrize("ray_chunks,n_rays", [(128, 768), (96, 480), (768, 768)])
def test_two_deterministic_train_steps_are_bit_identical(ray_chunks, n_rays):
    """6 chunks (grouped coarse weight-gradient launches of 3), 5 odd-sized chunks, one chunk"""
    P = make_problem(n_images=3, wh=16, weight_scale=1.5, bias_std=0.05)
    data = _batch(P, n_rays)
    runs = []
    for rep in range(2):
        ctx = _ctx(P, deterministic=1)
        assert ctx.get_option("deterministic") == 1.0
        g, loss = _step(ctx, data, ray_chunks)
        g2, loss2 = _step(ctx, data, ray_chunks)                      # same context, second launch: same bits
        assert torch.equal(g.view(torch.int32), g2.view(torch.int32)) and torch.equal(loss.view(torch.int32), loss2.view(torch.int32))
        ctx.apply_adam()
        w = np.concatenate([ctx.get_weights(0), ctx.get_weights(1)])
        runs.append((g, loss, w))
        ctx.close()
    (g0, l0, w0), (g1, l1, w1) = runs
    assert torch.equal(g0.view(torch.int32), g1.view(torch.int32)), float((g0 - g1).abs().max())
    assert torch.equal(l0.view(torch.int32), l1.view(torch.int32))
    assert np.array_equal(w0.view(np.int32), w1.view(np.int32))
    assert float(g0.abs().max()) > 0
    # against the default (atomic) mode: the same sums in another order
    ctx = _ctx(P, deterministic=0)
    ga, la = _step(ctx, data, ray_chunks)
    ctx.close()
    n = g0.numel() // 2
    for k, sl in (("coarse", slice(0, n)), ("fine", slice(n, 2 * n))):
        err = float((ga[sl] - g0[sl]).abs().max() / g0[sl].abs().max())
        log_stats(f"deterministic_vs_atomic_{k}_rc{ray_chunks}", rel_err=err)
        assert err < 2e-5, (k, err)
    assert float((la - l0).abs().max()) < 1e-6


def _train_procedural(steps, wh=32, batch=2, chunk=1024):
    """a NeRF trained for `steps` steps on the small procedural scene (HIP path, skipping on); returns it with the scene"""
    from keras_nerf_amd.model.nerf.nerf import NeRF
    from keras_nerf_amd.runtime import KnerfContext
    from tests.procedural_scene import make_scene
    c0 = KnerfContext(white_background=True)
    o, d, t, img = make_scene(c0, wh=wh, n_views=24, scale=1.6)
    c0.close()
    nerf = NeRF(seed=0)
    nerf.compile({"learning_rate": 5e-4}, "mse", batch_size=batch, image_height=wh, image_width=wh, ray_chunks=chunk, white_background=True,
                 skip_dead_tiles=True)
    order = np.random.default_rng(5).integers(0, 20, (steps, batch))
    fracs = []
    for s in range(steps):
        idx = torch.as_tensor(order[s], device="cuda")
        nerf.train_step((img[idx], (o[idx], d[idx], t[idx])), with_metrics=False)
        if (s + 1) % 100 == 0:
            live, total = nerf._ctx.tile_stats(reset=True)
            fracs.append(1.0 - live / max(total, 1))
    nerf._ctx.poll_nonfinite(wait=True)
    return nerf, (o, d, t, img), fracs


_TRAINED = {}


@pytest.mark.parametrize("sigma_bias_shift,min_dead", [(0.0, 0.005), (0.5, 0.05), (3.0, 0.12)])
def test_dead_tile_skipping_is_exact_on_a_trained_scene(sigma_bias_shift, min_dead):
    """the checkpoint as trained (a few per cent of the tiles are dead after 300 small steps; tools/convergence128.py --skip-dead
    measures the fraction over a real run, profiles/), and the same checkpoint with the sigma bias of both nets lowered by 0.5 and
    by 3, which empties the low-density regions -- the heavier-skipping paths (workgroups with few or no tiles, unbalanced ranges)
    with the real weights' structure (measured r03: 2.3 % / 9.0 % dead for shifts 0 / 0.5)"""
    from keras_nerf_amd.runtime import KnerfContext
    wh, batch, chunk = 32, 2, 512          # 4 chunks: grouped coarse launches too
    if not _TRAINED:
        nerf, scene, fracs = _train_procedural(300, wh=wh, batch=batch, chunk=1024)
        log_stats("dead_tile_frac_while_training_32x32", **{f"step{100 * (i + 1)}": f for i, f in enumerate(fracs)})
        _TRAINED.update(wc=nerf.coarse.get_flat_weights(), wf=nerf.fine.get_flat_weights(), scene=scene)
    o, d, t, img = _TRAINED["scene"]
    wc, wf = _TRAINED["wc"].copy(), _TRAINED["wf"].copy()
    sig_bias = sum(fi * fo + fo for _, fi, fo in O.layer_shapes(O.NerfConfig())[:8]) + 256      # index of sigma/bias in the flat vector
    assert O.layer_shapes(O.NerfConfig())[8][0] == "sigma"
    wc[sig_bias] -= sigma_bias_shift; wf[sig_bias] -= sigma_bias_shift
    N = batch * wh * wh
    data = (o[20:22].reshape(N, 3).contiguous(), d[20:22].reshape(N, 3).contiguous(), t[20:22].reshape(N, 64).contiguous(),
            img[20:22].reshape(N, 3).contiguous(), torch.rand((N, 128), device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)))
    res = {}
    for det in (1, 0):
        for skip in (0, 1):
            ctx = KnerfContext(white_background=True, lr=5e-4, options=dict(deterministic=det, skip_dead_tiles=skip))
            ctx.set_weights(0, wc); ctx.set_weights(1, wf)
            assert ctx.get_option("skip_dead_tiles_active") == float(skip)
            g, loss = _step(ctx, data, chunk)
            live, total = ctx.tile_stats()
            res[det, skip] = (g, loss, live, total)
            ctx.close()
    _, _, live, total = res[1, 1]
    assert total == N * 256 // 32 and res[1, 0][3] == 0            # both passes of every chunk counted; nothing counted with skipping off
    dead = 1.0 - live / total
    log_stats(f"dead_tile_frac_trained_32x32_step300_shift{sigma_bias_shift}", dead=dead, live=live, total=total)
    assert min_dead < dead < 0.999, dead                            # there are dead tiles to skip and live tiles left
    # deterministic mode: bit-identical with and without skipping
    g_ns, l_ns = res[1, 0][:2]; g_s, l_s = res[1, 1][:2]
    assert float(g_ns.abs().max()) > 0
    assert torch.equal(g_ns.view(torch.int32), g_s.view(torch.int32)), float((g_ns - g_s).abs().max())
    assert torch.equal(l_ns.view(torch.int32), l_s.view(torch.int32))
    # default mode (balanced split of the live tiles, atomics): equal up to the order of the sums
    n = g_ns.numel() // 2
    for sl in (slice(0, n), slice(n, 2 * n)):
        for g in (res[0, 0][0], res[0, 1][0]):
            assert float((g[sl] - g_ns[sl]).abs().max()) <= 2e-5 * float(g_ns[sl].abs().max())


def test_skipping_changes_nothing_when_nothing_is_dead_and_handles_all_dead():
    """random initial weights: (almost) every tile is live; a target equal to the render everywhere: every tile is dead and the
    gradient is exactly zero"""
    P = make_problem(n_images=2, wh=16, weight_scale=1.5, bias_std=0.05)
    data = _batch(P, 512)
    ctx = _ctx(P, deterministic=1, skip_dead_tiles=1)
    g, _ = _step(ctx, data, 256)
    live, total = ctx.tile_stats()
    ctx.close()
    ctx = _ctx(P, deterministic=1, skip_dead_tiles=0)
    g0, _ = _step(ctx, data, 256)
    ctx.close()
    assert total == 512 * 256 // 32 and live > 0.9 * total
    assert torch.equal(g.view(torch.int32), g0.view(torch.int32))
    # all dead: sigma's bias far below zero -> sigma = 0 everywhere -> white image; target white -> pixel error exactly 0
    cp = [p.copy() for p in P["cp"]]; fp = [p.copy() for p in P["fp"]]
    names = [n for n, _, _ in O.layer_shapes(P["cfg"])]
    si = 2 * names.index("sigma") + 1
    cp[si][:] = -1e3; fp[si][:] = -1e3
    o, d, t, img, u = data
    white = torch.ones_like(img)
    for skip in (0, 1):
        from keras_nerf_amd.runtime import KnerfContext
        ctx = KnerfContext(white_background=True, options=dict(skip_dead_tiles=skip))
        ctx.set_weights(0, O.flatten_params(cp)); ctx.set_weights(1, O.flatten_params(fp))
        g, loss = _step(ctx, (o, d, t, white, u), 256)
        assert float(g.abs().max()) == 0.0 and float(loss.abs().max()) == 0.0
        if skip:
            assert ctx.tile_stats() == (0, 512 * 256 // 32)
        ctx.close()


@pytest.mark.parametrize("shape", [dict(force_generic=True),                                              # the default shape on the general-shape kernels
                                   dict(n_layers=5, dense_units=192, skip_layer=2, pos_emb_xyz=6, pos_emb_dir=2, pad_width=False)])     # a shape only they cover
def test_dead_tile_skipping_on_the_general_shape_path_is_exact(shape):
    """Round 5: the general-shape kernels (csrc/generic.hip) walk the same list of live 32-sample tiles in every dgrad GEMM and every
    weight-gradient product.  A problem with a substantial dead fraction (sigma's bias lowered so that the ReLU on sigma is closed in
    most of space; a quarter of the rays get their own rendered pixel as target, i.e. exactly zero pixel error): in deterministic
    mode the gradients with and without skipping are BIT-identical (same sums, same order: flags + ordered compaction), in the
    default mode equal up to the atomics' order; the statistics count what the fused path counts."""
    from keras_nerf_amd.model.nerf.mlp import NeRFMLP
    from keras_nerf_amd.runtime import KnerfContext
    kw = {k: v for k, v in shape.items() if k != "force_generic"}
    nl, units, skip_l = kw.get("n_layers", 8), kw.get("dense_units", 256), kw.get("skip_layer", 4)
    lx, ld = kw.get("pos_emb_xyz", 10), kw.get("pos_emb_dir", 4)
    ws = []
    for net in (0, 1):
        m = NeRFMLP(nl, units, skip_l, seed=40 + net, xyz_dim=3 + 6 * lx, dir_dim=3 + 6 * ld)
        w = m.get_flat_weights() * 1.5
        names = [n for n, _, _ in m._shapes]
        off = sum(fi * fo + fo for _, fi, fo in m._shapes[:names.index("sigma")]) + m._shapes[names.index("sigma")][1]
        w[off] -= 0.6                                                     # sigma/bias: the gate on sigma closes in the low-density regions
        ws.append(w)
    P = make_problem(n_images=2, wh=16, weight_scale=1.5, bias_std=0.05)
    o, d, t, img, u = _batch(P, 512)
    res = {}
    for det in (1, 0):
        for skip in (0, 1):
            ctx = KnerfContext(white_background=True, force_generic=shape.get("force_generic"), options=dict(deterministic=det, skip_dead_tiles=skip), **kw)
            assert ctx.get_option("general_shape_path") == 1.0 and ctx.get_option("skip_dead_tiles_active") == float(skip)
            ctx.set_weights(0, ws[0]); ctx.set_weights(1, ws[1])
            if "tgt" not in res:          # every fourth ray: the coarse net's own pixel as target (the coarse pass of those rays is dead)
                ren = ctx.render_batch(o, d, t, u, ray_chunks=128)
                tgt = img.clone(); tgt[::4] = ren["c_image"][::4]
                res["tgt"] = tgt
            g, loss = _step(ctx, (o, d, t, res["tgt"], u), 128)
            res[det, skip] = (g, loss, ctx.tile_stats())
            ctx.close()
    live, total = res[1, 1][2]
    assert total == 512 * 256 // 32 and res[1, 0][2] == (0, 0) and res[0, 1][2] == (live, total)
    dead = 1.0 - live / total
    log_stats(f"generic_dead_tile_frac_{'default_shape' if shape.get('force_generic') else 'l5_u192'}", dead=dead, live=live, total=total)
    assert 0.05 < dead < 0.95, dead
    g_ns, l_ns, _ = res[1, 0]; g_s, l_s, _ = res[1, 1]
    assert float(g_ns.abs().max()) > 0 and bool(torch.isfinite(g_ns).all())
    assert torch.equal(g_ns.view(torch.int32), g_s.view(torch.int32)), float((g_ns - g_s).abs().max())
    assert torch.equal(l_ns.view(torch.int32), l_s.view(torch.int32))
    n = g_ns.numel() // 2
    for sl in (slice(0, n), slice(n, 2 * n)):
        for g in (res[0, 0][0], res[0, 1][0]):
            assert float((g[sl] - g_ns[sl]).abs().max()) <= 2e-5 * float(g_ns[sl].abs().max())
