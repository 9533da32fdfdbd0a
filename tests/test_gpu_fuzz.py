"""Seeded random configurations of the fused path against the oracle: ray counts that are not multiples of the 32-sample
tile or the 256-sample workgroup, odd sample counts, both backgrounds and out-of-range modes, several chunkings."""
import numpy as np
import pytest
from keras_nerf_amd.debug import debug_buffer
import torch

from oracle import nerf_oracle as O
from tests.problem import make_problem
from tests.test_gpu_forward import log_stats
from tests.test_gpu_train import per_tensor_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rays_per_chunk,chunks,group_max", [(37, 3, 4), (8, 5, 4), (129, 4, 4), (1, 6, 4), (64, 7, 3), (96, 4, 2)])
def test_grouped_train_batch_equals_chunk_by_chunk(rays_per_chunk, chunks, group_max):
    """knerf_train_batch (coarse weight gradients of a GROUP of chunks in one launch over several workspace regions) against the
    same chunks fed one by one through knerf_train_chunk, on ray counts that are multiples of nothing -- round 2's one-off
    tools/fuzz_more.py sweep as a test.  Equal up to the order of the fp32 atomics."""
    from keras_nerf_amd.runtime import KnerfContext
    cfg = O.NerfConfig()
    P = make_problem(n_images=3, wh=16, seed=5, weight_scale=1.5, bias_std=0.05, cfg=cfg)
    rc, C = rays_per_chunk, chunks
    R = rc * C
    o, d, t, img = (torch.as_tensor(P[k].reshape(P["N"], -1)[:R].copy(), device="cuda") for k in ("o", "d", "t", "img"))
    u = torch.as_tensor(P["u"].reshape(P["N"], -1)[:R].copy(), device="cuda")
    gs, ls = [], []
    for mode in ("batch", "chunks"):
        ctx = KnerfContext(white_background=True, options=dict(wgrad_group_max=group_max, merge_chunk_rays=0))
        ctx.set_weights(0, O.flatten_params(P["cp"])); ctx.set_weights(1, O.flatten_params(P["fp"]))
        ctx.zero_grads()
        loss = torch.zeros(2, device="cuda")
        if mode == "batch":
            ctx.train_batch(o, d, t, img, u, ray_chunks=rc, loss=loss)
            g0 = min(group_max, C); n_groups = -(-C // g0)
            assert ctx.get_option("wgrad_group") == -(-C // n_groups)          # the smallest group that needs no more launches
        else:
            for c in range(C):
                sl = slice(c * rc, (c + 1) * rc)
                ctx.train_chunk(o[sl].contiguous(), d[sl].contiguous(), t[sl].contiguous(), img[sl].contiguous(), u[sl].contiguous(),
                                ray_offset=c * rc, inv_chunks=1.0 / C, loss=loss)
        torch.cuda.synchronize()
        gs.append(ctx.grads_view().cpu().numpy().copy()); ls.append(loss.cpu().numpy().copy())
        ctx.close()
    n = gs[0].size // 2
    for sl in (slice(0, n), slice(n, 2 * n)):
        err = np.abs(gs[0][sl] - gs[1][sl]).max() / (np.abs(gs[1][sl]).max() + 1e-30)
        assert err < 1e-4, err
    np.testing.assert_allclose(ls[0], ls[1], atol=1e-6)


def _cases():
    rng = np.random.default_rng(20261003)
    out = []
    for i in range(8):
        nc = int(rng.integers(2, 97))
        nf = int(rng.integers(0, 130))
        out.append(dict(n_coarse=nc, n_fine=nf, rays=int(rng.integers(1, 200)), white=bool(rng.integers(0, 2)),
                        oob=["zero", "clamp"][int(rng.integers(0, 2))], seed=int(rng.integers(0, 1 << 30))))
    # round 3: the same with the run-time options on (sample counts that are multiples of 32 make skipping active)
    for i, (nc, nf) in enumerate(((64, 128), (32, 64), (96, 32), (33, 95))):
        out.append(dict(n_coarse=nc, n_fine=nf, rays=int(rng.integers(1, 200)), white=bool(i & 1), oob=["zero", "clamp"][i >> 1 & 1],
                        seed=int(rng.integers(0, 1 << 30)), det=int(i != 2), skip=1))
    # ... and on other built-in trunk shapes of the fused kernels (csrc/layout.h KNERF_FUSED_SHAPES): half width, several concats
    for i, (shape, nc, nf) in enumerate((((8, 4, 128), 33, 95), ((4, 2, 128), 64, 128), ((8, 2, 256), 17, 40), ((8, 4, 128), 96, 32))):
        out.append(dict(n_coarse=nc, n_fine=nf, rays=int(rng.integers(1, 200)), white=bool(i & 1), oob=["zero", "clamp"][i >> 1 & 1],
                        seed=int(rng.integers(0, 1 << 30)), det=int(i == 1), skip=int(i != 2), shape=shape))
    return out


@pytest.mark.parametrize("case", _cases(), ids=lambda c: f"nc{c['n_coarse']}_nf{c['n_fine']}_r{c['rays']}_{'w' if c['white'] else 'b'}_{c['oob']}" + ("_det" if c.get("det") else "") + ("_skip" if c.get("skip") else "") + ("_%dx%ds%d" % (c["shape"][0], c["shape"][2], c["shape"][1]) if c.get("shape") else ""))
def test_random_configuration_matches_oracle(case):
    from keras_nerf_amd.runtime import KnerfContext
    nl, sk, units = case.get("shape", (8, 4, 256))
    cfg = O.NerfConfig(n_coarse=case["n_coarse"], n_fine=case["n_fine"], n_layers=nl, skip_layer=sk, dense_units=units)
    P = make_problem(n_images=1, wh=16, seed=case["seed"], weight_scale=1.5, bias_std=0.05, cfg=cfg)
    R = case["rays"]
    o, d, t, img = (P[k].reshape(P["N"], -1)[:R].copy() for k in ("o", "d", "t", "img"))
    u = P["u"].reshape(P["N"], -1)[:R].copy() if case["n_fine"] else None
    ctx = KnerfContext(n_coarse=cfg.n_coarse, n_fine=cfg.n_fine, n_layers=nl, skip_layer=sk, dense_units=units, white_background=case["white"],
                       oob=case["oob"], options=dict(deterministic=case.get("det", 0), skip_dead_tiles=case.get("skip", 0)))
    assert ctx.get_option("general_shape_path") == 0.0
    if case.get("skip"):
        assert ctx.get_option("skip_dead_tiles_active") == float(cfg.n_coarse % 32 == 0 and (cfg.n_coarse + cfg.n_fine) % 32 == 0)
    ctx.set_weights(0, O.flatten_params(P["cp"])); ctx.set_weights(1, O.flatten_params(P["fp"]))
    ci, cd, cw = [x.cpu().numpy() for x in ctx.forward_chunk(0, o, d, t)]
    rc = O.predict_and_render_chunk_single(P["cp"], o, d, t, cfg, case["white"], emulate_bf16=O.FUSED)
    np.testing.assert_allclose(ci, rc["image"], atol=1e-2)
    np.testing.assert_allclose(cw, rc["weights"], atol=1e-2)
    if case["n_fine"] == 0:
        ctx.close(); return
    loss = torch.zeros(2, device="cuda")
    ctx.train_chunk(o, d, t, img, u, loss=loss)
    S = cfg.n_coarse + cfg.n_fine
    t_fine = debug_buffer(ctx, 5).view(torch.float32).cpu().numpy()[:R * S].reshape(R, S)
    w_c = debug_buffer(ctx, 6).view(torch.float32).cpu().numpy()[:R * cfg.n_coarse].reshape(R, cfg.n_coarse)
    np.testing.assert_array_equal(t_fine, O.fine_points(t, w_c, u, case["oob"]))          # sampler + sort: bit exact
    g = ctx.grads_view().cpu().numpy(); n = ctx.param_count
    _, lc, gc = O.chunk_loss_and_grads(P["cp"], o, d, t, img, cfg, case["white"], emulate_bf16=O.FUSED)
    _, lf, gf = O.chunk_loss_and_grads(P["fp"], o, d, t_fine, img, cfg, case["white"], emulate_bf16=O.FUSED)
    ec, ef = per_tensor_err(g[:n], O.flatten_params(gc), cfg), per_tensor_err(g[n:], O.flatten_params(gf), cfg)
    log_stats("fuzz_" + "_".join(f"{k}{v}" for k, v in case.items() if k != "seed"), coarse_worst=ec[0], fine_worst=ef[0])
    tol = 8e-2 if R * cfg.n_coarse < 2048 else 5e-2          # few samples: the bf16 roundings do not average out
    assert ec[0] < tol and ef[0] < tol, (ec, ef)
    assert abs(float(loss[0]) - float(lc)) < 2e-3 and abs(float(loss[1]) - float(lf)) < 2e-3
    ctx.close()


def test_more_chunks_per_batch_than_the_counter_ring_held():
    """every list of live tiles takes a counter from a ring that held 4,096 of them and was re-zeroed on wrap-around -- also
    while a GROUP's counter was still being appended to by later coarse passes, which would have dropped live tiles from that
    group's coarse weight gradients once a batch had more than ~1,800 chunks.  The ring is now sized per call (2 C + C / G + 8,
    grow-only).  2,200 chunks of 32 rays (train_single.py accepts any --ray_chunks that divides the batch): skipping on = skipping off
    up to the order of the fp32 atomics, on weights that have dead tiles (sigma's bias lowered)."""
    from keras_nerf_amd.runtime import KnerfContext
    cfg = O.NerfConfig()
    P = make_problem(n_images=1, wh=16, seed=9, weight_scale=1.5, bias_std=0.05, cfg=cfg)
    names = [n for n, _, _ in O.layer_shapes(cfg)]
    for params in (P["cp"], P["fp"]):
        params[2 * names.index("sigma") + 1][:] = -0.35              # closes sigma's gate on part of the samples: dead tiles exist
    C, rc = 2200, 32
    g = torch.Generator(device="cuda").manual_seed(3)
    idx = torch.randint(0, P["N"], (C * rc,), device="cuda", generator=g)
    o, d, t, img, u = (torch.as_tensor(P[k].reshape(P["N"], -1), device="cuda")[idx].contiguous() for k in ("o", "d", "t", "img", "u"))
    res = {}
    for skip in (1, 0):
        ctx = KnerfContext(white_background=True, options=dict(skip_dead_tiles=skip, merge_chunk_rays=0))      # 2,200 chunks of 32 rays AS chunks (the counter ring)
        ctx.set_weights(0, O.flatten_params(P["cp"])); ctx.set_weights(1, O.flatten_params(P["fp"]))
        loss = torch.zeros(2, device="cuda")
        ctx.train_batch(o, d, t, img, u, ray_chunks=rc, loss=loss)
        torch.cuda.synchronize()
        live, total = ctx.tile_stats()
        res[skip] = (ctx.grads_view().clone(), loss.clone(), live, total)
        assert ctx.get_option("wgrad_group") == 4
        ctx.close()
    assert res[1][3] == C * rc * (64 + 192) // 32 and 0 < res[1][2] < res[1][3]          # every pass was listed; some tiles are dead
    n = res[0][0].numel() // 2
    for sl in (slice(0, n), slice(n, 2 * n)):
        rel = float((res[1][0][sl] - res[0][0][sl]).abs().max() / res[0][0][sl].abs().max())
        assert rel < 1e-4, rel                                        # a zeroed group counter showed as missing coarse contributions (percent level)
    assert torch.allclose(res[1][1], res[0][1], rtol=1e-5)
    log_stats("counter_ring_2200_chunks", dead=1.0 - res[1][2] / res[1][3])
