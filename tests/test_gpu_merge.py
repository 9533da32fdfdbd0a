"""Option merge_chunk_rays (include/knerf.h, round 5): knerf_train_batch / knerf_render_batch run consecutive chunks as one set of
launches of up to 4,096 rays.  `ray_chunks` is the reference's memory knob (nerf.py:100, 332-473): the result must not depend on it
beyond the order of fp32 sums -- rendered outputs bit-identical, gradients and losses equal to the chunk-by-chunk launches."""
import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from tests.problem import make_problem
from tests.test_gpu_forward import log_stats

pytestmark = pytest.mark.gpu


def _ctx(P, **options):
    from keras_nerf_amd.runtime import KnerfContext
    ctx = KnerfContext(white_background=True, options=options)
    ctx.set_weights(0, O.flatten_params(P["cp"])); ctx.set_weights(1, O.flatten_params(P["fp"]))
    return ctx


def _data(P, n):
    N = P["N"]
    return tuple(torch.as_tensor(P[k].reshape(N, -1)[:n].copy(), device="cuda") for k in ("o", "d", "t", "img", "u"))


def test_default_is_4096_rays_and_the_option_round_trips():
    P = make_problem(n_images=1, wh=16)
    ctx = _ctx(P)
    assert ctx.get_option("merge_chunk_rays") == 4096.0 and ctx.get_option("merge_render_rays") == 65536.0
    ctx.set_option("merge_chunk_rays", 0)
    assert ctx.get_option("merge_chunk_rays") == 0.0
    with pytest.raises(ValueError):
        ctx.set_option("merge_chunk_rays", -1)
    ctx.close()


@pytest.mark.parametrize("given_u", [True, False])
def test_rendered_outputs_do_not_depend_on_the_merge(given_u):
    """per-ray work only: bit-identical, with injected u and with the library's own Philox numbers (keyed by the ray's index in the batch)"""
    P = make_problem(n_images=3, wh=16, weight_scale=1.5, bias_std=0.05)
    o, d, t, _, u = _data(P, 768)
    outs = []
    for merge in (0, 65536, 256):                   # every chunk its own launches / the default (here: all twelve chunks at once) / four chunks per launch
        ctx = _ctx(P, merge_render_rays=merge)
        assert ctx.get_option("merge_render_rays") == merge
        outs.append(ctx.render_batch(o, d, t, u if given_u else None, seed=11, ray_chunks=64))
        torch.cuda.synchronize()
        ctx.close()
    for other in outs[1:]:
        for k, v in outs[0].items():
            assert torch.equal(v, other[k]), k


@pytest.mark.parametrize("ray_chunks,n_rays,deterministic", [(64, 768, 0), (96, 480, 0), (32, 768, 1), (128, 768, 1)])
def test_merged_train_batch_equals_the_chunk_by_chunk_launches(ray_chunks, n_rays, deterministic):
    """12 / 5 / 24 / 6 chunks as one set of launches (768 and 480 rays fit 4,096) against the same chunks as their own launches and
    against an intermediate merge (256 rays: 4 / 1 / 8 / 2 chunks per launch): losses and every gradient tensor equal up to the order of
    fp32 sums; the per-ray images the training pass returns bit-identical."""
    P = make_problem(n_images=3, wh=16, weight_scale=1.5, bias_std=0.05)
    o, d, t, img, u = _data(P, n_rays)
    res = {}
    for merge in (0, 4096, 256):
        ctx = _ctx(P, merge_chunk_rays=merge, deterministic=deterministic)
        loss = torch.zeros(2, device="cuda"); ci = torch.empty((n_rays, 3), device="cuda"); fi = torch.empty((n_rays, 3), device="cuda")
        ctx.zero_grads()
        ctx.train_batch(o, d, t, img, u, ray_chunks=ray_chunks, loss=loss, c_image=ci, f_image=fi)
        torch.cuda.synchronize()
        res[merge] = (ctx.grads_view().cpu().numpy().copy(), loss.cpu().numpy().copy(), ci.clone(), fi.clone())
        if deterministic:                                   # a merged deterministic step is still bit-reproducible
            ctx.zero_grads(); loss2 = torch.zeros(2, device="cuda")
            ctx.train_batch(o, d, t, img, u, ray_chunks=ray_chunks, loss=loss2, c_image=ci, f_image=fi)
            torch.cuda.synchronize()
            assert np.array_equal(ctx.grads_view().cpu().numpy().view(np.int32), res[merge][0].view(np.int32))
            assert np.array_equal(loss2.cpu().numpy().view(np.int32), res[merge][1].view(np.int32))
        ctx.close()
    n = res[0][0].size // 2
    cfg = O.NerfConfig()
    for merge in (4096, 256):
        assert torch.equal(res[merge][2], res[0][2]) and torch.equal(res[merge][3], res[0][3])
        np.testing.assert_allclose(res[merge][1], res[0][1], rtol=2e-6)
        worst = 0.0
        for net in range(2):
            a = O.unflatten_params(res[merge][0][net * n:(net + 1) * n], cfg)
            b = O.unflatten_params(res[0][0][net * n:(net + 1) * n], cfg)
            for x, y in zip(a, b):
                worst = max(worst, float(np.abs(x - y).max() / max(np.abs(y).max(), 1e-30)))
        log_stats(f"merge{merge}_vs_chunks_rc{ray_chunks}_det{deterministic}", rel_err=worst)
        assert worst < 2e-5, (merge, worst)


def test_eager_diagnostics_keep_the_callers_chunks():
    """grad_diagnostics counts the LAST chunk's gradient (nerf.py:430-451): merging is off while it is on -- the counts equal those of a
    context that never merges"""
    P = make_problem(n_images=1, wh=16, weight_scale=1.5, bias_std=0.05)
    o, d, t, img, u = _data(P, 256)
    counts = []
    for merge in (4096, 0):
        ctx = _ctx(P, merge_chunk_rays=merge, grad_diagnostics=1)
        loss = torch.zeros(2, device="cuda")
        ctx.zero_grads()
        ctx.train_batch(o, d, t, img, u, ray_chunks=64, loss=loss)
        counts.append(tuple(ctx.grad_diagnostics(wait=True)[:2]))
        ctx.close()
    assert counts[0] == counts[1] and counts[0][0] > 0


def test_merged_launches_that_do_not_fit_fall_back_to_the_callers_chunks_and_the_step_succeeds():
    """`ray_chunks` is the reference's memory knob (nerf.py:95-100).  When the merged workspace cannot be allocated
    knerf_train_batch retries with the caller's own chunk size -- and the failed hipMalloc must not survive as the runtime's last
    error (every launch helper ends in hipGetLastError(): the first kernel after it would report out-of-memory although the retry
    succeeded).  Forced here by option workspace_limit_gb, which answers an over-limit request with a REAL failing hipMalloc: the
    step succeeds, equals the never-merged step bit for bit (deterministic mode), the failure is remembered (one fall-back for two
    steps), and the render path does the same."""
    P = make_problem(n_images=3, wh=16, weight_scale=1.5, bias_std=0.05)
    o, d, t, img, u = _data(P, 768)
    res = {}
    for name, opts in (("plain", dict(merge_chunk_rays=0, merge_render_rays=0)), ("fallback", dict(workspace_limit_gb=0.4))):
        ctx = _ctx(P, deterministic=1, **opts)
        outs = []
        for _ in range(2):
            loss = torch.zeros(2, device="cuda"); ci = torch.empty((768, 3), device="cuda"); fi = torch.empty((768, 3), device="cuda")
            ctx.zero_grads()
            ctx.train_batch(o, d, t, img, u, ray_chunks=64, loss=loss, c_image=ci, f_image=fi)      # 768 rays merged: ~1.5 GB; 64 rays: ~0.13 GB
            torch.cuda.synchronize()
            outs.append((ctx.grads_view().cpu().numpy().copy(), loss.cpu().numpy().copy(), ci.clone(), fi.clone()))
        if name == "fallback":
            assert ctx.get_option("merge_fallbacks") == 1.0                   # the second step did not try the merged size again
            assert ctx.get_option("merge_chunk_rays") == 4096.0
        if name == "fallback":                                                # rendering keeps no saved tensors: 768 rays are 3.2 MB, 64 rays 0.3 MB
            ctx.set_option("workspace_limit_gb", 0.002)
        rend = ctx.render_batch(o, d, t, u, seed=11, ray_chunks=64)
        rend2 = ctx.render_batch(o, d, t, u, seed=11, ray_chunks=64)
        torch.cuda.synchronize()
        if name == "fallback":
            assert ctx.get_option("merge_fallbacks") == 2.0                   # + ONE for the two renders
        res[name] = (outs, rend, rend2)
        ctx.close()
    for k in range(2):
        a, b = res["plain"][0][k], res["fallback"][0][k]
        assert np.array_equal(a[0].view(np.int32), b[0].view(np.int32)) and np.array_equal(a[1].view(np.int32), b[1].view(np.int32))
        assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    for key, v in res["plain"][1].items():
        assert torch.equal(v, res["fallback"][1][key]) and torch.equal(v, res["fallback"][2][key]), key
