"""The weight-gradient path in the regime the bench runs, against the ORACLE directly.

The other oracle comparisons of gradients use <= 256 rays: 1,536 sample tiles for ~252 weight-gradient workgroups, i.e. most
workgroups see one tile or none, the staging ring never cycles and a list of live tiles is a handful of entries.  Here ONE
coarse + fine `train_chunk` of 1,024 rays = 2,048 + 6,144 tiles of 32 samples: every workgroup of a 256 x 256 job walks > 200 tiles
(all six staging slots cycling many times, the counted vmcnt waits in steady state, the scalar list look-ups inside the pipeline),
compared per tensor with `oracle.nerf_oracle.chunk_loss_and_grads` in the kernels' arithmetic at the tolerance of the small tests
(GRAD_TOL_EMU), for BOTH instantiations the product can run: list mode (skip_dead_tiles = 1, the default and what bench.py runs:
`wgrad_kernel<..., true>`, `mlp_bwd_kernel` with a tile list) and contiguous mode (skip_dead_tiles = 0).  The NumPy oracle needs
about 1 TFLOP and 5 GB for this (a minute or two on the GPU box's host cores).  Replaces reference
keras_nerf/model/nerf/nerf.py:361-417 (two tapes, 48 gradient tensors, accumulation)."""
import numpy as np
import pytest
import torch

from keras_nerf_amd.debug import debug_buffer
from oracle import nerf_oracle as O
from tests.problem import make_problem
from tests.test_gpu_forward import log_stats
from tests.test_gpu_train import flat, new_ctx, per_tensor_err

pytestmark = pytest.mark.gpu
GRAD_TOL_EMU = 1.5e-2          # = tests/test_gpu_configs.py


def test_1024_ray_chunk_list_and_contiguous_wgrad_meet_the_oracle():
    P = make_problem(n_images=1, wh=32, weight_scale=1.5, bias_std=0.05)
    cfg, N = P["cfg"], P["N"]
    assert N == 1024
    o, d, t, u, img = flat(P)
    # (random weights and targets: next to no tile is dead, so the list holds (nearly) all 8,192 tiles -- the bench's situation;
    # lists WITH holes are compared bit for bit with the non-skipping launch in tests/test_gpu_det_skip.py)
    got = {}
    for skip in (1, 0):
        ctx = new_ctx(P, options=dict(skip_dead_tiles=skip))
        loss = torch.zeros(2, device="cuda")
        ci = torch.empty((N, 3), device="cuda"); fi = torch.empty_like(ci)
        ctx.tile_stats(reset=True)
        ctx.train_chunk(o, d, t, img, u, inv_chunks=1.0, loss=loss, c_image=ci, f_image=fi)
        torch.cuda.synchronize()
        live, total = ctx.tile_stats(reset=True)
        assert (total == 2048 + 6144 and 0 < live <= total) if skip else total == 0       # list mode really ran over 8,192 tiles
        t_fine = debug_buffer(ctx, 5).view(torch.float32).cpu().numpy()[:N * 192].reshape(N, 192).copy()
        got[skip] = dict(g=ctx.grads_view().cpu().numpy().copy(), loss=loss.cpu().numpy().copy(), t_fine=t_fine, ci=ci.cpu().numpy(), fi=fi.cpu().numpy())
        n = ctx.param_count
        ctx.close()
    np.testing.assert_array_equal(got[0]["t_fine"], got[1]["t_fine"])                    # the sampler does not depend on the option
    np.testing.assert_array_equal(got[0]["ci"], got[1]["ci"])
    rc, lc, gc = O.chunk_loss_and_grads(P["cp"], o, d, t, img, cfg, True, emulate_bf16=O.FUSED)
    rf, lf, gf = O.chunk_loss_and_grads(P["fp"], o, d, got[1]["t_fine"], img, cfg, True, emulate_bf16=O.FUSED)
    gc, gf = O.flatten_params(gc), O.flatten_params(gf)
    for skip in (1, 0):
        g = got[skip]["g"]
        ec, ef = per_tensor_err(g[:n], gc, cfg), per_tensor_err(g[n:], gf, cfg)
        log_stats(f"wgrad_regime_1024_rays_skip_dead_tiles_{skip}", coarse_worst=ec[0], fine_worst=ef[0],  
                  loss_c=abs(float(got[skip]["loss"][0]) - float(lc)), loss_f=abs(float(got[skip]["loss"][1]) - float(lf)))
        assert ec[0] < GRAD_TOL_EMU, (skip, ec)
        assert ef[0] < GRAD_TOL_EMU, (skip, ef)
        assert abs(float(got[skip]["loss"][0]) - float(lc)) < 2e-3 and abs(float(got[skip]["loss"][1]) - float(lf)) < 2e-3
        np.testing.assert_allclose(got[skip]["ci"], rc["image"], atol=1e-2)
        np.testing.assert_allclose(got[skip]["fi"], rf["image"], atol=1e-2)
    # the two instantiations against each other: the same sums in another order of fp32 atomics
    rel = np.abs(got[0]["g"] - got[1]["g"]).max() / np.abs(got[1]["g"]).max()
    log_stats("wgrad_regime_list_vs_contiguous", rel=rel)
    assert rel < 1e-4


def test_4096_ray_chunk_the_bench_launch_sizes_meet_the_oracle():
    """The chunk bench.py times -- 4,096 rays: 8,192 coarse + 24,576 fine tiles per launch, ~900 tiles per weight-gradient workgroup of a
    256 x 256 job -- per gradient tensor against the oracle (tools/oracle_check_chunk.py: the oracle runs in sub-batches of 512 rays
    to bound its memory; ~1.5 min of host time on the GPU box).  Measured (round 4): 6.9e-4 coarse / 1.0e-3 fine of each tensor's
    max |g| in both list and contiguous mode, images 1.2e-3 / 1.7e-3."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "oracle_check_chunk.py"), "--rays", "4096", "--sub", "512"],
                       capture_output=True, text=True, timeout=500)
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert lines, r.stderr[-2000:]
    out = json.loads(lines[-1])
    assert r.returncode == 0 and out["ok"], out
    assert out["tiles"] == {"coarse": 8192, "fine": 24576}
    for skip in (1, 0):
        z = out[f"skip_dead_tiles_{skip}"]
        log_stats(f"wgrad_regime_4096_rays_skip_dead_tiles_{skip}", coarse_worst=z["coarse_worst"], fine_worst=z["fine_worst"])
        assert z["coarse_worst"] < GRAD_TOL_EMU and z["fine_worst"] < GRAD_TOL_EMU and max(z["loss_err"]) < 2e-3
    assert max(out["image_max_abs_err"]) < 1e-2 and out["list_vs_contiguous_rel"] < 1e-4
