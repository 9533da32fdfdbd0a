"""A/B of build variants that must not change a single bit (the hand-counted fragment waits).

The chain kernels read their MFMA A fragments from LDS with inline-asm `ds_read_b128` that hipcc does not track and wait for them
with hand-counted `s_waitcnt lgkmcnt(N)` (chain.h Ring::frag_asm / frag_wait).  `-DKNERF_COMPILER_FRAGS` gives the same kernels
with compiler-tracked loads and compiler-inserted waits.  If the hand-counted scheme ever read a fragment early (a compiler update
that copies or re-materialises the destination registers between the asm read and its wait, a changed prefetch depth), the two
builds would differ somewhere in the 4096-ray chunk: every output, every saved activation block, every dZ block and -- in
deterministic mode -- the gradient itself are compared bit for bit.  The variant is built here (hipcc, ~1 min) when it is missing or
older than the sources."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _digests(lib=None):
    env = dict(os.environ)
    env.pop("KNERF_LIB", None)
    if lib:
        env["KNERF_LIB"] = lib
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "variant_digest.py")], capture_output=True, text=True, env=env, timeout=400)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def test_compiler_tracked_fragment_loads_give_identical_bits():
    from keras_nerf_amd import build as B
    lib = B.build(verbose=False, defines=("KNERF_COMPILER_FRAGS",), variant="cfrags")
    assert os.path.exists(lib)
    a, b = _digests(), _digests(lib)
    assert a == b, [k for k in a if a[k] != b[k]]
    assert len(a) >= 10


from tests.variant_shapes import XSHAPES      # noqa: E402  (also read by __graft_entry__.build)


def test_build_time_extra_shapes_run_on_the_fused_kernels():
    """`build.py --add-shape=NL,SK,U[,LX,LD]` (csrc/layout.h KNERF_EXTRA_SHAPES): a library built with two more width-128 trunks, four
    shapes with other positional-encoding depths (pos_emb_xyz 6 / 12 / 5 / 16, pos_emb_dir 2 / 4 / 1 / 3: four, six, four and eight
    encoding k-steps; h0 recomputed or saved), two width-64 trunks (two out tiles per layer: two waves across the weight-gradient
    kernel's output strips) and two shapes with pos_emb_dir 8 / 6 (51-d / 39-d direction encodings: four head k-steps) runs them on the fused kernels: host tables consistent, images / losses / gradients at
    the built-in shapes' tolerances against the oracle."""
    from keras_nerf_amd import build as B
    lib = B.build(verbose=False, variant="xshape", add_shapes=XSHAPES)
    env = dict(os.environ, KNERF_LIB=lib, KNERF_PROBE_LIB=lib.replace("libknerf_hip_", "libknerf_probe_"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "extra_shape_check.py")], capture_output=True, text=True, env=env, timeout=500)
    assert r.returncode == 0, r.stderr[-3000:]
    rows = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    want = [tuple(int(v) for v in x.split(",")) for x in XSHAPES]
    assert [tuple(x["shape"]) for x in rows] == [w if len(w) == 5 else w + (10, 4) for w in want]
    for x in rows:
        assert x["info"][:3] == x["shape"][:3] and x["info"][4:] == x["shape"][3:] and x["general_shape_path"] == 0.0, x
        assert x["coarse_worst"] < 1.5e-2 and x["fine_worst"] < 1.5e-2 and x["loss_err"] < 2e-3 and x["img_err"] < 1e-2, x
        # deterministic mode: same gradient up to the order of the sums, every tensor (a slab too small for the shape's largest job
        # table -- pos_emb_xyz 12 and 16 at width 256 -- showed as missing rows of the concat layer's gradient), bit-repeatable
        assert x["det_vs_atomic"] < 1e-5 and x["det_repeatable"] and x["det_coarse_worst"] < 1.5e-2 and x["det_fine_worst"] < 1.5e-2, x
    # the default library does not know them: same constructor arguments, general-shape kernels
    from keras_nerf_amd.runtime import KnerfContext
    for kw in (dict(n_layers=6, dense_units=128, skip_layer=3), dict(pos_emb_xyz=6, pos_emb_dir=2)):
        ctx = KnerfContext(white_background=True, **kw)
        assert ctx.get_option("general_shape_path") == 1.0
        ctx.close()


def test_a_trunk_that_ends_in_a_concat_in_the_weight_gradient_kernels_many_tile_regime():
    """Shape<9, 4, 256> (round 6: the head's weight-gradient job has eleven input tiles, the forward saves the enc blocks twice) at
    1,024 rays = 2,048 + 6,144 sample tiles, i.e. many tiles per workgroup in every job, list mode and contiguous mode, against the
    oracle in kernel arithmetic per gradient tensor (tools/oracle_check_chunk.py; the 4,096-ray run of the bench's launch sizes is
    profiles/r06_oracle_check_4096_rays_shape_9_4_256.json: 1.1e-3 coarse / 1.8e-3 fine)"""
    from keras_nerf_amd import build as B
    lib = B.build(verbose=False, variant="xshape", add_shapes=XSHAPES)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "oracle_check_chunk.py"), "--rays", "1024", "--sub", "256", "--shape", "9,4,256", "--lib", lib],
                       capture_output=True, text=True, timeout=400)
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert r.returncode == 0 and out["ok"] and out["general_shape_path"] == 0.0, out
    for mode in ("skip_dead_tiles_1", "skip_dead_tiles_0"):
        assert out[mode]["coarse_worst"] < 1.5e-2 and out[mode]["fine_worst"] < 1.5e-2 and max(out[mode]["loss_err"]) < 1e-5, out      # measured 2.4e-3 / 8.3e-3
    assert out["list_vs_contiguous_rel"] < 1e-5


AUTO_SHAPE = (6, 2, 128)      # coverable by the fused kernels, not in the built-in list; prebuilt as libknerf_hip_auto_6_2_128.so (build() of __graft_entry__)


def test_auto_build_puts_a_coverable_shape_on_the_fused_kernels_beside_the_product_library():
    """KNERF_AUTO_BUILD / KnerfContext(auto_build=True) (runtime.py): a shape the fused kernels could cover but the loaded library
    does not hold is compiled for them on first use (here: found built -- the variant travels with the tree, as the two above) and
    the context is re-created on THAT library, next to the product library in the same process: the shape leaves the general-shape
    kernels, its gradients meet the oracle at the built-in shapes' tolerance and the general-shape kernels' at theirs, and a
    default-shape context on the product library gives the same bits before and after."""
    import numpy as np
    import torch
    from oracle import nerf_oracle as O
    from keras_nerf_amd import _lib
    from keras_nerf_amd.runtime import KnerfContext
    from tests.problem import make_problem
    from tests.test_gpu_train import flat, per_tensor_err
    nl, sk, units = AUTO_SHAPE
    cfg = O.NerfConfig(n_layers=nl, dense_units=units, skip_layer=sk)
    P = make_problem(n_images=1, wh=16, weight_scale=1.5, bias_std=0.05, cfg=cfg)
    o, d, t, u, img = flat(P)
    Pd = make_problem(n_images=1, wh=16, weight_scale=1.5, bias_std=0.05)

    def default_shape_grads():
        ctx = KnerfContext(white_background=True, options=dict(deterministic=1))
        assert ctx.lib is _lib.load()
        ctx.set_weights(0, O.flatten_params(Pd["cp"])); ctx.set_weights(1, O.flatten_params(Pd["fp"]))
        loss = torch.zeros(2, device="cuda")
        ctx.train_chunk(*flat(Pd)[:3], flat(Pd)[4], flat(Pd)[3], inv_chunks=1.0, loss=loss)
        torch.cuda.synchronize()
        g = ctx.grads_view().clone(); ctx.close()
        return g

    before = default_shape_grads()
    res = {}
    for auto in (False, True):
        ctx = KnerfContext(n_layers=nl, dense_units=units, skip_layer=sk, white_background=True, auto_build=auto)
        assert ctx.get_option("general_shape_path") == (0.0 if auto else 1.0)
        assert (ctx.lib is _lib.load()) == (not auto)
        ctx.set_weights(0, O.flatten_params(P["cp"])); ctx.set_weights(1, O.flatten_params(P["fp"]))
        loss = torch.zeros(2, device="cuda")
        ctx.train_chunk(o, d, t, img, u, inv_chunks=1.0, loss=loss)
        torch.cuda.synchronize()
        res[auto] = (ctx.grads_view().cpu().numpy().copy(), loss.cpu().numpy().copy())
        if auto:
            ctx.apply_adam(); ctx.poll_nonfinite(wait=True)                  # the optimizer and the weight repack of that library too
            assert np.isfinite(ctx.get_weights(0)).all()
        ctx.close()
    n = res[True][0].size // 2
    rc, lc, gc = O.chunk_loss_and_grads(P["cp"], o, d, t, img, cfg, True, emulate_bf16=O.FUSED)
    assert per_tensor_err(res[True][0][:n], O.flatten_params(gc), cfg)[0] < 2e-2
    assert abs(float(res[True][1][0]) - float(lc)) < 2e-3
    # (coarse net only: the fine net's sample positions follow each implementation's own coarse weights)
    assert per_tensor_err(res[True][0][:n], res[False][0][:n], cfg)[0] < 4e-2
    assert np.abs(res[True][1] - res[False][1]).max() < 2e-3
    after = default_shape_grads()
    assert torch.equal(before.view(torch.int32), after.view(torch.int32))
