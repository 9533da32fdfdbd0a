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
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "variant_digest.py")], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def test_compiler_tracked_fragment_loads_give_identical_bits():
    from keras_nerf_amd import build as B
    lib = B.build(verbose=False, defines=("KNERF_COMPILER_FRAGS",), variant="cfrags")
    assert os.path.exists(lib)
    a, b = _digests(), _digests(lib)
    assert a == b, [k for k in a if a[k] != b[k]]
    assert len(a) >= 10


def test_build_time_extra_shapes_run_on_the_fused_kernels():
    """`build.py --add-shape=NL,SK,U` (csrc/layout.h KNERF_EXTRA_SHAPES): a library built with 6/3/128 and 8/2/128 beside the built-in
    list runs NeRF(n_layers=6, skip_layer=3, dense_units=128) and 8/2/128 on the fused kernels, at the built-in shapes' tolerances."""
    from keras_nerf_amd import build as B
    lib = B.build(verbose=False, variant="xshape", add_shapes=["6,3,128", "8,2,128"])
    env = dict(os.environ, KNERF_LIB=lib, KNERF_PROBE_LIB=lib.replace("libknerf_hip_", "libknerf_probe_"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "extra_shape_check.py")], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert [tuple(x["shape"]) for x in rows] == [(6, 3, 128), (8, 2, 128)]
    for x in rows:
        assert x["info"][:3] == x["shape"] and x["general_shape_path"] == 0.0, x
        assert x["coarse_worst"] < 1.5e-2 and x["fine_worst"] < 1.5e-2 and x["loss_err"] < 2e-3 and x["img_err"] < 1e-2, x
    # the default library does not know them: same constructor arguments, general-shape kernels
    from keras_nerf_amd.runtime import KnerfContext
    ctx = KnerfContext(n_layers=6, dense_units=128, skip_layer=3, white_background=True)
    assert ctx.get_option("general_shape_path") == 1.0
    ctx.close()
