"""Builds the HIP libraries (gfx950 only) in-tree with hipcc.  Cross-compiles without a GPU.

libknerf_hip.so    the product: the C ABI of include/knerf.h
libknerf_probe.so  diagnostics for tests/ and tools/ (include/knerf_debug.h): layout-table introspection, workspace views,
                   hardware-fact and bandwidth probes.  Links against libknerf_hip.so; the product never loads it.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libknerf_hip.so")
PROBE_LIB = os.path.join(HERE, "libknerf_probe.so")
SOURCES = ["knerf_api.hip", "mlp_fwd.hip", "mlp_bwd.hip", "wgrad.hip", "generic.hip", "composite.hip", "sampler.hip", "optim.hip",
           "raygen.hip", "utils_ops.hip"]
# The three big kernels are templates on the trunk shape (csrc/layout.h KNERF_FUSED_SHAPES): each of these sources is compiled once
# per shape with -DKNERF_SHAPE_SLICE=<index> (that translation unit then defines the kernels of its shape only; slice 0 also holds
# the run-time dispatchers), so the shapes build in parallel and the default shape's object is what it was before the others existed.
SLICED = {"mlp_fwd.hip", "mlp_bwd.hip", "wgrad.hip"}
N_SHAPE_SLICES = 12     # = kNumFusedShapes (knerf_api.hip static_asserts it)
PROBE_SOURCES = ["debug_api.hip", "probe.hip"]
HEADERS = ["chain.h", "ctx.h", "kernels.h", "layout.h", "bwd_body.h", "wgrad_body.h", "generic.h", os.path.join("..", "..", "include", "knerf.h"),
           os.path.join("..", "..", "include", "knerf_debug.h")]
# -ffp-contract=off: the parity-critical fp32 arithmetic (ray points, sampler, compositing) must round like the
# reference's separate mul/add ops; fused multiply-adds are written explicitly (__builtin_fmaf) where wanted.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-ffp-contract=off"]


def _newer(a: str, b: str) -> bool:
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def _compile(hipcc, sources, objdir, flags, force, verbose):
    hdr_paths = [os.path.join(CSRC, h) for h in HEADERS]
    objs, procs = [], []
    jobs = []
    for src in sources:
        if src in SLICED:
            jobs += [(src, src.replace(".hip", f"_s{k}.o"), [f"-DKNERF_SHAPE_SLICE={k}", f"-DKNERF_N_SHAPE_SLICES={N_SHAPE_SLICES}"]) for k in range(N_SHAPE_SLICES)]
        else:
            jobs.append((src, src.replace(".hip", ".o"), [f"-DKNERF_N_SHAPE_SLICES={N_SHAPE_SLICES}"]))
    todo = []
    for src, obj, extra in jobs:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, obj)
        objs.append(o)
        if force or _newer(s, o) or any(_newer(h, o) for h in hdr_paths):
            todo.append((obj, [hipcc, *flags, *extra, "-c", s, "-o", o]))
    limit = max(1, min(len(todo), (os.cpu_count() or 8)))      # hipcc processes in flight
    running = []
    while todo or running:
        while todo and len(running) < limit:
            name, cmd = todo.pop(0)
            if verbose:
                print(" ".join(cmd), flush=True)
            running.append((name, subprocess.Popen(cmd)))
            procs.append(name)
        name, p = running.pop(0)
        if p.wait() != 0:
            for _, q in running:
                q.kill()
            raise RuntimeError(f"hipcc failed on {name}")
    return objs, bool(procs)


def build(force: bool = False, verbose: bool = True, defines=(), variant: str = "") -> str:
    """defines/variant: experimental builds (-DNAME=VALUE ...) into libknerf_hip_<variant>.so, used by tools/kbench.py"""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build" + ("_" + variant if variant else ""))
    lib = LIB if not variant else os.path.join(HERE, f"libknerf_hip_{variant}.so")
    probe = PROBE_LIB if not variant else os.path.join(HERE, f"libknerf_probe_{variant}.so")
    flags = FLAGS + ["-D" + d for d in defines]
    os.makedirs(objdir, exist_ok=True)
    objs, changed = _compile(hipcc, SOURCES, objdir, flags, force, verbose)
    pobjs, pchanged = _compile(hipcc, PROBE_SOURCES, objdir, flags, force, verbose)
    if changed or not os.path.exists(lib):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        pchanged = True
    if pchanged or not os.path.exists(probe):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", probe, *pobjs, "-L" + HERE, "-l:" + os.path.basename(lib),
               "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    defs = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    var = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--variant=")), "")
    build(force="--force" in sys.argv, defines=defs, variant=var)
