"""Builds the HIP libraries (gfx950 only) in-tree with hipcc.  Cross-compiles without a GPU.

libknerf_hip.so    the product: the C ABI of include/knerf.h
libknerf_probe.so  diagnostics for tests/ and tools/ (include/knerf_debug.h): layout-table introspection, workspace views,
                   hardware-fact and bandwidth probes.  Links against libknerf_hip.so; the product never loads it.

    python keras_nerf_amd/build.py [--force] [-DNAME[=VALUE] ...] [--variant=NAME] [--add-shape=NL,SK,U[,LX,LD] ...]

--add-shape (or KNERF_ADD_SHAPES="NL,SK,U;NL,SK,U,LX,LD" in the environment): further NeRF(n_layers, skip_layer, dense_units [, pos_emb_xyz,
pos_emb_dir]) shapes for the fused kernels beside the built-in list of csrc/layout.h (dense_units 256, 128 or 64; encodings: the reference's
10 / 4 unless given, pos_emb_dir <= 8; since round 6 also trunks that end in a concat, e.g. 9,4,256); three more hipcc runs and about 1 MB of library each.  Shapes not in the list still work: they run on the general-shape kernels.
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
N_BUILTIN_SHAPES = 14   # = kNumBuiltinShapes (csrc/layout.h static_asserts it); knerf_api.hip checks the total against KNERF_N_SHAPE_SLICES
MAX_EXTRA_SHAPES = 36   # a build-time budget, not a limit of csrc/layout.h (one more instantiation of the three big kernels each)
PROBE_SOURCES = ["debug_api.hip", "probe.hip"]
# the extra trunk shapes of the `xshape` build variant (tests/test_gpu_variants.py builds and checks it; __graft_entry__.build() keeps an
# existing one up to date): round 4 added width 64 (8/4 and 4/2 are built in) and pos_emb_dir 5..8 (four head k-steps)
# round 6: a concat behind the LAST trunk layer ((n_layers - 1) % skip_layer == 0): the head takes [h ; xyz_enc ; dir_enc]
XSHAPES = ["6,3,128", "8,2,128", "8,4,256,6,2", "8,4,256,12,4", "8,4,128,5,1", "4,2,256,16,3",
           "6,3,64", "8,4,64,6,2", "8,4,256,10,8", "8,4,128,10,6", "9,4,256", "5,2,128", "5,4,64,6,2"]
HEADERS = ["chain.h", "ctx.h", "kernels.h", "layout.h", "bwd_body.h", "wgrad_body.h", "generic.h", os.path.join("..", "..", "include", "knerf.h"),
           os.path.join("..", "..", "include", "knerf_debug.h")]
# -ffp-contract=off: the parity-critical fp32 arithmetic (ray points, sampler, compositing) must round like the
# reference's separate mul/add ops; fused multiply-adds are written explicitly (__builtin_fmaf) where wanted.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-ffp-contract=off"]


# what the three big kernels are compiled from (hipcc's own dependency files of mlp_fwd_s0.o / mlp_bwd_s0.o / wgrad_s0.o say so):
# the digest of these files and of the compiler flags names the kernels' ISA -- a host-side edit (knerf_api.hip) does not change it
KERNEL_FILES = ["mlp_fwd.hip", "mlp_bwd.hip", "wgrad.hip", "chain.h", "bwd_body.h", "wgrad_body.h", "kernels.h", "layout.h"]


def kernel_digest(flags=None) -> str:
    """16 hex digits naming the source of the three big kernels + the flags they are built with; recorded beside every library
    (`<lib>.info.json`) and in every PMC summary (tools/pmc_report.py), so that bench.py quotes hardware counters only for the
    kernels it actually ran"""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_FILES:
        h.update(f.encode()); h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(" ".join(FLAGS if flags is None else flags).encode())
    return h.hexdigest()[:16]


def file_sha16(path: str) -> str:
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()[:16]


def git_head():
    """the commit of the tree, or None where there is no repository (the GPU boxes get a snapshot without .git)"""
    try:
        r = subprocess.run(["git", "-C", os.path.dirname(HERE), "rev-parse", "HEAD"], capture_output=True, text=True, timeout=10)
        dirty = subprocess.run(["git", "-C", os.path.dirname(HERE), "status", "--porcelain", "--untracked-files=no"], capture_output=True, text=True, timeout=10)
        return (r.stdout.strip() + ("+dirty" if dirty.stdout.strip() else "")) if r.returncode == 0 and r.stdout.strip() else None
    except Exception:           # noqa: BLE001 -- no git here
        return None


def _write_info(lib: str, flags) -> None:
    """<lib>.info.json: which kernels (kernel_digest of the sources it was built from) are inside WHICH file (sha of the library)"""
    import json
    import time
    info = {"lib_sha16": file_sha16(lib), "kernel_digest": kernel_digest(flags), "git_head_at_build": git_head(),
            "built_unix": int(time.time()), "flags": list(flags)}
    with open(lib + ".info.json", "w") as f:
        json.dump(info, f, indent=1)


def _newer(a: str, b: str) -> bool:
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def parse_shapes(specs) -> list:
    """["NL,SK,U" | "NL,SK,U,LX,LD", ...] (also ';'-separated inside one string) -> [(NL, SK, U) | (NL, SK, U, LX, LD), ...]; raises
    ValueError with the offending entry.  LX, LD = pos_emb_xyz, pos_emb_dir (omitted: the reference's 10, 4)"""
    out = []
    for spec in specs:
        for item in str(spec).split(";"):
            if not item.strip():
                continue
            try:
                v = tuple(int(x) for x in item.split(","))
            except ValueError:
                v = ()
            if len(v) not in (3, 5):
                raise ValueError(f"--add-shape wants n_layers,skip_layer,dense_units[,pos_emb_xyz,pos_emb_dir], got {item!r}")
            nl, sk, u = v[:3]
            lx, ld = v[3:] if len(v) == 5 else (10, 4)
            if u not in (64, 128, 256) or not 3 <= nl <= 16 or sk < 1 or not 1 <= lx <= 16 or not 1 <= ld <= 8:
                raise ValueError(f"shape {item!r} is not one the fused kernels cover: dense_units 64, 128 or 256, 3 <= n_layers <= 16, "
                                 f"skip_layer >= 1, 1 <= pos_emb_xyz <= 16, 1 <= pos_emb_dir <= 8")
            if u == 256 and lx == 16 and ld >= 5:
                # measured (round 5 compile sweep): eight encoding k-steps AND four direction k-steps at width 256 need 12 bytes of
                # scratch per lane in the training forward -- the spill check below would refuse the object after a minute of hipcc
                raise ValueError(f"shape {item!r}: pos_emb_xyz = 16 together with pos_emb_dir >= 5 does not fit the register file at "
                                 f"dense_units 256 (pos_emb_xyz <= 15 or pos_emb_dir <= 4 does; so does the same at width 128 / 64)")
            if (lx, ld) == (10, 4):
                v = v[:3]
            if v not in out:
                out.append(v)
    if len(out) > MAX_EXTRA_SHAPES:
        raise ValueError(f"at most {MAX_EXTRA_SHAPES} extra shapes")
    return out


def _spill_report(name: str, remarks: str):
    """None, or what is wrong: a kernel of this object uses scratch memory or spills vector registers (hipcc's kernel-resource-usage
    remarks).  Scalar-register spills are tolerated: they go to lanes of a vector register (v_writelane / v_readlane, ScratchSize stays
    0), not to memory, so they do not enter the vmcnt the kernels count (the 12-layer weight-gradient kernel has six of them)."""
    import re
    kernel, n_kernels = "?", 0
    for ln in remarks.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", ln)
        if m:
            kernel, n_kernels = m.group(1), n_kernels + 1
        m = re.search(r"remark:\s+(ScratchSize \[bytes/lane\]|VGPRs Spill): (\d+)", ln)
        if m and int(m.group(2)) != 0:
            return (f"{name}: kernel {kernel} reports {m.group(1)} = {m.group(2)}: the hand-counted waits of the fused kernels do not "
                    f"allow spills to memory (a shape given with --add-shape may simply be too large for the register file)")
    return None if n_kernels else f"{name}: hipcc printed no kernel-resource-usage remarks, the spill check cannot run"


def _deps(obj: str, fallback):
    """the headers an object was built from, from hipcc's own dependency file (`-MMD -MF <obj>.d`, written next to the object);
    without one (objects of an older build): every header of the tree"""
    d = obj + ".d"
    if not os.path.exists(d):
        return fallback
    txt = open(d).read().replace("\\\n", " ")
    parts = txt.split(":", 1)[1].split() if ":" in txt else []
    return [x for x in parts if x.endswith((".h", ".hpp")) and not x.startswith(("/opt/", "/usr/"))] or fallback


def _progress(line: str) -> None:
    """A build started from inside a test (tests/test_gpu_variants.py: up to 80 hipcc runs, minutes without a line on stdout under
    pytest's capture) leaves a trace where a watchdog that looks for signs of life can see it: the repository's scratch directory
    gpurun_out/, when there is one.  Never fails the build."""
    try:
        d = os.path.join(os.path.dirname(HERE), "gpurun_out")
        if os.path.isdir(d):
            with open(os.path.join(d, "build_progress.log"), "a") as f:
                f.write(line + "\n")
    except OSError:
        pass


def _compile(hipcc, sources, objdir, flags, force, verbose, n_slices):
    hdr_paths = [os.path.join(CSRC, h) for h in HEADERS]
    objs, procs = [], []
    jobs = []
    for src in sources:
        if src in SLICED:
            # -DKNERF_OWN_<k>=, : the lone comma csrc/layout.h KNERF_PICK looks for (which shape this translation unit instantiates)
            jobs += [(src, src.replace(".hip", f"_s{k}.o"), [f"-DKNERF_SHAPE_SLICE={k}", f"-DKNERF_OWN_{k}=,", f"-DKNERF_N_SHAPE_SLICES={n_slices}"]) for k in range(n_slices)]
        else:
            jobs.append((src, src.replace(".hip", ".o"), [f"-DKNERF_N_SHAPE_SLICES={n_slices}"]))
    todo = []
    for src, obj, extra in jobs:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, obj)
        objs.append(o)
        if force or _newer(s, o) or any(_newer(h, o) for h in _deps(o, hdr_paths)):
            # the three big kernels wait on hand-counted vmcnt / lgkmcnt values (chain.h StoreSched, frag_wait): a register spill --
            # scratch traffic counts in vmcnt -- would break them silently, so every instantiation's resource report is checked
            guard = ["-Rpass-analysis=kernel-resource-usage"] if src in SLICED else []
            todo.append((obj, [hipcc, *flags, *extra, *guard, "-MMD", "-MF", o + ".d", "-c", s, "-o", o], bool(guard)))
    limit = max(1, min(len(todo), (os.cpu_count() or 8)))      # hipcc processes in flight
    running = []
    while todo or running:
        while todo and len(running) < limit:
            name, cmd, guarded = todo.pop(0)
            if verbose:
                print(" ".join(cmd), flush=True)
            running.append((name, subprocess.Popen(cmd, stderr=subprocess.PIPE if guarded else None, text=True), guarded))
            procs.append(name)
        name, p, guarded = running.pop(0)
        err = p.communicate()[1] if guarded else None      # (communicate waits)
        if not guarded:
            p.wait()
        _progress(f"{name} rc={p.returncode} ({len(todo) + len(running)} to go)")
        bad = None
        if p.returncode != 0:
            bad = f"hipcc failed on {name}"
        elif guarded:
            bad = _spill_report(name, err)
        if guarded and err:          # pass hipcc's own diagnostics through, without the resource remarks
            rest = [ln for ln in err.splitlines() if "kernel-resource-usage" not in ln and not ln.lstrip().startswith(("|", "^")) and "__global__" not in ln]
            if rest and (verbose or bad):
                print("\n".join(rest), file=sys.stderr)
        if bad:
            for _, q, _g in running:
                q.kill()
            obj_path = os.path.join(objdir, name)
            if os.path.exists(obj_path):
                os.remove(obj_path)
            raise RuntimeError(bad)
    return objs, bool(procs)


def build(force: bool = False, verbose: bool = True, defines=(), variant: str = "", add_shapes=None) -> str:
    """defines/variant: experimental builds (-DNAME=VALUE ...) into libknerf_hip_<variant>.so, used by tools/kbench.py.
    add_shapes: ["NL,SK,U" | "NL,SK,U,LX,LD", ...] further trunk shapes for the fused kernels (default: $KNERF_ADD_SHAPES)"""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if add_shapes is None:
        add_shapes = [os.environ["KNERF_ADD_SHAPES"]] if os.environ.get("KNERF_ADD_SHAPES") else []
    extra = parse_shapes(add_shapes)
    n_slices = N_BUILTIN_SHAPES + len(extra)
    if extra:       # a function-like macro on the command line: KNERF_EXTRA_SHAPES(X) = X(14, NL, SK, U) X(15, ...) ...
        defines = tuple(defines) + ("KNERF_EXTRA_SHAPES(X)=" + " ".join(f"X({N_BUILTIN_SHAPES + i}, {', '.join(str(x) for x in v)})" for i, v in enumerate(extra)),)
    objdir = os.path.join(HERE, "build" + ("_" + variant if variant else ""))
    lib = LIB if not variant else os.path.join(HERE, f"libknerf_hip_{variant}.so")
    probe = PROBE_LIB if not variant else os.path.join(HERE, f"libknerf_probe_{variant}.so")
    flags = FLAGS + ["-D" + d for d in defines]
    os.makedirs(objdir, exist_ok=True)
    # objects are reused by modification time; a change of the flags (defines, extra shapes) rebuilds everything
    sig_file, sig = os.path.join(objdir, "flags.txt"), "\n".join(flags + [str(n_slices)])
    if not os.path.exists(sig_file) or open(sig_file).read() != sig:
        force = force or any(f.endswith(".o") for f in os.listdir(objdir))
        for f in os.listdir(objdir):          # objects of slices that no longer exist must not linger
            if f.endswith((".o", ".o.d")):
                os.remove(os.path.join(objdir, f))
    objs, changed = _compile(hipcc, SOURCES, objdir, flags, force, verbose, n_slices)
    pobjs, pchanged = _compile(hipcc, PROBE_SOURCES, objdir, flags, force, verbose, n_slices)
    open(sig_file, "w").write(sig)
    if changed or not os.path.exists(lib):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        pchanged = True
    if changed or not os.path.exists(lib + ".info.json"):
        # (an up-to-date library without a record: every object is newer than its sources, so the sources as they are now are
        # what it was built from)
        _write_info(lib, flags)
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
    shapes = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--add-shape=")]
    build(force="--force" in sys.argv, defines=defs, variant=var, add_shapes=shapes or None)
