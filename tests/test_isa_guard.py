"""Build-time guard for the hand-counted synchronisation of the three big kernels (no GPU needed: hipcc cross-compiles).

`mlp_fwd_kernel`, `mlp_bwd_kernel` and `wgrad_kernel` wait with COUNTED `s_waitcnt vmcnt(N)` / `lgkmcnt(N)` immediates that are
derived at compile time from tables of what the source issues (chain.h StoreSched / WaitTable, Ring::frag_asm / frag_wait, the
vmcnt arithmetic of wgrad_body.h).  Those counts are only right while the compiler emits exactly the memory operations the tables
assume: one register spill (scratch traffic counts in vmcnt), one store split in two or one LDS-DMA copy duplicated by an
unrolling decision would silently weaken a wait.  This test compiles the three files to gfx950 assembly and asserts, for every
instantiation: no VGPR/SGPR spills, no private segment, no scratch instruction, and the exact number of streaming stores,
LDS-DMA copies and MFMAs that the tables imply.

Expected counts (csrc/layout.h, chain.h, mlp_fwd.hip, bwd_body.h, wgrad_body.h):
  forward   MFMA = kFwdBlocks = 978.  LDS-DMA: 2 per page issue x (5 prologue pages + 61 barriers of wave group A + 61 of group B)
            = 254.  Stores: training 4 enc + (1 mask) + 4 x (8 x 2 + 1) + (8 x 2 + 1) + (8 x 2 + 1) + (8 x 2 + 3: mask + 2 dir) = 126 saved
            blocks + 1 raw output = 127; inference: the raw output only.
  dgrad     MFMA = kBwdBlocks = 904.  LDS-DMA: 2 x (5 + 56 + 56) = 234.  Stores: 1 dz_head block + 7 stages x 8 tiles x 2 = 113.
  wgrad     per instantiation, all nine jobs inlined: LDS-DMA = sum over jobs of (staging slots NS) x (copies per wave and tile G)
            = 6x3 (layer_0) + 4 x 4x4 (layers 2,3,4,6) + 4x5 (layer_5) + 6x4 (head) + 6x3 (layer_1) + 8x3 (layer_7) = 168;
            MFMA = 2 x accumulators per wave: 6 + 4 x 18 + 22 + 4 + (18 + 2 x 4 recompute sites) + (18 + 2 x 1) = 150.

Round 6, Shape<9, 4, 256> -- a trunk that ends in a concat, the head takes [h ; xyz_enc ; dir_enc] (build-time entry, compiled here as
slice 14): forward MFMA = 32 + 7 x 128 + 160 + (16 + 4 + 2) = 1110, LDS-DMA 2 x (5 + 69 + 69) = 286, stores 4 enc + 1 mask (layer_0) +
8 x 17 + (4 enc again + 2 dir) + 1 raw = 148; dgrad MFMA 8 + 8 x 128 = 1032, LDS-DMA 2 x (5 + 64 + 64) = 266, stores 1 + 8 x 16 = 129;
wgrad LDS-DMA 18 + 18 + 5 x 16 + 20 + 24 (layer_8, dz recomputed) + 6 x 4 (head: 22 + 2 blocks per tile) = 184, MFMA 6 + 26 + 5 x 18 + 22 + 20 +
4 (head: twelve rows over eight waves = two accumulators) = 168.
"""
import hashlib
import os
import re
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "keras_nerf_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _flags():
    import importlib.util
    spec = importlib.util.spec_from_file_location("knerf_build", os.path.join(ROOT, "keras_nerf_amd", "build.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    return b.FLAGS          # exactly what the library is built with


SLICE0 = ["-DKNERF_SHAPE_SLICE=0", "-DKNERF_OWN_0=,"]      # what build.py passes for slice 0 (csrc/layout.h KNERF_PICK)


CBL = ["-DKNERF_SHAPE_SLICE=14", "-DKNERF_OWN_14=,", "-DKNERF_EXTRA_SHAPES(X)=X(14, 9, 4, 256)"]      # build.py --add-shape=9,4,256 as slice 14


def _asm(name, slice_flags=None):
    """gfx950 assembly of csrc/<name>.hip, cached under the temp dir by the hash of every source and header it may include"""
    SLICE0 = slice_flags or globals()["SLICE0"]
    h = hashlib.sha1()
    for f in sorted(os.listdir(CSRC)) + ["../../include/knerf.h"]:
        p = os.path.join(CSRC, f)
        if os.path.isfile(p):
            h.update(f.encode()); h.update(open(p, "rb").read())
    h.update(" ".join(_flags() + SLICE0).encode())
    out = os.path.join(tempfile.gettempdir(), f"knerf_isa_{name}_{h.hexdigest()[:16]}.s")
    if not os.path.exists(out):
        tmp = out + f".{os.getpid()}"
        # slice 0 = the reference's default trunk shape (layout.h KNERF_FUSED_SHAPES): the object build.py links for it
        subprocess.run([HIPCC, *_flags(), *SLICE0, "-S", "--cuda-device-only", "-Wno-unused-command-line-argument", "-o", tmp,
                        os.path.join(CSRC, name + ".hip")], check=True, capture_output=True)
        os.replace(tmp, out)
    return open(out).read()


def _kernels(asm):
    """{mangled name: (body text, {metadata key: int})} for every kernel of an assembly file"""
    bodies = {}
    for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)^\.Lfunc_end\d+:", asm, re.S | re.M):
        bodies[m.group(1)] = m.group(2)
    meta = {}
    for m in re.finditer(r"\.name:\s+(_Z\w+)\n(.*?)(?=\n\s+- \.|\n\.\.\.|\Z)", asm, re.S):
        meta[m.group(1)] = {k: int(v) for k, v in re.findall(r"\.(\w+):\s+(\d+)\s*$", m.group(2), re.M)}
    return {k: (bodies[k], meta.get(k, {})) for k in bodies}


def _count(body, mnemonic):
    return len(re.findall(r"^\s+" + re.escape(mnemonic) + r"\b", body, re.M))


@pytest.fixture(scope="module")
def isa():
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    with ThreadPoolExecutor(3) as ex:
        fwd, bwd, wg = ex.map(_asm, ["mlp_fwd", "mlp_bwd", "wgrad"])
    return {"mlp_fwd": _kernels(fwd), "mlp_bwd": _kernels(bwd), "wgrad": _kernels(wg)}


@pytest.fixture(scope="module")
def isa_cbl():
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    with ThreadPoolExecutor(3) as ex:
        fwd, bwd, wg = ex.map(lambda n: _asm(n, CBL), ["mlp_fwd", "mlp_bwd", "wgrad"])
    return {"mlp_fwd": _kernels(fwd), "mlp_bwd": _kernels(bwd), "wgrad": _kernels(wg)}


EXPECT = {
    # kernel-name fragment: (file, instantiation tags, {mnemonic: count})
    # (kernels are templates on the trunk shape: Shape<8, 4, 256> = "5ShapeILi8ELi4ELi256EEE" in the mangled names)
    "mlp_fwd_kernelINS_5ShapeILi8ELi4ELi256EEELb1E": ("mlp_fwd", 2, {"v_mfma_f32_32x32x16_bf16": 978, "global_load_lds_dwordx4": 254, "global_store_dwordx4": 127}),
    "mlp_fwd_kernelINS_5ShapeILi8ELi4ELi256EEELb0E": ("mlp_fwd", 2, {"v_mfma_f32_32x32x16_bf16": 978, "global_load_lds_dwordx4": 254, "global_store_dwordx4": 1}),
    "mlp_bwd_kernelINS_5ShapeILi8ELi4ELi256EEE": ("mlp_bwd", 2, {"v_mfma_f32_32x32x16_bf16": 904, "global_load_lds_dwordx4": 234, "global_store_dwordx4": 113}),
    "wgrad_kernelINS_5ShapeILi8ELi4ELi256EEE": ("wgrad", 4, {"v_mfma_f32_32x32x16_bf16": 150, "global_load_lds_dwordx4": 168}),
}


EXPECT_CBL = {
    "mlp_fwd_kernelINS_5ShapeILi9ELi4ELi256EEELb1E": ("mlp_fwd", 2, {"v_mfma_f32_32x32x16_bf16": 1110, "global_load_lds_dwordx4": 286, "global_store_dwordx4": 148}),
    "mlp_fwd_kernelINS_5ShapeILi9ELi4ELi256EEELb0E": ("mlp_fwd", 2, {"v_mfma_f32_32x32x16_bf16": 1110, "global_load_lds_dwordx4": 286, "global_store_dwordx4": 1}),
    "mlp_bwd_kernelINS_5ShapeILi9ELi4ELi256EEE": ("mlp_bwd", 2, {"v_mfma_f32_32x32x16_bf16": 1032, "global_load_lds_dwordx4": 266, "global_store_dwordx4": 129}),
    "wgrad_kernelINS_5ShapeILi9ELi4ELi256EEE": ("wgrad", 4, {"v_mfma_f32_32x32x16_bf16": 168, "global_load_lds_dwordx4": 184}),
}


@pytest.mark.parametrize("frag", sorted(EXPECT_CBL))
def test_counted_waits_of_a_trunk_that_ends_in_a_concat(isa_cbl, frag):
    """the head stage with four more k-steps and the second copy of the enc blocks (round 6): same guard, the counts of Shape<9, 4, 256>"""
    # ten jobs in one kernel: two scalar registers go to lanes of a vector register (v_writelane, no memory: private segment 0, no
    # scratch instruction -- asserted below), which does not enter the vmcnt the waits count; build.py tolerates exactly this
    _check_counts(isa_cbl, frag, EXPECT_CBL, sgpr_to_lanes_ok=True)


@pytest.mark.parametrize("frag", sorted(EXPECT))
def test_counted_waits_still_match_the_emitted_instructions(isa, frag):
    _check_counts(isa, frag, EXPECT)


def _check_counts(isa, frag, EXPECT, sgpr_to_lanes_ok=False):
    fname, n_inst, counts = EXPECT[frag]
    ks = {k: v for k, v in isa[fname].items() if frag in k}
    assert len(ks) == n_inst, (frag, sorted(isa[fname]))
    for name, (body, meta) in ks.items():
        assert meta.get("vgpr_spill_count") == 0 and (meta.get("sgpr_spill_count") == 0 or sgpr_to_lanes_ok), (name, meta)
        assert meta.get("private_segment_fixed_size") == 0, (name, meta)
        assert not re.search(r"^\s+scratch_", body, re.M), f"{name}: scratch instruction (spill or stack object)"
        assert not re.search(r"^\s+buffer_(load|store)", body, re.M), f"{name}: buffer access (stack?)"
        for mnem, want in counts.items():
            got = _count(body, mnem)
            assert got == want, f"{name}: {got} x {mnem}, the wait tables assume {want}"
        assert meta.get("vgpr_count", 999) <= 256, (name, meta)


def test_chain_kernels_wait_with_counted_immediates(isa):
    """the counted waits are really there (a refactoring that drops back to vmcnt(0) / lgkmcnt(0) everywhere would pass the count
    test above and lose 10-20 % speed): the training forward has dozens of distinct vmcnt immediates"""
    for frag, fname in (("mlp_fwd_kernelINS_5ShapeILi8ELi4ELi256EEELb1E", "mlp_fwd"), ("mlp_bwd_kernelINS_5ShapeILi8ELi4ELi256EEE", "mlp_bwd")):
        for name, (body, _) in isa[fname].items():
            if frag not in name:
                continue
            vm = set(re.findall(r"s_waitcnt vmcnt\((\d+)\)", body))
            assert len(vm) >= 8 and "0" in vm, (name, sorted(vm, key=int))
            assert re.search(r"s_waitcnt lgkmcnt\(3\)", body), name
