"""Build-time shape list of the fused kernels (keras_nerf_amd/build.py --add-shape, csrc/layout.h KNERF_EXTRA_SHAPES), on the CPU:
argument parsing, and csrc/layout.h compiled host-only (g++) with two extra entries: the list grows, the new triples are found, their
parameter counts and stream sizes are the ones the oracle's layer shapes imply."""
import os
import subprocess
import sys

import pytest

from keras_nerf_amd import build as B
from oracle import nerf_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROG = r'''
#include <cstdio>
#include "layout.h"
using namespace knerf;
int main() {
    std::printf("%d %d\n", kNumBuiltinShapes, kNumFusedShapes);
    const int ids[4] = {fused_shape_id(6, 3, 128), fused_shape_id(8, 2, 128), fused_shape_id(8, 4, 256), fused_shape_id(7, 2, 256)};
    std::printf("%d %d %d %d %d %d\n", ids[0], ids[1], ids[2], ids[3], fused_shape_id(8, 4, 256, 12, 3), fused_shape_id(8, 4, 256, 12, 4));
    for (int k = kNumBuiltinShapes; k < kNumFusedShapes; ++k) {
        const ShapeInfo& s = shape_info(k);
        std::printf("%d %d %d %d %d %d %d %d %d %d\n", s.n_layers, s.skip, s.units, s.param_count, s.fwd_blocks, s.bwd_blocks, s.n_jobs, s.lx, s.ld, s.act_blocks);
    }
    PackTables pt;
    build_fwd<Shape<6, 3, 128>>(pt); build_bwd<Shape<6, 3, 128>>(pt);
    std::printf("%zu %zu %zu\n", pt.fwd.size(), pt.fwd_bias.size(), pt.bwd.size());
    return 0;
}
'''


def test_parse_shapes_accepts_covered_triples_and_names_the_others():
    assert B.parse_shapes(["6,3,128", "8,2,128;6,4,256", "6,3,128"]) == [(6, 3, 128), (8, 2, 128), (6, 4, 256)]
    assert B.parse_shapes([]) == [] and B.parse_shapes([""]) == []
    assert B.parse_shapes(["8,4,256,6,2;8,4,128,10,4"]) == [(8, 4, 256, 6, 2), (8, 4, 128)]           # the reference's encodings need no ShapeL entry
    # round 6: a concat behind the LAST layer ((n_layers - 1) % skip_layer == 0: 9/4, 4/3, 5/2 ...) is covered -- the head takes [h ; xyz_enc ; dir_enc]
    assert B.parse_shapes(["9,4,256", "4,3,256;5,2,128", "5,4,64,6,2"]) == [(9, 4, 256), (4, 3, 256), (5, 2, 128), (5, 4, 64, 6, 2)]
    for bad in ("8,4,96", "2,1,256", "8,4", "a,b,c", "8,0,256", "8,4,256,10", "8,4,256,17,4", "8,4,256,10,9", "8,4,256,0,4"):     # width, depth, arity, type, skip, arity, encodings (x3)
        with pytest.raises(ValueError):
            B.parse_shapes([bad])
    with pytest.raises(ValueError):
        B.parse_shapes([f"{nl},{sk},{u}" for nl in range(4, 17) for sk in (nl, nl + 1) for u in (128, 256)])       # 52 valid triples: over the build-time budget (MAX_EXTRA_SHAPES)
    assert len(B.parse_shapes([f"{nl},{nl},128" for nl in range(4, 17)])) == 13      # round 4: no cap at 12 any more (csrc/layout.h KNERF_PICK needs no per-index macro)
    # round 5 compile sweep of the corners: the one combination that does not fit the register file is named up front (the spill check
    # would refuse it after a minute of hipcc): eight encoding k-steps (pos_emb_xyz 16) with four direction k-steps (pos_emb_dir >= 5) at width 256
    for bad in ("8,4,256,16,5", "16,4,256,16,8"):
        with pytest.raises(ValueError, match="register file"):
            B.parse_shapes([bad])
    assert B.parse_shapes(["8,4,256,16,4", "8,4,256,15,8", "8,4,128,16,8", "7,4,64,16,8"]) == [(8, 4, 256, 16, 4), (8, 4, 256, 15, 8), (8, 4, 128, 16, 8), (7, 4, 64, 16, 8)]


def test_layout_header_with_extra_shapes(tmp_path):
    src = tmp_path / "shapes.cpp"
    src.write_text(PROG)
    exe = tmp_path / "shapes"
    r = subprocess.run(["g++", "-std=c++17", "-O0", "-I", os.path.join(ROOT, "keras_nerf_amd", "csrc"), "-DKNERF_EXTRA_SHAPES(X)=X(14, 6, 3, 128) X(15, 8, 2, 128) X(16, 8, 4, 256, 12, 3) X(17, 9, 4, 256) X(18, 5, 4, 64, 6, 2)",
                        str(src), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split("\n")
    assert out[0].split() == ["14", "19"]
    assert out[1].split() == ["14", "15", "0", "-1", "16", "-1"]
    for line, (nl, sk, u, lx, ld, qx, qd) in zip(out[2:7], ((6, 3, 128, 10, 4, 4, 2), (8, 2, 128, 10, 4, 4, 2), (8, 4, 256, 12, 3, 6, 2), (9, 4, 256, 10, 4, 4, 2),
                                                          (5, 4, 64, 6, 2, 4, 2))):
        v = [int(x) for x in line.split()]
        cfg = O.NerfConfig(n_layers=nl, dense_units=u, skip_layer=sk, pos_emb_xyz=lx, pos_emb_dir=ld)
        n_concat = sum(1 for name, i, o in O.layer_shapes(cfg) if name.startswith("layer_") and i == u + 3 + 6 * lx)
        ks, ot = u // 16, u // 32
        assert v[:3] == [nl, sk, u] and v[3] == O.param_count(cfg) and v[7:9] == [lx, ld]
        qt = qx if (nl - 1) % sk == 0 else 0            # a trunk that ends in a concat: the head takes [h ; xyz_enc ; dir_enc] (round 6)
        assert v[4] == qx * ot + (nl - 1) * ks * ot + n_concat * qx * ot + ks + qt + qd          # six encoding k-steps for pos_emb_xyz = 12
        assert v[5] == ot + (nl - 1) * ks * ot and v[6] == nl + 1
        saves_h0 = u != 256 or qx != 4
        assert v[9] == ks * (nl - 1 + saves_h0) + qx + qt + qd             # ... and its act run holds the enc blocks a second time, behind h_{NL-1}
    assert [int(x) for x in out[7].split()] == [(4 * 4 + 5 * 8 * 4 + 1 * 4 * 4 + 8 + 2) * 512, (4 * 6 + 1) * 32, (4 + 5 * 8 * 4) * 512]


def test_spill_report_of_the_build_guard():
    """build.py refuses an instantiation of the three big kernels that spills (their waits are hand-counted)"""
    ok = "a.hip:1:1: remark: Function Name: _Zk [-R]\na.hip:1:1: remark:     VGPRs Spill: 0 [-R]\na.hip:1:1: remark:     ScratchSize [bytes/lane]: 0 [-R]\n"
    assert B._spill_report("x.o", ok) is None
    assert "VGPRs Spill = 3" in B._spill_report("x.o", ok.replace("VGPRs Spill: 0", "VGPRs Spill: 3"))
    assert "ScratchSize" in B._spill_report("x.o", ok.replace("lane]: 0", "lane]: 16"))
    assert "cannot run" in B._spill_report("x.o", "")
    assert B._spill_report("x.o", ok + "a.hip:1:1: remark:     SGPRs Spill: 6 [-R]\n") is None      # to vector-register lanes, not to memory
