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
    const int ids[4] = {fused_shape_id(6, 3, 128), fused_shape_id(8, 2, 128), fused_shape_id(8, 4, 256), fused_shape_id(7, 3, 256)};
    std::printf("%d %d %d %d\n", ids[0], ids[1], ids[2], ids[3]);
    for (int k = 12; k < kNumFusedShapes; ++k) {
        const ShapeInfo& s = shape_info(k);
        std::printf("%d %d %d %d %d %d %d\n", s.n_layers, s.skip, s.units, s.param_count, s.fwd_blocks, s.bwd_blocks, s.n_jobs);
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
    for bad in ("8,4,64", "2,1,256", "9,4,256", "4,3,256", "8,4", "a,b,c", "8,0,256"):     # width, depth, concat behind the last layer (x2), arity, type, skip
        with pytest.raises(ValueError):
            B.parse_shapes([bad])
    with pytest.raises(ValueError):
        B.parse_shapes([f"{nl},{nl},128" for nl in range(4, 17)])       # 13 valid triples: more than the slice macros cover


def test_layout_header_with_extra_shapes(tmp_path):
    src = tmp_path / "shapes.cpp"
    src.write_text(PROG)
    exe = tmp_path / "shapes"
    r = subprocess.run(["g++", "-std=c++17", "-O0", "-I", os.path.join(ROOT, "keras_nerf_amd", "csrc"), "-DKNERF_EXTRA_SHAPES(X)=X(12, 6, 3, 128) X(13, 8, 2, 128)",
                        str(src), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split("\n")
    assert out[0].split() == ["12", "14"]
    assert out[1].split() == ["12", "13", "0", "-1"]
    for line, (nl, sk, u) in zip(out[2:4], ((6, 3, 128), (8, 2, 128))):
        v = [int(x) for x in line.split()]
        cfg = O.NerfConfig(n_layers=nl, dense_units=u, skip_layer=sk)
        n_concat = sum(1 for name, i, o in O.layer_shapes(cfg) if name.startswith("layer_") and i == u + 63)
        ks, ot = u // 16, u // 32
        assert v[:3] == [nl, sk, u] and v[3] == O.param_count(cfg)
        assert v[4] == 4 * ot + (nl - 1) * ks * ot + n_concat * 4 * ot + ks + 2
        assert v[5] == ot + (nl - 1) * ks * ot and v[6] == nl + 1
    assert [int(x) for x in out[4].split()] == [(4 * 4 + 5 * 8 * 4 + 1 * 4 * 4 + 8 + 2) * 512, (4 * 6 + 1) * 32, (4 + 5 * 8 * 4) * 512]
