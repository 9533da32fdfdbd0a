"""Host tables of EVERY built-in trunk shape of the fused kernels (csrc/layout.h KNERF_FUSED_SHAPES: widths 256 and 128), on the CPU:
the weight-stream packing tables address each trunk parameter exactly once and the composed head behind the parameters, the dgrad
table is a permutation of what it streams, and the weight-gradient destination tables hit every trunk / sigma gradient element and
every head-accumulator element exactly once.  (The default shape is additionally replayed lane by lane in test_layout_sim.py; the
other shapes run against the oracle on the GPU, test_gpu_fused_shapes.py.)"""
import numpy as np
import pytest

from keras_nerf_amd import _lib
from keras_nerf_amd import debug as D
from oracle import nerf_oracle as O

AUX_S = (256 + 99 + 51) * 3    # csrc/layout.h kAuxS: the head accumulator keeps s behind the largest M (width 256, pos_emb_xyz 16 in a trunk that ends in a concat, pos_emb_dir 8)


def _n_shapes():
    n = 0
    while True:
        try:
            D.debug_table(5, n)
        except _lib.KnerfError:
            return n
        n += 1


def test_the_list_has_the_default_shape_first_and_all_three_widths():
    n = _n_shapes()
    assert n >= 14
    infos = [tuple(D.debug_table(5, k)[:3]) for k in range(n)]
    assert infos[0] == (8, 4, 256)
    assert len(set(infos)) == n                                   # no triple twice
    assert {u for _, _, u in infos} == {256, 128, 64}
    assert (8, 4, 64) in infos and (4, 2, 64) in infos and (8, 4, 128) in infos and (4, 2, 128) in infos and (8, 2, 256) in infos and (12, 4, 256) in infos


@pytest.mark.parametrize("k", range(14))
def test_tables_of_shape(k):
    _check_tables(k)


def _enc_q(L):
    """k-steps of an encoding of depth L: 2 + 3 L features per lane half, 8 per k-step, rounded up to even (csrc/layout.h enc_q)"""
    return ((2 + 3 * L + 7) // 8 + 1) // 2 * 2


def _check_tables(k):
    nl, sk, U, n, lx, ld = (int(v) for v in D.debug_table(5, k))
    cfg = O.NerfConfig(n_layers=nl, dense_units=U, skip_layer=sk, pos_emb_xyz=lx, pos_emb_dir=ld)
    xd, dd, qx, qd = 3 + 6 * lx, 3 + 6 * ld, _enc_q(lx), _enc_q(ld)
    assert n == O.param_count(cfg)
    shapes = O.layer_shapes(cfg)
    n_trunk = sum(i * o + o for name, i, o in shapes if name.startswith("layer_"))
    fwd, bias, bwd = D.debug_table(0, k), D.debug_table(1, k), D.debug_table(2, k)
    n_concat = sum(1 for name, i, o in shapes if name.startswith("layer_") and i == U + xd)
    ks, ot = U // 16, U // 32
    # a trunk that ends in a concat ((nl - 1) % sk == 0, mlp.py:36-38): sigma / features take [h ; xyz_enc] -- xt more real rows and
    # qt more k-steps in the composed head (round 6)
    cbl = nl > 1 and (nl - 1) % sk == 0
    xt, qt = (xd, qx) if cbl else (0, 0)
    assert dict((name, i) for name, i, o in shapes)["sigma"] == U + xt and dict((name, i) for name, i, o in shapes)["features"] == U + xt
    assert fwd.size == (qx * ot + (nl - 1) * ks * ot + n_concat * qx * ot + (ks + qt + qd)) * 512       # layer_0, U-wide layers, concat extras, head
    assert bias.size == (ot * nl + 1) * 32
    assert bwd.size == (ot + (nl - 1) * ks * ot) * 512
    # forward stream + bias: every trunk parameter once, nothing of the four tensors behind the trunk, the composed head behind n
    used = np.zeros(n, np.int32)
    np.add.at(used, fwd[(fwd >= 0) & (fwd < n)], 1)
    np.add.at(used, bias[(bias >= 0) & (bias < n)], 1)
    assert (used[:n_trunk] == 1).all() and not used[n_trunk:].any()
    head = np.concatenate([fwd[fwd >= n], bias[bias >= n]]) - n
    assert sorted(head) == sorted([r * 4 + c for r in range(U + xt + dd) for c in range(4)] + [(U + 16 * qt + 16 * qd) * 4 + c for c in range(4)])
    # dgrad stream: layers 1 .. NL-1 (their first U input rows: no gradient flows into the encodings) and the head's h rows, each once
    vals, counts = np.unique(bwd[bwd >= 0], return_counts=True)
    assert counts.max() == 1
    assert sorted(bwd[bwd >= n] - n) == [r * 4 + c for r in range(U) for c in range(4)]
    exp = []
    off = 0
    for name, i, o in shapes:
        if name.startswith("layer_") and name != "layer_0":
            exp.append(np.arange(off, off + U * o))                 # rows 0 .. U-1 of kernel[in, out]
        off += i * o + o
    assert np.array_equal(np.sort(bwd[(bwd >= 0) & (bwd < n)]), np.concatenate(exp))
    # weight-gradient destinations: trunk + sigma gradients once each; head accumulator M [(U [+63] +27) x 3] and s [3] once each
    dst, job_off = D.debug_table(3, k), D.debug_table(4, k)
    assert job_off.size == nl + 2 and job_off[0] == 0 and job_off[-1] == dst.size and (np.diff(job_off) > 0).all()
    g = dst[(dst >= 0) & (dst < n)]
    assert np.array_equal(np.sort(g), np.arange(n_trunk + U + xt + 1))
    aux = np.sort(dst[dst >= n] - n)
    assert np.array_equal(aux, np.concatenate([np.arange((U + xt + dd) * 3), AUX_S + np.arange(3)]))
    assert dst.min() >= -1
