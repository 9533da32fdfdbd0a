"""The general-shape path's relu bits (csrc/generic.hip GemmArgs::mask_out / mask_in, round 5): a lane-level model of the forward
GEMM's epilogue writing them and of the dgrad GEMM's epilogue reading them, for every pair of column-block widths NT (writer) and
NT' (reader) -- the two products of one tensor may run with different NT (launch_gemm picks it from K) -- on random activations.

Layout under test: [32-row tile][feature // 8][32 bytes], byte 16 hh + i of a (tile, octet) = row (i & 3) + 8 (i >> 2) + 4 hh, bit =
feature % 8.  The MFMA accumulator of lane (r, hh) holds rows (i & 3) + 8 (i >> 2) + 4 hh of column r for i = 0..15, and the kernels
give column r of column tile t the feature n0 + NT r + t (gemm_epilogue's column assignment)."""
import numpy as np
import pytest


def _row(i, hh):
    return (i & 3) + 8 * (i >> 2) + 4 * hh


def write_bits(act, NT):
    """act: [32 rows][N] post-relu values of one tile; returns the mask bytes [N // 8][32] as the forward epilogue stores them"""
    N = act.shape[1]
    out = np.full((N // 8, 32), -1, dtype=np.int64)
    stores = 0
    for n0 in range(0, N, 32 * NT):
        w4 = np.zeros((64, 4), dtype=np.uint64)                      # per lane: four dwords
        for lane in range(64):
            r, hh = lane & 31, lane >> 5
            f0 = n0 + NT * r
            sh = f0 & 7
            for i in range(16):
                bits = 0
                for t in range(NT):
                    bits |= int(act[_row(i, hh), f0 + t] > 0) << t
                w4[lane, i >> 2] |= np.uint64(bits << (8 * (i & 3) + sh))
        o = 1
        while o < 8 // NT:                                           # __shfl_xor butterfly over the 8 / NT lanes of an octet
            w4 = w4 | w4[np.arange(64) ^ o]
            o <<= 1
        for lane in range(64):
            r, hh = lane & 31, lane >> 5
            if r & (8 // NT - 1):
                continue
            f0 = n0 + NT * r
            by = np.array([(int(w4[lane, q]) >> (8 * k)) & 0xFF for q in range(4) for k in range(4)])
            assert (out[f0 >> 3, 16 * hh:16 * hh + 16] == -1).all(), "two lanes store the same 16 bytes"
            out[f0 >> 3, 16 * hh:16 * hh + 16] = by
            stores += 1
    assert (out >= 0).all(), "bytes never written"
    assert stores == (N // 8) * 2
    return out.astype(np.uint8)


def read_bits(mask, N, NT):
    """what the dgrad epilogue with column blocks of NT tiles decides per (row, feature)"""
    got = np.zeros((32, N), dtype=bool)
    for n0 in range(0, N, 32 * NT):
        for lane in range(64):
            r, hh = lane & 31, lane >> 5
            f0 = n0 + NT * r
            sh = f0 & 7
            by = mask[f0 >> 3, 16 * hh:16 * hh + 16]                 # ONE 16-byte load
            w4 = [int(by[4 * q]) | int(by[4 * q + 1]) << 8 | int(by[4 * q + 2]) << 16 | int(by[4 * q + 3]) << 24 for q in range(4)]
            for i in range(16):
                bits = w4[i >> 2] >> (8 * (i & 3) + sh)
                for t in range(NT):
                    got[_row(i, hh), f0 + t] = bool((bits >> t) & 1)
    return got


@pytest.mark.parametrize("NTW", [1, 2, 4, 8])
@pytest.mark.parametrize("NTR", [1, 2, 4, 8])
def test_relu_bits_round_trip_between_any_two_column_block_widths(NTW, NTR):
    rng = np.random.default_rng(NTW * 10 + NTR)
    N = 256
    act = np.maximum(rng.standard_normal((32, N)), 0.0)
    act[rng.random((32, N)) < 0.2] = 0.0
    mask = write_bits(act, NTW)
    # the stored layout itself: bit f % 8 of byte [f // 8][16 hh + i] is row (i & 3) + 8 (i >> 2) + 4 hh
    for hh in range(2):
        for i in range(16):
            row = _row(i, hh)
            want = np.packbits((act[row] > 0).reshape(N // 8, 8)[:, ::-1], axis=1)[:, 0]
            assert (mask[:, 16 * hh + i] == want).all()
    assert (read_bits(mask, N, NTR) == (act > 0)).all()


def test_every_row_of_a_tile_is_addressed_once():
    assert sorted(_row(i, hh) for hh in range(2) for i in range(16)) == list(range(32))
