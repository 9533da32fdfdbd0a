"""Lane-level NumPy model of v_mfma_f32_32x32x16_bf16 and of the fused chain kernels' dataflow (tests only).

Operand maps (cdna guide section 3, restated in keras_nerf_amd/csrc/layout.h):
  A: lane l (r=l&31, h=l>>5), element j -> A[r][8h+j];   B: lane l (c=l&31, h), element j -> B[8h+j][c]
  C/D: lane l, reg i -> D[(i&3) + 8*(i>>2) + 4*(l>>5)][l&31]
"""
import numpy as np

from oracle.nerf_oracle import round_bf16

LANES = np.arange(64)
R_, H_ = LANES & 31, LANES >> 5
ROW_OF = (np.arange(16)[None, :] & 3) + 8 * (np.arange(16)[None, :] >> 2) + 4 * H_[:, None]   # [64,16]


def mfma(a_frag, b_frag, c):
    """a_frag, b_frag [64,8] (bf16-representable fp32), c [64,16] fp32 -> d [64,16]"""
    A = np.zeros((32, 16), np.float32); B = np.zeros((16, 32), np.float32)
    for j in range(8):
        A[R_, 8 * H_ + j] = a_frag[:, j]
        B[8 * H_ + j, R_] = b_frag[:, j]
    D = (A.astype(np.float64) @ B.astype(np.float64)).astype(np.float32)
    return c + D[ROW_OF, R_[:, None]]


def pack_acc(acc):
    """f32 accumulator [64,16] -> two bf16 B-operand k-steps [64,8]"""
    b = round_bf16(acc)
    return b[:, :8], b[:, 8:]


def enc_slots(p, L, nq):
    """B-operand blocks of the positional encoding of p [32,3]: list of nq arrays [64,8]
    (half 0: x, y, sin(2^i p_c); half 1: z, 0, cos(2^i p_c); slot m = 2 + 3i + c)"""
    e = np.zeros((64, nq * 8), np.float32)
    for l in range(64):
        s, h = l & 31, l >> 5
        e[l, 0] = p[s, 2] if h else p[s, 0]
        e[l, 1] = 0.0 if h else p[s, 1]
        for i in range(L):
            for c in range(3):
                a = np.float32(2.0 ** i) * p[s, c]
                e[l, 2 + 3 * i + c] = np.cos(a) if h else np.sin(a)
    e = round_bf16(e)
    return [e[:, 8 * q:8 * q + 8] for q in range(nq)]


def gather_blocks(table, flat):
    """int32 table [nblocks*512] -> bf16-rounded fragments [nblocks,64,8]"""
    w = np.where(table >= 0, flat[np.maximum(table, 0)], np.float32(0)).astype(np.float32)
    return round_bf16(w).reshape(-1, 64, 8)


def bias_acc(bias_tab, flat, tile):
    b = np.where(bias_tab >= 0, flat[np.maximum(bias_tab, 0)], np.float32(0)).astype(np.float32).reshape(-1, 32)
    return b[tile][ROW_OF]    # [64,16]


FWD_STAGES = [(0, 4, 8), (32, 16, 8), (160, 16, 8), (288, 16, 8), (416, 16, 8), (544, 20, 8), (704, 16, 8), (832, 16, 8),
              (960, 16, 9), (1104, 18, 4), (1176, 8, 1)]


def forward_chain(fwd_tab, bias_tab, flat, p, d):
    """Mirror of mlp_fwd_kernel for one wave (32 samples).  Returns rgb [32,3], sigma [32], saved act blocks, masks."""
    frags = gather_blocks(fwd_tab, flat)
    enc = enc_slots(p, 10, 4)
    dirc = enc_slots(d, 4, 2)
    act = {}
    masks = {}
    btile = 0

    def stage(st, inputs, relu):
        nonlocal btile
        b0, nks, n_ot = FWD_STAGES[st]
        outs, raw = [], []
        for ot in range(n_ot):
            acc = bias_acc(bias_tab, flat, btile); btile += 1
            for ks in range(nks):
                acc = mfma(frags[b0 + ot * nks + ks], inputs[ks], acc)
            raw.append(acc)
            a2 = np.maximum(acc, 0) if relu else acc
            lo, hi = pack_acc(a2)
            outs += [lo, hi]
        return outs, raw
    x, raw = stage(0, enc, True); act[0] = x; masks[0] = [r > 0 for r in raw]
    for st in range(1, 8):
        inp = x + enc if st == 5 else x
        x, raw = stage(st, inp, True); act[st] = x; masks[st] = [r > 0 for r in raw]
    fs, raw = stage(8, x, False)
    feat = fs[:16]
    sigma = np.maximum(raw[8][:32, 0], 0)          # half 0, reg 0 = row 0
    f2, _ = stage(9, feat + dirc, False)
    _, raw = stage(10, f2, False)
    z = raw[0][:32, :3]
    rgb = 1.0 / (1.0 + np.exp(-z))
    return rgb, sigma, dict(enc=enc, dirc=dirc, h=act, feat=feat, f2=f2, masks=masks)
