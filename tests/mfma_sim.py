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
              (960, 18, 1)]
N_PARAMS, HEAD_ROWS, AUX_M, AUX_S, AUX_COUNT = 595844, 288, 0, (256 + 99 + 51) * 3, 1224      # csrc/layout.h kAuxS / kAuxCount (round 6: room for the xyz rows of a trunk that ends in a concat)


def extended_weights(params, cfg):
    """[flat parameters | composed head matrix [288,4] | head bias [4]]: what optim.hip head_compose leaves in a net's
    weight buffer (csrc/layout.h "collapsed head"), here from the oracle's head_compose"""
    from oracle import nerf_oracle as O
    H, hb = O.head_compose(params, cfg)
    Hp = np.zeros((HEAD_ROWS, 4), np.float32); Hp[:H.shape[0]] = H
    return np.concatenate([O.flatten_params(params), Hp.reshape(-1), hb.astype(np.float32)])


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
    _, raw = stage(8, x + dirc, False)             # head: one out tile, rows r, g, b, sigma (half 0, regs 0..3)
    z = raw[0][:32, :4]
    rgb = 1.0 / (1.0 + np.exp(-z[:, :3]))
    sigma = np.maximum(z[:, 3], 0)
    return rgb, sigma, dict(enc=enc, dirc=dirc, h=act, masks=masks)


# ---------------------------------------------------------------------------------------------------------------
# backward: dgrad chain + saved blocks + wgrad (transposed LDS reads), mirroring mlp_bwd.hip / wgrad.hip
# ---------------------------------------------------------------------------------------------------------------
BWD_STAGES = [(0, 1, 8)] + [(8 + 128 * i, 16, 8) for i in range(7)]
ACT_H = lambda l: 16 * (l - 1) if l <= 4 else 68 + 16 * (l - 5)      # l = 1..7: h0 is not saved (recomputed by the layer_1 wgrad job)
K_ACT_ENC, K_ACT_H7, K_ACT_DIR, K_ACT_BLOCKS = 64, 100, 116, 118
K_DZ_HEAD, K_DZ_BLOCKS = 128, 130


def saved_block_image(frag, blk):
    """[64,8] B-operand fragment -> 512 bf16 in memory order (layout.h saved_off)"""
    img = np.zeros(512, np.float32)
    for l in range(64):
        s, h = l & 31, l >> 5
        off = (2 * (s ^ ((blk & 1) << 2)) + h) * 16
        img[off // 2: off // 2 + 8] = frag[l]
    return img


def act_run(saved):
    """forward_chain's saved dict -> [118*512] memory image of one tile's act run (h1..h4, enc, h5..h7, dir)"""
    run = np.zeros(K_ACT_BLOCKS * 512, np.float32)

    def put(b0, frags):
        for i, f in enumerate(frags):
            run[(b0 + i) * 512:(b0 + i + 1) * 512] = saved_block_image(f, b0 + i)
    for l in range(1, 8):
        put(ACT_H(l), saved["h"][l])
    put(K_ACT_ENC, saved["enc"]); put(K_ACT_DIR, saved["dirc"])
    return run


def mask_block_words(mask_tiles):
    """one layer's relu mask block as the forward stores it (mlp_fwd.hip relu_epi): [64 lanes][4 words]; out tile ot, accumulator
    register i of lane l -> word ot >> 1, bit (ot & 1) * 8 + (i >> 1) + 16 * (i & 1)"""
    w = np.zeros((64, 4), np.uint32)
    for ot, m in enumerate(mask_tiles):
        for i in range(16):
            bit = (ot & 1) * 8 + (i >> 1) + 16 * (i & 1)
            w[:, ot >> 1] |= (m[:, i].astype(np.uint32) << np.uint32(bit))
    return w


def recompute_dz7_frags(zhead_img, bwd_frags, mask_words, wo):
    """wgrad_l7_recompute: D[sample][feature strip wo] = dz_head . H^T (dz_head as the A operand, the dgrad stream's block wo as B),
    masked per (feature, sample) from the forward's mask block, bf16: the two B-operand fragments of output strip wo"""
    z = saved_block_frag(zhead_img, K_DZ_HEAD)
    acc = mfma(z, bwd_frags[wo], np.zeros((64, 16), np.float32))
    for l in range(64):
        c, hh = l & 31, l >> 5
        hf, i = (c >> 2) & 1, (c & 3) + 4 * (c >> 3)
        bit = (wo & 1) * 8 + (i >> 1) + 16 * (i & 1)
        for r in range(16):
            smp = (r & 3) + 8 * (r >> 2) + 4 * hh
            if not (int(mask_words[smp + 32 * hf, wo >> 1]) >> bit) & 1:
                acc[l, r] = 0.0
    return pack_acc(acc)


def backward_chain(bwd_tab, flat, rgb, sigma, drgb, dsigma, masks):
    """Mirror of mlp_bwd_kernel for one wave.  Returns the dz run memory image [130*512]."""
    frags = gather_blocks(bwd_tab, flat)
    run = np.zeros(K_DZ_BLOCKS * 512, np.float32)

    def put(b0, fr):
        for i, f in enumerate(fr):
            run[(b0 + i) * 512:(b0 + i + 1) * 512] = saved_block_image(f, b0 + i)
    zhead = np.zeros((64, 8), np.float32)
    zhead[:32, :3] = drgb * rgb * (1 - rgb)
    zhead[:32, 3] = np.where(sigma > 0, dsigma, 0)
    zhead = round_bf16(zhead)
    put(K_DZ_HEAD, [zhead])

    def stage(st, inputs, mask=None):
        b0, nks, n_ot = BWD_STAGES[st]
        outs = []
        for ot in range(n_ot):
            acc = np.zeros((64, 16), np.float32)
            for ks in range(nks):
                acc = mfma(frags[b0 + ot * nks + ks], inputs[ks], acc)
            if mask is not None:
                acc = np.where(mask[ot], acc, 0)
            lo, hi = pack_acc(acc)
            outs += [lo, hi]
        return outs
    dz = stage(0, [zhead], masks[7])                 # dz7 stays in registers: it is not written (the layer_7 wgrad job recomputes it)
    for st in range(1, 8):
        layer = 6 - (st - 1)
        dz = stage(st, dz, masks[layer]); put(16 * layer, dz)
    return run


def tr_read(img, addr):
    """ds_read_b64_tr_b16 (pinned on hardware by tests/test_gpu_probe.py): img = fp32 view of a bf16 LDS image
    indexed by element, addr [64] byte offsets -> [64,4]"""
    out = np.zeros((64, 4), np.float32)
    for l in range(64):
        g, i = l >> 4, l & 15
        for q in range(4):
            src = 16 * g + 4 * q + (i >> 2)
            out[l, q] = img[addr[src] // 2 + (i & 3)]
    return out


def tr_frag(region, pair, kk, permuted=False):
    """permuted: the sample order sigma(hh, j) = (j&3) + 8(j>>2) + 4hh of an accumulator tile used as an operand
    (wgrad_body.h wgrad_l1_recompute) instead of 8 hh + j"""
    lane_off = np.zeros((2, 64), np.int64)
    for l in range(64):
        grp, il = l >> 4, l & 15
        par, h, q, p = grp & 1, grp >> 1, il >> 2, il & 3
        for r in range(2):
            spos = 4 * (h ^ par) + 8 * r + q if permuted else 8 * h + 4 * (r ^ par) + q
            lane_off[r, l] = par * 1024 + (2 * spos + (p & 1)) * 16 + (p >> 1) * 8
    base = pair * 2048 + kk * 512
    return np.concatenate([tr_read(region, base + lane_off[0]), tr_read(region, base + lane_off[1])], axis=1)


WGRAD_JOBS = {0: (K_ACT_ENC, 2, 0, 8), 1: (K_ACT_ENC, 8, 16, 8), 5: (48, 10, 80, 8), 8: (K_ACT_H7, 9, K_DZ_HEAD, 1)}
for _j in (2, 3, 4, 6, 7):
    WGRAD_JOBS[_j] = (ACT_H(_j - 1), 8, 16 * _j, 8)


def saved_block_frag(img, blk):
    """inverse of saved_block_image: 512 bf16 in memory order -> [64,8] fragment (what a lane reads back with its saved_off)"""
    frag = np.zeros((64, 8), np.float32)
    for l in range(64):
        s, h = l & 31, l >> 5
        off = (2 * (s ^ ((blk & 1) << 2)) + h) * 16
        frag[l] = img[off // 2: off // 2 + 8]
    return frag


def recompute_h0_frags(enc_img, fwd_frags, bias_tab, w_ext):
    """wgrad_l1_recompute: per h0 feature tile, D[sample][feature] = bias + sum_ks enc(ks) . W_0(tile, ks) with enc as the A
    operand; relu, bf16; registers 0..7 / 8..15 are the two A-operand fragments (lane = feature, samples in sigma order)"""
    enc = [saved_block_frag(enc_img[q * 512:(q + 1) * 512], K_ACT_ENC + q) for q in range(4)]
    b = np.where(bias_tab >= 0, w_ext[np.maximum(bias_tab, 0)], np.float32(0)).astype(np.float32).reshape(-1, 32)
    out = []
    for it in range(8):
        acc = np.tile(b[it][R_][:, None], (1, 16)).astype(np.float32)          # every row of a column holds that feature's bias
        for ks in range(4):
            acc = mfma(enc[ks], fwd_frags[4 * it + ks], acc)
        lo, hi = pack_acc(np.maximum(acc, 0))
        out.append((lo, hi))
    return out


def head_expand(w_ext, aux, grad):
    """Mirror of optim.hip head_expand_kernel: aux sums -> += gradients of features / rgb_features / rgb (fp32)"""
    f = np.float32
    o = 63 * 256 + 256 + 4 * (256 * 256 + 256) + (319 * 256 + 256) + 2 * (256 * 256 + 256)
    k_bs = o + 256; k_wf = k_bs + 1; k_bf = k_wf + 256 * 256; k_wr = k_bf + 256; k_br = k_wr + 283 * 128; k_wc = k_br + 128; k_bc = k_wc + 384
    assert k_bc + 3 == N_PARAMS
    Wf = w_ext[k_wf:k_bf].reshape(256, 256); bf = w_ext[k_bf:k_wr]; Wr = w_ext[k_wr:k_br].reshape(283, 128); br = w_ext[k_br:k_wc]
    Wc = w_ext[k_wc:k_bc].reshape(128, 3)
    M = aux[AUX_M:AUX_M + 283 * 3].reshape(283, 3).astype(f); s3 = aux[AUX_S:AUX_S + 3].astype(f)
    P1 = Wr[:256] @ Wc
    Q = Wf.T @ M[:256] + np.outer(bf, s3)
    grad[k_wf:k_bf] += (M[:256] @ P1.T).reshape(-1); grad[k_bf:k_wr] += s3 @ P1.T
    grad[k_wr:k_br] += (np.concatenate([Q, M[256:]], 0) @ Wc.T).reshape(-1); grad[k_br:k_wc] += s3 @ Wc.T
    grad[k_wc:k_bc] += (Wr[:256].T @ Q + Wr[256:].T @ M[256:] + np.outer(br, s3)).reshape(-1); grad[k_bc:k_bc + 3] += s3
    return grad


def wgrad(act_runs, dz_runs, dst_tab, job_off, n_params, w_ext, fwd_tab=None, bias_tab=None, bwd_tab=None, mask7_words=None):
    """Mirror of wgrad_kernel over a list of tiles + head_expand: returns the flat gradient.  Destinations >= n_params
    address the head accumulator (csrc/layout.h kAuxBase).  Job 1 recomputes h0 from enc (fwd_tab / bias_tab needed), job 7
    recomputes dz7 from dz_head and the layer-7 mask block (bwd_tab / mask7_words: one [64,4] word array per tile)."""
    grad = np.zeros(n_params + AUX_COUNT, np.float64)
    fwd_frags = gather_blocks(fwd_tab, w_ext) if fwd_tab is not None else None
    bwd_frags = gather_blocks(bwd_tab, w_ext) if bwd_tab is not None else None
    ones = np.ones((64, 8), np.float32)
    for jb, (ab, n_it, db, n_ot) in WGRAD_JOBS.items():
        dst = dst_tab[job_off[jb]:job_off[jb + 1]].reshape(n_it * 32 + 1, n_ot * 32)
        acc = {(it, ot): np.zeros((64, 16), np.float32) for it in range(n_it + 1) for ot in range(n_ot)}
        for ti, (act, dz) in enumerate(zip(act_runs, dz_runs)):
            in_reg = act[ab * 512:(ab + 2 * n_it) * 512]
            dz_reg = dz[db * 512:(db + 2 * n_ot) * 512]
            h0 = recompute_h0_frags(act[ab * 512:(ab + 4) * 512], fwd_frags, bias_tab, w_ext) if jb == 1 else None
            dz7 = [recompute_dz7_frags(dz[K_DZ_HEAD * 512:(K_DZ_HEAD + 1) * 512], bwd_frags, mask7_words[ti], ot) for ot in range(8)] if jb == 7 else None
            for kk in range(2):
                for ot in range(n_ot):
                    b = dz7[ot][kk] if jb == 7 else tr_frag(dz_reg, ot, kk, permuted=(jb == 1))
                    for it in range(n_it + 1):
                        if it == n_it:
                            a = ones
                        elif jb == 1:
                            a = h0[it][kk]
                        else:
                            a = tr_frag(in_reg, it, kk, permuted=(jb == 7))
                        acc[(it, ot)] = mfma(a, b, acc[(it, ot)])
        for (it, ot), A in acc.items():
            for l in range(64):
                c, hh = l & 31, l >> 5
                if it < n_it:
                    for i in range(16):
                        row = 32 * it + (i & 3) + 8 * (i >> 2) + 4 * hh
                        d = dst[row, 32 * ot + c]
                        if d >= 0:
                            grad[d] += A[l, i]
                elif hh == 0:
                    d = dst[n_it * 32, 32 * ot + c]
                    if d >= 0:
                        grad[d] += A[l, 0]
    return head_expand(w_ext, grad[n_params:].astype(np.float32), grad[:n_params].astype(np.float32))
