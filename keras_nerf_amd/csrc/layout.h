// layout.h -- host-side description of every device data layout of the fused NeRF MLP kernels.
//
// Pure C++ (no HIP): included by the HIP library and exercised on the CPU by tests/test_layout_sim.py through
// knerf_debug_* entry points, where a NumPy lane-level model of v_mfma_f32_32x32x16_bf16 replays the kernels'
// dataflow against the oracle.  Everything the kernels assume about operand order lives here, once.
//
// Architecture covered: NeRFMLP(n_layers=8, dense_units=256, skip_layer=4) with 63-d / 27-d encodings
// (reference keras_nerf/model/nerf/mlp.py:5-50, nerf.py:118-130).  Other shapes are rejected by knerf_create.
//
// MFMA 32x32x16 bf16 operand maps (gfx950):
//   A: lane l (r = l&31, h = l>>5) element j holds A[row r][k = 8h+j]
//   B: lane l (c = l&31, h = l>>5) element j holds B[k = 8h+j][col c]
//   C/D: lane l, reg i holds D[row = (i&3) + 8*(i>>2) + 4*(l>>5)][col = l&31]
// The kernels compute H^T = W^T X^T: rows = features, cols = samples (one sample per lane&31).  A 32x32 f32
// result converted pairwise to bf16 IS the B operand of the next layer (regs 8s..8s+7 -> k-step s), with the
// k order permuted: element j of lane-half h is row 16s + 8(j>>2) + 4h + (j&3).  The packed weights carry the
// same permutation, so no lane ever moves data.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace knerf {

constexpr int kUnits = 256;
constexpr int kLx = 10, kLd = 4;
constexpr int kXyzDim = 63, kDirDim = 27;
constexpr int kNumTensors = 24;

// ---- flat fp32 parameter buffer: Keras trainable_variables order (mlp.py:11-27), kernel[in,out] row-major + bias
struct TensorInfo { int offset, rows, cols; };  // bias: rows = 1
inline std::vector<TensorInfo> tensor_table() {
    const int fi[12] = {63, 256, 256, 256, 256, 319, 256, 256, 256, 256, 283, 128};
    const int fo[12] = {256, 256, 256, 256, 256, 256, 256, 256, 1, 256, 128, 3};
    std::vector<TensorInfo> t;
    int off = 0;
    for (int l = 0; l < 12; ++l) {
        t.push_back({off, fi[l], fo[l]}); off += fi[l] * fo[l];
        t.push_back({off, 1, fo[l]});     off += fo[l];
    }
    return t;
}
constexpr int kParamCount = 595844;
enum Layer { L0 = 0, L1, L2, L3, L4, L5, L6, L7, LSIG, LFEAT, LRF, LRGB };

// ---- collapsed head ------------------------------------------------------------------------------------------------
// In this reference `features` and `rgb_features` are LINEAR Dense layers (mlp.py:21-24,44-46: no activation argument), so
// everything between the trunk output h7 and the sigmoid is one affine map of (h7, dir_enc):
//   rgb_pre = h7 (W_f W_r1 W_c) + dir_enc (W_r2 W_c) + ((b_f W_r1 + b_r) W_c + b_c),   sigma_pre = h7 w_s + b_s
// (W_r1 = rows 0..255 of the rgb_features kernel, W_r2 = its 27 dir rows).  The kernels evaluate that map as ONE out tile
// with 4 real rows (r, g, b, sigma) instead of three dense stages (144 + 72 + 8 MFMAs per 32 samples -> 18), and its
// backward as one 4-channel dZ.  The 283x4 "head" matrix H and its bias are derived from the fp32 master weights after
// every weight change (optim.hip head_compose) and live behind the 595,844 parameters in the same buffer, so the packing
// tables address them like any other tensor.  The gradients of the six head tensors are recovered exactly (chain rule on
// the same identity) from M = [h7 ; dir_enc]^T dz_rgb (283x3) and s = sum dz_rgb, which the wgrad head job accumulates
// in an auxiliary buffer (optim.hip head_expand).
constexpr int kHeadRows = 288;                       // 256 h7 features + 32 dir slots (27 real)
constexpr int kHeadOff = kParamCount;                // H[row][c], c = 0..2 rgb, 3 sigma
constexpr int kHeadBiasOff = kHeadOff + kHeadRows * 4;
constexpr int kExtParamCount = kHeadBiasOff + 4;     // floats in a net's weight buffer
// auxiliary gradient buffer of one net: M[row][c] (row = 0..255 h7, 256..282 dir; c = 0..2), then s[c]
constexpr int kAuxM = 0, kAuxS = 283 * 3, kAuxCount = 864;
constexpr int kAuxBase = kParamCount;                // wgrad destination indices >= kAuxBase address the aux buffer

// ---- slot maps: which reference feature sits in (k-step q, lane-half h, element j) of a B-operand block
// encoded position: 64 slots (4 k-steps).  half 0: x, y, sin(2^i p_c);  half 1: z, pad, cos(2^i p_c).
inline int enc_feature(int q, int h, int j, int L) {
    int m = 8 * q + j;                  // 0..(8*nq-1) within the lane half
    if (m == 0) return h == 0 ? 0 : 2;  // x | z
    if (m == 1) return h == 0 ? 1 : -1; // y | pad
    int i = (m - 2) / 3, c = (m - 2) % 3;
    if (i >= L) return -1;
    return 3 + 6 * i + (h ? 3 : 0) + c; // positional_encoding order: [x, sin(2^0 x), cos(2^0 x), ...] (utils.py:176-186)
}
// hidden activations: slot (q,h,j) of a 256-wide tensor (16 k-steps)
inline int hid_feature(int q, int h, int j) { return 16 * q + 8 * (j >> 2) + 4 * h + (j & 3); }

// ---- description of one fused "dense" stage: list of input k-steps and output tiles
struct KStep { int kind; int q; };   // kind: 0 hidden(prev out), 1 enc, 2 dir, 3 rgb-channels(bwd), 4 sigma(bwd); q = k-step inside that tensor
struct Stage {
    std::vector<KStep> ks;
    int n_ot;                 // output tiles of 32 rows
    // element source: param index for (k-step, half, j, out row) or -1
};

// index into the flat param buffer for kernel tensor `layer` at [in_row][out_col], -1 when out of range
inline int kidx(const std::vector<TensorInfo>& tt, int layer, int in_row, int out_col) {
    const TensorInfo& k = tt[2 * layer];
    if (in_row < 0 || out_col < 0 || in_row >= k.rows || out_col >= k.cols) return -1;
    return k.offset + in_row * k.cols + out_col;
}
inline int bidx(const std::vector<TensorInfo>& tt, int layer, int out_col) {
    const TensorInfo& b = tt[2 * layer + 1];
    if (out_col < 0 || out_col >= b.cols) return -1;
    return b.offset + out_col;
}

// =====================================================================================================
// forward stream: 978 blocks of 1 KiB (64 lanes x 8 bf16).  Order: stage, out tile, k-step.
// stage list: L0 | L1..L4 | L5 (16 hidden + 4 enc k-steps) | L6 L7 | HEAD (16 h7 + 2 dir k-steps, 1 tile: rows r,g,b,sigma)
// =====================================================================================================
constexpr int kFwdBlocks = 32 + 4 * 128 + 160 + 2 * 128 + 18;   // 978
constexpr int kFwdBiasTiles = 8 * 8 + 1;                         // 65 tiles of 32 fp32

struct PackTables {
    std::vector<int32_t> fwd;       // kFwdBlocks*512 entries: index into the extended weight buffer or -1
    std::vector<int32_t> fwd_bias;  // kFwdBiasTiles*32
    std::vector<int32_t> bwd;       // kBwdBlocks*512
};

// row of the head matrix for k-step ks of the HEAD stage: h7 features, then the dir encoding slots
inline int head_in_row(int ks, int h, int j) {
    if (ks < 16) return hid_feature(ks, h, j);
    const int e = enc_feature(ks - 16, h, j, kLd);
    return e < 0 ? -1 : 256 + e;
}
inline int hidx(int row, int c) { return (row < 0 || row >= 283 || c < 0 || c > 3) ? -1 : kHeadOff + row * 4 + c; }

// input feature (row of the layer's kernel) for forward stage `st` k-step `ks`, half h, element j
inline int fwd_in_row(int st, int ks, int h, int j) {
    switch (st) {
        case 0: return enc_feature(ks, h, j, kLx);                       // layer_0: 63 inputs in 4 k-steps
        case 5: return ks < 16 ? hid_feature(ks, h, j)                   // layer_5: [h(256), xyz_enc(63)] (mlp.py:36-38)
                               : (enc_feature(ks - 16, h, j, kLx) < 0 ? -1 : 256 + enc_feature(ks - 16, h, j, kLx));
        case 8: return head_in_row(ks, h, j);
        default: return hid_feature(ks, h, j);                           // 256-wide
    }
}
struct FwdStage { int layer; int nks; int n_ot; };
// stage index: 0..7 trunk, 8 = HEAD (layer = -1: addresses the head matrix)
inline FwdStage fwd_stage(int st) {
    switch (st) {
        case 0: return {L0, 4, 8};
        case 5: return {L5, 20, 8};
        case 8: return {-1, 18, 1};
        default: return {st, 16, 8};
    }
}
constexpr int kFwdStages = 9;

inline void build_fwd(PackTables& pt) {
    auto tt = tensor_table();
    pt.fwd.assign((size_t)kFwdBlocks * 512, -1);
    pt.fwd_bias.assign((size_t)kFwdBiasTiles * 32, -1);
    size_t blk = 0; int btile = 0;
    for (int st = 0; st < kFwdStages; ++st) {
        FwdStage s = fwd_stage(st);
        for (int ot = 0; ot < s.n_ot; ++ot) {
            for (int r = 0; r < 32; ++r)
                pt.fwd_bias[(size_t)btile * 32 + r] = s.layer < 0 ? (r < 4 ? kHeadBiasOff + r : -1) : bidx(tt, s.layer, 32 * ot + r);
            ++btile;
            for (int ks = 0; ks < s.nks; ++ks, ++blk) {
                for (int l = 0; l < 64; ++l) {
                    int r = l & 31, h = l >> 5;
                    for (int j = 0; j < 8; ++j)
                        pt.fwd[blk * 512 + l * 8 + j] = s.layer < 0 ? hidx(fwd_in_row(st, ks, h, j), r)
                                                                     : kidx(tt, s.layer, fwd_in_row(st, ks, h, j), 32 * ot + r);
                }
            }
        }
    }
}

// =====================================================================================================
// backward (dgrad) stream: A = W (rows = the layer's INPUT features, k = its OUTPUT features).
// stage order (reverse of forward):
//   B0: dz_head (1 k-step: channels r,g,b,sigma at half 0, j<4) -> dh7 (8 tiles)    H[h7 feature][channel]
//   B1..B7: dz_l (16) -> dh_{l-1} (8 tiles) for l = 7,6,5,4,3,2,1   (layer 5: rows 0..255 of its 319)
// =====================================================================================================
constexpr int kBwdBlocks = 8 + 7 * 128;   // 904
constexpr int kBwdStages = 8;
struct BwdStage { int nks; int n_ot; };
inline BwdStage bwd_stage(int st) { return st == 0 ? BwdStage{1, 8} : BwdStage{16, 8}; }
inline void build_bwd(PackTables& pt) {
    auto tt = tensor_table();
    pt.bwd.assign((size_t)kBwdBlocks * 512, -1);
    size_t blk = 0;
    for (int st = 0; st < kBwdStages; ++st) {
        BwdStage s = bwd_stage(st);
        for (int ot = 0; ot < s.n_ot; ++ot)
            for (int ks = 0; ks < s.nks; ++ks, ++blk)
                for (int l = 0; l < 64; ++l) {
                    int r = l & 31, h = l >> 5, row = 32 * ot + r;   // row = input feature of the layer
                    for (int j = 0; j < 8; ++j) {
                        int v = -1;
                        if (st == 0) v = hidx(row, h == 0 && j < 4 ? j : -1);
                        else v = kidx(tt, L7 - (st - 1), row, hid_feature(ks, h, j));
                        pt.bwd[blk * 512 + l * 8 + j] = v;
                    }
                }
    }
}

// =====================================================================================================
// saved tensors (training): per sample tile (32 samples) a run of 1 KiB B-operand blocks; lane (h, s) of block b
// stores its 8 bf16 at byte offset saved_off(b, h, s) = (2*(s ^ 4*(b&1)) + h) * 16: the two feature halves of a
// sample are adjacent (32 B per sample) and odd blocks rotate their sample quads, which makes the wgrad kernel's
// ds_read_b64_tr_b16 transposed reads bank-conflict free while the store stays one coalesced 1 KiB write.
// forward "act" run (118 blocks):  h1 h2 h3 h4 enc h5 h6 h7 dir   -- h0 is NOT saved: layer_0 has only 64 input slots, so the
//   layer_1 wgrad job recomputes h0 = relu(W_0 enc + b_0) from the 4 enc blocks (4 MFMAs per wave and tile) instead of reading
//   16 blocks that the forward would have had to write (wgrad_body.h wgrad_l1_recompute)
// backward "dz" run  (130 blocks): dz0 .. dz7  dz_head(2: channels r,g,b,sigma in the first block, the second stays zero);
//   the 16 blocks of dz7 are RESERVED BUT NEVER WRITTEN: dz7 = mask7 * (H dz_head) has 4 input channels, so the layer_7 wgrad
//   job recomputes its 32-column strip with one MFMA per wave and tile (wgrad_body.h wgrad_l7_recompute)
// so that every wgrad job reads ONE contiguous range of each:  e.g. layer_5: act[h4..enc] x dz5, head: act[h7..dir] x dz_head.
// =====================================================================================================
constexpr int saved_off(int b, int h, int s) { return (2 * (s ^ ((b & 1) << 2)) + h) * 16; }
constexpr int kActH1 = 0, kActH4 = 48, kActEnc = 64, kActH5 = 68, kActH7 = 100, kActDir = 116, kActBlocks = 118;
constexpr int act_h(int l) { return l <= 4 ? 16 * (l - 1) : 68 + 16 * (l - 5); }      // l = 1..7 (h0 is not saved)
constexpr int kDzHead = 128, kDzBlocks = 130;
constexpr int kMaskBlocks = 8;   // relu masks: one 1 KiB block per trunk layer per tile (16 B per lane = 128 bits)
// Byte stride between consecutive sample tiles of each saved run.  All waves of the chip write the same block of their
// own tile at about the same time, so a stride that is a large power of two times a small odd number (e.g. 156 KiB =
// 2^12 * 39) concentrates those writes on a few memory channels; an odd number of 256-B units spreads them.
#ifndef KNERF_TILE_SKEW
#define KNERF_TILE_SKEW 256
#endif
// KNERF_SAVED_GROUP = G (a power of two, default 1): G consecutive tiles form a group that is stored BLOCK-major -- block b of the
// group's G tiles is one contiguous G KiB piece, so the 8 waves of a chain workgroup (G = 8) write 8 KiB per block instead of
// eight 1 KiB pieces 118 KiB apart.  G = 1 is the tile-major layout described above.  (r03 A/B experiment)
#ifndef KNERF_SAVED_GROUP
#define KNERF_SAVED_GROUP 1
#endif
constexpr int kSavedGroup = KNERF_SAVED_GROUP;
static_assert(kSavedGroup >= 1 && (kSavedGroup & (kSavedGroup - 1)) == 0, "KNERF_SAVED_GROUP must be a power of two");
constexpr int kSavedBlockStride = kSavedGroup * 1024;        // bytes between consecutive blocks of one tile
constexpr size_t kActTileBytes = (size_t)kActBlocks * 1024 + KNERF_TILE_SKEW;      // per-tile footprint at G = 1; G x blocks + skew per group
constexpr size_t kDzTileBytes = (size_t)kDzBlocks * 1024 + KNERF_TILE_SKEW;
constexpr size_t kMaskTileBytes = (size_t)kMaskBlocks * 1024 + KNERF_TILE_SKEW;
constexpr size_t saved_group_bytes(int blocks) { return (size_t)kSavedGroup * blocks * 1024 + KNERF_TILE_SKEW; }
// byte offset of block 0 of tile `tile` in a saved run of `blocks` blocks per tile
constexpr size_t saved_tile_off(size_t tile, int blocks) {
    return kSavedGroup == 1 ? tile * ((size_t)blocks * 1024 + KNERF_TILE_SKEW)
                            : (tile / kSavedGroup) * saved_group_bytes(blocks) + (tile % kSavedGroup) * 1024;
}
constexpr size_t act_tile_off(size_t tile) { return saved_tile_off(tile, kActBlocks); }
constexpr size_t dz_tile_off(size_t tile) { return saved_tile_off(tile, kDzBlocks); }
constexpr size_t mask_tile_off(size_t tile) { return saved_tile_off(tile, kMaskBlocks); }
// bytes of a region of `tiles` tiles (whole groups)
constexpr size_t saved_region_bytes(size_t tiles, int blocks) { return (tiles + kSavedGroup - 1) / kSavedGroup * saved_group_bytes(blocks); }

// wgrad jobs: dW[in_row][out_col] += sum_s act[s][in] * dz[s][out], db[out_col] += sum_s dz[s][out]
struct WgradJob {
    int act_blk, n_it;    // first act block, number of 32-row input tiles (2 blocks each)
    int dz_blk, n_ot;     // first dz block, number of 32-col output tiles
    int layer;            // destination kernel/bias; -1: the head job (sigma column -> LSIG, rgb columns -> aux buffer)
};
constexpr int kWgradJobs = 9;
constexpr int kHeadJob = 8;
inline WgradJob wgrad_job(int j) {
    switch (j) {
        case 0: return {kActEnc, 2, 0, 8, L0};
        case 5: return {kActH4, 10, 16 * 5, 8, L5};             // [h4 ; enc]
        case 8: return {kActH7, 9, kDzHead, 1, -1};             // [h7 ; dir] x (r,g,b,sigma)
        case 1: return {kActEnc, 8, 16, 8, L1};                 // h0 recomputed from enc; the table rows are h0 features
        default: return {act_h(j - 1), 8, 16 * j, 8, j};        // layers 2-4, 6, 7
    }
}
// in_row of job jb for tile-row index tr (0..32*n_it-1) in *natural tr-read order*: the transposed read un-permutes
// hidden tensors (row = feature), and presents enc/dir blocks in slot order (q = tr>>4, c16 = tr&15 ->
// h = (c16>>2)&1, j = 4*(c16>>3) + (c16&3)).
inline int slot_from_c16_h(int c16) { return (c16 >> 2) & 1; }
inline int slot_from_c16_j(int c16) { return 4 * (c16 >> 3) + (c16 & 3); }
inline int wgrad_in_row(int jb, int tr) {
    auto encrow = [](int tr_, int L) { int q = tr_ >> 4, c = tr_ & 15; return enc_feature(q, slot_from_c16_h(c), slot_from_c16_j(c), L); };
    switch (jb) {
        case 0: return encrow(tr, kLx);
        case 5: return tr < 256 ? tr : (encrow(tr - 256, kLx) < 0 ? -1 : 256 + encrow(tr - 256, kLx));
        case 8: return tr < 256 ? tr : (encrow(tr - 256, kLd) < 0 ? -1 : 256 + encrow(tr - 256, kLd));
        default: return tr;
    }
}
// destination of wgrad output element (tile-row tr, column tc): index into the flat gradient, kAuxBase + index into the
// aux buffer, or -1; tr == -2 selects the bias row (column sums of dz)
inline int wgrad_dst(const std::vector<TensorInfo>& tt, int jb, int tr, int tc) {
    WgradJob J = wgrad_job(jb);
    if (J.layer < 0) {                       // head: columns 0..2 = dz_rgb, 3 = dz_sigma
        if (tc > 3) return -1;
        if (tr == -2) return tc == 3 ? bidx(tt, LSIG, 0) : kAuxBase + kAuxS + tc;
        const int row = wgrad_in_row(jb, tr);
        if (row < 0) return -1;
        if (tc == 3) return row < 256 ? kidx(tt, LSIG, row, 0) : -1;
        return kAuxBase + kAuxM + row * 3 + tc;
    }
    if (tr == -2) return bidx(tt, J.layer, tc);
    return kidx(tt, J.layer, wgrad_in_row(jb, tr), tc);
}

}  // namespace knerf
