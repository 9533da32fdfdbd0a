// layout.h -- host-side description of every device data layout of the fused NeRF MLP kernels.
//
// Pure C++ (no HIP): included by the HIP library and exercised on the CPU by tests/test_layout_sim.py through
// knerf_debug_* entry points, where a NumPy lane-level model of v_mfma_f32_32x32x16_bf16 replays the kernels'
// dataflow against the oracle.  Everything the kernels assume about operand order lives here, once.
//
// Architecture covered: NeRFMLP(n_layers, dense_units=256, skip_layer) with 63-d / 27-d encodings (reference
// keras_nerf/model/nerf/mlp.py:5-50, nerf.py:118-130), see Shape below; the reference's default is Shape<8, 4>.
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

constexpr int kLx = 10, kLd = 4;               // the reference's defaults (nerf.py:11-14): 63-d / 27-d encodings
constexpr int kMaxLd = 8;                      // largest pos_emb_dir of a fused shape: 51-d direction encoding, four head k-steps (kAuxS, optim.hip)
constexpr int kMaxLx = 16;                     // largest pos_emb_xyz: 99-d position encoding, eight k-steps

// ---- trunk shape ---------------------------------------------------------------------------------------------------------
// The fused kernels cover NeRFMLP(n_layers = NL, dense_units = 256, skip_layer = SK) with 63-d / 27-d encodings (mlp.py:5-50):
// layer l (l >= 2) takes [h_{l-1} ; xyz_enc] when the reference concatenated behind layer l-1, i.e. (l-1) % SK == 0 (mlp.py:36-38).
// Shape<8, 4> is the reference's default and the shape every number in DESIGN.md is quoted for; the other instantiations
// (knerf_api.hip kFusedShapes) share every line of kernel code with it.  Not covered (-> general-shape path, generic.hip): other
// widths, fewer than 3 layers.  A concat behind the LAST layer ((NL - 1) % SK == 0: 9 / 4, 5 / 4, 5 / 2, 7 / 3 ...; mlp.py:36-38 has no
// "not the last layer" condition) makes the trunk output [h ; xyz_enc]: the sigma and features kernels have U + kXyzDim rows and the
// composed head takes [h ; xyz_enc ; dir_enc] (round 6: kTrunkX / kTrunkXQ below).
// ShapeImpl carries every constant; Shape<NL, SK, U> = the reference's encodings (the names the built-in kernels are mangled with),
// ShapeL<NL, SK, U, LX, LD> = other positional-encoding depths (NeRF(pos_emb_xyz=LX, pos_emb_dir=LD), build-time entries only).
template <int NL_, int SK_, int U_, int LX_, int LD_>
struct ShapeImpl {
    static constexpr int NL = NL_, SK = SK_, U = U_, LX = LX_, LD = LD_;
    static constexpr int kKs = U / 16, kOt = U / 32;        // k-steps / out tiles of a U-wide layer: 16 / 8 at 256, 8 / 4 at 128, 4 / 2 at 64
    // encodings: 3 + 6 L features in slots of 16 per k-step (8 per lane half: x, y | z, pad, then sin | cos of 2^i p_c), the k-step
    // count rounded up to EVEN -- the weight-gradient kernel takes its inputs in tiles of 32 rows = two blocks
    static constexpr int kXyzDim = 3 + 6 * LX, kDirDim = 3 + 6 * LD;
    static constexpr int enc_q(int L) { return ((2 + 3 * L + 7) / 8 + 1) / 2 * 2; }
    static constexpr int kEncQ = enc_q(LX), kDirQ = enc_q(LD);          // 4 and 2 for the reference's L = 10 / 4
    static constexpr bool concat_in(int l) { return l >= 2 && l < NL && (l - 1) % SK == 0; }
    static constexpr bool kConcatBehindLast = (NL - 1) % SK == 0 && NL - 1 > 0;
    // the trunk's output behind its last layer: h, and xyz_enc once more when the reference concatenates behind that layer too
    static constexpr int kTrunkX = kConcatBehindLast ? kXyzDim : 0, kTrunkXQ = kConcatBehindLast ? kEncQ : 0;   // extra rows / head k-steps
    static constexpr int kTrunkOut = U + kTrunkX;            // rows of the sigma and features kernels (mlp.py:19-22 built on the concat)
    static constexpr bool kSupported = NL >= 3 && NL <= 16 && SK >= 1 && (U == 256 || U == 128 || U == 64) &&
                                       LX >= 1 && LX <= kMaxLx && LD >= 1 && LD <= kMaxLd;  // LX, LD: the head accumulator holds U + 99 + 51 rows (kAuxS)
    static constexpr int first_concat() { for (int l = 2; l < NL; ++l) if (concat_in(l)) return l; return 0; }
    static constexpr int kFirstConcat = first_concat();
    // flat fp32 parameter buffer: Keras trainable_variables order (mlp.py:11-27), kernel[in,out] row-major + bias; behind the trunk
    // sigma [U,1], features [U,U], rgb_features [U+27, U/2], rgb [U/2, 3]
    static constexpr int fan_in(int l) { return l == 0 ? kXyzDim : (concat_in(l) ? U + kXyzDim : U); }
    static constexpr int trunk_params() { int n = 0; for (int l = 0; l < NL; ++l) n += fan_in(l) * U + U; return n; }
    static constexpr int kTrunkParams = trunk_params();      // = offset of the sigma kernel
    static constexpr int kHeadReal = U + kDirDim;            // rows of the rgb_features kernel = real rows of the composed head matrix
    static constexpr int kParamCount = kTrunkParams + (kTrunkOut + 1) + (kTrunkOut * U + U) + (kHeadReal * (U / 2) + U / 2) + ((U / 2) * 3 + 3);
    static constexpr int kHeadMatReal = kTrunkOut + kDirDim;  // real rows of the COMPOSED head matrix: h, [xyz_enc,] dir_enc
    // forward stream: stage st = trunk layer st (st < NL) or the head (st == NL); order: stage, out tile, k-step
    static constexpr int kFwdStages = NL + 1;
    static constexpr int fwd_nks(int st) { return st == 0 ? kEncQ : st == NL ? kKs + kTrunkXQ + kDirQ : (concat_in(st) ? kKs + kEncQ : kKs); }
    static constexpr int fwd_not(int st) { return st == NL ? 1 : kOt; }
    static constexpr int fwd_b0(int st) { int n = 0; for (int q = 0; q < st; ++q) n += fwd_nks(q) * fwd_not(q); return n; }
    static constexpr int kFwdBlocks = fwd_b0(NL + 1);
    static constexpr int kFwdBiasTiles = kOt * NL + 1;       // tiles of 32 fp32: kOt per trunk layer, 1 for the head
    // backward (dgrad) stream: stage 0 = the head (1 k-step x kOt tiles), stage q = layer NL-q (kKs x kOt), q = 1 .. NL-1
    static constexpr int kBwdStages = NL;
    static constexpr int bwd_b0(int st) { return st == 0 ? 0 : kOt + (st - 1) * kKs * kOt; }
    static constexpr int kBwdBlocks = bwd_b0(NL);
    // Two recomputations instead of saved tensors exist for the 256-wide trunk only (their weight-gradient jobs are written for 8
    // waves = 8 column strips): h0 from the encoding (layer_1's job) and the last layer's dZ from dz_head.  At width 128 the forward
    // saves h0 and dgrad writes the last dZ like any other; the h0 job also assumes the reference's four encoding blocks.
    static constexpr bool kSaveH0 = U != 256 || kEncQ != 4;
    // saved runs (see "saved tensors" below).  act: [h0] h1 .. h_{NL-1} (kKs blocks each), the 4 enc blocks directly behind
    // h_{c-1} for the FIRST concat layer c (its weight-gradient job then reads ONE contiguous range; later concat layers read two), in
    // front of everything when there is no concat layer, and the 2 dir blocks at the end, directly behind h_{NL-1} (the head job's
    // range) -- with a SECOND copy of the enc blocks between the two when the head takes [h ; xyz_enc ; dir_enc] (kActHeadEnc: the head
    // job still reads one contiguous range).  dz: dz_0 .. dz_{NL-1} (kKs each), dz_head (2).
    static constexpr int kFirstSaved = kSaveH0 ? 0 : 1;
    static constexpr int kActEnc = kFirstConcat ? kKs * (kFirstConcat - kFirstSaved) : 0;
    static constexpr int act_h(int l) { return kKs * (l - kFirstSaved) + ((kFirstConcat == 0 || l >= kFirstConcat) ? kEncQ : 0); }   // l = kFirstSaved .. NL-1
    static constexpr int kActHeadEnc = kKs * (NL - kFirstSaved) + kEncQ;      // = act_h(NL - 1) + kKs
    static constexpr int kActDir = kActHeadEnc + kTrunkXQ, kActBlocks = kActDir + kDirQ;
    static constexpr int kDzHead = kKs * NL, kDzBlocks = kDzHead + 2;
    // dz of the last trunk layer is mask * (H dz_head) with 4 input channels: at width 256 its weight-gradient job recomputes it
    // (wgrad_body.h wgrad_last_recompute) and dgrad does not write it -- unless that layer takes [h ; xyz_enc] (then the job is the
    // plain two-range one and dgrad writes the block run like any other)
    static constexpr bool kSaveLastDz = concat_in(NL - 1) || U != 256;
    static constexpr int kMaskBlocks = NL;                   // relu masks: one 1 KiB block per trunk layer per tile (16 B per lane = 128 bits)
    static constexpr int kWgradJobs = NL + 1, kHeadJob = NL; // job j = trunk layer j; the last one = the head
    // collapsed head (below): U h features [+ 16 kEncQ xyz slots] + 16 kDirQ dir slots (32, 27 of them real, for LD = 4); the matrix
    // is stored by REAL row (kHeadMatReal of them), the slot count only sizes its region
    static constexpr int kHeadRows = U + 16 * kTrunkXQ + 16 * kDirQ;
    static constexpr int kHeadOff = kParamCount;             // H[row][c], c = 0..2 rgb, 3 sigma
    static constexpr int kHeadBiasOff = kHeadOff + kHeadRows * 4;
    static constexpr int kExtParamCount = kHeadBiasOff + 4;  // floats in a net's weight buffer
    static constexpr int kAuxBase = kParamCount;             // wgrad destination indices >= kAuxBase address the aux buffer
    static_assert(kSupported, "trunk shape not covered by the fused kernels");
};
template <int NL_, int SK_, int U_ = 256> struct Shape : ShapeImpl<NL_, SK_, U_, kLx, kLd> {};
template <int NL_, int SK_, int U_, int LX_, int LD_> struct ShapeL : ShapeImpl<NL_, SK_, U_, LX_, LD_> {};
using DefaultShape = Shape<8, 4>;
// entry arguments -> type: (NL, SK, U) -> Shape<...>, (NL, SK, U, LX, LD) -> ShapeL<...>
#define KNERF_SHAPE_SEL(a1, a2, a3, a4, a5, NAME, ...) NAME
#define KNERF_SHAPE_T(...) KNERF_SHAPE_SEL(__VA_ARGS__, ShapeL, ShapeBadArity, Shape, ShapeBadArity, ShapeBadArity)<__VA_ARGS__>
// The trunk shapes the library is built for, X(index, n_layers, skip_layer, dense_units [, pos_emb_xyz, pos_emb_dir]); index 0 is the reference's default.  Every entry costs
// one more instantiation of the three big kernels: build.py compiles mlp_fwd / mlp_bwd / wgrad once per entry with
// -DKNERF_SHAPE_SLICE=<index>, and a translation unit built that way defines the kernels of its own shape only (explicit
// instantiation; `extern template` for the others) -- slice 0 also holds the run-time dispatchers.
#define KNERF_BUILTIN_SHAPES(X) X(0, 8, 4, 256) X(1, 8, 2, 256) X(2, 6, 3, 256) X(3, 4, 2, 256) X(4, 12, 4, 256) X(5, 8, 3, 256) X(6, 8, 5, 256) \
    X(7, 6, 2, 256) X(8, 6, 4, 256) X(9, 10, 5, 256) X(10, 8, 4, 128) X(11, 4, 2, 128) X(12, 8, 4, 64) X(13, 4, 2, 64)
// Further entries chosen at BUILD time: `python keras_nerf_amd/build.py --add-shape=NL,SK,U[,LX,LD] ...` defines KNERF_EXTRA_SHAPES(X) as
// X(14, NL, SK, U) X(15, NL, SK, U, LX, LD) ... (indices continue the built-in list).  A triple the
// kernels do not cover fails to compile on Shape's static_assert; one that repeats an earlier entry is never selected.
#ifndef KNERF_EXTRA_SHAPES
#define KNERF_EXTRA_SHAPES(X)
#endif
#define KNERF_FUSED_SHAPES(X) KNERF_BUILTIN_SHAPES(X) KNERF_EXTRA_SHAPES(X)
#define KNERF_X(I, ...) +1
constexpr int kNumBuiltinShapes = 0 KNERF_BUILTIN_SHAPES(KNERF_X);
constexpr int kNumFusedShapes = 0 KNERF_FUSED_SHAPES(KNERF_X);
#undef KNERF_X
static_assert(kNumBuiltinShapes == 14, "keras_nerf_amd/build.py N_BUILTIN_SHAPES");
// index of a shape in that list, -1 when the fused kernels do not cover it (-> general-shape path)
template <class S> constexpr bool shape_is(int n_layers, int skip_layer, int dense_units, int lx, int ld) {
    return n_layers == S::NL && skip_layer == S::SK && dense_units == S::U && lx == S::LX && ld == S::LD;
}
constexpr int fused_shape_id(int n_layers, int skip_layer, int dense_units = 256, int pos_emb_xyz = kLx, int pos_emb_dir = kLd) {
#define KNERF_X(I, ...) if (shape_is<KNERF_SHAPE_T(__VA_ARGS__)>(n_layers, skip_layer, dense_units, pos_emb_xyz, pos_emb_dir)) return I;
    KNERF_FUSED_SHAPES(KNERF_X)
#undef KNERF_X
    return -1;
}
// KNERF_PICK(I, DEF, EXT) -> DEF when this translation unit owns shape I (no slicing: every shape), EXT otherwise.  build.py compiles
// a sliced source with -DKNERF_SHAPE_SLICE=<k> AND -DKNERF_OWN_<k>=, (a lone comma): `KNERF_OWN_##I 1, 0` then starts with a comma for the
// owned index only, which shifts the argument that KNERF_PICK_2ND returns from 0 to 1 -- any number of shapes, no per-index macro
// (rounds 2-3 carried 24 hand-expanded KNERF_SLICE_n blocks here).
#ifndef KNERF_SHAPE_SLICE
#define KNERF_PICK(I, DEF, EXT) DEF
#define KNERF_HAS_DISPATCH 1
#else
#define KNERF_PICK_2ND(a, b, ...) b
#define KNERF_PICK_SEL(...) KNERF_PICK_2ND(__VA_ARGS__)
#define KNERF_PICK_1(DEF, EXT) DEF
#define KNERF_PICK_0(DEF, EXT) EXT
#define KNERF_PICK_CAT(a, b) a##b
#define KNERF_PICK_GO(bit) KNERF_PICK_CAT(KNERF_PICK_, bit)
#define KNERF_PICK(I, DEF, EXT) KNERF_PICK_GO(KNERF_PICK_SEL(KNERF_OWN_##I 1, 0))(DEF, EXT)
#define KNERF_HAS_DISPATCH (KNERF_SHAPE_SLICE == 0)
#endif
constexpr int kParamCount = DefaultShape::kParamCount;      // 595,844: knerf_param_count()
static_assert(kParamCount == 595844 && DefaultShape::kFwdBlocks == 978 && DefaultShape::kBwdBlocks == 904 && DefaultShape::kActBlocks == 118 &&
              DefaultShape::kActEnc == 64 && DefaultShape::act_h(5) == 68 && DefaultShape::kDzBlocks == 130, "the default shape's layout");

struct TensorInfo { int offset, rows, cols; };  // bias: rows = 1
template <class S>
inline std::vector<TensorInfo> tensor_table() {
    std::vector<TensorInfo> t;
    int off = 0;
    auto add = [&](int fi, int fo) { t.push_back({off, fi, fo}); off += fi * fo; t.push_back({off, 1, fo}); off += fo; };
    for (int l = 0; l < S::NL; ++l) add(S::fan_in(l), S::U);
    add(S::kTrunkOut, 1); add(S::kTrunkOut, S::U); add(S::kHeadReal, S::U / 2); add(S::U / 2, 3);       // sigma, features, rgb_features, rgb
    return t;
}
// index of a tensor in the table: trunk layer l -> l; then sigma, features, rgb_features, rgb
template <class S> constexpr int LSIG = S::NL;

// ---- collapsed head ------------------------------------------------------------------------------------------------
// In this reference `features` and `rgb_features` are LINEAR Dense layers (mlp.py:21-24,44-46: no activation argument), so
// everything between the trunk output h and the sigmoid is one affine map of (h, dir_enc):
//   rgb_pre = h (W_f W_r1 W_c) + dir_enc (W_r2 W_c) + ((b_f W_r1 + b_r) W_c + b_c),   sigma_pre = h w_s + b_s
// (W_r1 = rows 0..255 of the rgb_features kernel, W_r2 = its 27 dir rows).  The kernels evaluate that map as ONE out tile
// with 4 real rows (r, g, b, sigma) instead of three dense stages (144 + 72 + 8 MFMAs per 32 samples -> 18), and its
// backward as one 4-channel dZ.  The 283x4 "head" matrix H and its bias are derived from the fp32 master weights after
// every weight change (optim.hip head_compose) and live behind the parameters in the same buffer (Shape::kHeadOff), so the
// packing tables address them like any other tensor.  The gradients of the six head tensors are recovered exactly (chain rule on
// the same identity) from M = [h ; dir_enc]^T dz_rgb (283x3) and s = sum dz_rgb, which the wgrad head job accumulates
// in an auxiliary buffer (optim.hip head_expand).
// auxiliary gradient buffer of one net: M[row][c] (row = 0..U-1 h, [then the 3 + 6 LX xyz rows of a trunk that ends in a concat,] then
// the 3 + 6 LD dir rows; c = 0..2), then s[c] at kAuxS (sized for U = 256, LX = kMaxLx, LD = kMaxLd)
constexpr int kMaxDirDim = 3 + 6 * kMaxLd, kMaxXyzDim = 3 + 6 * kMaxLx;        // 51, 99
constexpr int kAuxM = 0, kAuxS = (256 + kMaxXyzDim + kMaxDirDim) * 3, kAuxCount = 1224;      // 1218 + 3 sums, rounded up

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
// forward stream: 1 KiB blocks (64 lanes x 8 bf16).  Order: stage, out tile, k-step.  Default shape, 978 blocks:
// L0 | L1..L4 | L5 (16 hidden + 4 enc k-steps) | L6 L7 | HEAD (16 h7 + 2 dir k-steps, 1 tile: rows r,g,b,sigma)
// =====================================================================================================
struct PackTables {
    std::vector<int32_t> fwd;       // kFwdBlocks*512 entries: index into the extended weight buffer or -1
    std::vector<int32_t> fwd_bias;  // kFwdBiasTiles*32
    std::vector<int32_t> bwd;       // kBwdBlocks*512
};

// row of the head matrix for k-step ks of the HEAD stage: h features, [the xyz encoding slots of a trunk that ends in a concat,] then
// the dir encoding slots
template <class S>
inline int head_in_row(int ks, int h, int j) {
    if (ks < S::kKs) return hid_feature(ks, h, j);
    if (ks < S::kKs + S::kTrunkXQ) {
        const int e = enc_feature(ks - S::kKs, h, j, S::LX);
        return e < 0 ? -1 : S::U + e;
    }
    const int e = enc_feature(ks - S::kKs - S::kTrunkXQ, h, j, S::LD);
    return e < 0 ? -1 : S::kTrunkOut + e;
}
template <class S> inline int hidx(int row, int c) { return (row < 0 || row >= S::kHeadMatReal || c < 0 || c > 3) ? -1 : S::kHeadOff + row * 4 + c; }

// input feature (row of the layer's kernel) for forward stage `st` k-step `ks`, half h, element j
template <class S>
inline int fwd_in_row(int st, int ks, int h, int j) {
    if (st == 0) return enc_feature(ks, h, j, S::LX);                        // layer_0: 63 inputs in 4 k-steps (LX = 10)
    if (st == S::NL) return head_in_row<S>(ks, h, j);
    if (ks < S::kKs) return hid_feature(ks, h, j);                           // U-wide
    const int e = enc_feature(ks - S::kKs, h, j, S::LX);                     // concat layer: [h(U), xyz_enc] (mlp.py:36-38)
    return e < 0 ? -1 : S::U + e;
}

template <class S>
inline void build_fwd(PackTables& pt) {
    auto tt = tensor_table<S>();
    pt.fwd.assign((size_t)S::kFwdBlocks * 512, -1);
    pt.fwd_bias.assign((size_t)S::kFwdBiasTiles * 32, -1);
    size_t blk = 0; int btile = 0;
    for (int st = 0; st < S::kFwdStages; ++st) {
        const bool head = st == S::NL;
        for (int ot = 0; ot < S::fwd_not(st); ++ot) {
            for (int r = 0; r < 32; ++r)
                pt.fwd_bias[(size_t)btile * 32 + r] = head ? (r < 4 ? S::kHeadBiasOff + r : -1) : bidx(tt, st, 32 * ot + r);
            ++btile;
            for (int ks = 0; ks < S::fwd_nks(st); ++ks, ++blk) {
                for (int l = 0; l < 64; ++l) {
                    int r = l & 31, h = l >> 5;
                    for (int j = 0; j < 8; ++j)
                        pt.fwd[blk * 512 + l * 8 + j] = head ? hidx<S>(fwd_in_row<S>(st, ks, h, j), r)
                                                             : kidx(tt, st, fwd_in_row<S>(st, ks, h, j), 32 * ot + r);
                }
            }
        }
    }
}

// =====================================================================================================
// backward (dgrad) stream: A = W (rows = the layer's INPUT features, k = its OUTPUT features).
// stage order (reverse of forward):
//   B0: dz_head (1 k-step: channels r,g,b,sigma at half 0, j<4) -> dh_{NL-1} (8 tiles)    H[h feature][channel]
//   Bq: dz_l (16) -> dh_{l-1} (8 tiles) for l = NL-q, q = 1 .. NL-1   (a concat layer: rows 0..255 of its 319)
// =====================================================================================================
template <class S>
inline void build_bwd(PackTables& pt) {
    auto tt = tensor_table<S>();
    pt.bwd.assign((size_t)S::kBwdBlocks * 512, -1);
    size_t blk = 0;
    for (int st = 0; st < S::kBwdStages; ++st) {
        const int nks = st == 0 ? 1 : S::kKs;
        for (int ot = 0; ot < S::kOt; ++ot)
            for (int ks = 0; ks < nks; ++ks, ++blk)
                for (int l = 0; l < 64; ++l) {
                    int r = l & 31, h = l >> 5, row = 32 * ot + r;   // row = input feature of the layer
                    for (int j = 0; j < 8; ++j) {
                        int v = -1;
                        if (st == 0) v = hidx<S>(row, h == 0 && j < 4 ? j : -1);
                        else v = kidx(tt, S::NL - st, row, hid_feature(ks, h, j));
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
// forward "act" run (default shape, 118 blocks):  h1 h2 h3 h4 enc h5 h6 h7 dir   -- h0 is NOT saved: layer_0 has only 64 input slots, so the
//   layer_1 wgrad job recomputes h0 = relu(W_0 enc + b_0) from the 4 enc blocks (4 MFMAs per wave and tile) instead of reading
//   16 blocks that the forward would have had to write (wgrad_body.h wgrad_l1_recompute)
// backward "dz" run  (default shape, 130 blocks): dz0 .. dz7  dz_head(2: channels r,g,b,sigma in the first block, the second stays zero);
//   the 16 blocks of dz7 are RESERVED BUT NEVER WRITTEN: dz7 = mask7 * (H dz_head) has 4 input channels, so the layer_7 wgrad
//   job recomputes its 32-column strip with one MFMA per wave and tile (wgrad_body.h wgrad_last_recompute)
// so that every wgrad job reads ONE contiguous range of each:  e.g. layer_5: act[h4..enc] x dz5, head: act[h7..dir] x dz_head.
// =====================================================================================================
constexpr int saved_off(int b, int h, int s) { return (2 * (s ^ ((b & 1) << 2)) + h) * 16; }
// relu-mask blocks (one per trunk layer and tile; 16 B = 128 mask bits per forward lane): lane (feature half hf = lane >> 5, sample
// s = lane & 31) keeps its 16 bytes at mask_lane_off = lane * 16.  The layer-7 weight-gradient job reads one mask word per (sample
// row, hf) with ds_read_b32 -- two addresses per 32-lane group, 512 B apart = the same bank of 32: 2-way, 61.5 % of the kernel's LDS
// bank-conflict cycles (tests/test_lds_bank_model.py, profiles/r05_wgrad_lds_conflicts.json).  KNERF_MASK_LAYOUT=1 puts a sample's two
// halves ADJACENT (32 B per sample, as in the saved activation blocks; forward store and dgrad load stay whole 1 KiB blocks per wave):
// those reads become conflict-free (share 7.7 % -> 3.1 % by PMC) -- and the fine launch came out 0.4-1.0 % SLOWER in every pairing of
// two A/B calls (profiles/r05_mask_layout_ab.json): measured, not adopted, kept as a build knob.
#ifndef KNERF_MASK_LAYOUT
#define KNERF_MASK_LAYOUT 0
#endif
constexpr int mask_lane_off(int lane) { return KNERF_MASK_LAYOUT ? ((lane & 31) * 2 + (lane >> 5)) * 16 : lane * 16; }
// byte offset inside a mask block of word w of (sample s, feature half hf)
constexpr int mask_word_off(int s, int hf, int w) { return mask_lane_off(hf * 32 + s) + w * 4; }
// (block offsets of the runs: Shape::kActEnc, act_h(l), kActDir, kActBlocks, kDzHead, kDzBlocks, kMaskBlocks)
// Byte stride between consecutive sample tiles of each saved run.  All waves of the chip write the same block of their
// own tile at about the same time, so a stride that is a large power of two times a small odd number (e.g. 156 KiB =
// 2^12 * 39) concentrates those writes on a few memory channels; an odd number of 256-B units spreads them.
#ifndef KNERF_TILE_SKEW
#define KNERF_TILE_SKEW 256
#endif
// KNERF_SAVED_GROUP = G (a power of two, default 1): G consecutive tiles form a group that is stored BLOCK-major -- block b of the
// group's G tiles is one contiguous G KiB piece, so the 8 waves of a chain workgroup (G = 8) write 8 KiB per block instead of
// eight 1 KiB pieces 118 KiB apart.  G = 1 is the tile-major layout described above.  (r03 A/B experiment, DESIGN.md 5.0: no gain.)
#ifndef KNERF_SAVED_GROUP
#define KNERF_SAVED_GROUP 1
#endif
constexpr int kSavedGroup = KNERF_SAVED_GROUP;
static_assert(kSavedGroup >= 1 && (kSavedGroup & (kSavedGroup - 1)) == 0, "KNERF_SAVED_GROUP must be a power of two");
constexpr int kSavedBlockStride = kSavedGroup * 1024;        // bytes between consecutive blocks of one tile
constexpr size_t saved_group_bytes(int blocks) { return (size_t)kSavedGroup * blocks * 1024 + KNERF_TILE_SKEW; }
// byte offset of block 0 of tile `tile` in a saved run of `blocks` blocks per tile
constexpr size_t saved_tile_off(size_t tile, int blocks) {
    return kSavedGroup == 1 ? tile * ((size_t)blocks * 1024 + KNERF_TILE_SKEW)
                            : (tile / kSavedGroup) * saved_group_bytes(blocks) + (tile % kSavedGroup) * 1024;
}
template <class S> constexpr size_t act_tile_off(size_t tile) { return saved_tile_off(tile, S::kActBlocks); }
template <class S> constexpr size_t dz_tile_off(size_t tile) { return saved_tile_off(tile, S::kDzBlocks); }
template <class S> constexpr size_t mask_tile_off(size_t tile) { return saved_tile_off(tile, S::kMaskBlocks); }
// bytes of a region of `tiles` tiles (whole groups)
constexpr size_t saved_region_bytes(size_t tiles, int blocks) { return (tiles + kSavedGroup - 1) / kSavedGroup * saved_group_bytes(blocks); }

// wgrad jobs: dW[in_row][out_col] += sum_s act[s][in] * dz[s][out], db[out_col] += sum_s dz[s][out].  Job j = trunk layer j, job NL
// = the head.  Kinds (wgrad_body.h picks the body): 0 first layer (enc x dz_0), 1 layer_1 with h0 recomputed from enc, 2 plain
// 256 x 256, 3 concat layer ([h ; enc] x dz: input tiles 0..7 from act_blk, 8..9 from act_blk2), 4 last layer with its dz recomputed,
// 5 head ([h ; dir] x dz_head).
struct WgradJob {
    int kind;
    int act_blk, act_blk2, n_it;   // first act block (of input tiles 0..7 / of tiles 8..), number of 32-row input tiles (2 blocks each)
    int dz_blk, n_ot;              // first dz block, number of 32-col output tiles
    int layer;                     // destination kernel/bias; -1: the head job (sigma column -> LSIG, rgb columns -> aux buffer)
};
template <class S>
constexpr WgradJob wgrad_job(int j) {
    constexpr int K = S::kKs, T = S::kOt;
    if (j == 0) return {0, S::kActEnc, S::kActEnc, S::kEncQ / 2, 0, T, 0};
    if (j == S::NL) return {5, S::act_h(S::NL - 1), S::act_h(S::NL - 1), T + (S::kTrunkXQ + S::kDirQ) / 2, S::kDzHead, 1, -1};   // [h ; (enc ;) dir] x (r,g,b,sigma): one range behind h
    if (j == 1 && !S::kSaveH0) return {1, S::kActEnc, S::kActEnc, T, K, T, 1};                        // h0 recomputed: the table rows are h0 features
    if (S::concat_in(j)) return {3, S::act_h(j - 1), S::kActEnc - K, T + S::kEncQ / 2, K * j, T, j};             // input tile T + k -> block act_blk2 + K + 2k
    if (j == S::NL - 1 && !S::kSaveLastDz) return {4, S::act_h(j - 1), S::act_h(j - 1), T, K * j, T, j};
    return {2, S::act_h(j - 1), S::act_h(j - 1), T, K * j, T, j};
}
// floats of the largest job table of a shape ((32 n_it + 1) rows x 32 n_ot columns): the per-workgroup slab of the deterministic
// mode (wgrad_body.h flush_acc, wgrad.hip wgrad_reduce_kernel).  82,176 = layer_5's (320 + 1) x 256 for the default shape; a width-256
// shape with pos_emb_xyz >= 11 has 11 or 12 input tiles in its concat job (a fixed stride overflowed there).
template <class S>
constexpr int wgrad_partial_stride() {
    int m = 0;
    for (int j = 0; j < S::kWgradJobs; ++j) {
        const WgradJob J = wgrad_job<S>(j);
        const int e = (J.n_it * 32 + 1) * J.n_ot * 32;
        if (e > m) m = e;
    }
    return m;
}
static_assert(wgrad_partial_stride<DefaultShape>() == (10 * 32 + 1) * 256, "the default shape's largest job table is layer_5's");
// in_row of job jb for tile-row index tr (0..32*n_it-1) in *natural tr-read order*: the transposed read un-permutes
// hidden tensors (row = feature), and presents enc/dir blocks in slot order (q = tr>>4, c16 = tr&15 ->
// h = (c16>>2)&1, j = 4*(c16>>3) + (c16&3)).
inline int slot_from_c16_h(int c16) { return (c16 >> 2) & 1; }
inline int slot_from_c16_j(int c16) { return 4 * (c16 >> 3) + (c16 & 3); }
template <class S>
inline int wgrad_in_row(int jb, int tr) {
    auto encrow = [](int tr_, int L) { int q = tr_ >> 4, c = tr_ & 15; return enc_feature(q, slot_from_c16_h(c), slot_from_c16_j(c), L); };
    const int kind = wgrad_job<S>(jb).kind;
    if (kind == 0) return encrow(tr, S::LX);
    if (kind == 3) return tr < S::U ? tr : (encrow(tr - S::U, S::LX) < 0 ? -1 : S::U + encrow(tr - S::U, S::LX));
    if (kind == 5) {
        if (tr < S::U) return tr;
        if (tr < S::U + 16 * S::kTrunkXQ) return encrow(tr - S::U, S::LX) < 0 ? -1 : S::U + encrow(tr - S::U, S::LX);
        const int e = encrow(tr - S::U - 16 * S::kTrunkXQ, S::LD);
        return e < 0 ? -1 : S::kTrunkOut + e;
    }
    return tr;
}
// destination of wgrad output element (tile-row tr, column tc): index into the flat gradient, kAuxBase + index into the
// aux buffer, or -1; tr == -2 selects the bias row (column sums of dz)
template <class S>
inline int wgrad_dst(const std::vector<TensorInfo>& tt, int jb, int tr, int tc) {
    const WgradJob J = wgrad_job<S>(jb);
    if (J.layer < 0) {                       // head: columns 0..2 = dz_rgb, 3 = dz_sigma
        if (tc > 3) return -1;
        if (tr == -2) return tc == 3 ? bidx(tt, LSIG<S>, 0) : S::kAuxBase + kAuxS + tc;
        const int row = wgrad_in_row<S>(jb, tr);
        if (row < 0) return -1;
        if (tc == 3) return row < S::kTrunkOut ? kidx(tt, LSIG<S>, row, 0) : -1;
        return S::kAuxBase + kAuxM + row * 3 + tc;
    }
    if (tr == -2) return bidx(tt, J.layer, tc);
    return kidx(tt, J.layer, wgrad_in_row<S>(jb, tr), tc);
}

// ---- run-time view of a shape (host): what knerf_api.hip needs to size buffers and pick instantiations
struct ShapeInfo {
    int id, n_layers, skip, units, lx, ld, dir_dim, dir_slots;
    int trunk_x, trunk_x_slots;     // rows / slots of xyz_enc in the trunk's output (a concat behind the last layer), else 0
    int param_count, ext_param_count, trunk_params, head_off, head_bias_off;
    int fwd_blocks, fwd_bias_tiles, bwd_blocks;
    int act_blocks, dz_blocks, mask_blocks;
    int n_jobs;
    int job_kind[17];
    int partial_stride;             // wgrad_partial_stride<S>(): floats per workgroup slab of the deterministic mode
};
template <class S>
inline ShapeInfo make_shape_info(int id) {
    ShapeInfo i{};
    i.id = id; i.n_layers = S::NL; i.skip = S::SK; i.units = S::U; i.lx = S::LX; i.ld = S::LD; i.dir_dim = S::kDirDim; i.dir_slots = 16 * S::kDirQ;
    i.trunk_x = S::kTrunkX; i.trunk_x_slots = 16 * S::kTrunkXQ;
    i.param_count = S::kParamCount; i.ext_param_count = S::kExtParamCount; i.trunk_params = S::kTrunkParams;
    i.head_off = S::kHeadOff; i.head_bias_off = S::kHeadBiasOff;
    i.fwd_blocks = S::kFwdBlocks; i.fwd_bias_tiles = S::kFwdBiasTiles; i.bwd_blocks = S::kBwdBlocks;
    i.act_blocks = S::kActBlocks; i.dz_blocks = S::kDzBlocks; i.mask_blocks = S::kMaskBlocks;
    i.n_jobs = S::kWgradJobs;
    for (int j = 0; j < S::kWgradJobs; ++j) i.job_kind[j] = wgrad_job<S>(j).kind;
    i.partial_stride = wgrad_partial_stride<S>();
    return i;
}
inline const ShapeInfo& shape_info(int id) {
    static const ShapeInfo all[kNumFusedShapes] = {
#define KNERF_X(I, ...) make_shape_info<KNERF_SHAPE_T(__VA_ARGS__)>(I),
        KNERF_FUSED_SHAPES(KNERF_X)
#undef KNERF_X
    };
    return all[id >= 0 && id < kNumFusedShapes ? id : 0];
}

}  // namespace knerf
