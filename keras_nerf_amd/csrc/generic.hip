// generic.hip -- general-shape NeRFMLP forward / backward on gfx950 (see generic.h).
//
// Reference semantics restated: NeRFMLP.call (mlp.py:29-50): relu Dense x n_layers with [h ; xyz_enc] concatenated after
// layer i when i % skip == 0 and i > 0; sigma = relu(Dense(1)); features = Dense(units); [features ; dir_enc];
// rgb_features = Dense(units/2) (linear); rgb = sigmoid(Dense(3)).  Inputs: NeRFUtils.encode_position_and_directions
// (utils.py:188-210).  Backward is the chain rule of that graph (tape.gradient at nerf.py:370-377).
//
// Layout: every activation is a row-major bf16 matrix [Mp][ld] (one row per ray sample, Mp = samples rounded up to 128,
// every segment padded to a multiple of 32 columns with zeros), so that both MFMA operands of `out = in . W` are
// K-contiguous 16-byte loads: A fragment = 8 consecutive features of a sample, B fragment = 8 consecutive input weights
// of an output unit (weights are packed transposed, Wt[out][in]).  dgrad uses the same kernel with Wd[in][out].
// wgrad contracts over samples (the row index of both operands).  Layers of at least 128 x 128: a workgroup stages [32][256]
// slabs of X and dZ in LDS and reads sample-major operand fragments with ds_read_b64_tr_b16 (wgrad_coop_kernel).  Smaller
// layers: each 32x32 block is transposed in registers by two MFMAs against identity fragments -- D = A.[I|0] + A'.[0|I]
// leaves lane = feature, registers = samples, an MFMA operand with a fixed sample permutation shared by both factors --
// so neither LDS nor a workgroup barrier is involved (wgrad_kernel).
#include "generic.h"
#include "kernels.h"

#include <cmath>

namespace knerf {
namespace gen {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned short u16;
typedef __attribute__((ext_vector_type(4))) short s16x4g;

namespace {

constexpr int r32(int v) { return (v + 31) / 32 * 32; }

__device__ __forceinline__ u16 to_bf16(float v) {
    const __bf16 b = (__bf16)v;                       // round to nearest even (v_cvt_pk_bf16_f32)
    return __builtin_bit_cast(u16, b);
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// ---- positional encoding (utils.py:176-210): rows [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)] --------
// One thread per (sample row, group of 8 columns): a row's groups are adjacent threads, so the 16-byte stores of a wavefront cover
// whole rows back to back (round 5; rounds 2-4 had one thread per sample writing its 192 bytes two at a time: 7 % of a train chunk).
// Values as before: sinf / cosf of 2^l x with the scaling exact.
__device__ __forceinline__ float enc_column(int col, int dim, const float (&v)[3]) {
    if (col < 3) return v[col];
    if (col >= dim) return 0.f;
    const int k = col - 3, l = k / 6, r = k % 6;
    const float arg = __int_as_float((127 + l) << 23) * v[r % 3];          // 2^l * x (l <= 32)
    return r < 3 ? sinf(arg) : cosf(arg);
}
__global__ __launch_bounds__(256) void encode_kernel(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ t,
                                                    long long n, long long mp, int S, int lx, int ld, u16* __restrict__ ex, int kxp,
                                                    u16* __restrict__ ed, int kdp) {
    const int gx = kxp / 8, gd = kdp / 8, gpr = gx + gd;                    // 16-byte groups per row: xyz buffer, dir buffer
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= mp * gpr) return;
    const long long m = idx / gpr;
    const int grp = (int)(idx % gpr);
    const bool is_dir = grp >= gx;
    u16* dst = is_dir ? ed + (size_t)m * kdp + (size_t)(grp - gx) * 8 : ex + (size_t)m * kxp + (size_t)grp * 8;
    u16 out[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (m < n) {
        const long long ray = m / S;
        float v[3];
        if (is_dir) {
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = d[ray * 3 + c];
        } else {
            const float tv = t[m];
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = __fadd_rn(o[ray * 3 + c], __fmul_rn(d[ray * 3 + c], tv));   // o + d*t, separate mul and add like the reference
        }
        const int col0 = (is_dir ? grp - gx : grp) * 8, dim = 3 + 6 * (is_dir ? ld : lx);
#pragma unroll
        for (int j = 0; j < 8; ++j) out[j] = to_bf16(enc_column(col0 + j, dim, v));
    }
    uint4 pk;
    __builtin_memcpy(&pk, out, 16);
    *reinterpret_cast<uint4*>(dst) = pk;
}

// already-encoded fp32 rows [n][dim] -> zero-padded bf16 rows [mp][ld]   (NeRFMLP.__call__ on encoded inputs, mlp.py:29-31)
__global__ __launch_bounds__(256) void convert_rows_kernel(const float* __restrict__ src, int dim, long long n, long long mp, u16* __restrict__ dst, int ld) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= mp * ld) return;
    const long long m = i / ld;
    const int c = (int)(i % ld);
    dst[i] = (m < n && c < dim) ? to_bf16(src[m * dim + c]) : (u16)0;
}

// dst[:, 0:cols) = src[:, 0:cols)   (cols multiple of 8, 16-byte granules): the concat halves
__global__ __launch_bounds__(256) void copy_cols_kernel(const u16* __restrict__ src, int lds, u16* __restrict__ dst, int ldd, long long rows, int cols) {
    const int per_row = cols / 8;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * per_row) return;
    const long long r = i / per_row;
    const int c = (int)(i % per_row) * 8;
    *reinterpret_cast<uint4*>(dst + (size_t)r * ldd + c) = *reinterpret_cast<const uint4*>(src + (size_t)r * lds + c);
}

// ---- C[M][N] = A[M][K] . Bt[N][K]^T (+bias) (relu) (* relu bits) ------------------------------------------------------
struct GemmArgs {
    const u16* A; int lda;
    const u16* Bt; int ldb;
    long long M; int N, K;
    const float* bias; int n_real;       // bias[col] for col < n_real (fp32 master weights), else 0
    int relu;
    // relu bits of a trunk layer's output, one bit per (row, feature): [32-row tile][feature / 8][32 bytes], the 32 bytes of a
    // (tile, feature octet) in the order the MFMA accumulator holds the tile's rows -- byte 16 hh + i is row (i & 3) + 8 (i >> 2)
    // + 4 hh -- so that a lane's sixteen rows are ONE 16-byte access.  The forward GEMM writes them (mask_out; 1/16 of the
    // activation's bytes), the dgrad GEMM of the same features reads them (mask_in) instead of the saved bf16 activation: a
    // third of that GEMM's traffic, and its only load with a use right behind it
    unsigned char* mask_out; const unsigned char* mask_in; int mask_cols;      // mask_cols = padded units / 8
    u16* Cb; int ldc;                    // bf16 output (may be null)
    float* Cf; int ldcf;                 // fp32 output (may be null)
    // dead-tile skipping (backward only): the 32-row tiles to process = list entries [0, *n_live) (composite.hip appends the tiles
    // whose dL/d(rgb, sigma) is not all zero); rows of other tiles are neither read nor written.  null: every tile of M
    const int* live; const int* n_live;
};

// epilogue shared by the two GEMM kernels: bias, relu, relu mask of the dgrad, bf16 / fp32 stores.
// Column assignment: column r of the column block's tile t is output feature n0 + NT r + t (the kernels stage weight row
// n0 + NT r + t at LDS row 32 t + r), NOT n0 + 32 t + r: lane r then holds NT CONSECUTIVE features of a sample row across its NT
// accumulators, so a row's values leave as one 2 NT-byte store per lane -- for NT = 8 one instruction writes two whole 512-byte
// rows, 16 store instructions per 32-row tile instead of 128 two-byte ones that each touch two rows (the epilogue used to issue as
// many stores as the tile has MFMAs).  The launch-uniform switches (fp32 or bf16 output, mask or not) are taken ONCE around the loops.
template <int NT> struct PackedRow;                    // NT bf16 values as one store
template <> struct PackedRow<1> { typedef u16 type; };
template <> struct PackedRow<2> { typedef unsigned int type; };
template <> struct PackedRow<4> { typedef __attribute__((ext_vector_type(2))) unsigned int type; };
template <> struct PackedRow<8> { typedef __attribute__((ext_vector_type(4))) unsigned int type; };
template <int NT>
__device__ __forceinline__ typename PackedRow<NT>::type pack_row(const float (&v)[NT], unsigned& bits) {
    u16 hw[NT];
    bits = 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) { hw[t] = to_bf16(v[t]); bits |= ((short)hw[t] > 0 ? 1u : 0u) << t; }     // the bit = what a reader of the STORED value would decide
    typename PackedRow<NT>::type out;
    __builtin_memcpy(&out, hw, sizeof(out));
    return out;
}
// this lane's relu bits of the tile at row m0 (dgrad): requested at the top of the tile, used in its epilogue
template <int NT>
__device__ __forceinline__ uint4 relu_bits_load(const GemmArgs& g, long long m0, int n0, int r, int h) {
    return *reinterpret_cast<const uint4*>(g.mask_in + ((size_t)(m0 >> 5) * g.mask_cols + ((n0 + NT * r) >> 3)) * 32 + 16 * h);
}
template <int NT>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, const f32x16 (&acc)[NT], long long m0, int n0, int r, int h, uint4 mk) {
    // D[row][col]: lane (col = r, hh = h), register i -> row (i&3) + 8(i>>2) + 4hh; this lane's features: f0 .. f0 + NT - 1
    const size_t row0 = (size_t)(m0 + 4 * h);
    const int f0 = n0 + NT * r;
    const unsigned sh = f0 & 7;                    // where this lane's NT bits sit in their octet
    float b[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) b[t] = (g.bias && f0 + t < g.n_real) ? g.bias[f0 + t] : 0.f;
    if (g.Cf) {                                   // fp32 head outputs (no mask, no relu)
        float* p = g.Cf + row0 * g.ldcf + f0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float* q = p + (size_t)((i & 3) + 8 * (i >> 2)) * g.ldcf;
#pragma unroll
            for (int t = 0; t < NT; ++t) q[t] = acc[t][i] + b[t];
        }
    } else if (g.mask_in) {                       // dgrad: dz = [relu bit] * acc
        typedef typename PackedRow<NT>::type P;
        u16* p = g.Cb + row0 * g.ldc + f0;
        const unsigned w4[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const unsigned bits = w4[i >> 2] >> (8 * (i & 3) + sh);
            float v[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) v[t] = (bits >> t) & 1u ? acc[t][i] : 0.f;
            unsigned unused;
            *reinterpret_cast<P*>(p + (size_t)((i & 3) + 8 * (i >> 2)) * g.ldc) = pack_row<NT>(v, unused);
        }
    } else {
        typedef typename PackedRow<NT>::type P;
        const bool relu = g.relu != 0;
        u16* p = g.Cb + row0 * g.ldc + f0;
        unsigned w4[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float v[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) { v[t] = acc[t][i] + b[t]; v[t] = relu ? fmaxf(v[t], 0.f) : v[t]; }
            unsigned bits;
            *reinterpret_cast<P*>(p + (size_t)((i & 3) + 8 * (i >> 2)) * g.ldc) = pack_row<NT>(v, bits);
            w4[i >> 2] |= bits << (8 * (i & 3) + sh);
        }
        if (g.mask_out) {
            // an octet's bits sit in 8 / NT neighbouring lanes: OR them together, the first lane of the group stores
#pragma unroll
            for (int o = 1; o < 8 / NT; o <<= 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) w4[q] |= (unsigned)__shfl_xor((int)w4[q], o, 64);
            }
            if ((r & (8 / NT - 1)) == 0)
                *reinterpret_cast<uint4*>(g.mask_out + ((size_t)(m0 >> 5) * g.mask_cols + (f0 >> 3)) * 32 + 16 * h) = uint4{w4[0], w4[1], w4[2], w4[3]};
        }
    }
}

// Workgroup = 8 waves x 32 rows; the Bt slab of the current 64-wide K chunk ([32 NT][64] bf16) is staged in LDS once per
// workgroup (rows padded to 144 B: the 16-byte fragment reads of 16 lanes then cover all 64 banks exactly once) and
// double-buffered, so the weights cross L2 once per 256 rows instead of once per 32; A fragments come straight from
// global memory (each wave owns its rows) and are fetched one chunk ahead.
constexpr int kGemmWaves = 8, kGemmKC = 64, kGemmRowB = 144;

template <int NT>
__global__ __launch_bounds__(kGemmWaves * 64) void gemm_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    constexpr int NB = NT * 32;                       // staged Bt rows
    constexpr int BUF = NB * kGemmRowB;               // bytes per buffer
    constexpr int PER_T = (NB * 8 + kGemmWaves * 64 - 1) / (kGemmWaves * 64);   // 16-byte granules per thread per chunk
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long long my_tile = (long long)blockIdx.x * kGemmWaves + wave;
    const bool live = g.live ? my_tile < (long long)*g.n_live : my_tile * 32 < g.M;     // M is a multiple of 128, the workgroup covers 256 rows
    long long m0 = live ? (g.live ? (long long)g.live[my_tile] : my_tile) * 32 : 0;    // idle waves still help staging and hit the barriers
    const int n0 = blockIdx.y * NB;
    const u16* A = g.A + (size_t)(m0 + r) * g.lda + 8 * h;
    const int nchunks = (g.K + kGemmKC - 1) / kGemmKC;
    uint4 mk = uint4{0, 0, 0, 0};
    if (g.mask_in && live) mk = relu_bits_load<NT>(g, m0, n0, r, h);

    uint4 stage[PER_T];
    auto fetch_b = [&](int kc) {                      // global -> registers: granule q = (row, 16-byte column chunk)
#pragma unroll
        for (int p = 0; p < PER_T; ++p) {
            const int q = p * kGemmWaves * 64 + tid;
            const int row = q >> 3, cc = q & 7;
            const int k = kc * kGemmKC + cc * 8;
            // LDS row 32 t + r holds weight row n0 + NT r + t (column assignment of gemm_epilogue)
            stage[p] = (row < NB && k < g.K) ? *reinterpret_cast<const uint4*>(g.Bt + (size_t)(n0 + NT * (row & 31) + (row >> 5)) * g.ldb + k) : uint4{0, 0, 0, 0};
        }
    };
    auto put_b = [&](int buf) {
#pragma unroll
        for (int p = 0; p < PER_T; ++p) {
            const int q = p * kGemmWaves * 64 + tid;
            const int row = q >> 3, cc = q & 7;
            if (row < NB) *reinterpret_cast<uint4*>(gsm + buf * BUF + row * kGemmRowB + cc * 16) = stage[p];
        }
    };
    bf16x8 a_cur[4], a_nxt[4];
    auto fetch_a = [&](int kc, bf16x8 (&a)[4]) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int k = kc * kGemmKC + ks * 16;
            if (k < g.K) a[ks] = *reinterpret_cast<const bf16x8*>(A + k);
        }
    };
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = zero16();
    fetch_b(0); fetch_a(0, a_cur);
    put_b(0);
    __syncthreads();
    for (int kc = 0; kc < nchunks; ++kc) {
        const int buf = kc & 1;
        const bool more = kc + 1 < nchunks;
        if (more) { fetch_b(kc + 1); fetch_a(kc + 1, a_nxt); }
        const char* B = gsm + buf * BUF + r * kGemmRowB + h * 16;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (kc * kGemmKC + ks * 16 < g.K) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const bf16x8 b = *reinterpret_cast<const bf16x8*>(B + t * 32 * kGemmRowB + ks * 32);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[ks], b, acc[t], 0, 0, 0);
                }
            }
        }
        if (more) {
            put_b(buf ^ 1);                           // the other buffer was last read before the previous barrier
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) a_cur[ks] = a_nxt[ks];
        }
        __syncthreads();
    }
    if (!live) return;
    gemm_epilogue<NT>(g, acc, m0, n0, r, h, mk);
}

// Weights-stationary variant for the common case that a whole Bt slab [32 NT][K] fits in LDS: it is staged ONCE per workgroup
// (K zero-padded to a multiple of 64; row pitch 2 K64 + 16 bytes: pitch/4 = 4 (odd) -> the 16-byte fragment reads of 16
// lanes hit 64 distinct banks), the workgroup is persistent over row tiles and its waves never synchronise again: each
// wave streams its own rows (A fragments one 64-wide chunk ahead, across tile boundaries) against the resident weights.
// The chunk body is branch-free and fully unrolled with the B fragments read 4 MFMAs ahead (left to itself hipcc emits
// ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma per fragment: 48 cycles per MFMA instead of 8).
// Column blocks (N wider than one slab: widths above 256, or K = 320 behind a concat) re-read the SAME rows of A once each.  The grid is
// one-dimensional and maps workgroup w to (row group bx, column block by) such that the gy column blocks of a row group are
// neighbours in one XCD's dispatch order (workgroups are dealt round-robin over the 8 XCDs: w and w + 8 share one): they walk the
// same row tiles at the same pace, so after the first of them the rows come from that XCD's L2 instead of HBM.  gx is a multiple
// of 8 whenever gy > 1 (launch_gemm_ws).  Placement is a speed matter only; nothing depends on it.
template <int NT>
__global__ __launch_bounds__(kGemmWaves * 64) void gemm_ws_kernel(GemmArgs g, int pitch, int gx, int gy) {
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    constexpr int NB = NT * 32;
    constexpr int NF = 4 * NT;                        // fragments per 64-wide chunk
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    int bx = blockIdx.x, by = 0;
    if (gy > 1) {
        const int xcd = bx & 7, local = bx >> 3;
        by = local % gy;
        bx = (local / gy) * 8 + xcd;
    }
    const int n0 = by * NB;
    const int K64 = (g.K + 63) / 64 * 64;
    const int gpr = K64 / 8;                          // 16-byte granules per staged row
    for (int q = tid; q < NB * gpr; q += kGemmWaves * 64) {
        const int row = q / gpr, cc = q % gpr;
        *reinterpret_cast<uint4*>(gsm + row * pitch + cc * 16) =
            cc * 8 < g.K ? *reinterpret_cast<const uint4*>(g.Bt + (size_t)(n0 + NT * (row & 31) + (row >> 5)) * g.ldb + cc * 8) : uint4{0, 0, 0, 0};   // gemm_epilogue's column assignment
    }
    __syncthreads();
    const long long n_tiles = g.live ? (long long)*g.n_live : g.M / 32;
    const char* B = gsm + r * pitch + h * 16;
    const long long stride = (long long)gx * kGemmWaves;
    long long tile = (long long)bx * kGemmWaves + wave;
    auto row_tile = [&](long long i) -> long long { return g.live ? (long long)g.live[i] : i; };      // entry i of the tiles to process
    bf16x8 a_cur[4], a_nxt[4];
    auto fetch_a = [&](long long t, int k, bf16x8 (&a)[4]) {          // one 64-wide chunk of this wave's 32 rows (t = a row tile)
        const u16* A = g.A + (size_t)(t * 32 + r) * g.lda + 8 * h + k;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (k + ks * 16 < g.K) a[ks] = *reinterpret_cast<const bf16x8*>(A + ks * 16);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) a[ks][j] = (__bf16)0.f;     // K tail: the staged weights there are zero as well
            }
        }
    };
    auto frag = [&](int k, int f) { return *reinterpret_cast<const bf16x8*>(B + (f % NT) * 32 * pitch + (k + (f / NT) * 16) * 2); };
    long long rt = tile < n_tiles ? row_tile(tile) : 0, rt_next = 0;
    if (tile < n_tiles) fetch_a(rt, 0, a_cur);
    for (; tile < n_tiles; tile += stride, rt = rt_next) {
        const long long m0 = rt * 32;
        if (tile + stride < n_tiles) rt_next = row_tile(tile + stride);
        uint4 mk = uint4{0, 0, 0, 0};
        if (g.mask_in) mk = relu_bits_load<NT>(g, m0, n0, r, h);
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = zero16();
        bf16x8 pf[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) pf[f] = frag(0, f);
        for (int k = 0; k < K64; k += kGemmKC) {
            const bool more = k + kGemmKC < K64;
            const bool next_tile = !more && tile + stride < n_tiles;
            if (more) fetch_a(rt, k + kGemmKC, a_nxt);
            else if (next_tile) fetch_a(rt_next, 0, a_nxt);
            const int kn = more ? k + kGemmKC : k;          // where the ring's look-ahead reads at the end of this chunk
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const bf16x8 cur = pf[f % 4];
                pf[f % 4] = f + 4 < NF ? frag(k, f + 4) : frag(kn, f + 4 - NF);
                acc[f % NT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[f / NT], cur, acc[f % NT], 0, 0, 0);
            }
            if (more || next_tile) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) a_cur[ks] = a_nxt[ks];
            }
        }
        gemm_epilogue<NT>(g, acc, m0, n0, r, h, mk);
    }
}

constexpr size_t kGemmWsLds = 150 * 1024;
template <int NT>
hipError_t launch_gemm_ws(const GemmArgs& g, int gy, hipStream_t s) {
    const int pitch = (g.K + 63) / 64 * 64 * 2 + 16;
    const size_t lds = (size_t)NT * 32 * pitch;
    static AttrOnce once;
    hipError_t ae = once([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ws_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGemmWsLds); });
    if (ae != hipSuccess) return ae;
    long long gx = (g.M / 32 + kGemmWaves - 1) / kGemmWaves;
    const long long cap = 512 / gy > 0 ? 512 / gy : 1;           // persistent: about two workgroups per CU in all
    if (gx > cap) gx = cap;
    if (gy > 1) gx = gx >= 8 ? gx / 8 * 8 : 8;                   // the XCD mapping of the kernel wants whole groups of eight row groups
    hipLaunchKernelGGL((gemm_ws_kernel<NT>), dim3((unsigned)(gx * gy)), dim3(kGemmWaves * 64), lds, s, g, pitch, (int)gx, gy);
    return hipGetLastError();
}

template <int NT>
hipError_t launch_gemm_nt(const GemmArgs& g, int gy, hipStream_t s) {
    const size_t lds = 2 * (size_t)NT * 32 * kGemmRowB;
    static AttrOnce once;
    if (lds > 64 * 1024) {
        hipError_t ae = once([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
        if (ae != hipSuccess) return ae;
    }
    const unsigned gx = (unsigned)((g.M + kGemmWaves * 32 - 1) / (kGemmWaves * 32));
    hipLaunchKernelGGL((gemm_kernel<NT>), dim3(gx, gy), dim3(kGemmWaves * 64), lds, s, g);
    return hipGetLastError();
}

hipError_t launch_gemm(const GemmArgs& g, hipStream_t s) {
    const int nt = g.N / 32;
    const size_t row_bytes = (size_t)(g.K + 63) / 64 * 64 * 2 + 16;
#ifndef KNERF_GEN_NO_WS
    // the widest column block whose weight slab fits in LDS, resident for the whole launch
    if (nt % 8 == 0 && 256 * row_bytes <= kGemmWsLds) return launch_gemm_ws<8>(g, nt / 8, s);
    if (nt % 4 == 0 && 128 * row_bytes <= kGemmWsLds) return launch_gemm_ws<4>(g, nt / 4, s);
    if (nt % 2 == 0 && 64 * row_bytes <= kGemmWsLds) return launch_gemm_ws<2>(g, nt / 2, s);
    if (32 * row_bytes <= kGemmWsLds) return launch_gemm_ws<1>(g, nt, s);
#endif
    if (nt % 8 == 0) return launch_gemm_nt<8>(g, nt / 8, s);
    if (nt % 4 == 0) return launch_gemm_nt<4>(g, nt / 4, s);
    if (nt % 2 == 0) return launch_gemm_nt<2>(g, nt / 2, s);
    return launch_gemm_nt<1>(g, nt, s);
}
// ---- heads ----------------------------------------------------------------------------------------------------------
// raw[m] = (sigmoid(z[m][0..2]), relu(z[m][3])) from the head GEMM's pre-activations   (mlp.py:42-49)
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ z, long long n, float* __restrict__ raw) {
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    if (m >= n) return;
    float4 v;
    v.x = 1.f / (1.f + expf(-z[m * 32 + 0]));
    v.y = 1.f / (1.f + expf(-z[m * 32 + 1]));
    v.z = 1.f / (1.f + expf(-z[m * 32 + 2]));
    v.w = fmaxf(z[m * 32 + 3], 0.f);
    reinterpret_cast<float4*>(raw)[m] = v;
}
// dZ_head [Mp][32]: columns 0..2 = d_rgb * y (1 - y) (sigmoid), column 3 = d_sigma * [sigma > 0] (relu); rows >= n and the
// padding columns are zero
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ draw, long long n, long long mp,
                                                      u16* __restrict__ dzh) {
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    if (m >= mp) return;
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    if (m < n) {
        const float4 y = reinterpret_cast<const float4*>(raw)[m];
        const float4 dy = reinterpret_cast<const float4*>(draw)[m];
        g[0] = dy.x * y.x * (1.f - y.x);
        g[1] = dy.y * y.y * (1.f - y.y);
        g[2] = dy.z * y.z * (1.f - y.z);
        g[3] = y.w > 0.f ? dy.w : 0.f;
    }
    u16* r = dzh + (size_t)m * 32;
    for (int c = 0; c < 32; ++c) r[c] = c < 4 ? to_bf16(g[c]) : 0;
}

// ---- collapsed head (layout.h / DESIGN.md section 2.0) for any shape --------------------------------------------------
// Offsets of the four head tensors in the flat parameters and the geometry of the head's input buffer.
struct HeadGeom {
    int ws, bs, wf, bf, wr, br, wc, bc;      // kernel / bias offsets of sigma, features, rgb_features, rgb
    int U, u2, dir_dim, xyz_dim, Tr;         // units, units/2, encodings, rows of the sigma/features kernels
    int up, K, dir_col0;                     // padded units, columns of the head buffer, first dir column
    int cat_last;                            // 1: [h ; xyz_enc] precedes the dir columns
};
// kernel row of sigma/features for buffer column c, or -1 (padding / dir column)
__device__ __forceinline__ int head_trunk_row(const HeadGeom& g, int c) {
    if (c < g.up) return c < g.U ? c : -1;
    if (g.cat_last && c < g.dir_col0) return (c - g.up) < g.xyz_dim ? g.U + (c - g.up) : -1;
    return -1;
}
__device__ __forceinline__ int head_col_of_trunk_row(const HeadGeom& g, int i) { return i < g.U ? i : g.up + (i - g.U); }

// head[c][0..2] = (W_f (W_r1 W_c))[row(c)] or (W_r2 W_c)[c - dir_col0], head[c][3] = w_s[row(c)] or 0; bias behind it;
// scratch P = W_r W_c [U + dir_dim][3] behind that (kept for expand_head: the weights do not change in between).
__global__ __launch_bounds__(1024) void head_compose_kernel(const float* __restrict__ w, float* __restrict__ head, HeadGeom g) {
    float* H = head;
    float* hb = head + (size_t)g.K * 4;
    float* P = hb + 4;
    const int tid = threadIdx.x;
    for (int r = tid; r < g.U + g.dir_dim; r += 1024) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        const float* wr = w + g.wr + (size_t)r * g.u2;
        for (int k = 0; k < g.u2; ++k) { const float v = wr[k]; a0 += v * w[g.wc + k * 3]; a1 += v * w[g.wc + k * 3 + 1]; a2 += v * w[g.wc + k * 3 + 2]; }
        P[r * 3] = a0; P[r * 3 + 1] = a1; P[r * 3 + 2] = a2;
    }
    __syncthreads();
    for (int c = tid; c < g.K; c += 1024) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        const int i = head_trunk_row(g, c);
        if (i >= 0) {
            const float* wf = w + g.wf + (size_t)i * g.U;
            for (int j = 0; j < g.U; ++j) { const float v = wf[j]; a0 += v * P[j * 3]; a1 += v * P[j * 3 + 1]; a2 += v * P[j * 3 + 2]; }
            a3 = w[g.ws + i];
        } else if (c >= g.dir_col0 && c - g.dir_col0 < g.dir_dim) {
            const int m = g.U + (c - g.dir_col0);
            a0 = P[m * 3]; a1 = P[m * 3 + 1]; a2 = P[m * 3 + 2];
        }
        H[c * 4] = a0; H[c * 4 + 1] = a1; H[c * 4 + 2] = a2; H[c * 4 + 3] = a3;
    }
    if (tid < 3) {
        float c = w[g.bc + tid];
        for (int j = 0; j < g.U; ++j) c += w[g.bf + j] * P[j * 3 + tid];
        for (int k = 0; k < g.u2; ++k) c += w[g.br + k] * w[g.wc + k * 3 + tid];
        hb[tid] = c;
    }
    if (tid == 3) hb[3] = w[g.bs];
}

// gaux = M [K][4] (buffer-column order), s [4].  Phase a (grid over blocks of 64 features): Q = W_f^T M1 + b_f (x) s into the
// scratch behind P -- 64 columns x 16 slices of the Tr rows per workgroup, the slices summed in a fixed order through LDS (one
// workgroup walking all Tr rows in one loop, as rounds 1-4 had it, took 216 us per call: 5 % of a chunk).  Phase c (one wave per
// element): the rgb kernel / bias, which need all of Q.  Phase b (grid): everything that is one short dot product per element.
__global__ __launch_bounds__(1024) void head_expand_a_kernel(const float* __restrict__ w, const float* __restrict__ gaux, float* __restrict__ head, HeadGeom g) {
    __shared__ float part[16][64][3];
    const float* M = gaux;
    const float* sv = gaux + (size_t)g.K * 4;
    float* P = head + (size_t)g.K * 4 + 4;
    float* Q = P + (size_t)(g.U + g.dir_dim) * 3;
    const int jj = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + jj;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    if (j < g.U) {
        for (int i = sl; i < g.Tr; i += 16) {
            const float v = w[g.wf + (size_t)i * g.U + j];
            const float* m = M + (size_t)head_col_of_trunk_row(g, i) * 4;
            a0 += v * m[0]; a1 += v * m[1]; a2 += v * m[2];
        }
    }
    part[sl][jj][0] = a0; part[sl][jj][1] = a1; part[sl][jj][2] = a2;
    __syncthreads();
    if (sl == 0 && j < g.U) {
        const float bf = w[g.bf + j];
        a0 = bf * sv[0]; a1 = bf * sv[1]; a2 = bf * sv[2];
        for (int q = 0; q < 16; ++q) { a0 += part[q][jj][0]; a1 += part[q][jj][1]; a2 += part[q][jj][2]; }
        Q[j * 3] = a0; Q[j * 3 + 1] = a1; Q[j * 3 + 2] = a2;
    }
}
__global__ __launch_bounds__(256) void head_expand_c_kernel(const float* __restrict__ w, const float* __restrict__ gaux, const float* __restrict__ head,
                                                           float* __restrict__ grad, HeadGeom g) {
    const float* M = gaux;
    const float* sv = gaux + (size_t)g.K * 4;
    const float* P = head + (size_t)g.K * 4 + 4;
    const float* Q = P + (size_t)(g.U + g.dir_dim) * 3;
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);    // rgb kernel [u2][3] += W_r1^T Q + W_r2^T M2 + b_r (x) s: one wave per element
    if (e < g.u2 * 3) {
        const int k = e / 3, c = e % 3;
        float a = 0.f;
        for (int j = lane; j < g.U; j += 64) a += w[g.wr + (size_t)j * g.u2 + k] * Q[j * 3 + c];
        for (int m = lane; m < g.dir_dim; m += 64) a += w[g.wr + (size_t)(g.U + m) * g.u2 + k] * M[(size_t)(g.dir_col0 + m) * 4 + c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        if (lane == 0) grad[g.wc + e] += a + w[g.br + k] * sv[c];
    }
    if (blockIdx.x == 0 && threadIdx.x < 3) grad[g.bc + threadIdx.x] += sv[threadIdx.x];
    if (blockIdx.x == 0 && threadIdx.x == 3) grad[g.bs] += sv[3];
}
__global__ __launch_bounds__(256) void head_expand_b_kernel(const float* __restrict__ w, const float* __restrict__ gaux, const float* __restrict__ head,
                                                           float* __restrict__ grad, HeadGeom g) {
    const float* M = gaux;
    const float* sv = gaux + (size_t)g.K * 4;
    const float* P = head + (size_t)g.K * 4 + 4;
    const float* Q = P + (size_t)(g.U + g.dir_dim) * 3;
    long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n_f = (long long)g.Tr * g.U, n_r = (long long)(g.U + g.dir_dim) * g.u2;
    if (e < n_f) {                                        // features kernel [Tr][U] += M1 P1^T
        const int i = (int)(e / g.U), j = (int)(e % g.U);
        const float* m = M + (size_t)head_col_of_trunk_row(g, i) * 4;
        grad[g.wf + e] += m[0] * P[j * 3] + m[1] * P[j * 3 + 1] + m[2] * P[j * 3 + 2];
        return;
    }
    e -= n_f;
    if (e < g.U) { grad[g.bf + e] += sv[0] * P[e * 3] + sv[1] * P[e * 3 + 1] + sv[2] * P[e * 3 + 2]; return; }
    e -= g.U;
    if (e < n_r) {                                        // rgb_features kernel [U + dir][u2] += [Q ; M2] W_c^T
        const int r = (int)(e / g.u2), k = (int)(e % g.u2);
        const float* v = r < g.U ? Q + (size_t)r * 3 : M + (size_t)(g.dir_col0 + r - g.U) * 4;
        grad[g.wr + e] += v[0] * w[g.wc + k * 3] + v[1] * w[g.wc + k * 3 + 1] + v[2] * w[g.wc + k * 3 + 2];
        return;
    }
    e -= n_r;
    if (e < g.u2) { grad[g.br + e] += sv[0] * w[g.wc + e * 3] + sv[1] * w[g.wc + e * 3 + 1] + sv[2] * w[g.wc + e * 3 + 2]; return; }
    e -= g.u2;
    if (e < g.Tr) grad[g.ws + e] += M[(size_t)head_col_of_trunk_row(g, (int)e) * 4 + 3];   // sigma kernel [Tr][1]
}

// ---- weight packing -------------------------------------------------------------------------------------------------
struct PackArgs {
    const float* w;             // flat fp32 parameters of the net
    u16* dst;
    int w_off, n_real;          // kernel[k][n] = w[w_off + k * n_real + n]
    int K, np;                  // padded input width (buffer ld), padded output width
    int n_seg; Seg seg[2];
    int transpose;              // 0: dst[n][k] (ld K)   1: dst[k][col0 + n] (ld ldd)
    int ldd, col0;
};
__global__ __launch_bounds__(256) void pack_kernel(PackArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)a.K * a.np) return;
    const int n = (int)(i / a.K), kc = (int)(i % a.K);
    int wrow = -1;
    for (int s = 0; s < a.n_seg; ++s)
        if (kc >= a.seg[s].col0 && kc < a.seg[s].col0 + a.seg[s].width) wrow = a.seg[s].wrow0 + kc - a.seg[s].col0;
    const float v = (wrow >= 0 && n < a.n_real) ? a.w[a.w_off + (size_t)wrow * a.n_real + n] : 0.f;
    if (a.transpose) a.dst[(size_t)kc * a.ldd + a.col0 + n] = to_bf16(v);
    else a.dst[(size_t)n * a.K + kc] = to_bf16(v);
}

// ---- wgrad: dW[k][n] += sum_m X[m][k] dZ[m][n], db[n] += sum_m dZ[m][n] -----------------------------------------------
struct WgradArgs {
    const u16* X; int ldx;
    const u16* Z; int ldz;
    long long steps;            // Mp / 32
    float* grad;                // flat fp32 gradient of the net
    int w_off, b_off, n_real;
    int n_seg; Seg seg[2];
    // deterministic mode (knerf_set_option "deterministic"): no fp32 atomics -- every accumulating unit (a wave of wgrad_kernel, a
    // workgroup of wgrad_coop_kernel) stores its partial tile to its OWN slab and wgrad_reduce_kernel adds the slabs of a block in
    // slab order into grad.  The step -> unit assignment is a function of the grid alone, so two launches give the same bits.
    // Block (bx, by), slab sl: partial + ((bx gy + by) n_sl + sl) tile_floats; a tile is [TK][TN] sums then [bias_halves][TN].
    float* partial;             // null: atomics
    const int* live; const int* n_live;     // dead-tile skipping: the 32-sample steps to contract over = list entries [0, *n_live); null: all
};

// 32 samples x 32 features (row-major; raw = the two 16-byte row pieces each lane loaded) -> two operand fragments with
// lane = feature: f[0] = samples sigma(hh, j) (first 16), f[1] = 16 + sigma(hh, j), sigma(hh, j) = (j&3) + 8(j>>2) + 4hh.
struct RawTile { bf16x8 a1, a2; };
__device__ __forceinline__ RawTile load_tile(const u16* P, int ld, int r, int h) {
    RawTile t;
    t.a1 = *reinterpret_cast<const bf16x8*>(P + (size_t)r * ld + 8 * h);
    t.a2 = *reinterpret_cast<const bf16x8*>(P + (size_t)r * ld + 16 + 8 * h);
    return t;
}
__device__ __forceinline__ void transpose_tile(const RawTile& t, const bf16x8& ilo, const bf16x8& ihi, bf16x8 (&f)[2], float* colsum) {
    f32x16 T = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t.a1, ilo, zero16(), 0, 0, 0);
    T = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t.a2, ihi, T, 0, 0, 0);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float lo = T[j], hi = T[8 + j];
        f[0][j] = (__bf16)lo;              // exact: the values are bf16 times one
        f[1][j] = (__bf16)hi;
        s += lo + hi;
    }
    if (colsum) *colsum += s;
}

// Which 32-sample steps an accumulating unit (unit `u` of `units`) contracts over, as (first, stride, count) over INDICES that
// step_of() maps to sample tiles.  Atomic mode: units interleave (u, u + units, ...) over all steps or over the live list -- the best
// balance.  Deterministic mode (g.partial): unit u owns the CONTIGUOUS tile range [u T / units, (u + 1) T / units) of the pass's T
// tiles and walks it in ascending order -- all of it, or (dead-tile skipping) its live tiles, found by binary search in the
// ascending list the compaction kernel made: the dead tiles would only have added exact zeros, so a launch with skipping forms the
// same sums in the same order as one without (bit-identical gradients, tests/test_gpu_det_skip.py).
struct UnitSched { long long first, stride, count; };
__device__ __forceinline__ UnitSched unit_schedule(const WgradArgs& g, long long u, long long units) {
    const long long n_live = g.live ? (long long)*g.n_live : g.steps;
    if (!g.partial) return {u, units, n_live > u ? (n_live - u + units - 1) / units : 0};
    const long long lo = u * g.steps / units, hi = (u + 1) * g.steps / units;
    if (!g.live) return {lo, 1, hi - lo};
    auto lower = [&](long long v) {                 // first list index whose tile is >= v
        long long a = 0, b = n_live;
        while (a < b) { const long long m = (a + b) >> 1; if ((long long)g.live[m] < v) a = m + 1; else b = m; }
        return a;
    };
    const long long i0 = lower(lo), i1 = lower(hi);
    return {i0, 1, i1 - i0};
}
__device__ __forceinline__ long long step_of(const WgradArgs& g, long long idx) { return g.live ? (long long)g.live[idx] : idx; }

#ifndef KNERF_GEN_WG_DEPTH
#define KNERF_GEN_WG_DEPTH 3
#endif
constexpr int kWgDepth = KNERF_GEN_WG_DEPTH;
template <int KT, int NT>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs g) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int k0 = blockIdx.x * KT * 32, n0 = blockIdx.y * NT * 32;
    bf16x8 ilo, ihi;          // B[k][c] = [k == c] for c < 16 / [k == c - 16] for c >= 16, k = 8h + j
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        ilo[j] = (__bf16)((8 * h + j == r) ? 1.f : 0.f);
        ihi[j] = (__bf16)((8 * h + j == r - 16) ? 1.f : 0.f);
    }
    f32x16 acc[KT][NT];
#pragma unroll
    for (int a = 0; a < KT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = zero16();
    float bsum[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b) bsum[b] = 0.f;
    const bool do_bias = blockIdx.x == 0;
    const UnitSched us = unit_schedule(g, (long long)blockIdx.z * 4 + wave, (long long)gridDim.z * 4);
    const long long stride = us.stride, n_steps = us.first + us.count * us.stride;      // indices first, first + stride, ... < n_steps
    long long s = us.first;
    // kWgDepth m-steps are in flight: one step of this wave is 16 MFMAs (~0.5 us with its SIMD partner), a first-touch
    // HBM load takes several times that
    constexpr int D = kWgDepth;
    RawTile xr[D][KT], zr[D][NT];
    auto fetch = [&](long long st_i, RawTile (&x)[KT], RawTile (&z)[NT]) {
        const long long st = step_of(g, st_i);
        const u16* X = g.X + (size_t)st * 32 * g.ldx + k0;
        const u16* Z = g.Z + (size_t)st * 32 * g.ldz + n0;
#pragma unroll
        for (int a = 0; a < KT; ++a) x[a] = load_tile(X + 32 * a, g.ldx, r, h);
#pragma unroll
        for (int b = 0; b < NT; ++b) z[b] = load_tile(Z + 32 * b, g.ldz, r, h);
    };
#pragma unroll
    for (int dd = 0; dd < D; ++dd)
        if (s + dd * stride < n_steps) fetch(s + dd * stride, xr[dd], zr[dd]);
    for (; s < n_steps; s += D * stride) {
#pragma unroll
        for (int dd = 0; dd < D; ++dd) {
            const long long st = s + dd * stride;
            if (st < n_steps) {                       // wave-uniform
                bf16x8 xa[KT][2], zb[NT][2];
#pragma unroll
                for (int a = 0; a < KT; ++a) transpose_tile(xr[dd][a], ilo, ihi, xa[a], nullptr);
#pragma unroll
                for (int b = 0; b < NT; ++b) transpose_tile(zr[dd][b], ilo, ihi, zb[b], do_bias ? &bsum[b] : nullptr);
                if (st + D * stride < n_steps) fetch(st + D * stride, xr[dd], zr[dd]);
#pragma unroll
                for (int a = 0; a < KT; ++a)
#pragma unroll
                    for (int b = 0; b < NT; ++b) {
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[a][0], zb[b][0], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[a][1], zb[b][1], acc[a][b], 0, 0, 0);
                    }
            }
        }
    }
    // flush: lane (col = r, hh = h), register i -> input column k0 + 32a + (i&3) + 8(i>>2) + 4hh
    if (g.partial) {            // deterministic mode: this wave's tile to its own slab (every element, also of a wave without steps)
        constexpr int TK = KT * 32, TN = NT * 32, TF = TK * TN + 2 * TN;
        const int n_sl = gridDim.z * 4, sl = blockIdx.z * 4 + wave;
        float* tile = g.partial + ((size_t)(blockIdx.x * gridDim.y + blockIdx.y) * n_sl + sl) * TF;
#pragma unroll
        for (int b = 0; b < NT; ++b) {
#pragma unroll
            for (int a = 0; a < KT; ++a)
#pragma unroll
                for (int i = 0; i < 16; ++i) tile[(32 * a + (i & 3) + 8 * (i >> 2) + 4 * h) * TN + 32 * b + r] = acc[a][b][i];
            if (do_bias) tile[TK * TN + h * TN + 32 * b + r] = bsum[b];       // the two lane halves hold different samples: two rows
        }
        return;
    }
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        const int col = n0 + 32 * b + r;
        if (col >= g.n_real) continue;
#pragma unroll
        for (int a = 0; a < KT; ++a)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int kc = k0 + 32 * a + (i & 3) + 8 * (i >> 2) + 4 * h;
                int wrow = -1;
                for (int sg = 0; sg < g.n_seg; ++sg)
                    if (kc >= g.seg[sg].col0 && kc < g.seg[sg].col0 + g.seg[sg].width) wrow = g.seg[sg].wrow0 + kc - g.seg[sg].col0;
                if (wrow >= 0) atomicAdd(g.grad + g.w_off + (size_t)wrow * g.n_real + col, acc[a][b][i]);
            }
        if (do_bias) atomicAdd(g.grad + g.b_off + col, bsum[b]);     // both lane halves hold different samples
    }
}

// Cooperative wgrad (every layer larger than one 64 x 64 tile, coop_serves below): a workgroup of 8 waves owns a 256 (inputs) x 256 (outputs) block of dW
// and walks 32-sample steps; per step the X and dZ slabs [32][256] are staged ONCE in LDS and every wave builds its operand
// fragments with ds_read_b64_tr_b16 -- lane 4q+p of a 16-lane group supplies the address of sample row q / feature quad p and
// receives feature `lane` of the four samples, i.e. exactly the MFMA operand layout with samples as k (tests/test_gpu_probe.py
// pins the instruction).  Against the per-wave kernel above: operands cross L2 once per 256 x 256 block instead of 4 x, and no
// MFMA is spent on transposing.  Wave (wa, wb) = (wave >> 1, wave & 1): input tiles 2wa, 2wa+1, output tiles 4wb..4wb+3.
// Round 5: the slabs arrive by LDS-DMA (global_load_lds_dwordx4: no staging registers, so THREE steps = 96 KiB per workgroup are in
// flight instead of one -- rounds 2-4 ran this launch at 2.6 TB/s, latency-bound), four 32 KiB buffers.  A DMA instruction lands
// 1 KiB contiguously (two unpadded 512-byte rows), so the rows cannot be padded apart: instead the 16-byte chunks of row r sit at
// chunk ^ 4 (r & 3) -- the four rows a transposed read touches then cover the 64 banks exactly once (tests/test_lds_bank_model.py).
// Each lane fetches the chunk that belongs at its LDS position.  Waits are counted by hand (4 DMA instructions per wave and step).
constexpr int kCoopSlab = 32 * 512, kCoopBuf = 2 * kCoopSlab, kCoopDepth = 4;
__device__ __forceinline__ void coop_glds16(const void* gsrc, unsigned lds_dst) {       // as wgrad_body.h glds16: invisible to hipcc's wait counting
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__global__ __launch_bounds__(512) void wgrad_coop_kernel(WgradArgs g, int K, int N, int gx, int gy, int gz) {
    // one-dimensional grid -> (bx, by, bz).  A layer wider than one 256 x 256 block has gx gy blocks per unit of samples, which read
    // the SAME X and dZ slabs (X once per column block, dZ once per row block): the blocks of a unit are neighbours in one XCD's
    // dispatch order (workgroups w and w + 8 share an XCD; gz is a multiple of 8 then, launch_wgrad), so the repeats come from that
    // XCD's L2.  Speed only; nothing depends on the placement.
    int bx = 0, by = 0, bz = blockIdx.x;
    if (gx * gy > 1) {
        const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, q = local % (gx * gy);
        bx = q % gx; by = q / gx; bz = (local / (gx * gy)) * 8 + xcd;
    }
    extern __shared__ __attribute__((aligned(16))) char csm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave >> 1, wb = wave & 1;
    const int k0 = bx * 256, n0 = by * 256;
    const int r = lane & 31, h = lane >> 5;
    // transposed-read lane offsets: group gq = lane >> 4 -> feature half (gq & 1), k half (gq >> 1); i = lane & 15 -> q = i >> 2, p = i & 3.
    // Row 8 (gq >> 1) + q (+ 16 kk, + 4 for the second read: the low two bits stay q), chunk 4 tile + 2 (gq & 1) + (p >> 1), swizzled by q
    const int gq = lane >> 4, il = lane & 15, tq = il >> 2;
    const int tr_off = (8 * (gq >> 1) + tq) * 512 + (2 * (gq & 1) + ((il & 3) >> 1)) * 16 + (il & 1) * 8;
    typedef __attribute__((address_space(3))) s16x4g* lds_s16x4g_ptr;
    auto frag = [&](const char* slab, int tile, int kk) {
        const char* p = slab + tr_off + (16 * kk) * 512 + ((tile ^ tq) << 6);
        const s16x4g lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4g_ptr)p);
        const s16x4g hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4g_ptr)(p + 4 * 512));
        typedef __attribute__((ext_vector_type(8))) short s16x8g;
        const s16x8g v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };
    const UnitSched us = unit_schedule(g, (long long)bz, (long long)gz);
    const long long cnt = us.count;
    // entry j of this unit's schedule -> sample tile; the list look-up is a SCALAR load in asm (a compiler-visible vector load
    // inside the loop would make hipcc drain vmcnt(0), i.e. the whole DMA pipeline, every step)
    auto tile_of = [&](long long j) -> long long {
        const long long idx = us.first + (j < cnt ? j : cnt - 1) * us.stride;      // past the end: a harmless re-read keeps the vmcnt arithmetic uniform
        if (!g.live) return idx;
        const unsigned long long p = (unsigned long long)(g.live + idx);
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)p), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32));
        int t;
        asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "s"(((unsigned long long)hi << 32) | lo) : "memory");
        return (long long)t;
    };
    // this lane's part of a step: instruction m = wave + 8 i (i = 0, 1) of each slab covers rows 2m, 2m+1; the lane sits at LDS slot
    // (row, lane & 31) and fetches chunk (lane & 31) ^ 4 (row & 3) of its row (clamped into the layer: tiles past K / N are never used)
    const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)csm;
    const int kmax = (K - k0 + 7) / 8 - 1 < 31 ? (K - k0 + 7) / 8 - 1 : 31, nmax = (N - n0 + 7) / 8 - 1 < 31 ? (N - n0 + 7) / 8 - 1 : 31;
    auto issue = [&](long long st, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = wave + 8 * i, row = 2 * m + h;
            const int c = r ^ ((row & 3) << 2);
            const unsigned dst = smem_base + buf * kCoopBuf + m * 1024;
            coop_glds16(g.X + (size_t)(st * 32 + row) * g.ldx + k0 + (c < kmax ? c : kmax) * 8, __builtin_amdgcn_readfirstlane(dst));
            coop_glds16(g.Z + (size_t)(st * 32 + row) * g.ldz + n0 + (c < nmax ? c : nmax) * 8, __builtin_amdgcn_readfirstlane(dst + kCoopSlab));
        }
    };
    f32x16 acc[2][4], acc_b[4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = zero16();
#pragma unroll
    for (int b = 0; b < 4; ++b) acc_b[b] = zero16();
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
    const bool do_bias = bx == 0 && wa == 0;                     // wave-uniform
    const bool k_ok[2] = {k0 + 32 * (2 * wa) < K, k0 + 32 * (2 * wa + 1) < K};
    const bool n_ok[4] = {n0 + 32 * (4 * wb) < N, n0 + 32 * (4 * wb + 1) < N, n0 + 32 * (4 * wb + 2) < N, n0 + 32 * (4 * wb + 3) < N};
    if (cnt > 0) {
#pragma unroll
        for (int d = 0; d < kCoopDepth - 1; ++d) issue(tile_of(d), d);          // steps 0 .. D-2
    }
    for (long long i = 0; i < cnt; ++i) {
        const int buf = (int)(i % kCoopDepth);
        const long long t_next = tile_of(i + kCoopDepth - 1);                     // (its scalar load waits here, before the counted wait)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (kCoopDepth - 2)) : "memory");    // step i landed (mine) ...
        __builtin_amdgcn_s_barrier();                                             // ... everyone's; the buffer of step i-1 is free (its reads fed MFMAs already)
        issue(t_next, (int)((i + kCoopDepth - 1) % kCoopDepth));
        const char* xs = csm + buf * kCoopBuf;
        const char* zs = xs + kCoopSlab;
        if (k_ok[0] && n_ok[0]) {                                        // a wave whose first tiles are out of range has nothing to do
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 zb[4], xa[2];
#pragma unroll
                for (int b = 0; b < 4; ++b) zb[b] = frag(zs, 4 * wb + b, kk);
#pragma unroll
                for (int a = 0; a < 2; ++a) xa[a] = frag(xs, 2 * wa + a, kk);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        if (k_ok[a] && n_ok[b]) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[a], zb[b], acc[a][b], 0, 0, 0);
                if (do_bias) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc_b[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, zb[b], acc_b[b], 0, 0, 0);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                             // the trailing (unused) copies
    // flush: lane (col = r, hh = h), register i -> input column 32 tile + (i&3) + 8(i>>2) + 4hh
    if (g.partial) {            // deterministic mode: the workgroup's 256 x 256 block (its eight waves own disjoint parts) to its own slab
        constexpr int TN = 256, TF = 256 * 256 + TN;
        float* tile = g.partial + ((size_t)(bx * gy + by) * gz + bz) * TF;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int cl = 32 * (4 * wb + b) + r;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int i = 0; i < 16; ++i) tile[(32 * (2 * wa + a) + (i & 3) + 8 * (i >> 2) + 4 * h) * TN + cl] = acc[a][b][i];
            if (do_bias && h == 0) tile[256 * 256 + cl] = acc_b[b][0];
        }
        return;
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int col = n0 + 32 * (4 * wb + b) + r;
        if (col >= g.n_real) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int kc = k0 + 32 * (2 * wa + a) + (i & 3) + 8 * (i >> 2) + 4 * h;
                int wrow = -1;
                for (int sg = 0; sg < g.n_seg; ++sg)
                    if (kc >= g.seg[sg].col0 && kc < g.seg[sg].col0 + g.seg[sg].width) wrow = g.seg[sg].wrow0 + kc - g.seg[sg].col0;
                if (wrow >= 0) atomicAdd(g.grad + g.w_off + (size_t)wrow * g.n_real + col, acc[a][b][i]);
            }
        if (do_bias && h == 0) atomicAdd(g.grad + g.b_off + col, acc_b[b][0]);   // every row of the ones product holds the column sum
    }
}

// deterministic mode, second pass: element e of block (bx, by) = the sum of its n_sl slabs IN SLAB ORDER, added to the gradient by
// the one thread that owns the destination (launches follow each other on the stream, so the accumulation over passes and chunks
// is ordered too)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgradArgs g, int n_sl, int TK, int TN, int bias_halves, int gy) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int tf = TK * TN + bias_halves * TN;
    if (e >= TK * TN + TN) return;
    const int bx = blockIdx.y / gy, by = blockIdx.y % gy;
    const float* base = g.partial + (size_t)blockIdx.y * n_sl * tf;
    if (e < TK * TN) {
        const int kc = bx * TK + e / TN, col = by * TN + e % TN;
        if (col >= g.n_real) return;
        int wrow = -1;
        for (int sg = 0; sg < g.n_seg; ++sg)
            if (kc >= g.seg[sg].col0 && kc < g.seg[sg].col0 + g.seg[sg].width) wrow = g.seg[sg].wrow0 + kc - g.seg[sg].col0;
        if (wrow < 0) return;
        float sum = 0.f;
        for (int sl = 0; sl < n_sl; ++sl) sum += base[(size_t)sl * tf + e];
        g.grad[g.w_off + (size_t)wrow * g.n_real + col] += sum;
    } else {
        const int cl = e - TK * TN, col = by * TN + cl;
        if (bx != 0 || col >= g.n_real) return;
        float sum = 0.f;
        for (int sl = 0; sl < n_sl; ++sl)
            for (int hh = 0; hh < bias_halves; ++hh) sum += base[(size_t)sl * tf + TK * TN + hh * TN + cl];
        g.grad[g.b_off + col] += sum;
    }
}

// Which of the two weight-gradient kernels serves a [K] x [N] layer: the cooperative one, always (since round 5).  The per-wave
// kernel re-reads X once per column block and dZ once per row block of its (at most 64 x 64) tiles -- the first layer (K = 64)
// ran at 1.6 TB/s on it (306 us per fine launch, 105 us now), the head's [head_K] x [32] sums took 200 us (159 us now) -- and where
// one tile covers the whole layer (widths <= 64) its 1,024 workgroups x 4 waves flush the SAME few thousand addresses with
// atomics: 410-510 us per 64 x 64 layer for 40 us of traffic, 80 % of a width-64 train chunk.  Tiles past K or N cost the
// cooperative kernel nothing but idle waves (k_ok / n_ok) and DMA copies of clamped chunks that hit in cache; it flushes once per
// workgroup (256 of them).  For A/B runs: -DKNERF_GEN_COOP_MIN_TILES=n keeps layers with fewer than n tiles on BOTH sides on the
// per-wave kernel, -DKNERF_GEN_NO_COOP everything.
#ifndef KNERF_GEN_COOP_MIN_TILES
#define KNERF_GEN_COOP_MIN_TILES 1      // kt or nt >= this
#endif
#ifndef KNERF_GEN_COOP_WGS
#define KNERF_GEN_COOP_WGS 256      // one workgroup per CU: each flushes a whole 256 x 256 block with atomics (512: +8 %, 1024: +23 %)
#endif
inline bool coop_serves(int kt, int nt) {
#ifdef KNERF_GEN_NO_COOP
    return false;
#else
    return kt >= KNERF_GEN_COOP_MIN_TILES || nt >= KNERF_GEN_COOP_MIN_TILES;
#endif
}

// slab floats one weight-gradient launch over a [K] x [N] layer can need (the grid's upper bound; independent of the sample count)
size_t wgrad_partial_floats_for(int K, int N) {
    const int kt = K / 32, nt = N / 32;
    if (coop_serves(kt, nt)) {
        const long long gx = (K + 255) / 256, gy = (N + 255) / 256;
        long long gz = KNERF_GEN_COOP_WGS / (gx * gy); if (gz < 1) gz = 1;
        if (gx * gy > 1 && gz < 8) gz = 8;                       // launch_wgrad's whole groups of eight units
        return (size_t)(gx * gy * gz) * (256 * 256 + 256);
    }
    const int KT = kt % 2 == 0 ? 2 : 1, NT = nt % 4 == 0 ? 4 : (nt % 2 == 0 ? 2 : 1);      // the largest tiles launch_wgrad may pick
    const long long gx = kt / KT, gy = nt / NT;
    long long gz = 1024 / (gx * gy); if (gz < 1) gz = 1;
    return (size_t)(gx * gy * gz * 4) * (KT * 32 * NT * 32 + 2 * NT * 32);
}

hipError_t launch_wgrad(const WgradArgs& g, int K, int N, hipStream_t s) {
    const int kt = K / 32, nt = N / 32;
    if (coop_serves(kt, nt)) {
        static AttrOnce once;
        const size_t lds = (size_t)kCoopDepth * kCoopBuf;
        hipError_t ae = once([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_coop_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
        if (ae != hipSuccess) return ae;
        const int gx = (K + 255) / 256, gy = (N + 255) / 256;
        long long gz = KNERF_GEN_COOP_WGS / ((long long)gx * gy);
        if (gz > g.steps) gz = g.steps;
        if (gz < 1) gz = 1;
        if (gx * gy > 1) gz = gz >= 8 ? gz / 8 * 8 : 8;         // the kernel's XCD mapping wants whole groups of eight units (idle units are harmless)
        hipLaunchKernelGGL(wgrad_coop_kernel, dim3((unsigned)(gx * gy * gz)), dim3(512), lds, s, g, K, N, gx, gy, (int)gz);
        if (g.partial)
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((256 * 256 + 256 + 255) / 256, gx * gy), dim3(256), 0, s, g, (int)gz, 256, 256, 1, gy);
        return hipGetLastError();
    }
#ifndef KNERF_GEN_MAXKT
#define KNERF_GEN_MAXKT 2
#endif
#ifndef KNERF_GEN_MAXNT
#define KNERF_GEN_MAXNT 2      // measured: 2x2 blocks (more waves in flight) 17.6 ms per default-shape chunk, 2x4 19.1, 1x4 18.7
#endif
    const int KT = (kt % 2 == 0 && KNERF_GEN_MAXKT >= 2) ? 2 : 1;
    const int NT = (nt % 4 == 0 && KNERF_GEN_MAXNT >= 4) ? 4 : ((nt % 2 == 0 && KNERF_GEN_MAXNT >= 2) ? 2 : 1);
    const int gx = kt / KT, gy = nt / NT;
    long long gz = 1024 / ((long long)gx * gy);
    const long long max_z = (g.steps + 3) / 4;
    if (gz > max_z) gz = max_z;
    if (gz < 1) gz = 1;
    const dim3 grid(gx, gy, (unsigned)gz), block(256);
#define KNERF_GEN_WG(KT_, NT_) hipLaunchKernelGGL((wgrad_kernel<KT_, NT_>), grid, block, 0, s, g)
    if (KT == 2 && NT == 4) KNERF_GEN_WG(2, 4);
    else if (KT == 2 && NT == 2) KNERF_GEN_WG(2, 2);
    else if (KT == 2 && NT == 1) KNERF_GEN_WG(2, 1);
    else if (KT == 1 && NT == 4) KNERF_GEN_WG(1, 4);
    else if (KT == 1 && NT == 2) KNERF_GEN_WG(1, 2);
    else KNERF_GEN_WG(1, 1);
#undef KNERF_GEN_WG
    if (g.partial)
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((KT * 32 * NT * 32 + NT * 32 + 255) / 256, gx * gy), dim3(256), 0, s, g, (int)gz * 4, KT * 32, NT * 32, 2, gy);
    return hipGetLastError();
}

inline unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

// ---- host: plan -----------------------------------------------------------------------------------------------------
bool is_default_shape(int n_layers, int units, int skip, int lx, int ld) {
    return n_layers == 8 && units == 256 && skip == 4 && lx == 10 && ld == 4;
}

int param_count(int n_layers, int units, int skip, int lx, int ld) {
    return build_plan(n_layers, units, skip, lx, ld).n_params;
}

Plan build_plan(int n_layers, int units, int skip, int lx, int ld) {
    Plan p = build_plan_widths(n_layers, units, skip, 3 + 6 * lx, 3 + 6 * ld);
    p.lx = lx; p.ld = ld;
    return p;
}

// Keras Dense builds its kernel from the LAST dimension of its first input (mlp.py:11-27 gives no input size), so a stand-alone
// NeRFMLP takes any two input widths: the plan only ever needs the widths (encode_kernel alone uses lx / ld, and a plan made here
// has none: lx = ld = -1, forward() refuses it)
Plan build_plan_widths(int n_layers, int units, int skip, int xyz_dim, int dir_dim) {
    Plan p{};
    p.n_layers = n_layers; p.units = units; p.skip = skip; p.lx = -1; p.ld = -1;
    p.xyz_dim = xyz_dim; p.dir_dim = dir_dim;
    p.kxp = r32(p.xyz_dim); p.kdp = r32(p.dir_dim); p.up = r32(units);
    const int u2 = units / 2;
    p.u2p = r32(u2);
    auto new_buf = [&](int ldv) { p.buf_ld.push_back(ldv); return (int)p.buf_ld.size() - 1; };
    auto new_dz = [&](int ldv) { p.dz_ld.push_back(ldv); return (int)p.dz_ld.size() - 1; };
    p.buf_encx = new_buf(p.kxp);
    p.buf_encd = new_buf(p.kdp);
    int off = 0;
    size_t packed = 0;
    int prev = p.buf_encx, fan_in = p.xyz_dim, n_seg = 1;
    Seg segs[2] = {{0, p.xyz_dim, 0}, {0, 0, 0}};
    auto add_layer = [&](int n_real, int out_buf, int out_col0, int head, int relu, int dz_buf, int dz_col0) -> Layer& {
        Layer L{};
        L.k_real = fan_in; L.n_real = n_real;
        L.w_off = off; L.b_off = off + fan_in * n_real; off += fan_in * n_real + n_real;
        L.n_seg = n_seg; L.seg[0] = segs[0]; L.seg[1] = segs[1];
        L.in_buf = prev; L.out_buf = out_buf; L.out_col0 = out_col0; L.head = head; L.relu = relu;
        L.np = r32(n_real);
        L.dz_buf = dz_buf; L.dz_col0 = dz_col0;
        const int K = p.buf_ld[prev];
        L.wt_off = packed; packed += (size_t)L.np * K;
        L.wd_off = packed; L.wd_ld = L.np; L.wd_col0 = 0; packed += (size_t)K * L.np;
        p.layers.push_back(L);
        return p.layers.back();
    };
    bool cat_last = false;
    for (int i = 0; i < n_layers; ++i) {
        const bool cat = (i % skip == 0) && i > 0;                 // mlp.py:36-38
        const bool last = i == n_layers - 1;                        // the trunk output doubles as the head's input: + dir columns
        if (last) { cat_last = cat; p.head_dir_col0 = p.up + (cat ? p.kxp : 0); }
        const int out = new_buf(p.up + (cat ? p.kxp : 0) + (last ? p.kdp : 0));
        add_layer(units, out, 0, -1, 1, new_dz(p.up), 0);
        p.concat_after.push_back(cat ? 1 : 0);
        prev = out;
        segs[0] = {0, units, 0};
        if (cat) { segs[1] = {p.up, p.xyz_dim, units}; n_seg = 2; fan_in = units + p.xyz_dim; }
        else { n_seg = 1; fan_in = units; }
    }
    p.buf_trunk = prev;
    p.head_K = p.buf_ld[p.buf_trunk];
    p.trunk_real = units + (cat_last ? p.xyz_dim : 0);
    p.dz_head = new_dz(32);
    p.head_wt_off = packed; packed += (size_t)32 * p.head_K;
    p.head_wd_off = packed; packed += (size_t)p.head_K * 32;
    // the four head layers in Keras order (mlp.py:19-27): parameter offsets and shapes only -- they are evaluated together as the
    // composed head and own neither buffers nor packed weights
    auto add_head_layer = [&](int k_real, int n_real, int head, int nseg, Seg s0, Seg s1) {
        Layer L{};
        L.k_real = k_real; L.n_real = n_real;
        L.w_off = off; L.b_off = off + k_real * n_real; off += k_real * n_real + n_real;
        L.n_seg = nseg; L.seg[0] = s0; L.seg[1] = s1;
        L.in_buf = p.buf_trunk; L.out_buf = -1; L.head = head; L.relu = 0; L.np = r32(n_real); L.dz_buf = p.dz_head;
        p.layers.push_back(L);
    };
    const Seg t0 = {0, units, 0}, t1 = {p.up, p.xyz_dim, units}, none = {0, 0, 0};
    add_head_layer(p.trunk_real, 1, 0, cat_last ? 2 : 1, t0, cat_last ? t1 : none);            // sigma
    add_head_layer(p.trunk_real, units, -1, cat_last ? 2 : 1, t0, cat_last ? t1 : none);        // features
    add_head_layer(units + p.dir_dim, u2, -1, 2, Seg{0, units, 0}, Seg{p.up, p.dir_dim, units}); // rgb_features: [features ; dir_enc]
    add_head_layer(u2, 3, 1, 1, Seg{0, u2, 0}, none);                                            // rgb
    p.n_params = off;
    p.packed_elems = packed;
    p.act_elems_per_row = 0; for (int v : p.buf_ld) p.act_elems_per_row += v;
    p.dz_elems_per_row = 0; for (int v : p.dz_ld) p.dz_elems_per_row += v;
    return p;
}

size_t padded_rows(long long n) { return (size_t)((n + 127) / 128 * 128); }
size_t relu_bits_bytes_per_row(const Plan& p) { return (size_t)p.up / 8; }
size_t relu_bits_bytes(const Plan& p, size_t mp) { return (size_t)p.n_layers * relu_bits_bytes_per_row(p) * mp; }

namespace {
u16* act_buf(const Plan& p, const Workspace& ws, int b) {
    size_t o = 0;
    for (int i = 0; i < b; ++i) o += p.buf_ld[i];
    return ws.act + o * ws.mp;
}
u16* dz_buf(const Plan& p, const Workspace& ws, int b) {
    size_t o = 0;
    for (int i = 0; i < b; ++i) o += p.dz_ld[i];
    return ws.dz + o * ws.mp;
}
#define GENCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return e_; } while (0)
}  // namespace

namespace {
HeadGeom head_geom(const Plan& p) {
    const int nl = p.n_layers;
    const Layer &Ls = p.layers[nl], &Lf = p.layers[nl + 1], &Lr = p.layers[nl + 2], &Lc = p.layers[nl + 3];
    HeadGeom g{};
    g.ws = Ls.w_off; g.bs = Ls.b_off; g.wf = Lf.w_off; g.bf = Lf.b_off; g.wr = Lr.w_off; g.br = Lr.b_off; g.wc = Lc.w_off; g.bc = Lc.b_off;
    g.U = p.units; g.u2 = p.units / 2; g.dir_dim = p.dir_dim; g.xyz_dim = p.xyz_dim; g.Tr = p.trunk_real;
    g.up = p.up; g.K = p.head_K; g.dir_col0 = p.head_dir_col0; g.cat_last = p.trunk_real > p.units ? 1 : 0;
    return g;
}
}  // namespace

size_t head_floats(const Plan& p) { return (size_t)p.head_K * 4 + 4 + (size_t)(p.units + p.dir_dim) * 3 + (size_t)p.units * 3; }
size_t aux_floats(const Plan& p) { return (size_t)p.head_K * 4 + 4; }

hipError_t pack_weights(const Plan& p, const float* w_flat, const NetDev& net, hipStream_t s) {
    unsigned short* packed = net.packed;
    for (int li = 0; li < p.n_layers; ++li) {
        const Layer& L = p.layers[li];
        PackArgs a{};
        a.w = w_flat; a.w_off = L.w_off; a.n_real = L.n_real; a.K = p.buf_ld[L.in_buf]; a.np = L.np;
        a.n_seg = L.n_seg; a.seg[0] = L.seg[0]; a.seg[1] = L.seg[1];
        a.dst = packed + L.wt_off; a.transpose = 0; a.ldd = a.K; a.col0 = 0;
        hipLaunchKernelGGL(pack_kernel, dim3(blocks_for((long long)a.K * a.np)), dim3(256), 0, s, a);
        a.dst = packed + L.wd_off; a.transpose = 1; a.ldd = L.wd_ld; a.col0 = L.wd_col0;
        hipLaunchKernelGGL(pack_kernel, dim3(blocks_for((long long)a.K * a.np)), dim3(256), 0, s, a);
    }
    // the composed head: fp32 [head_K][4] in buffer-column order, then its two bf16 packings
    hipLaunchKernelGGL(head_compose_kernel, dim3(1), dim3(1024), 0, s, w_flat, net.head, head_geom(p));
    PackArgs a{};
    a.w = net.head; a.w_off = 0; a.n_real = 4; a.K = p.head_K; a.np = 32; a.n_seg = 1; a.seg[0] = Seg{0, p.head_K, 0};
    a.dst = packed + p.head_wt_off; a.transpose = 0; a.ldd = a.K; a.col0 = 0;
    hipLaunchKernelGGL(pack_kernel, dim3(blocks_for((long long)a.K * a.np)), dim3(256), 0, s, a);
    a.dst = packed + p.head_wd_off; a.transpose = 1; a.ldd = 32; a.col0 = 0;
    hipLaunchKernelGGL(pack_kernel, dim3(blocks_for((long long)a.K * a.np)), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t expand_head(const Plan& p, const NetDev& net, const float* w_flat, float* grad_flat, hipStream_t s) {
    const HeadGeom g = head_geom(p);
    hipLaunchKernelGGL(head_expand_a_kernel, dim3((g.U + 63) / 64), dim3(1024), 0, s, w_flat, net.gaux, net.head, g);
    hipLaunchKernelGGL(head_expand_c_kernel, dim3((g.u2 * 3 + 3) / 4), dim3(256), 0, s, w_flat, net.gaux, net.head, grad_flat, g);
    const long long total = (long long)g.Tr * g.U + g.U + (long long)(g.U + g.dir_dim) * g.u2 + g.u2 + g.Tr;
    hipLaunchKernelGGL(head_expand_b_kernel, dim3(blocks_for(total)), dim3(256), 0, s, w_flat, net.gaux, net.head, grad_flat, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return hipMemsetAsync(net.gaux, 0, aux_floats(p) * sizeof(float), s);
}

namespace {
// every Dense layer over the encodings already sitting in the enc_x / enc_d buffers, then the heads
hipError_t run_layers(const Plan& p, const Workspace& ws, const NetDev& net, const float* w_flat, long long n, long long mp, float* raw, hipStream_t s) {
    u16* ex = act_buf(p, ws, p.buf_encx);
    u16* ed = act_buf(p, ws, p.buf_encd);
    for (int li = 0; li < p.n_layers; ++li) {
        const Layer& L = p.layers[li];
        GemmArgs g{};
        g.A = act_buf(p, ws, L.in_buf); g.lda = p.buf_ld[L.in_buf];
        g.Bt = net.packed + L.wt_off; g.ldb = g.lda;
        g.M = mp; g.N = L.np; g.K = g.lda;
        g.bias = w_flat + L.b_off; g.n_real = L.n_real; g.relu = L.relu;
        g.Cb = act_buf(p, ws, L.out_buf) + L.out_col0; g.ldc = p.buf_ld[L.out_buf];
        if (ws.mask) { g.mask_out = ws.mask + (size_t)li * relu_bits_bytes_per_row(p) * mp; g.mask_cols = p.up / 8; }      // training workspace: the dgrad's relu bits
        GENCHK(launch_gemm(g, s));
        if (p.concat_after[li]) {
            hipLaunchKernelGGL(copy_cols_kernel, dim3(blocks_for(mp * (p.kxp / 8))), dim3(256), 0, s, ex, p.kxp,
                               act_buf(p, ws, L.out_buf) + p.up, p.buf_ld[L.out_buf], mp, p.kxp);
            GENCHK(hipGetLastError());
        }
    }
    // head: [h ; (xyz_enc) ; dir_enc] . H -> (r, g, b, sigma) pre-activations, one GEMM on the composed matrix (generic.h)
    u16* trunk = act_buf(p, ws, p.buf_trunk);
    hipLaunchKernelGGL(copy_cols_kernel, dim3(blocks_for(mp * (p.kdp / 8))), dim3(256), 0, s, ed, p.kdp, trunk + p.head_dir_col0, p.head_K, mp, p.kdp);
    GENCHK(hipGetLastError());
    {
        GemmArgs g{};
        g.A = trunk; g.lda = p.head_K; g.Bt = net.packed + p.head_wt_off; g.ldb = p.head_K;
        g.M = mp; g.N = 32; g.K = p.head_K;
        g.bias = net.head + (size_t)p.head_K * 4; g.n_real = 4; g.relu = 0;
        g.Cf = ws.zc; g.ldcf = 32;
        GENCHK(launch_gemm(g, s));
    }
    hipLaunchKernelGGL(head_fwd_kernel, dim3(blocks_for(n)), dim3(256), 0, s, ws.zc, n, raw);
    return hipGetLastError();
}
}  // namespace

hipError_t forward(const Plan& p, const Workspace& ws, const NetDev& net, const float* w_flat, const float* o, const float* d,
                   const float* t, long long n, int S, float* raw, hipStream_t s) {
    const long long mp = (long long)padded_rows(n);
    if ((size_t)mp > ws.mp || p.lx < 0 || p.ld < 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(encode_kernel, dim3(blocks_for(mp * ((p.kxp + p.kdp) / 8))), dim3(256), 0, s, o, d, t, n, mp, S, p.lx, p.ld,
                       act_buf(p, ws, p.buf_encx), p.kxp, act_buf(p, ws, p.buf_encd), p.kdp);
    GENCHK(hipGetLastError());
    return run_layers(p, ws, net, w_flat, n, mp, raw, s);
}

hipError_t forward_encoded(const Plan& p, const Workspace& ws, const NetDev& net, const float* w_flat, const float* xyz_enc,
                           const float* dir_enc, long long n, float* raw, hipStream_t s) {
    const long long mp = (long long)padded_rows(n);
    if ((size_t)mp > ws.mp) return hipErrorInvalidValue;
    hipLaunchKernelGGL(convert_rows_kernel, dim3(blocks_for(mp * p.kxp)), dim3(256), 0, s, xyz_enc, p.xyz_dim, n, mp, act_buf(p, ws, p.buf_encx), p.kxp);
    hipLaunchKernelGGL(convert_rows_kernel, dim3(blocks_for(mp * p.kdp)), dim3(256), 0, s, dir_enc, p.dir_dim, n, mp, act_buf(p, ws, p.buf_encd), p.kdp);
    GENCHK(hipGetLastError());
    return run_layers(p, ws, net, w_flat, n, mp, raw, s);
}

size_t wgrad_partial_floats(const Plan& p) {
    size_t m = wgrad_partial_floats_for(p.head_K, 32);
    for (int li = 0; li < p.n_layers; ++li) {
        const size_t v = wgrad_partial_floats_for(p.buf_ld[p.layers[li].in_buf], p.layers[li].np);
        if (v > m) m = v;
    }
    return m;
}

namespace {
// running totals of the dead-tile statistics (knerf_tile_stats): stats[0] += live tiles of this pass, stats[1] += all its tiles
__global__ void tile_stats_kernel(const int* n_live, long long total, long long* stats) {
    stats[0] += (long long)*n_live; stats[1] += total;
}
}  // namespace

hipError_t backward(const Plan& p, const Workspace& ws, const NetDev& net, const float* raw, const float* draw, long long n,
                    float* grad_flat, hipStream_t s, float* partial, const int* live, const int* n_live, long long* stats) {
    const long long mp = (long long)padded_rows(n);
    if ((size_t)mp > ws.mp) return hipErrorInvalidValue;
    const int nl = p.n_layers;
    u16* dzh = dz_buf(p, ws, p.dz_head);
    hipLaunchKernelGGL(head_bwd_kernel, dim3(blocks_for(mp)), dim3(256), 0, s, raw, draw, n, mp, dzh);
    GENCHK(hipGetLastError());
    if (live && stats) {
        hipLaunchKernelGGL(tile_stats_kernel, dim3(1), dim3(1), 0, s, n_live, (n + 31) / 32, stats);
        GENCHK(hipGetLastError());
    }
    // dead-tile skipping: every GEMM and weight-gradient product below walks the list of live 32-row tiles instead of all of mp
    if (!ws.mask) return hipErrorInvalidValue;       // the forward that preceded must have run on a training workspace
    auto dgrad = [&](const u16* A, int lda, int K, const u16* Wd, int ldb, int N, int layer, u16* C, int ldc) {
        GemmArgs g{};
        g.A = A; g.lda = lda; g.Bt = Wd; g.ldb = ldb; g.M = mp; g.N = N; g.K = K;
        g.mask_in = ws.mask + (size_t)layer * relu_bits_bytes_per_row(p) * mp; g.mask_cols = p.up / 8; g.Cb = C; g.ldc = ldc;
        g.live = live; g.n_live = n_live;
        return launch_gemm(g, s);
    };
    // d trunk = dZ_head . H^T (the same product W_f (W_r1 (W_c dz_rgb)) + w_s dz_sigma the tape forms), masked by the last
    // trunk layer's relu; only the h columns carry a gradient (xyz_enc and dir_enc are constants)
    {
        const Layer& Ll = p.layers[nl - 1];
        GENCHK(dgrad(dzh, 32, 32, net.packed + p.head_wd_off, 32, p.up, nl - 1, dz_buf(p, ws, Ll.dz_buf), p.up));
    }
    for (int i = nl - 2; i >= 0; --i) {
        const Layer& Ln = p.layers[i + 1];
        const Layer& Li = p.layers[i];
        GENCHK(dgrad(dz_buf(p, ws, Ln.dz_buf), p.up, p.up, net.packed + Ln.wd_off, Ln.wd_ld, p.up, i, dz_buf(p, ws, Li.dz_buf), p.up));
    }
    for (int li = 0; li < nl; ++li) {
        const Layer& L = p.layers[li];
        WgradArgs w{};
        w.X = act_buf(p, ws, L.in_buf); w.ldx = p.buf_ld[L.in_buf];
        w.Z = dz_buf(p, ws, L.dz_buf) + L.dz_col0; w.ldz = p.dz_ld[L.dz_buf];
        w.steps = mp / 32; w.grad = grad_flat; w.w_off = L.w_off; w.b_off = L.b_off; w.n_real = L.n_real;
        w.n_seg = L.n_seg; w.seg[0] = L.seg[0]; w.seg[1] = L.seg[1];
        w.partial = partial;                    // one slab arena serves every launch: they follow each other on the stream
        w.live = live; w.n_live = n_live;
        GENCHK(launch_wgrad(w, w.ldx, L.np, s));
    }
    {   // head sums M = [h ; (xyz) ; dir]^T dZ_head [head_K][4] and s = column sums, into the aux buffer (expand_head)
        WgradArgs w{};
        w.X = act_buf(p, ws, p.buf_trunk); w.ldx = p.head_K;
        w.Z = dzh; w.ldz = 32;
        w.steps = mp / 32; w.grad = net.gaux; w.w_off = 0; w.b_off = p.head_K * 4; w.n_real = 4;
        w.n_seg = 1; w.seg[0] = Seg{0, p.head_K, 0};
        w.partial = partial;
        w.live = live; w.n_live = n_live;
        GENCHK(launch_wgrad(w, w.ldx, 32, s));
    }
    return hipSuccess;
}

}  // namespace gen
}  // namespace knerf
