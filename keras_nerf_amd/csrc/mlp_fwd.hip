// mlp_fwd.hip -- fused positional encoding + 8x256 ReLU trunk + collapsed sigma / rgb head for one NeRF MLP (gfx950).
//
// Replaces, per sample: NeRFUtils.encode_position_and_directions (reference keras_nerf/model/nerf/utils.py:188-210),
// NeRFUtils.positional_encoding (utils.py:176-186) and NeRFMLP.call (mlp.py:29-50).
// Input: ray origins/directions [R,3] and t-values [R,S] (fp32).  Output: raw[R*S] = (r,g,b,sigma) fp32 after
// sigmoid / relu.  SAVE additionally writes every trunk layer's bf16 activations (B-operand blocks, layout.h) and the
// ReLU masks for the backward kernels.  The three linear layers behind the trunk (features, rgb_features, rgb; mlp.py:44-48)
// and the sigma head are evaluated as ONE 4-row stage on the composed matrix (layout.h "collapsed head").
#include "chain.h"
#include "kernels.h"
#include "layout.h"

namespace knerf {

// sin / cos of 2^i * x, i = 0..L-1, for the three components, written straight into B-operand slots.
// Range reduction is exact: r = x/(2pi) as hi+lo floats, fract(2^i * r_hi) is exact in fp32, the v_sin_f32 argument
// is in revolutions.  cos = sin shifted by a quarter revolution, so both lane halves run the same instruction.
template <int L, int NQ>
__device__ __forceinline__ void encode(float x, float y, float z, int h, bf16x8 (&out)[NQ]) {
    const float C1 = 0.15915494f;             // fl(1/(2 pi))
    const float C2 = 6.4206383e-09f;          // 1/(2 pi) - C1   (0.15915494309189535 - 0.15915493667125702)
    float v[3] = {x, y, z};
    float rh[3], rl[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        rh[c] = v[c] * C1;
        rl[c] = __builtin_fmaf(v[c], C1, -rh[c]) + v[c] * C2;
    }
    const float phase = h ? 0.25f : 0.0f;
    float e[NQ * 8];
#pragma unroll
    for (int m = 0; m < NQ * 8; ++m) e[m] = 0.f;
    e[0] = h ? z : x;
    e[1] = h ? 0.f : y;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const float s = (float)(1 << i);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float a = rh[c] * s;
            float f = a - __builtin_floorf(a);
            float arg = f + (rl[c] * s + phase);
            e[2 + 3 * i + c] = __builtin_amdgcn_sinf(arg);
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) out[q][j] = (__bf16)e[8 * q + j];
}

// store schedule of the SAVE variant: 4 enc blocks up front, 2 blocks per out tile (none for layer_0) and one mask block per trunk layer, the
// 2 dir blocks right behind layer_7 (counted as that stage's end-of-stage stores); the head stage stores nothing
#ifndef KNERF_STORE_BURST
#define KNERF_STORE_BURST 2      // saved blocks written per burst: 2 = behind every out tile; 4 / 8 / 16 = every 2nd / 4th / 8th tile
#endif
constexpr int kBurstTiles = KNERF_STORE_BURST / 2;          // out tiles per burst
// StoreSched models a burst as a pseudo-stage of kBurstTiles out tiles whose stores all come at its end
#if KNERF_STORE_BURST == 2
constexpr StoreSched<9> kFwdStores = {{{0, 4, 8, 0, 1, 0}, {32, 16, 8, 2, 1, 0}, {160, 16, 8, 2, 1, 0}, {288, 16, 8, 2, 1, 0},
                                       {416, 16, 8, 2, 1, 0}, {544, 20, 8, 2, 1, 0}, {704, 16, 8, 2, 1, 0}, {832, 16, 8, 2, 3, 0},
                                       {960, 18, 1, 0, 0, 0}}, 4};
constexpr int kFwdStoreStages = 9;
#else
constexpr int kParts = 8 / kBurstTiles;                      // bursts per trunk stage
constexpr int kFwdStoreStages = 8 * kParts + 1;
constexpr StoreSched<kFwdStoreStages> make_fwd_burst_sched() {
    StoreSched<kFwdStoreStages> s{};
    const int b0[8] = {0, 32, 160, 288, 416, 544, 704, 832}, nks[8] = {4, 16, 16, 16, 16, 20, 16, 16};
    for (int l = 0; l < 8; ++l)
        for (int p = 0; p < kParts; ++p) {
            const int extra = p == kParts - 1 ? (l == 7 ? 3 : 1) : 0;      // mask block (+ the 2 dir blocks behind layer_7)
            s.st[l * kParts + p] = StoreStage{b0[l] + p * kBurstTiles * nks[l], nks[l], kBurstTiles, 0, (l == 0 ? 0 : 2 * kBurstTiles) + extra, 0};
        }
    s.st[8 * kParts] = StoreStage{960, 18, 1, 0, 0, 0};
    s.initial = 4;
    return s;
}
constexpr StoreSched<kFwdStoreStages> kFwdStores = make_fwd_burst_sched();
#endif
constexpr StoreSched<1> kNoStores = {{{0, 1, 0, 0, 0, 0}}, 0};
struct FwdWaitSave { static constexpr WaitTable<kFwdBlocks> tab = make_wait_table<kFwdStoreStages, kFwdBlocks>(kFwdStores); };
struct FwdWaitPlain { static constexpr WaitTable<kFwdBlocks> tab = make_wait_table<1, kFwdBlocks>(kNoStores); };

#ifdef KNERF_FWD_STAMPS     // diagnostic build only (tools/fwd_stamps.py): per-workgroup s_memtime at entry / first MFMA / exit + HW_ID of the fine inference launch
__device__ unsigned long long g_fwd_stamps[4096 * 4];
__device__ __forceinline__ unsigned long long fwd_stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define FWD_STAMP(v) const unsigned long long v = fwd_stamp()
#else
#define FWD_STAMP(v)
#endif

// NET only names the instantiation (0 = coarse pass, 1 = fine pass) for profiler summaries
template <bool SAVE, int NET>
__global__ __launch_bounds__(kThreads, 2) void mlp_fwd_kernel(FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    FWD_STAMP(st0);
    float* bias_lds = reinterpret_cast<float*>(smem + kRingBytes);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the per-tile base pointers stay in SGPRs
    const int grp = wave >> 2;                                    // stagger group (chain.h): 0 = waves 0-3, 1 = waves 4-7
    const int col = lane & 31, h = lane >> 5;

    // biases -> LDS (plain loads, before any LDS-DMA is in flight; requesting the ring's first pages ahead of these loads shortens
    // the measured ramp by 0.3 us per workgroup and lengthens the body by 0.15: not worth a second ordering rule)
    for (int i = tid; i < kFwdBiasTiles * 32; i += kThreads) bias_lds[i] = a.bias[i];

    const long long tile = (long long)blockIdx.x * kWaves + wave;
    long long g = tile * kTile + col;
    const bool valid = g < a.n_samples;
    if (!valid) g = a.n_samples - 1;
    const long long ray = g / a.S;
    const float t = a.t[g];
    const float ox = a.o[ray * 3 + 0], oy = a.o[ray * 3 + 1], oz = a.o[ray * 3 + 2];
    const float dx = a.d[ray * 3 + 0], dy = a.d[ray * 3 + 1], dz = a.d[ray * 3 + 2];
    // p = o + d * t  (two roundings, as the reference's mul then add; utils.py:193-194)
    const float px = __fadd_rn(ox, __fmul_rn(dx, t)), py = __fadd_rn(oy, __fmul_rn(dy, t)), pz = __fadd_rn(oz, __fmul_rn(dz, t));
    __syncthreads();

    Ring ring{a.stream, smem, tid, wave};
    ring.prologue_issue();
    asm volatile("" ::: "memory");            // every store below stays behind the prologue's LDS-DMA (StoreSched counts on it)

    // the encodings are recomputed where they are consumed (layer_0 and layer_5; rgb_features) instead of pinning 24
    // VGPRs across the trunk: 42 v_sin per re-encode vs ~1200 MFMAs per tile
    bf16x8 enc[4];
    encode<kLx, 4>(px, py, pz, h, enc);

    char* act = nullptr; char* maskp = nullptr;
    if (SAVE) {
        act = a.act + act_tile_off((size_t)tile);
        maskp = a.mask + mask_tile_off((size_t)tile);
#ifndef KNERF_ABLATE_ENC_IO      // timing experiment only (with -DKNERF_CONSERVATIVE_WAIT): the upper bound of re-deriving the encodings in wgrad
#pragma unroll
        for (int q = 0; q < 4; ++q) store_block(act, kActEnc + q, lane, enc[q]);
#endif
    }

    ring.prologue_wait();
    FWD_STAMP(st1);
    Prefetch pf;
    pf.start<kFwdBlocks>(ring, lane);
#ifdef KNERF_CONSERVATIVE_WAIT
    FwdWaitPlain waits;
#else
    std::conditional_t<SAVE, FwdWaitSave, FwdWaitPlain> waits;
#endif

    bf16x8 x[16], y[16];
    // relu epilogue of a trunk layer: out -> y (or x), activations + mask saved in training
    auto relu_epi = [&](bf16x8 (&out)[16], int layer, unsigned (&mbits)[4]) {
        return [&, layer](int ot, f32x16 acc) {
            pack_acc(acc, out[2 * ot], out[2 * ot + 1]);
            out[2 * ot] = relu_packed(out[2 * ot]);
            out[2 * ot + 1] = relu_packed(out[2 * ot + 1]);
            if (SAVE) {
                // the layer's output stays in registers until the next layer has read it, so the blocks of kBurstTiles out
                // tiles can leave together: longer contiguous bursts per wave (2 KiB x kBurstTiles) for the same registers
                // (h0 is not saved at all: the layer_1 wgrad job recomputes it from enc, layout.h)
                if (layer > 0 && (ot + 1) % kBurstTiles == 0) {
#pragma unroll
                    for (int q = ot + 1 - kBurstTiles; q <= ot; ++q) {
                        store_block(act, act_h(layer) + 2 * q, lane, out[2 * q]);
                        store_block(act, act_h(layer) + 2 * q + 1, lane, out[2 * q + 1]);
                    }
                }
                // mask word of tile ot in byte lanes: even tile -> bits 0-7 / 16-23, odd tile -> bits 8-15 / 24-31
                const unsigned m = relu_mask_bits(out[2 * ot], out[2 * ot + 1]);
                if (ot & 1) mbits[ot >> 1] |= m << 8; else mbits[ot >> 1] = m;
                if (ot == 7) store16_wt(maskp, (unsigned)(layer * kSavedBlockStride + lane * 16), u32x4{mbits[0], mbits[1], mbits[2], mbits[3]});
            }
        };
    };
    unsigned mb[4];
    int btile = 0;
    auto bias_init = [&](int base) { return [&, base](int ot) { return bias_acc(bias_lds, base + ot, h); }; };

    // layer_0: 63 -> 256
    dense_stage<0, 4, 8, kFwdBlocks>(ring, pf, lane, grp, waits, bias_init(0), [&](int ks) { return enc[ks]; }, relu_epi(x, 0, mb));
    // layer_1..4
    dense_stage<32, 16, 8, kFwdBlocks>(ring, pf, lane, grp, waits, bias_init(8), [&](int ks) { return x[ks]; }, relu_epi(y, 1, mb));
    dense_stage<160, 16, 8, kFwdBlocks>(ring, pf, lane, grp, waits, bias_init(16), [&](int ks) { return y[ks]; }, relu_epi(x, 2, mb));
    dense_stage<288, 16, 8, kFwdBlocks>(ring, pf, lane, grp, waits, bias_init(24), [&](int ks) { return x[ks]; }, relu_epi(y, 3, mb));
    dense_stage<416, 16, 8, kFwdBlocks>(ring, pf, lane, grp, waits, bias_init(32), [&](int ks) { return y[ks]; }, relu_epi(x, 4, mb));
    // layer_5: [h4, xyz_enc] -> 256   (skip concat: h first, input second; mlp.py:36-38)
    bf16x8 enc5[4];
    encode<kLx, 4>(px, py, pz, h, enc5);
    dense_stage<544, 20, 8, kFwdBlocks>(ring, pf, lane, grp, waits, bias_init(40), [&](int ks) { return ks < 16 ? x[ks < 16 ? ks : 0] : enc5[ks >= 16 ? ks - 16 : 0]; },
                            relu_epi(y, 5, mb));
    dense_stage<704, 16, 8, kFwdBlocks>(ring, pf, lane, grp, waits, bias_init(48), [&](int ks) { return y[ks]; }, relu_epi(x, 6, mb));
    dense_stage<832, 16, 8, kFwdBlocks>(ring, pf, lane, grp, waits, bias_init(56), [&](int ks) { return x[ks]; }, relu_epi(y, 7, mb));
    (void)btile;
    // head: [h7, dir_enc] -> (r, g, b, sigma) pre-activations, one out tile on the composed matrix (layout.h); lanes of
    // half 0 hold rows 0-3 in acc[0..3].  sigmoid on rgb (mlp.py:26-27,48), relu on sigma (mlp.py:19-20,42).
    bf16x8 dirc[2];
    encode<kLd, 2>(dx, dy, dz, h, dirc);
#ifndef KNERF_ABLATE_ENC_IO
    if (SAVE) {
#pragma unroll
        for (int q = 0; q < 2; ++q) store_block(act, kActDir + q, lane, dirc[q]);
    }
#endif
    dense_stage<960, 18, 1, kFwdBlocks>(ring, pf, lane, grp, waits, bias_init(64), [&](int ks) { return ks < 16 ? y[ks < 16 ? ks : 0] : dirc[ks >= 16 ? ks - 16 : 0]; },
                            [&](int, f32x16 acc) {
                                if (valid && h == 0) {
                                    f32x4 r;
                                    r[0] = 1.f / (1.f + expf(-acc[0]));
                                    r[1] = 1.f / (1.f + expf(-acc[1]));
                                    r[2] = 1.f / (1.f + expf(-acc[2]));
                                    r[3] = acc[3] > 0.f ? acc[3] : 0.f;
                                    reinterpret_cast<f32x4*>(a.raw)[g] = r;
                                }
                            });
    ring_finish<kFwdBlocks>(ring, grp);
#ifdef KNERF_FWD_STAMPS
    if (!SAVE && NET == 1 && threadIdx.x == 0 && blockIdx.x < 4096) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* o = g_fwd_stamps + blockIdx.x * 4;
        o[0] = st0; o[1] = st1; o[2] = fwd_stamp(); o[3] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
}

#ifdef KNERF_FWD_STAMPS
extern "C" int knerf_debug_fwd_stamps(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_fwd_stamps), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif

hipError_t launch_mlp_fwd(const FwdArgs& a, bool save, hipStream_t stream) {
    const long long tiles = (a.n_samples + kTile - 1) / kTile;
    const int grid = (int)((tiles + kWaves - 1) / kWaves);
    const size_t lds = kRingBytes + kFwdBiasTiles * 32 * sizeof(float);
    static AttrOnce once;
    hipError_t ae = once([&]() -> hipError_t {
        const void* fns[4] = {reinterpret_cast<const void*>(mlp_fwd_kernel<false, 0>), reinterpret_cast<const void*>(mlp_fwd_kernel<false, 1>),
                              reinterpret_cast<const void*>(mlp_fwd_kernel<true, 0>), reinterpret_cast<const void*>(mlp_fwd_kernel<true, 1>)};
        for (const void* f : fns) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    });
    if (ae != hipSuccess) return ae;
    const dim3 g(grid), b(kThreads);
    if (save) {
        if (a.net == 0) hipLaunchKernelGGL((mlp_fwd_kernel<true, 0>), g, b, lds, stream, a);
        else hipLaunchKernelGGL((mlp_fwd_kernel<true, 1>), g, b, lds, stream, a);
    } else {
        if (a.net == 0) hipLaunchKernelGGL((mlp_fwd_kernel<false, 0>), g, b, lds, stream, a);
        else hipLaunchKernelGGL((mlp_fwd_kernel<false, 1>), g, b, lds, stream, a);
    }
    return hipGetLastError();
}

}  // namespace knerf
