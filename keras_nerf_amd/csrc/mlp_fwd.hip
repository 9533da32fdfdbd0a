// mlp_fwd.hip -- fused positional encoding + 8x256 ReLU trunk + collapsed sigma / rgb head for one NeRF MLP (gfx950).
//
// Replaces, per sample: NeRFUtils.encode_position_and_directions (reference keras_nerf/model/nerf/utils.py:188-210),
// NeRFUtils.positional_encoding (utils.py:176-186) and NeRFMLP.call (mlp.py:29-50).
// Input: ray origins/directions [R,3] and t-values [R,S] (fp32).  Output: raw[R*S] = (r,g,b,sigma) fp32 after
// sigmoid / relu.  SAVE additionally writes every trunk layer's bf16 activations (B-operand blocks, layout.h) and the
// ReLU masks for the backward kernels.  The three linear layers behind the trunk (features, rgb_features, rgb; mlp.py:44-48)
// and the sigma head are evaluated as ONE 4-row stage on the composed matrix (layout.h "collapsed head").
#ifdef KNERF_ABLATE_FWD_STORES     // timing experiment only (r04): the training forward WITHOUT its saved-tensor stores -- the
#define KNERF_ABLATE_STORES         // upper bound of a forward that does not save what the backward will skip.  Waits as if no store
#define KNERF_CONSERVATIVE_WAIT     // had been issued; results of the backward are garbage, its timing is what it is.
#endif
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
    static_assert(2 + 3 * L <= NQ * 8, "encode: NQ k-steps hold 8 NQ features per lane half");
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

// store schedule of the SAVE variant: the kEncQ (4) enc blocks up front; per trunk layer 2 blocks behind every out tile (none for
// layer_0 where h0 is not saved) and one mask block at its end, the kDirQ (2) dir blocks right behind the last trunk layer (counted as
// that stage's end-of-stage stores) -- preceded by a second copy of the kEncQ enc blocks when the head takes [h ; xyz_enc ; dir_enc]
// (Shape::kTrunkXQ, a concat behind the last layer); the head stage stores nothing
template <class S>
constexpr StoreSched<S::kFwdStages> make_fwd_stores() {
    StoreSched<S::kFwdStages> t{};
    for (int st = 0; st < S::kFwdStages; ++st)
        t.st[st] = StoreStage{S::fwd_b0(st), S::fwd_nks(st), S::fwd_not(st), (st == S::NL || (st == 0 && !S::kSaveH0)) ? 0 : 2,
                              st == S::NL ? 0 : (st == S::NL - 1 ? 1 + S::kTrunkXQ + S::kDirQ : 1), 0};
    t.initial = S::kEncQ;
    return t;
}
constexpr StoreSched<1> kNoStores = {{{0, 1, 0, 0, 0, 0}}, 0};
template <class S> struct FwdWaitSave { static constexpr WaitTable<S::kFwdBlocks> tab = make_wait_table<S::kFwdStages, S::kFwdBlocks>(make_fwd_stores<S>()); };
template <class S> struct FwdWaitPlain { static constexpr WaitTable<S::kFwdBlocks> tab = make_wait_table<1, S::kFwdBlocks>(kNoStores); };

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

// NET only names the instantiation (0 = coarse pass, 1 = fine pass) for profiler summaries; S = the trunk shape (layout.h)
template <class S, bool SAVE, int NET>
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
    for (int i = tid; i < S::kFwdBiasTiles * 32; i += kThreads) bias_lds[i] = a.bias[i];

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
    constexpr int QX = S::kEncQ, QD = S::kDirQ;     // k-steps of the two encodings (4 / 2 for the reference's L = 10 / 4)
    bf16x8 enc[QX];
    encode<S::LX, QX>(px, py, pz, h, enc);

    char* act = nullptr; char* maskp = nullptr;
    if (SAVE) {
        act = a.act + act_tile_off<S>((size_t)KNERF_STORE_TILE(tile));
        maskp = a.mask + mask_tile_off<S>((size_t)KNERF_STORE_TILE(tile));
#ifndef KNERF_ABLATE_ENC_IO      // timing experiment only (with -DKNERF_CONSERVATIVE_WAIT): the upper bound of re-deriving the encodings in wgrad
#pragma unroll
        for (int q = 0; q < QX; ++q) store_block(act, S::kActEnc + q, lane, enc[q]);
#endif
    }

    ring.prologue_wait();
    FWD_STAMP(st1);
    Prefetch pf;
    pf.start<S::kFwdBlocks>(ring, lane);
#ifdef KNERF_CONSERVATIVE_WAIT
    FwdWaitPlain<S> waits;
#else
    std::conditional_t<SAVE, FwdWaitSave<S>, FwdWaitPlain<S>> waits;
#endif

    constexpr int K = S::kKs, T = S::kOt;        // k-steps and out tiles of a U-wide layer (16 / 8 at width 256)
    bf16x8 x[K], y[K];
    // relu epilogue of a trunk layer: out -> y (or x), activations + mask saved in training
    auto relu_epi = [&](bf16x8 (&out)[K], int layer, unsigned (&mbits)[4]) {
        return [&, layer](int ot, f32x16 acc) {
            pack_acc(acc, out[2 * ot], out[2 * ot + 1]);
            out[2 * ot] = relu_packed(out[2 * ot]);
            out[2 * ot + 1] = relu_packed(out[2 * ot + 1]);
            if (SAVE) {
                // two blocks per out tile, right behind it (h0 is not saved at all: the layer_1 wgrad job recomputes it from enc,
                // layout.h); longer bursts per wave were measured in round 2 and are slower
                if (layer > 0 || S::kSaveH0) {
                    store_block(act, S::act_h(layer) + 2 * ot, lane, out[2 * ot]);
                    store_block(act, S::act_h(layer) + 2 * ot + 1, lane, out[2 * ot + 1]);
                }
                // mask word of tile ot in byte lanes: even tile -> bits 0-7 / 16-23, odd tile -> bits 8-15 / 24-31
                const unsigned m = relu_mask_bits(out[2 * ot], out[2 * ot + 1]);
                if (ot & 1) mbits[ot >> 1] |= m << 8; else mbits[ot >> 1] = m;
                if (ot == T - 1) store16_wt(maskp, (unsigned)(layer * kSavedBlockStride + mask_lane_off(lane)), u32x4{mbits[0], mbits[1], mbits[2], mbits[3]});
            }
        };
    };
    unsigned mb[4] = {0u, 0u, 0u, 0u};      // T / 2 words carry bits (width 128: two of the four)
    auto bias_init = [&](int base) { return [&, base](int ot) { return bias_acc(bias_lds, base + ot, h); }; };

    // layer_0: 63 -> 256, into x
    dense_stage<0, QX, T, S::kFwdBlocks>(ring, pf, lane, grp, waits, bias_init(0), [&](int ks) { return enc[ks]; }, relu_epi(x, 0, mb));
    // layers 1 .. NL-1, ping-pong x -> y -> x ...: even layers write x, odd layers y.  A concat layer takes [h, xyz_enc] (skip concat:
    // h first, input second; mlp.py:36-38) -- the encoding is recomputed where it is consumed instead of pinning 16 VGPRs
    // across the trunk: 42 v_sin per re-encode vs ~1000 MFMAs per tile
    static_for<S::NL - 1>([&](auto l_) {
        constexpr int l = decltype(l_)::value + 1;
        auto run = [&](bf16x8 (&in)[K], bf16x8 (&out)[K]) {
            if constexpr (S::concat_in(l)) {
                bf16x8 encc[QX];
                encode<S::LX, QX>(px, py, pz, h, encc);
                dense_stage<S::fwd_b0(l), K + QX, T, S::kFwdBlocks>(ring, pf, lane, grp, waits, bias_init(T * l),
                                                                   [&](int ks) { return ks < K ? in[ks < K ? ks : 0] : encc[ks >= K ? ks - K : 0]; },
                                                                   relu_epi(out, l, mb));
            } else {
                dense_stage<S::fwd_b0(l), K, T, S::kFwdBlocks>(ring, pf, lane, grp, waits, bias_init(T * l), [&](int ks) { return in[ks]; },
                                                               relu_epi(out, l, mb));
            }
        };
        if constexpr (l % 2) run(x, y); else run(y, x);
    });
    // head: [h_{NL-1}, dir_enc] -> (r, g, b, sigma) pre-activations, one out tile on the composed matrix (layout.h); lanes of
    // half 0 hold rows 0-3 in acc[0..3].  sigmoid on rgb (mlp.py:26-27,48), relu on sigma (mlp.py:19-20,42).  A trunk that ends in a
    // concat (mlp.py:36-38 behind the LAST layer) hands [h, xyz_enc] to sigma and features: QH more k-steps between h and dir_enc,
    // the encoding recomputed once more and saved a second time right behind h_{NL-1}, where the head's weight-gradient job reads it
    constexpr int QH = S::kTrunkXQ;
    bf16x8 ench[QH > 0 ? QH : 1];
    if constexpr (QH > 0) {
        encode<S::LX, QX>(px, py, pz, h, ench);
#ifndef KNERF_ABLATE_ENC_IO
        if (SAVE) {
#pragma unroll
            for (int q = 0; q < QH; ++q) store_block(act, S::kActHeadEnc + q, lane, ench[q]);
        }
#endif
    }
    bf16x8 dirc[QD];
    encode<S::LD, QD>(dx, dy, dz, h, dirc);
#ifndef KNERF_ABLATE_ENC_IO
    if (SAVE) {
#pragma unroll
        for (int q = 0; q < QD; ++q) store_block(act, S::kActDir + q, lane, dirc[q]);
    }
#endif
    auto head = [&](bf16x8 (&in)[K]) {
        dense_stage<S::fwd_b0(S::NL), K + QH + QD, 1, S::kFwdBlocks>(ring, pf, lane, grp, waits, bias_init(T * S::NL),
                                [&](int ks) { return ks < K ? in[ks < K ? ks : 0] : (ks < K + QH ? ench[(ks >= K && ks < K + QH) ? ks - K : 0] : dirc[ks >= K + QH ? ks - K - QH : 0]); },
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
    };
    if constexpr ((S::NL - 1) % 2) head(y); else head(x);
    ring_finish<S::kFwdBlocks>(ring, grp);
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

template <class S>
hipError_t launch_mlp_fwd_t(const FwdArgs& a, bool save, hipStream_t stream) {
    const long long tiles = (a.n_samples + kTile - 1) / kTile;
    const int grid = (int)((tiles + kWaves - 1) / kWaves);
    const size_t lds = kRingBytes + S::kFwdBiasTiles * 32 * sizeof(float);
    static AttrOnce once;
    hipError_t ae = once([&]() -> hipError_t {
        const void* fns[4] = {reinterpret_cast<const void*>(mlp_fwd_kernel<S, false, 0>), reinterpret_cast<const void*>(mlp_fwd_kernel<S, false, 1>),
                              reinterpret_cast<const void*>(mlp_fwd_kernel<S, true, 0>), reinterpret_cast<const void*>(mlp_fwd_kernel<S, true, 1>)};
        for (const void* f : fns) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    });
    if (ae != hipSuccess) return ae;
    const dim3 g(grid), b(kThreads);
    if (save) {
        if (a.net == 0) hipLaunchKernelGGL((mlp_fwd_kernel<S, true, 0>), g, b, lds, stream, a);
        else hipLaunchKernelGGL((mlp_fwd_kernel<S, true, 1>), g, b, lds, stream, a);
    } else {
        if (a.net == 0) hipLaunchKernelGGL((mlp_fwd_kernel<S, false, 0>), g, b, lds, stream, a);
        else hipLaunchKernelGGL((mlp_fwd_kernel<S, false, 1>), g, b, lds, stream, a);
    }
    return hipGetLastError();
}

// explicit instantiation of this translation unit's shape(s), `extern template` for the others (layout.h KNERF_FUSED_SHAPES)
#define KNERF_X(I, ...) KNERF_PICK(I, template, extern template) hipError_t launch_mlp_fwd_t<KNERF_SHAPE_T(__VA_ARGS__)>(const FwdArgs&, bool, hipStream_t);
KNERF_FUSED_SHAPES(KNERF_X)
#undef KNERF_X

#if KNERF_HAS_DISPATCH
hipError_t launch_mlp_fwd(const FwdArgs& a, bool save, hipStream_t stream) {
    switch (a.shape) {
#define KNERF_X(I, ...) case I: return launch_mlp_fwd_t<KNERF_SHAPE_T(__VA_ARGS__)>(a, save, stream);
        KNERF_FUSED_SHAPES(KNERF_X)
#undef KNERF_X
        default: return hipErrorInvalidValue;
    }
}
#endif

}  // namespace knerf
