// bwd_body.h -- fused data-gradient (dgrad) chain of one NeRF MLP: dL/d(rgb,sigma) per sample -> dZ of every layer.
//
// Backward of NeRFMLP.call (reference keras_nerf/model/nerf/mlp.py:29-50) as TensorFlow's tape computes it at
// nerf.py:376-377 / 405-406, restricted to what the weight gradients need (no gradient flows to the inputs):
//   dz_rgb = drgb * rgb(1-rgb);  dz_sig = dsigma * [sigma>0];  dh7 = H [dz_rgb ; dz_sig]  with the composed head matrix
//   H = [W_f W_r1 W_c | w_s] (layout.h "collapsed head": the same product W_f (W_r1 (W_c dz_rgb)) + w_s dz_sig the tape forms);
//   dz_l = dh_l * [h_l > 0];  dh_{l-1} = W_l dz_l  (layer_5: rows 0..255 only).
// Same register-resident structure as the forward chain (chain.h) with A = W instead of W^T.  Every dZ is written
// as B-operand blocks (layout.h "dz run") for the wgrad kernel; ReLU masks come from the forward pass.
#pragma once
#include "chain.h"
#include "kernels.h"
#include "layout.h"

namespace knerf {

// store schedule: 2 dZ blocks per out tile of every stage but the first (the last layer's dz is recomputed by wgrad, not stored --
// unless that layer is a concat layer, Shape::kSaveLastDz; the dz_head block is stored BEFORE the ring's prologue, i.e. it is older
// than every LDS-DMA and never counted)
template <class S>
constexpr StoreSched<S::kBwdStages> make_bwd_stores() {
    StoreSched<S::kBwdStages> t{};
    for (int st = 0; st < S::kBwdStages; ++st)
        t.st[st] = StoreStage{S::bwd_b0(st), st == 0 ? 1 : S::kKs, S::kOt, (st == 0 && !S::kSaveLastDz) ? 0 : 2, 0, 0};
    t.initial = 0;
    return t;
}
#ifdef KNERF_CONSERVATIVE_WAIT
constexpr StoreSched<1> kNoStoresB = {{{0, 1, 0, 0, 0, 0}}, 0};
template <class S> struct BwdWait { static constexpr WaitTable<S::kBwdBlocks> tab = make_wait_table<1, S::kBwdBlocks>(kNoStoresB); };
#else
template <class S> struct BwdWait { static constexpr WaitTable<S::kBwdBlocks> tab = make_wait_table<S::kBwdStages, S::kBwdBlocks>(make_bwd_stores<S>()); };
#endif

// dgrad chain of the 8 sample tiles (8 waves x 32 samples) of workgroup tile `wg_tile`
template <class S>
__device__ __forceinline__ void mlp_bwd_tile(const BwdArgs& a, char* smem, long long wg_tile) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));    // opaque per call: in a persistent loop nothing derived from it may be hoisted (and spilled)
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: per-tile base pointers stay in SGPRs
    const int grp = wave >> 2;                                    // stagger group (chain.h): 0 = waves 0-3, 1 = waves 4-7
    const int col = lane & 31, h = lane >> 5;
    long long tile = wg_tile * kWaves + wave;
    if (a.live) {
        // dead-tile skipping (composite.hip tile flags -> compact_tiles): the grid covers every tile, workgroups past the live
        // count leave at once; the last live workgroup's spare waves redo its last tile (identical stores)
        const long long n_all = (a.n_samples + kTile - 1) / kTile;
        const int n_live = *a.n_live < n_all ? *a.n_live : (int)n_all;      // never more entries than the pass has tiles
        if (a.stats && wg_tile == 0 && tid == 0) {                            // running totals for knerf_tile_stats
            atomicAdd(reinterpret_cast<unsigned long long*>(a.stats), (unsigned long long)n_live);
            atomicAdd(reinterpret_cast<unsigned long long*>(a.stats) + 1, (unsigned long long)(a.n_samples / kTile));
        }
        if (wg_tile * kWaves >= n_live) return;
        tile = a.live[tile < n_live ? tile : n_live - 1];
#ifdef KNERF_LIST_GUARD     // diagnostic build: a list entry outside the pass is counted and replaced instead of faulting
        const long long nt = (a.n_samples + kTile - 1) / kTile;
        if (tile < 0 || tile >= nt || n_live > nt) {
            if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(a.stats) + 2, 1ull);
            tile = 0;
        }
#endif
    }
    long long g = tile * kTile + col;
    const bool valid = g < a.n_samples;
    if (!valid) g = a.n_samples - 1;

    // everything this wave reads with ordinary loads is fetched (and waited for) before the LDS-DMA ring starts
    u32x4 mk[S::NL];
    const char* maskp = a.mask + mask_tile_off<S>((size_t)tile) + mask_lane_off(lane);
#pragma unroll
    for (int l = 0; l < S::NL; ++l) mk[l] = *reinterpret_cast<const u32x4*>(maskp + l * kSavedBlockStride);
    const f32x4 raw = reinterpret_cast<const f32x4*>(a.raw)[g];
    f32x4 dr = reinterpret_cast<const f32x4*>(a.draw)[g];
    if (!valid) dr = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int l = 0; l < S::NL; ++l) asm volatile("" ::"v"(mk[l]));   // pin the waits here, not inside the pipelined loop

    bf16x8 zhead;
#pragma unroll
    for (int j = 0; j < 8; ++j) zhead[j] = (__bf16)0.f;
    if (h == 0) {
        zhead[0] = (__bf16)(dr[0] * raw[0] * (1.f - raw[0]));
        zhead[1] = (__bf16)(dr[1] * raw[1] * (1.f - raw[1]));
        zhead[2] = (__bf16)(dr[2] * raw[2] * (1.f - raw[2]));
        zhead[3] = (__bf16)(raw[3] > 0.f ? dr[3] : 0.f);
    }
    char* dz = a.dz + dz_tile_off<S>((size_t)KNERF_STORE_TILE(tile));
    store_block(dz, S::kDzHead, lane, zhead);     // block kDzHead+1 stays zero (the buffer is zero-initialised)

    asm volatile("" ::: "memory");            // the store above stays ahead of the first LDS-DMA in program order
    Ring ring{a.stream, smem, tid, wave};
    ring.prologue_issue();
    ring.prologue_wait();
    Prefetch pf;
    pf.start<S::kBwdBlocks>(ring, lane);
    BwdWait<S> waits;

    constexpr int K = S::kKs, T = S::kOt;
    bf16x8 x[K], y[K];
    auto zero_init = [](int) { return zero_acc(); };
    // masked epilogue: dz_l = dh_l * [h_l > 0], applied to the packed bf16 pairs with the forward's bit mask
    // (tile ot -> word ot>>1, byte lane ot&1; chain.h relu_mask_bits / apply_mask_packed)
    auto mask_epi = [&](auto& out, int layer) {
        return [&, layer](int ot, f32x16 acc) {
            pack_acc(acc, out[2 * ot], out[2 * ot + 1]);
#ifndef KNERF_ABLATE_MASK      // timing experiment only
            apply_mask_packed(out[2 * ot], out[2 * ot + 1], mk[layer][ot >> 1] >> ((ot & 1) * 8));
#endif
            // the last layer's dz is not written: it is mask * (H dz_head) with only 4 input channels, which its wgrad job recomputes
            // from the dz_head block and the mask block (wgrad_body.h wgrad_last_recompute) -- unless that job is the two-range
            // one of a concat layer (Shape::kSaveLastDz)
            if (layer != S::NL - 1 || S::kSaveLastDz) {
                store_block(dz, K * layer + 2 * ot, lane, out[2 * ot]);
                store_block(dz, K * layer + 2 * ot + 1, lane, out[2 * ot + 1]);
            }
        };
    };
    // B0: dz_head (r, g, b, sigma) -> dh_{NL-1} -> dz_{NL-1} (y)
    dense_stage<0, 1, T, S::kBwdBlocks>(ring, pf, lane, grp, waits, zero_init, [&](int) { return zhead; }, mask_epi(y, S::NL - 1));
    // Bq: dz_l -> dz_{l-1}, l = NL-q; odd stages read y and write x, even stages the other way round
    static_for<S::NL - 1>([&](auto q_) {
        constexpr int q = decltype(q_)::value + 1;
        if constexpr (q % 2) dense_stage<S::bwd_b0(q), K, T, S::kBwdBlocks>(ring, pf, lane, grp, waits, zero_init, [&](int ks) { return y[ks]; }, mask_epi(x, S::NL - 1 - q));
        else dense_stage<S::bwd_b0(q), K, T, S::kBwdBlocks>(ring, pf, lane, grp, waits, zero_init, [&](int ks) { return x[ks]; }, mask_epi(y, S::NL - 1 - q));
    });
    ring_finish<S::kBwdBlocks>(ring, grp);
}


}  // namespace knerf
