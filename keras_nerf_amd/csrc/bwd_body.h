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

// store schedule: 2 dZ blocks per out tile of every stage but the first (dz7 is recomputed by wgrad, not stored; the dz_head
// block is stored BEFORE the ring's prologue, i.e. it is older than every LDS-DMA and never counted)
#ifndef KNERF_STORE_BURST
#define KNERF_STORE_BURST 2      // dZ blocks written per burst (see mlp_fwd.hip)
#endif
constexpr int kBwdBurstTiles = KNERF_STORE_BURST / 2;
#if KNERF_STORE_BURST == 2
constexpr StoreSched<8> kBwdStores = {{{0, 1, 8, 0, 0, 0}, {8, 16, 8, 2, 0, 0}, {136, 16, 8, 2, 0, 0}, {264, 16, 8, 2, 0, 0},
                                       {392, 16, 8, 2, 0, 0}, {520, 16, 8, 2, 0, 0}, {648, 16, 8, 2, 0, 0}, {776, 16, 8, 2, 0, 0}}, 0};
constexpr int kBwdStoreStages = 8;
#else
constexpr int kBwdParts = 8 / kBwdBurstTiles;
constexpr int kBwdStoreStages = 8 * kBwdParts;
constexpr StoreSched<kBwdStoreStages> make_bwd_burst_sched() {
    StoreSched<kBwdStoreStages> s{};
    for (int st = 0; st < 8; ++st)
        for (int p = 0; p < kBwdParts; ++p) {
            const int b0 = st == 0 ? 0 : 8 + 128 * (st - 1), nks = st == 0 ? 1 : 16;
            s.st[st * kBwdParts + p] = StoreStage{b0 + p * kBwdBurstTiles * nks, nks, kBwdBurstTiles, 0, st == 0 ? 0 : 2 * kBwdBurstTiles, 0};
        }
    s.initial = 0;
    return s;
}
constexpr StoreSched<kBwdStoreStages> kBwdStores = make_bwd_burst_sched();
#endif
#ifdef KNERF_CONSERVATIVE_WAIT
constexpr StoreSched<1> kNoStoresB = {{{0, 1, 0, 0, 0, 0}}, 0};
struct BwdWait { static constexpr WaitTable<kBwdBlocks> tab = make_wait_table<1, kBwdBlocks>(kNoStoresB); };
#else
struct BwdWait { static constexpr WaitTable<kBwdBlocks> tab = make_wait_table<kBwdStoreStages, kBwdBlocks>(kBwdStores); };
#endif

// dgrad chain of the 8 sample tiles (8 waves x 32 samples) of workgroup tile `wg_tile`
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
    u32x4 mk[8];
    const char* maskp = a.mask + mask_tile_off((size_t)tile) + lane * 16;
#pragma unroll
    for (int l = 0; l < 8; ++l) mk[l] = *reinterpret_cast<const u32x4*>(maskp + l * kSavedBlockStride);
    const f32x4 raw = reinterpret_cast<const f32x4*>(a.raw)[g];
    f32x4 dr = reinterpret_cast<const f32x4*>(a.draw)[g];
    if (!valid) dr = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int l = 0; l < 8; ++l) asm volatile("" ::"v"(mk[l]));   // pin the waits here, not inside the pipelined loop

    bf16x8 zhead;
#pragma unroll
    for (int j = 0; j < 8; ++j) zhead[j] = (__bf16)0.f;
    if (h == 0) {
        zhead[0] = (__bf16)(dr[0] * raw[0] * (1.f - raw[0]));
        zhead[1] = (__bf16)(dr[1] * raw[1] * (1.f - raw[1]));
        zhead[2] = (__bf16)(dr[2] * raw[2] * (1.f - raw[2]));
        zhead[3] = (__bf16)(raw[3] > 0.f ? dr[3] : 0.f);
    }
    char* dz = a.dz + dz_tile_off((size_t)tile);
    store_block(dz, kDzHead, lane, zhead);     // block kDzHead+1 stays zero (the buffer is zero-initialised)

    asm volatile("" ::: "memory");            // the store above stays ahead of the first LDS-DMA in program order
    Ring ring{a.stream, smem, tid, wave};
    ring.prologue_issue();
    ring.prologue_wait();
    Prefetch pf;
    pf.start<kBwdBlocks>(ring, lane);
    BwdWait waits;

    bf16x8 x[16], y[16];
    auto zero_init = [](int) { return zero_acc(); };
    // masked epilogue: dz_l = dh_l * [h_l > 0], applied to the packed bf16 pairs with the forward's bit mask
    // (tile ot -> word ot>>1, byte lane ot&1; chain.h relu_mask_bits / apply_mask_packed)
    auto mask_epi = [&](auto& out, int layer) {
        return [&, layer](int ot, f32x16 acc) {
            pack_acc(acc, out[2 * ot], out[2 * ot + 1]);
#ifndef KNERF_ABLATE_MASK      // timing experiment only
            apply_mask_packed(out[2 * ot], out[2 * ot + 1], mk[layer][ot >> 1] >> ((ot & 1) * 8));
#endif
            // dz7 is not written: it is mask7 * (H dz_head) with only 4 input channels, which the layer_7 wgrad job recomputes from
            // the dz_head block and the mask block (wgrad_body.h wgrad_l7_recompute)
            if (layer != 7 && (ot + 1) % kBwdBurstTiles == 0) {        // bursts of kBwdBurstTiles out tiles (the layer's dZ stays in registers anyway)
#pragma unroll
                for (int q = ot + 1 - kBwdBurstTiles; q <= ot; ++q) {
                    store_block(dz, 16 * layer + 2 * q, lane, out[2 * q]);
                    store_block(dz, 16 * layer + 2 * q + 1, lane, out[2 * q + 1]);
                }
            }
        };
    };
    // B0: dz_head (r, g, b, sigma) -> dh7 -> dz7 (y)
    dense_stage<0, 1, 8, kBwdBlocks>(ring, pf, lane, grp, waits, zero_init, [&](int) { return zhead; }, mask_epi(y, 7));
    // B1..B7: dz_l -> dz_{l-1}
    dense_stage<8, 16, 8, kBwdBlocks>(ring, pf, lane, grp, waits, zero_init, [&](int ks) { return y[ks]; }, mask_epi(x, 6));
    dense_stage<136, 16, 8, kBwdBlocks>(ring, pf, lane, grp, waits, zero_init, [&](int ks) { return x[ks]; }, mask_epi(y, 5));
    dense_stage<264, 16, 8, kBwdBlocks>(ring, pf, lane, grp, waits, zero_init, [&](int ks) { return y[ks]; }, mask_epi(x, 4));
    dense_stage<392, 16, 8, kBwdBlocks>(ring, pf, lane, grp, waits, zero_init, [&](int ks) { return x[ks]; }, mask_epi(y, 3));
    dense_stage<520, 16, 8, kBwdBlocks>(ring, pf, lane, grp, waits, zero_init, [&](int ks) { return y[ks]; }, mask_epi(x, 2));
    dense_stage<648, 16, 8, kBwdBlocks>(ring, pf, lane, grp, waits, zero_init, [&](int ks) { return x[ks]; }, mask_epi(y, 1));
    dense_stage<776, 16, 8, kBwdBlocks>(ring, pf, lane, grp, waits, zero_init, [&](int ks) { return y[ks]; }, mask_epi(x, 0));
    ring_finish<kBwdBlocks>(ring, grp);
}


}  // namespace knerf
