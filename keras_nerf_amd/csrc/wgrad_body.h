// wgrad_body.h -- weight and bias gradients of one NeRF MLP from the saved activations and dZ blocks (gfx950).
//
// dW_l[in][out] += sum_samples act_l[s][in] * dz_l[s][out],  db_l[out] += sum_samples dz_l[s][out]
// i.e. what tape.gradient(...) returns for the 24 trainable tensors at reference keras_nerf/model/nerf/nerf.py:376-377 /
// 405-406, accumulated straight into the flat fp32 accumulator that nerf.py:383-384 / 412-413 maintain (the 1/C chunk
// factor is already folded into dL/dimage by the compositing kernel).
//
// The contraction runs over SAMPLES, which the chain kernels keep on the lane axis, so both MFMA operands need the
// transposed view.  The saved 1 KiB blocks are copied verbatim into LDS (LDS-DMA) and read back with
// ds_read_b64_tr_b16: each lane ends up holding 8 consecutive samples of one feature.  HBM-bound by construction
// (128 FLOP per byte at width 256): a workgroup owns one layer ("job") and a contiguous range of sample tiles, keeps
// the layer's gradient in registers (one 32-column strip per wave) and adds it to global memory with fp32 atomics
// once at the end.
#pragma once
#include <hip/hip_runtime.h>
#include "chain.h"
#include "kernels.h"
#include "layout.h"

namespace knerf {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

constexpr int kWgThreads = 512, kWgWaves = 8;
#ifdef KNERF_ABLATE_HALF_WGRAD_MATH     // timing experiment only: one of a tile's two k-steps in the recomputing jobs as well
constexpr int kWgKSteps = 1;
#else
constexpr int kWgKSteps = 2;            // k-steps of 16 samples per 32-sample tile
#endif
constexpr int kWgScratch = kWgWaves * 1024;   // landing zone of padding LDS-DMA copies (never read)

// LDS-DMA of 16 B per lane, invisible to hipcc's wait-count pass (the builtin form makes it drain vmcnt(0) before
// every transposed LDS read).  lds_dst = wave-uniform LDS byte address; completion is counted by hand (vmcnt).
#ifndef KNERF_WGRAD_LOAD_POLICY
#define KNERF_WGRAD_LOAD_POLICY "nt"     // once-read streams (measured -2.5 %)
#endif
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off " KNERF_WGRAD_LOAD_POLICY "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// the same copy without the compiler-level memory fence: for a copy issued in the middle of a tile's products, whose LDS reads
// (of other slots) may move across it.  Ordering against the reads of its own slot comes from the s_waitcnt/s_barrier pair.
__device__ __forceinline__ void glds16_nofence(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off " KNERF_WGRAD_LOAD_POLICY "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst));
}
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst, bool late) {
    if (late) glds16_nofence(gsrc, lds_dst); else glds16(gsrc, lds_dst);
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}

// which block of a staged run a copy fetches
#ifdef KNERF_ABLATE_HALF_SAVED     // timing experiment only (chain.h store8_wt): every odd block is fetched from its even neighbour's
#define KNERF_SRC_BLOCK(b) ((b) & ~1)   // address (an L2 hit: a sister wave reads it in the same iteration), so HBM delivers HALF the bytes at an unchanged instruction count
#else
#define KNERF_SRC_BLOCK(b) (b)
#endif

#ifdef KNERF_WGRAD_STAMPS   // diagnostic build only (tools/kbench.py --stamps): per-workgroup cycle totals of the loop phases
__device__ unsigned long long g_wgrad_stamps[1024 * 8];
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define STAMP(var) const unsigned long long var = stamp()
#else
#define STAMP(var)
#endif

// KNERF_WGRAD_ABLATE_LDS (PMC experiments only, DESIGN.md 2.5 "LDS bank conflicts"; results are wrong, the counters are the point):
// bit 0: the sixteen ds_read_b32 of the layer-7 mask words per tile become one broadcast read; bit 1: the row-wise ds_read_b128 of
// sample-major blocks (enc in the layer_1 job, dz_head in the layer_7 job) read linearly (lane * 16) instead of through saved_off
#ifndef KNERF_WGRAD_ABLATE_LDS
#define KNERF_WGRAD_ABLATE_LDS 0
#endif

struct WgradPlan {          // one entry per workgroup, built on the host (knerf_api.hip)
    int job, split, nsplit, pad;
};

// Flush of one 32x32 accumulator tile: element i of lane (c, hh) goes to dst[(row0 + (i&3) + 8(i>>2)) * ncols + col] (an index into
// the gradient buffer, >= kAuxBase: into the head sums, < 0: dropped).  The 16 table look-ups are requested together and the
// atomics follow (element by element the look-up's latency was exposed 16 times per tile: 29 us per launch, 1030 atomics per wave).
// Deterministic mode (a.partial != null, knerf_set_option "deterministic"): no atomics -- the workgroup's sums go to its own slab of
// `partial` with plain stores (element index = the one the destination table is read at) and wgrad_reduce_kernel (wgrad.hip) adds
// the slabs of a job in split order, so the gradient is bit-reproducible from run to run.
// `stride` = layout.h wgrad_partial_stride<S>(): floats per workgroup slab = the shape's largest job table (82,176 for the default shape)
__device__ __forceinline__ void flush_acc(const WgradArgs& a, const int* dst, int row0, int ncols, int col, const f32x16& acc, int stride) {
    if (a.partial) {
        float* p = a.partial + (size_t)blockIdx.x * stride;
#pragma unroll
        for (int i = 0; i < 16; ++i) p[(row0 + (i & 3) + 8 * (i >> 2)) * ncols + col] = acc[i];
        return;
    }
    int d[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = __builtin_nontemporal_load(dst + (row0 + (i & 3) + 8 * (i >> 2)) * ncols + col);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
#ifdef KNERF_WGRAD_ABLATE_FLUSH      // timing experiment only: what the atomic flush costs
        asm volatile("" ::"v"(acc[i]), "v"(d[i]));
#else
        if (d[i] >= 0) atomicAdd(d[i] < a.aux_base ? a.grad + d[i] : a.aux + (d[i] - a.aux_base), acc[i]);
#endif
    }
}

// bias row (column sums of dz): element `idx` of the job's table, held by the half-0 lanes
__device__ __forceinline__ void flush_bias(const WgradArgs& a, const int* dst, int idx, int hh, float v, int stride) {
    if (a.partial) {
        if (hh == 0) a.partial[(size_t)blockIdx.x * stride + idx] = v;
        return;
    }
    const int d = dst[idx];
    if (d >= 0 && hh == 0) atomicAdd(d < a.aux_base ? a.grad + d : a.aux + (d - a.aux_base), v);
}

// 8 consecutive samples (k-step kk of the tile, MFMA half h) of feature (lane&31) of tile-pair `pair` in a staged
// region: two transposed reads.  lane_off[r] = per-lane byte offset of read r inside the pair's two blocks.
__device__ __forceinline__ bf16x8 tr_frag(const char* region, int pair, int kk, const int (&lane_off)[2]) {
    const char* base = region + pair * 2048 + kk * 512;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + lane_off[0]));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + lane_off[1]));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// The job body walks a contiguous range [t0, t1) of sample tiles ...
struct ContigSeq {
    static constexpr bool kIsList = false;
    int t0, t1;
    __device__ __forceinline__ int count() const { return t1 - t0; }
    __device__ __forceinline__ int tile(int i) const { return t0 + i; }
};
// ... or entries [i0, i1) of the list of LIVE tiles (dead-tile skipping: composite.hip flags -> compact_tiles).  The index is
// wave-uniform, so the look-up is a SCALAR load -- written as inline asm, because hipcc would make it a vector load here (it cannot
// prove the list read-only next to the kernel's atomics and asm copies), and a compiler-visible VMEM load inside the hand-counted
// vmcnt pipeline makes it drain vmcnt(0) every tile.  SMEM shares lgkmcnt with LDS, so load and wait sit in ONE asm statement at a
// point where no LDS read is pending: tile_sync (prologue), and tile_wait below, which wraps an iteration's own vmcnt wait and
// barrier so that the look-up's latency hides behind them.
struct ListSeq {
    static constexpr bool kIsList = true;
    const int* list;
    int i0, i1;
    __device__ __forceinline__ int count() const { return i1 - i0; }
    __device__ __forceinline__ unsigned long long addr(int i) const {
        const unsigned long long p = (unsigned long long)(list + i0 + i);
        // NB the builtin returns int: without the casts the low half is SIGN-extended into the high one (a list above a 2 GiB
        // boundary then faults at 0xffffffff........ -- r03, found the hard way)
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)p), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32));
        return ((unsigned long long)hi << 32) | (unsigned long long)lo;
    }
    __device__ __forceinline__ int tile(int i) const {          // load and wait: for places outside the pipelined loop
        int t;
        asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "s"(addr(i)) : "memory");
        return t;
    }
};

// Top of a pipelined iteration: wait until at most VM of this wave's LDS-DMA copies are in flight (and, with LGKM0, for its own LDS
// writes), take the workgroup barrier, and return the tile that this iteration stages (entry `idx` of the sequence).
template <int VM, bool LGKM0, class Seq>
__device__ __forceinline__ int wait_barrier_next(const Seq& seq, int idx) {
    if constexpr (Seq::kIsList) {
        int t;
        if constexpr (LGKM0)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_load_dword %0, %1, 0x0\n\ts_waitcnt vmcnt(%2)\n\ts_barrier\n\ts_waitcnt lgkmcnt(0)"
                         : "=s"(t) : "s"(seq.addr(idx)), "n"(VM) : "memory");
        else
            asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt vmcnt(%2)\n\ts_barrier\n\ts_waitcnt lgkmcnt(0)"
                         : "=s"(t) : "s"(seq.addr(idx)), "n"(VM) : "memory");
        return t;
    } else {
        if constexpr (LGKM0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(VM) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM) : "memory");
        __builtin_amdgcn_s_barrier();
        return seq.tile(idx);
    }
}

// act_blk: first act block of the input's h tiles; act_blk2 + S::kKs: first block of the tiles behind them (a concat layer's encoding: directly behind
// its h for the first concat layer of a shape -- act_blk2 == act_blk, one contiguous range -- elsewhere for later ones)
template <class S, int NI, int NO, class Seq>
__device__ __forceinline__ void wgrad_job_body(const WgradArgs& a, const int job, const int act_blk, const int act_blk2, const int dz_blk,
                                               const Seq seq, char* smem) {
    constexpr int WO = NO >= 8 ? 8 : (NO >= 4 ? 4 : (NO >= 2 ? 2 : 1));     // waves across output tiles (8 / 4 / 2 at width 256 / 128 / 64; 1: the head)
    constexpr int WI = kWgWaves / WO;                        // waves across input tiles (+ the bias row)
    constexpr int ROWS = NI + 1;
    constexpr int NACC = (ROWS + WI - 1) / WI;
    constexpr int BLK_IN = 2 * NI, BLK_DZ = 2 * NO;
    constexpr int TILE_BYTES = (BLK_IN + BLK_DZ) * 1024;
    constexpr int G_IN = (BLK_IN + kWgWaves - 1) / kWgWaves, G_DZ = (BLK_DZ + kWgWaves - 1) / kWgWaves;
    constexpr int G = G_IN + G_DZ;                           // LDS-DMA instructions per wave per iteration (uniform)
    // staged tiles: as many as LDS holds, at most KNERF_WGRAD_MAX_SLOTS -- a workgroup's rate is (bytes in flight) / (HBM
    // latency), so the jobs with small tiles (layer_0, head: 20 KiB) need more of them in flight to keep pace with the 32 KiB ones
#ifndef KNERF_WGRAD_MAX_SLOTS
#define KNERF_WGRAD_MAX_SLOTS 6
#endif
    constexpr int NS_FIT = (160 * 1024 - kWgScratch) / TILE_BYTES;
    constexpr int NS = NS_FIT >= KNERF_WGRAD_MAX_SLOTS ? KNERF_WGRAD_MAX_SLOTS : (NS_FIT >= 4 ? 4 : 3);
    static_assert(G * (NS - 2) <= 60, "vmcnt immediate");
    static_assert(NS * TILE_BYTES + kWgScratch <= 160 * 1024, "LDS budget");
    static_assert(NO == WO, "one output tile per wave column");

    const int cnt = seq.count();
    if (cnt <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: keeps the per-wave tile branches uniform
    const int wo = WO == 1 ? 0 : wave % WO, wi = WI == 1 ? 0 : wave / WO;
    const unsigned smem_base = lds_addr(smem);
    const unsigned scratch = smem_base + NS * TILE_BYTES + wave * 1024;

    // per-lane offsets of the two transposed reads (see layout.h saved_off and the header of this file)
    int lane_off[2];
    {
        const int grp = lane >> 4, par = grp & 1, h = grp >> 1, il = lane & 15, q = il >> 2, p = il & 3;
#pragma unroll
        for (int r = 0; r < 2; ++r)
            lane_off[r] = par * 1024 + (2 * (8 * h + 4 * (r ^ par) + q) + (p & 1)) * 16 + (p >> 1) * 8;
    }

    // sample tile number i of the sequence, clamped at the end (a harmless re-read keeps the vmcnt arithmetic uniform)
    auto tile_at = [&](int i) { return seq.tile(i < cnt ? i : cnt - 1); };
    // stage sample tile t into `slot`
    auto issue = [&](int t, int slot, bool late = false) {
#ifdef KNERF_LIST_GUARD     // diagnostic build: a list entry outside the launch is counted and replaced instead of faulting
        if (t < 0 || t >= a.n_tiles) { if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(a.stats) + 3, 1ull); t = 0; }
#endif
        const char* tile_in = a.act + act_tile_off<S>((size_t)t) + lane * 16;
        const char* src_dz = a.dz + dz_tile_off<S>((size_t)t) + (size_t)dz_blk * kSavedBlockStride + lane * 16;
        const unsigned dst = smem_base + slot * TILE_BYTES;
#pragma unroll
        for (int r = 0; r < G_IN; ++r) {
            const int b = r * kWgWaves + wave;
            const bool ok = b < BLK_IN;
            const int blk = (ok ? KNERF_SRC_BLOCK(b) : 0) + ((ok && b >= S::kKs) ? act_blk2 : act_blk);     // block of the tile's act run (h: S::kKs blocks)
#ifdef KNERF_ABLATE_ENC_IO     // timing experiment only: the enc / dir blocks come from tile 0 (L2 hits) -- what wgrad would gain if it re-derived them for free
            const bool is_enc = (blk >= S::kActEnc && blk < S::kActEnc + S::kEncQ) || blk >= S::kActDir;
            const char* base = is_enc ? a.act + lane * 16 : tile_in;
            glds16(base + (size_t)blk * kSavedBlockStride, __builtin_amdgcn_readfirstlane(ok ? dst + b * 1024 : scratch), late);
#else
            glds16(tile_in + (size_t)blk * kSavedBlockStride, __builtin_amdgcn_readfirstlane(ok ? dst + b * 1024 : scratch), late);
#endif
        }
#pragma unroll
        for (int r = 0; r < G_DZ; ++r) {
            const int b = r * kWgWaves + wave;
            const bool ok = b < BLK_DZ;
            glds16(src_dz + (ok ? KNERF_SRC_BLOCK(b) : 0) * kSavedBlockStride, __builtin_amdgcn_readfirstlane(ok ? dst + (BLK_IN + b) * 1024 : scratch), late);
        }
    };
    f32x16 acc[NACC];
#pragma unroll
    for (int n = 0; n < NACC; ++n) acc[n] = zero_acc();
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;

    // prologue: tiles 0..NS-2
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(tile_at(s), s);
    int slot = 0;
#ifdef KNERF_WGRAD_STAMPS
    unsigned long long c_wait = 0, c_bar = 0, c_issue = 0, c_comp = 0;
#endif
    for (int i = 0; i < cnt; ++i) {
        STAMP(s0);
        // tile i landed (mine) ... everyone's (barrier); tile i-1 is free; t_next = the tile this iteration stages
        const int t_next = wait_barrier_next<G * (NS - 2), false>(seq, i + NS - 1 < cnt ? i + NS - 1 : cnt - 1);
        STAMP(s1);
        STAMP(s2);
        int nslot = slot + NS - 1; if (nslot >= NS) nslot -= NS;
#ifdef KNERF_WGRAD_EARLY_ISSUE     // A/B knob: the older order (copies issued right behind the barrier)
        issue(t_next, nslot);
#endif
        STAMP(s3);
        const char* in_reg = smem + slot * TILE_BYTES;
        const char* dz_reg = in_reg + BLK_IN * 1024;
#ifdef KNERF_WGRAD_ABLATE_COMPUTE     // timing experiment only: pure streaming
        if (cnt < 0)
#endif
        {
        // branch-free rows: a wave whose row index is the bias row (it == NI) swaps in the all-ones tile, rows past
        // it compute on a clamped (valid) tile and are dropped at the flush -- no control flow between the
        // transposed reads, so they issue back to back
        auto read_k = [&](int kk, bf16x8& b, bf16x8 (&afr)[NACC]) {
            b = tr_frag(dz_reg, wo, kk, lane_off);
#pragma unroll
            for (int n = 0; n < NACC; ++n) {
                const int it = wi + n * WI;
                afr[n] = tr_frag(in_reg, it < NI ? it : NI - 1, kk, lane_off);
                if (WI * NACC > NI && it >= NI) afr[n] = ones;
            }
        };
#ifndef KNERF_WGRAD_EARLY_ISSUE
        // Order inside a tile: both halves' fragments are requested first, the copies of tile i+NS-1 are issued behind the first
        // half's MFMAs (all eight waves queue their copies at once -- about 500 cycles of the CU's address path per 32 KiB tile --
        // and a wave stalls on its own: with products already queued the matrix pipe works through that stall), then the rest.
        constexpr bool kBoth = NACC <= 9;      // register budget: 16 NACC accumulators + 2 x 4 (NACC + 1) operands
        bf16x8 b0, b1, afr0[NACC], afr1[NACC];
        read_k(0, b0, afr0);
#ifdef KNERF_ABLATE_HALF_WGRAD_MATH     // timing experiment only: half the transposed reads and MFMAs per tile, as an 8-bit
        (void)b1; (void)afr1; (void)kBoth;       // operand format would need (v_mfma_f32_32x32x64_f8f6f4 contracts 64 samples in the time of two bf16 k-steps)
#pragma unroll
        for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr0[n], b0, acc[n], 0, 0, 0);
        issue(t_next, nslot, true);
#else
        if (kBoth) read_k(1, b1, afr1);
#pragma unroll
        for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr0[n], b0, acc[n], 0, 0, 0);
        issue(t_next, nslot, true);
        if (!kBoth) read_k(1, b1, afr1);
#pragma unroll
        for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr1[n], b1, acc[n], 0, 0, 0);
#endif
#else
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 b, afr[NACC];
            read_k(kk, b, afr);
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[n], b, acc[n], 0, 0, 0);
        }
#endif
        }
        slot = slot + 1 == NS ? 0 : slot + 1;
#ifdef KNERF_WGRAD_STAMPS
        STAMP(s4);
        c_wait += s1 - s0; c_bar += s2 - s1; c_issue += s3 - s2; c_comp += s4 - s3;
#endif
    }
#ifdef KNERF_WGRAD_STAMPS
    const unsigned long long t_loop_end = stamp();
    if (threadIdx.x == 0 && blockIdx.x < 1024) {
        unsigned long long* o = g_wgrad_stamps + blockIdx.x * 8;
        o[0] = c_wait; o[1] = c_bar; o[2] = c_issue; o[3] = c_comp; o[4] = (unsigned long long)cnt; o[5] = job;
        o[6] = t_loop_end - (c_wait + c_bar + c_issue + c_comp) - o[6];     // entry (stored by wgrad_kernel) -> first iteration
        o[7] = t_loop_end;                                                   // wgrad_kernel turns this into loop end -> exit
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();      // nobody may restage LDS for the next job while a wave still reads this one

    // flush: acc[n] reg i of lane (c, hh) is dW[row 32*it + (i&3) + 8(i>>2) + 4hh][col 32*wo + c]
    const int* dst = a.dst + a.job_off[job];
    constexpr int NCOLS = NO * 32;
    const int c = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int n = 0; n < NACC; ++n) {
        const int it = wi + n * WI;
        if (it < NI) {
            flush_acc(a, dst, 32 * it + 4 * hh, NCOLS, 32 * wo + c, acc[n], wgrad_partial_stride<S>());
        } else if (it == NI) {
            flush_bias(a, dst, NI * 32 * NCOLS + 32 * wo + c, hh, acc[n][0], wgrad_partial_stride<S>());   // bias row: every row of the ones-tile holds the column sums
        }
    }
}

// layer_1 with h0 RECOMPUTED.  h0 = relu(W_0 enc + b_0) has only 64 input slots, so the forward does not save it (16 KiB per
// tile less to write there and to read here): this job stages enc (4 blocks) + dz1 (16 blocks) per tile, every wave recomputes
// ONE 32-feature tile of h0 for the tile's 32 samples with 4 MFMAs in the orientation D[sample][feature] = enc . W_0 -- whose
// result already has lane = feature, registers = samples, i.e. it IS an A operand of the weight-gradient product, with the
// sample order sigma(hh, j) = (j&3) + 8(j>>2) + 4hh (+16 kk) -- and publishes its two fragments in LDS; after a barrier all
// waves read the 16 fragments and accumulate dW_1 = h0^T dz1 as before.  The dz1 fragments are read with the SAME sample
// permutation (only the per-lane offsets of the transposed reads change; the bank pattern stays conflict-free: 32 q + 128
// (h ^ par) covers the 64 banks in two passes).  W_0's fragments are the forward stream's own layer_0 blocks: an A fragment of
// W^T and a B fragment of W hold the same registers (layout.h: the 32x32x16 operand maps are symmetric).
template <class S, class Seq>
__device__ __forceinline__ void wgrad_l1_recompute(const WgradArgs& a, const Seq seq, char* smem) {
    constexpr int NI = 8, BLK_IN = 4, BLK_DZ = 16;
    constexpr int TILE_BYTES = (BLK_IN + BLK_DZ) * 1024;
    constexpr int G_IN = 1, G_DZ = 2, G = G_IN + G_DZ;
    constexpr int NS = 6;
    constexpr int kXch = 2 * 16 * 1024;                      // h0 fragments of two tiles (the one being consumed, the next one): [parity][it][kk][1 KiB]
    static_assert(NS * TILE_BYTES + kXch + kWgScratch <= 160 * 1024, "LDS budget");
    constexpr int job = 1, NACC = NI + 1, NCOLS = 256;

    const int cnt = seq.count();
    if (cnt <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wo = wave;                                     // output strip (32 columns of dW_1) AND the h0 tile this wave recomputes
    const unsigned smem_base = lds_addr(smem);
    char* xch = smem + NS * TILE_BYTES;
    const unsigned scratch = smem_base + NS * TILE_BYTES + kXch + wave * 1024;

    // W_0 fragments of h0 tile `wave` (forward stream blocks 4*wave .. 4*wave+3) and its bias column
    bf16x8 w0[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) w0[ks] = *reinterpret_cast<const bf16x8*>(a.fwd_stream + (size_t)(4 * wave + ks) * 1024 + lane * 16);
    const float b0 = a.bias[wave * 32 + (lane & 31)];

    // transposed reads of dz1 in the permuted sample order: samples 4 (h ^ par) + 8 r + q of the 16-sample group kk
    int lane_off[2];
    int enc_off;                                             // this lane's 16 bytes of an (even) saved block; odd blocks: ^ 128
    {
        const int grp = lane >> 4, par = grp & 1, h = grp >> 1, il = lane & 15, q = il >> 2, p = il & 3;
#pragma unroll
        for (int r = 0; r < 2; ++r)
            lane_off[r] = par * 1024 + (2 * (4 * (h ^ par) + 8 * r + q) + (p & 1)) * 16 + (p >> 1) * 8;
        enc_off = (2 * (lane & 31) + (lane >> 5)) * 16;      // saved_off(b even, h, s); odd b: s ^ 4  <=>  byte offset ^ 128
    }
    auto tile_at = [&](int i) { return seq.tile(i < cnt ? i : cnt - 1); };
    auto issue = [&](int t, int slot, bool late = false) {
#ifdef KNERF_LIST_GUARD
        if (t < 0 || t >= a.n_tiles) { if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(a.stats) + 3, 1ull); t = 0; }
#endif
#ifdef KNERF_ABLATE_ENC_IO
        const char* src_in = a.act + (size_t)S::kActEnc * kSavedBlockStride + lane * 16;
#else
        const char* src_in = a.act + act_tile_off<S>((size_t)t) + (size_t)S::kActEnc * kSavedBlockStride + lane * 16;
#endif
        const char* src_dz = a.dz + dz_tile_off<S>((size_t)t) + (size_t)16 * kSavedBlockStride + lane * 16;
        const unsigned dst = smem_base + slot * TILE_BYTES;
        const bool ok = wave < BLK_IN;
        glds16(src_in + (ok ? KNERF_SRC_BLOCK(wave) : 0) * kSavedBlockStride, __builtin_amdgcn_readfirstlane(ok ? dst + wave * 1024 : scratch), late);
#pragma unroll
        for (int r = 0; r < G_DZ; ++r) {
            const int b = r * kWgWaves + wave;
            glds16(src_dz + KNERF_SRC_BLOCK(b) * kSavedBlockStride, __builtin_amdgcn_readfirstlane(dst + (BLK_IN + b) * 1024), late);
        }
    };
    f32x16 acc[NACC];
#pragma unroll
    for (int n = 0; n < NACC; ++n) acc[n] = zero_acc();
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
    static_assert((S::kActEnc & 1) == 0, "enc block parity");

    // my tile of h0 for the sample tile staged in `reg`: acc = b_0, += enc(ks) . W_0(ks), relu, bf16 (the forward's own order of
    // operations, mlp_fwd.hip), published as the two A-operand fragments of h0 tile `wave`
    auto recompute = [&](const char* reg, int parity) {
        f32x16 h;
#pragma unroll
        for (int r = 0; r < 16; ++r) h[r] = b0;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 e = *reinterpret_cast<const bf16x8*>(reg + ks * 1024 + ((KNERF_WGRAD_ABLATE_LDS & 2) ? lane * 16 : (enc_off ^ ((ks & 1) << 7))));
            h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(e, w0[ks], h, 0, 0, 0);
        }
        bf16x8 lo, hi;
        pack_acc(h, lo, hi);
        char* x = xch + parity * (kXch / 2) + wave * 2048 + lane * 16;
        *reinterpret_cast<bf16x8*>(x) = relu_packed(lo);
        *reinterpret_cast<bf16x8*>(x + 1024) = relu_packed(hi);
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(tile_at(s), s);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G * (NS - 2)) : "memory");       // tile 0 landed
    __builtin_amdgcn_s_barrier();
    recompute(smem, 0);
    int slot = 0;
    // One barrier per tile: iteration i consumes tile i (its h0 fragments were written during iteration i-1) and recomputes
    // h0 of tile i+1 into the other half of the exchange buffer, so tile i+1 must have landed too (one tile less in flight).
    for (int i = 0; i < cnt; ++i) {
        // tiles <= i+1 landed (mine) and my h0 fragments of tile i are written (lgkmcnt: the ds_writes of the recompute) ...
        // ... everyone's (barrier); tile i-1 and h0(i-1) are free
        const int t_next = wait_barrier_next<G * (NS - 3), true>(seq, i + NS - 1 < cnt ? i + NS - 1 : cnt - 1);
        int nslot = slot + NS - 1; if (nslot >= NS) nslot -= NS;
        const char* dz_reg = smem + slot * TILE_BYTES + BLK_IN * 1024;
        const char* xr = xch + (int)(i & 1) * (kXch / 2);
#pragma unroll
        for (int kk = 0; kk < kWgKSteps; ++kk) {
            const bf16x8 b = tr_frag(dz_reg, wo, kk, lane_off);
            bf16x8 afr[NACC];
#pragma unroll
            for (int n = 0; n < NI; ++n) afr[n] = *reinterpret_cast<const bf16x8*>(xr + (n * 2 + kk) * 1024 + lane * 16);
            afr[NI] = ones;
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[n], b, acc[n], 0, 0, 0);
            if (kk == 0) issue(t_next, nslot, true);          // behind the first half's MFMAs (see wgrad_job_body)
        }
        slot = slot + 1 == NS ? 0 : slot + 1;
        recompute(smem + slot * TILE_BYTES, (int)((i + 1) & 1));               // past the end: the clamped re-read of the last tile, unused
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const int* dst = a.dst + a.job_off[job];
    const int c = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int n = 0; n < NACC; ++n) {
        if (n < NI) {
            flush_acc(a, dst, 32 * n + 4 * hh, NCOLS, 32 * wo + c, acc[n], wgrad_partial_stride<S>());
        } else {
            flush_bias(a, dst, NI * 32 * NCOLS + 32 * wo + c, hh, acc[n][0], wgrad_partial_stride<S>());
        }
    }
}

// layer_7 with dz7 RECOMPUTED.  dz7 = mask7 * (H dz_head) has only 4 input channels (the collapsed head, layout.h), so the dgrad
// kernel does not write it: this job stages h6 (16 blocks), the dz_head block and the layer-7 mask block per tile (18 KiB); every
// wave recomputes the 32 columns of dz7 that its output strip needs with ONE MFMA in the orientation D[sample][feature] =
// dz_head . H^T (lane = feature, registers = samples: a B operand of the weight-gradient product as it stands, in the
// accumulator's sample order), clears the entries whose forward activation was not positive (one 32-bit LDS read of the mask
// block per accumulator register) and rounds to bf16 exactly as the dgrad chain does.  The h6 fragments are read with the same
// sample permutation as in wgrad_l1_recompute.  H's fragment is the dgrad stream's own block `wo` (stage B0).
// The recompute of tile i+1 runs inside tile i's MFMA sequence (its VALU selects and conversions issue between MFMAs), so
// the strip is a register operand by the time the tile's own products start.
template <class S, class Seq>
__device__ __forceinline__ void wgrad_last_recompute(const WgradArgs& a, const Seq seq, char* smem) {
    constexpr int NI = 8, BLK_IN = 16, BLK_X = 2;           // x: [0] dz_head block, [1] mask7 block
    constexpr int TILE_BYTES = (BLK_IN + BLK_X) * 1024;
    constexpr int G = 3;                                     // 2 h6 copies + 1 (dz_head / mask / padding) per wave and tile
    constexpr int NS = 8;                                    // 18 KiB tiles: eight slots, six tiles in flight behind the two in use
    static_assert(NS * TILE_BYTES + kWgScratch <= 160 * 1024, "LDS budget");
    constexpr int job = S::NL - 1, NACC = NI + 1, NCOLS = 256;      // the last trunk layer

    const int cnt = seq.count();
    if (cnt <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wo = wave;
    const unsigned smem_base = lds_addr(smem);
    const unsigned scratch = smem_base + NS * TILE_BYTES + wave * 1024;

    const bf16x8 hfrag = *reinterpret_cast<const bf16x8*>(a.bwd_stream + (size_t)wo * 1024 + lane * 16);   // H[feature 32 wo + (lane&31)][channel 8h+j]
    int lane_off[2];
    int zoff;                   // this lane's 16 bytes of the (even) dz_head block: lane = (sample, half)
    int moff, mbit;             // mask word of (feature column of this lane, sample 0) and the bit inside it
    {
        const int grp = lane >> 4, par = grp & 1, h = grp >> 1, il = lane & 15, q = il >> 2, p = il & 3;
#pragma unroll
        for (int r = 0; r < 2; ++r)
            lane_off[r] = par * 1024 + (2 * (4 * (h ^ par) + 8 * r + q) + (p & 1)) * 16 + (p >> 1) * 8;
        zoff = (2 * (lane & 31) + (lane >> 5)) * 16;
        // forward (mlp_fwd.hip relu_epi): the lane of sample s and half hf = (c >> 2) & 1 holds feature 32 ot + c of out tile ot in
        // accumulator register i = (c & 3) + 4 (c >> 3); its mask bit sits in word ot >> 1 at (ot & 1) * 8 + (i >> 1) + 16 (i & 1)
        const int c = lane & 31, hf = (c >> 2) & 1, i = (c & 3) + 4 * (c >> 3);
        moff = mask_word_off(0, hf, wo >> 1);             // + the sample row's stride below
        mbit = (wo & 1) * 8 + (i >> 1) + 16 * (i & 1);
    }
    static_assert((S::kDzHead & 1) == 0, "dz_head block parity");
    auto tile_at = [&](int i) { return seq.tile(i < cnt ? i : cnt - 1); };
    auto issue = [&](int t, int slot, bool late = false) {
#ifdef KNERF_LIST_GUARD
        if (t < 0 || t >= a.n_tiles) { if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(a.stats) + 3, 1ull); t = 0; }
#endif
        const char* src_in = a.act + act_tile_off<S>((size_t)t) + (size_t)S::act_h(S::NL - 2) * kSavedBlockStride + lane * 16;
        const unsigned dst = smem_base + slot * TILE_BYTES;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int b = r * kWgWaves + wave;
            glds16(src_in + KNERF_SRC_BLOCK(b) * kSavedBlockStride, __builtin_amdgcn_readfirstlane(dst + b * 1024), late);
        }
        const char* src_x = wave == 1 ? a.mask + mask_tile_off<S>((size_t)t) + (S::NL - 1) * kSavedBlockStride + lane * 16
                                      : a.dz + dz_tile_off<S>((size_t)t) + (size_t)S::kDzHead * kSavedBlockStride + lane * 16;
        glds16(src_x, __builtin_amdgcn_readfirstlane(wave < 2 ? dst + (BLK_IN + wave) * 1024 : scratch), late);
    };
    const int hh = lane >> 5;
    // my strip of dz7 for one tile's 32 samples, in two steps so that the loop can put the second one behind MFMAs of the
    // current tile: (1) the MFMA and the mask words, (2) select + round
    auto dz7_mfma = [&](const char* x_reg, f32x16& dzf, unsigned (&mw)[16]) {
        const bf16x8 z = *reinterpret_cast<const bf16x8*>(x_reg + ((KNERF_WGRAD_ABLATE_LDS & 2) ? lane * 16 : zoff));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int smp = (KNERF_WGRAD_ABLATE_LDS & 1) ? 0 : (r & 3) + 8 * (r >> 2) + 4 * hh;             // sample row of accumulator register r
            mw[r] = *reinterpret_cast<const unsigned*>(x_reg + 1024 + smp * (mask_word_off(1, 0, 0) - mask_word_off(0, 0, 0)) + ((KNERF_WGRAD_ABLATE_LDS & 1) ? 0 : moff));
        }
        dzf = __builtin_amdgcn_mfma_f32_32x32x16_bf16(z, hfrag, zero_acc(), 0, 0, 0);
    };
    auto dz7_pack = [&](f32x16& dzf, const unsigned (&mw)[16], bf16x8 (&bfr)[2]) {
#pragma unroll
        for (int r = 0; r < 16; ++r) dzf[r] = ((mw[r] >> mbit) & 1u) ? dzf[r] : 0.f;
        pack_acc(dzf, bfr[0], bfr[1]);
    };
    f32x16 acc[NACC];
#pragma unroll
    for (int n = 0; n < NACC; ++n) acc[n] = zero_acc();
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
    asm volatile("" ::"v"(hfrag));   // the wait for this global load is pinned here: inside the loop it would be a vmcnt(0) that drains the LDS-DMA pipeline every tile

#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(tile_at(s), s);
    bf16x8 bnext[2];
    {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G * (NS - 2)) : "memory");   // tile 0 landed
        __builtin_amdgcn_s_barrier();
        f32x16 dzf; unsigned mw[16];
        dz7_mfma(smem + BLK_IN * 1024, dzf, mw);
        dz7_pack(dzf, mw, bnext);
    }
    int slot = 0;
    for (int i = 0; i < cnt; ++i) {
        // tile i+1 landed (mine) ... everyone's (barrier); tile i-1 is free
        const int t_next = wait_barrier_next<G * (NS - 3), false>(seq, i + NS - 1 < cnt ? i + NS - 1 : cnt - 1);
        int nslot = slot + NS - 1; if (nslot >= NS) nslot -= NS;
        const char* in_reg = smem + slot * TILE_BYTES;
        slot = slot + 1 == NS ? 0 : slot + 1;
        bf16x8 bfr[2] = {bnext[0], bnext[1]};
        f32x16 dzf; unsigned mw[16];
        dz7_mfma(smem + slot * TILE_BYTES + BLK_IN * 1024, dzf, mw);           // tile i+1 (past the end: the clamped re-read, unused)
#pragma unroll
        for (int kk = 0; kk < kWgKSteps; ++kk) {
            bf16x8 afr[NACC];
#pragma unroll
            for (int n = 0; n < NI; ++n) afr[n] = tr_frag(in_reg, n, kk, lane_off);
            afr[NI] = ones;
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[n], bfr[kk], acc[n], 0, 0, 0);
            if (kk == 0) {
                issue(t_next, nslot, true);                    // behind the first half's MFMAs (see wgrad_job_body)
                dz7_pack(dzf, mw, bnext);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const int* dst = a.dst + a.job_off[job];
    const int c = lane & 31;
#pragma unroll
    for (int n = 0; n < NACC; ++n) {
        if (n < NI) {
            flush_acc(a, dst, 32 * n + 4 * hh, NCOLS, 32 * wo + c, acc[n], wgrad_partial_stride<S>());
        } else {
            flush_bias(a, dst, NI * 32 * NCOLS + 32 * wo + c, hh, acc[n][0], wgrad_partial_stride<S>());
        }
    }
}

// one job of the plan over the given tile range: job j = trunk layer j, job NL = the head; the body follows from the job's kind
// (layout.h wgrad_job): every branch below is resolved at compile time but the uniform `job == j`
template <class S, class Seq>
__device__ __forceinline__ void wgrad_dispatch(const WgradArgs& a, int job, const Seq& seq, char* smem) {
    static_for<S::kWgradJobs>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        constexpr WgradJob J = wgrad_job<S>(j);
        if (job == j) {
            if constexpr (J.kind == 1) wgrad_l1_recompute<S>(a, seq, smem);                                    // h0 recomputed from enc (width 256)
            else if constexpr (J.kind == 4) wgrad_last_recompute<S>(a, seq, smem);                             // dz recomputed from dz_head and the mask (width 256)
            // first layer <2, T>, plain <T, T>, concat <T + 2, T>, head <T + 1, 1>  (T = U / 32 tiles: <2,8> <8,8> <10,8> <9,1> at width 256)
            else wgrad_job_body<S, J.n_it, J.n_ot>(a, j, J.act_blk, J.act_blk2, J.dz_blk, seq, smem);
        }
    });
}

}  // namespace knerf
