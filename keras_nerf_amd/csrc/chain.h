// chain.h -- device machinery shared by the fused forward and dgrad MLP chain kernels (gfx950 only).
//
// One workgroup = 8 waves = 256 samples; each wave owns 32 samples (one per lane&31; the two lane halves hold
// different feature rows of the same sample).  Activations never leave registers between layers: the 32x32 f32
// MFMA result, converted to bf16, is the next layer's B operand (layout.h).  Weights arrive as a linear stream of
// 1 KiB A-fragment blocks pulled by LDS-DMA (global_load_lds_dwordx4) through a ring of 16 KiB pages.
//
// Stagger: the two waves that share a SIMD (wave w and w+4) run the same program; in lockstep their epilogues and
// barrier waits coincide and the matrix pipe idles.  Waves 0-3 ("group A") therefore take the ring's barrier in the
// middle of a page (block 16k+8) and waves 4-7 ("group B") at the start of the next one (block 16k+16): B runs half a
// page ahead, so one wave's epilogue overlaps its SIMD partner's MFMAs.  Barrier k guarantees pages <= k+2 landed
// (B prefetches into page k+2 before barrier k+1) and recycles the slot of page k-1 for page k+kSlots-1.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>

#include "layout.h"

namespace knerf {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;

constexpr int kWaves = 8;                 // waves per workgroup
constexpr int kThreads = kWaves * 64;
constexpr int kTile = 32;                 // samples per wave
#ifndef KNERF_PAGE_BLOCKS
#define KNERF_PAGE_BLOCKS 16
#endif
constexpr int kPageBlocks = KNERF_PAGE_BLOCKS;   // 1 KiB blocks per ring page
constexpr int kPageBytes = kPageBlocks * 1024;
#ifndef KNERF_SLOTS
#define KNERF_SLOTS 6
#endif
#ifndef KNERF_PREFETCH
#define KNERF_PREFETCH 4
#endif
constexpr int kSlots = KNERF_SLOTS;       // ring slots of 16 KiB
constexpr int kRingBytes = kSlots * kPageBytes;
constexpr int kGldsPerPage = kPageBytes / (kThreads * 16);   // 2 LDS-DMA instructions per thread per page
constexpr int kWaitInFlight = kGldsPerPage * (kSlots - 4);   // LDS-DMA ops that may still fly at a barrier: pages k+3..k+kSlots-2
constexpr int kTailPages = kSlots;        // dummy pages appended to every stream so that issue never needs a guard

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N-1>{})
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Static schedule of the global stores a chain kernel issues between its MFMA blocks (they share the in-order vmcnt
// queue with the ring's LDS-DMA).  A stage writes `per_tile` stores in the epilogue of every out tile (`skip_last`: not
// for its last tile) and `at_end` more after its last tile; `initial` stores precede block 0.
struct StoreStage { int b0, nks, n_ot, per_tile, at_end, skip_last; };
template <int NST>
struct StoreSched {
    StoreStage st[NST];
    int initial;
    // stores issued (program order) before the MFMA of block b
    constexpr int before(int b) const {
        if (b < 0) return 0;
        int n = initial;
        for (int i = 0; i < NST; ++i) {
            const StoreStage& s = st[i];
            for (int ot = 0; ot < s.n_ot; ++ot) {
                const int last = s.b0 + ot * s.nks + s.nks - 1;     // the epilogue follows this block
                if (last < b) {
                    if (!(s.skip_last && ot == s.n_ot - 1)) n += s.per_tile;
                    if (ot == s.n_ot - 1) n += s.at_end;
                }
            }
        }
        return n;
    }
    // vmcnt immediate of the barrier a wave takes in front of block b (b % 8 == 0): the page that must have landed was
    // issued right after this wave's barrier kSlots-3 pages earlier (or in the prologue); everything issued since then
    // may still be in flight
    constexpr int wait_at(int b) const {
        const int prev = b - (kSlots - 3) * kPageBlocks;
        const int n = kWaitInFlight + before(b) - (prev < 0 ? 0 : before(prev));
        return n > 60 ? 60 : n;
    }
};
template <int NBLK>
struct WaitTable { int n[NBLK / 8 + 3]; };        // indexed by b / 8
template <int NST, int NBLK>
constexpr WaitTable<NBLK> make_wait_table(const StoreSched<NST>& s) {
    WaitTable<NBLK> t{};
    for (int i = 0; i < NBLK / 8 + 3; ++i) t.n[i] = s.wait_at(i * 8);
    return t;
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

struct Ring {
    const char* stream;    // global: packed bf16 A-fragments, page after page
    char* lds;             // LDS base of the ring (16-byte aligned, offset 0 of the dynamic segment)
    int tid;
    int wave;              // scalar (readfirstlane) wave index

    // every thread moves 2 x 16 B of page `page` into slot page % kSlots.  Uniform base + 32-bit lane offset keeps the
    // address arithmetic on the scalar unit.
    __device__ __forceinline__ void issue(int page) const {
        char* dst = lds + (page % kSlots) * kPageBytes + wave * 1024;         // wave-uniform base; HW adds lane*16
#pragma unroll
        for (int i = 0; i < kGldsPerPage; ++i) {
            const char* base = stream + (size_t)page * kPageBytes + i * kThreads * 16;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + (unsigned)(tid * 16)), (lds_ptr_t)(dst + i * kThreads * 16), 16, 0, 0);
        }
    }
    __device__ __forceinline__ void prologue_issue() const {
#pragma unroll
        for (int p = 0; p < kSlots - 1; ++p) issue(p);
    }
    __device__ __forceinline__ void prologue_wait() const {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kGldsPerPage * (kSlots - 3)) : "memory");   // pages 0 and 1 landed (mine)
        __builtin_amdgcn_s_barrier();                                                          // ... and everyone's
    }
    // barrier k (group A: in front of block 16k+8, group B: in front of block 16k+16): pages <= k+2 readable,
    // the slot of page k-1 is recycled for page k+kSlots-1
    template <int WAIT>
    __device__ __forceinline__ void sync(int k) const {
        wait_vmcnt<WAIT>();
        __builtin_amdgcn_s_barrier();
        issue(k + kSlots - 1);
    }
    __device__ __forceinline__ void drain() const { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

    __device__ __forceinline__ bf16x8 frag(int b, int lane) const {
        const char* p = lds + ((b / kPageBlocks) % kSlots) * kPageBytes + (b % kPageBlocks) * 1024 + lane * 16;
        return *reinterpret_cast<const bf16x8*>(p);
    }
#ifndef KNERF_COMPILER_FRAGS
    // The same read as an instruction hipcc does not track (the default; -DKNERF_COMPILER_FRAGS restores the plain load): with
    // the compiler's own bookkeeping every use of a prefetched fragment waits for lgkmcnt(0), i.e. also for the kPrefetch-1
    // younger reads behind it -- counted waits: inference chain -4 %, training forward / dgrad -3 % (r02 A/B, two runs each).
    // B is a compile-time block index; the 16-bit offset field covers three slots, `hi` selects the upper three.
    template <int B>
    __device__ __forceinline__ bf16x8 frag_asm(unsigned lane_lo, unsigned lane_hi) const {
        constexpr int slot = (B / kPageBlocks) % kSlots;
        constexpr int off = (slot % 3) * kPageBytes + (B % kPageBlocks) * 1024;
        static_assert(kSlots <= 6 && off < 65536, "ds offset field");
        bf16x8 v;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(slot < 3 ? lane_lo : lane_hi), "n"(off));
        return v;
    }
#endif
};
#ifndef KNERF_COMPILER_FRAGS
// the fragment becomes usable once at most N younger LDS operations are outstanding (LDS returns in order)
template <int N>
__device__ __forceinline__ void frag_wait(bf16x8& v) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N)); }
#endif

// bias tile -> accumulator init.  LDS holds fp32 [tile][32]; reg i of lane-half h is row (i&3) + 8*(i>>2) + 4h.
__device__ __forceinline__ f32x16 bias_acc(const float* bias_lds, int tile, int h) {
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 v = *reinterpret_cast<const f32x4*>(bias_lds + tile * 32 + 8 * g + 4 * h);
        acc[4 * g + 0] = v[0]; acc[4 * g + 1] = v[1]; acc[4 * g + 2] = v[2]; acc[4 * g + 3] = v[3];
    }
    return acc;
}

__device__ __forceinline__ f32x16 zero_acc() {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    return acc;
}

// f32x16 accumulator -> two bf16x8 B-operand k-steps (regs 0..7 -> k-step 0, regs 8..15 -> k-step 1); explicit pair
// conversions so that each dword is ONE v_cvt_pk_bf16_f32
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ void pack_acc(const f32x16& a, bf16x8& lo, bf16x8& hi) {
    u32x4 l, h;
#pragma unroll
    for (int k = 0; k < 4; ++k) { l[k] = cvt_pk_bf16(a[2 * k], a[2 * k + 1]); h[k] = cvt_pk_bf16(a[8 + 2 * k], a[8 + 2 * k + 1]); }
    lo = __builtin_bit_cast(bf16x8, l); hi = __builtin_bit_cast(bf16x8, h);
}

// ReLU on PACKED bf16: as int16 a negative float is a negative integer, so one v_pk_max_i16 per pair does it
// (relu(round(x)) == round(relu(x)): rounding keeps the sign).
__device__ __forceinline__ bf16x8 relu_packed(const bf16x8& v) {
    typedef __attribute__((ext_vector_type(2))) short s16x2;
    u32x4 x = __builtin_bit_cast(u32x4, v);
    const s16x2 z = {0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned xk = x[k];                       // scalar copy: see the bit_cast note below
        x[k] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, xk), z));   // v_pk_max_i16
    }
    return __builtin_bit_cast(bf16x8, x);
}

// ReLU mask of one out tile from its two packed (already ReLU'd) k-steps: flag = (half != 0) via v_pk_min_u16(x, 1),
// packed pair k (k = 0..7: lo dwords then hi dwords) -> bit k (even element) and bit 16+k (odd element).
__device__ __forceinline__ unsigned relu_mask_bits(const bf16x8& lo, const bf16x8& hi) {
    const u32x4 a = __builtin_bit_cast(u32x4, lo), b = __builtin_bit_cast(u32x4, hi);
    unsigned f[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(f[k]) : "v"(a[k]));
        asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(f[4 + k]) : "v"(b[k]));
    }
    unsigned w = f[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) w |= f[k] << k;       // v_lshl_or_b32
    return w;
}

// dgrad side: zero the halves of packed dZ whose forward activation was not positive.  w = relu_mask_bits of that tile.
__device__ __forceinline__ void apply_mask_packed(bf16x8& lo, bf16x8& hi, unsigned w) {
    u32x4 a = __builtin_bit_cast(u32x4, lo), b = __builtin_bit_cast(u32x4, hi);
    typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned fa = (w >> k) & 0x00010001u, fb = (w >> (4 + k)) & 0x00010001u;      // 0/1 per half
        // NB scalar copies first: __builtin_bit_cast applied to a vector ELEMENT reads element 0 (hipcc 7.2)
        const unsigned ak = a[k], bk = b[k];
        a[k] = __builtin_bit_cast(unsigned, (u16x2)(__builtin_bit_cast(u16x2, ak) * __builtin_bit_cast(u16x2, fa)));
        b[k] = __builtin_bit_cast(unsigned, (u16x2)(__builtin_bit_cast(u16x2, bk) * __builtin_bit_cast(u16x2, fb)));
    }
    lo = __builtin_bit_cast(bf16x8, a); hi = __builtin_bit_cast(bf16x8, b);
}

// A-fragment prefetch ring: block b's fragment is read from LDS kPrefetch MFMAs before it is used.  kPrefetch <= 8
// keeps every read inside the half page that the last sync made readable (see Ring::sync).
constexpr int kPrefetch = KNERF_PREFETCH;
struct Prefetch {
    bf16x8 a[kPrefetch];
    unsigned lane_lo, lane_hi;          // LDS byte address of this lane's 16 bytes in slot 0 / slot 3
    template <int NBLOCKS>
    __device__ __forceinline__ void start(const Ring& ring, int lane) {
#ifndef KNERF_COMPILER_FRAGS
        lane_lo = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring.lds + lane * 16;
        lane_hi = lane_lo + 3 * kPageBytes;
        static_for<kPrefetch>([&](auto i_) { constexpr int i = decltype(i_)::value; a[i] = ring.template frag_asm<(i < NBLOCKS ? i : 0)>(lane_lo, lane_hi); });
#else
#pragma unroll
        for (int i = 0; i < kPrefetch; ++i) a[i] = ring.frag(i < NBLOCKS ? i : 0, lane);
#endif
    }
};

// number of ring barriers each group takes over a stream of NBLOCKS blocks (A: b = 8, 24, ...; B: b = 16, 32, ...)
constexpr int barriers_a(int nblocks) { return nblocks > kPageBlocks / 2 ? (nblocks - kPageBlocks / 2 - 1) / kPageBlocks + 1 : 0; }
constexpr int barriers_b(int nblocks) { return nblocks > kPageBlocks ? (nblocks - kPageBlocks - 1) / kPageBlocks + 1 : 0; }
static_assert(kPrefetch <= kPageBlocks / 2 && kSlots >= 4 && kPageBlocks % 16 == 0, "ring geometry");
// after its last block group B takes the barriers group A still has (the counts differ by at most one)
template <int NBLOCKS>
__device__ __forceinline__ void ring_finish(const Ring& ring, int grp) {
    if (grp == 1) {
#pragma unroll
        for (int i = barriers_b(NBLOCKS); i < barriers_a(NBLOCKS); ++i) __builtin_amdgcn_s_barrier();
    }
    ring.drain();
}

// One dense stage: for every out tile, acc = init(ot); acc += A(block) x in(ks) over the stage's k-steps; epi(ot, acc).
// B0 = index of the stage's first block in the stream, NBLOCKS the stream length; every index is a compile-time
// constant (static_for).  W::tab.n[b/8] = vmcnt immediate of a barrier in front of block b (WaitTable).
// grp = 0 for waves 0-3 (barrier at b % 16 == 8), 1 for waves 4-7 (barrier at b % 16 == 0); scalar.
template <int B0, int NKS, int NOT, int NBLOCKS, class W, class Init, class In, class Epi>
__device__ __forceinline__ void dense_stage(const Ring& ring, Prefetch& pf, int lane, int grp, W, Init&& init, In&& in, Epi&& epi) {
    static_for<NOT>([&](auto ot_) {
        constexpr int ot = decltype(ot_)::value;
        f32x16 acc = init(ot);
        static_for<NKS>([&](auto ks_) {
            constexpr int ks = decltype(ks_)::value;
            constexpr int b = B0 + ot * NKS + ks;
            if constexpr (b % kPageBlocks == kPageBlocks / 2) {
                if (grp == 0) ring.template sync<W::tab.n[b / 8]>(b / kPageBlocks);
            } else if constexpr (b % kPageBlocks == 0 && b > 0) {
                if (grp == 1) ring.template sync<W::tab.n[b / 8]>(b / kPageBlocks - 1);
            }
            bf16x8 cur = pf.a[b % kPrefetch];
#ifndef KNERF_COMPILER_FRAGS
            {   // younger fragment reads behind block b's: kPrefetch-1, fewer at the end of the stream
                constexpr int younger = NBLOCKS - 1 - b < kPrefetch - 1 ? NBLOCKS - 1 - b : kPrefetch - 1;
                frag_wait<younger>(cur);
            }
            if constexpr (b + kPrefetch < NBLOCKS) pf.a[b % kPrefetch] = ring.template frag_asm<b + kPrefetch>(pf.lane_lo, pf.lane_hi);
#else
            if constexpr (b + kPrefetch < NBLOCKS) pf.a[b % kPrefetch] = ring.frag(b + kPrefetch, lane);
#endif
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur, in(ks), acc, 0, 0, 0);
        });
        epi(ot, acc);
    });
}

// 16-byte streaming store.  The saved activations / dZ are a pure output stream of several GB per launch; as plain
// stores they evict the 1.2 MB weight stream that every workgroup re-reads from L2, and those re-reads then compete
// with the stores for HBM (measured: forward 1.22 ms plain, 0.88 ms sc1 or nt; dgrad 1.02 ms sc1, 0.81 ms nt -- nt is
// the default).  Inline asm: hipcc does not count it in vmcnt (the StoreSched table does) and the trailing s_nop keeps
// the data registers intact until they have been read.
__device__ __forceinline__ void store16_wt(char* base, unsigned off, const u32x4& v) {
#ifdef KNERF_ABLATE_STORES      // timing experiment only: keeps the value alive, skips the store
    asm volatile("" ::"v"(v));
#elif defined(KNERF_ABLATE_STORE_EXEC0)   // timing experiment only (r04): the store is ISSUED (same vmcnt accounting, same address VALU) with no lane active -- no data leaves the CU
    unsigned long long keep;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 0\n\tglobal_store_dwordx4 %1, %2, %3 nt\n\ts_mov_b64 exec, %0\n\ts_nop 1" : "=&s"(keep) : "v"(off), "v"(v), "s"(base) : "memory");
#elif defined(KNERF_PLAIN_STORES)
    *reinterpret_cast<u32x4*>(base + off) = v;
#else
#ifndef KNERF_STORE_POLICY
#define KNERF_STORE_POLICY "nt"
#endif
    asm volatile("global_store_dwordx4 %0, %1, %2 " KNERF_STORE_POLICY "\n\ts_nop 1" ::"v"(off), "v"(v), "s"(base) : "memory");
#endif
}

#ifdef KNERF_ABLATE_HALF_SAVED
// TIMING EXPERIMENT ONLY (r04, DESIGN.md 5.5: the ceiling of 8-bit saved tensors): a saved block keeps its 1 KiB slot but only
// its first 512 bytes are written (8 B per lane), and wgrad fetches only every second block from HBM (wgrad_body.h KNERF_SRC_BLOCK) -- half the HBM bytes of every saved
// activation / dZ tensor at an unchanged instruction count (the StoreSched tables stay valid).  The backward's RESULTS are garbage.
__device__ __forceinline__ void store8_wt(char* base, unsigned off, const u32x4& v) {
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
    const u32x2 w = {v[0] ^ v[2], v[1] ^ v[3]};
    asm volatile("global_store_dwordx2 %0, %1, %2 nt\n\ts_nop 1" ::"v"(off), "v"(w), "s"(base) : "memory");
}
#endif
// tile whose slot of the saved runs a chain wave writes: its own -- or, in a timing experiment, one of 256 slots that
// stay resident in the caches, so that the stores travel the whole on-chip path but (mostly) not to HBM
#ifdef KNERF_ABLATE_STORE_L2
#define KNERF_STORE_TILE(tile) ((tile) & 255)
#else
#define KNERF_STORE_TILE(tile) (tile)
#endif
// saved B-operand block (layout.h saved_off): lane (h = lane>>5, s = lane&31) -> (2*(s ^ 4*(block&1)) + h) * 16.
// `base` must be wave-uniform (it lives in SGPRs).
__device__ __forceinline__ void store_block(char* base, int block, int lane, const bf16x8& v) {
    const int s = lane & 31, h = lane >> 5;
#ifdef KNERF_ABLATE_HALF_SAVED
    store8_wt(base, (unsigned)(block * kSavedBlockStride + (2 * (s ^ ((block & 1) << 2)) + h) * 8), __builtin_bit_cast(u32x4, v));
#else
    store16_wt(base, (unsigned)(block * kSavedBlockStride + (2 * (s ^ ((block & 1) << 2)) + h) * 16), __builtin_bit_cast(u32x4, v));
#endif
}

}  // namespace knerf
