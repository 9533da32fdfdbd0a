// chain.h -- device machinery shared by the fused forward and dgrad MLP chain kernels (gfx950 only).
//
// One workgroup = 8 waves = 256 samples; each wave owns 32 samples (one per lane&31; the two lane halves hold
// different feature rows of the same sample).  Activations never leave registers between layers: the 32x32 f32
// MFMA result, converted to bf16, is the next layer's B operand (layout.h).  Weights arrive as a linear stream of
// 1 KiB A-fragment blocks pulled by LDS-DMA (global_load_lds_dwordx4) through a ring of 16 KiB pages.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace knerf {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int kWaves = 8;                 // waves per workgroup
constexpr int kThreads = kWaves * 64;
constexpr int kTile = 32;                 // samples per wave
constexpr int kPageBlocks = 16;           // 1 KiB blocks per ring page
constexpr int kPageBytes = kPageBlocks * 1024;
constexpr int kSlots = 6;                 // ring slots (96 KiB)
constexpr int kRingBytes = kSlots * kPageBytes;
constexpr int kGldsPerPage = kPageBytes / (kThreads * 16);   // 2 LDS-DMA instructions per thread per page
constexpr int kWaitInFlight = kGldsPerPage * (kSlots - 3);   // vmcnt at a mid-page sync: pages P+2..P+kSlots-2 may fly
constexpr int kTailPages = kSlots;        // dummy pages appended to every stream so that issue never needs a guard

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

struct Ring {
    const char* stream;    // global: packed bf16 A-fragments, page after page
    char* lds;             // LDS base of the ring (16-byte aligned, offset 0 of the dynamic segment)
    int tid;

    // every thread moves 2 x 16 B of page `page` into slot page % kSlots
    __device__ __forceinline__ void issue(int page) const {
        const char* src = stream + (size_t)page * kPageBytes + tid * 16;
        char* dst = lds + (page % kSlots) * kPageBytes + (tid & ~63) * 16;   // wave-uniform base; HW adds lane*16
#pragma unroll
        for (int i = 0; i < kGldsPerPage; ++i)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + i * kThreads * 16), (lds_ptr_t)(dst + i * kThreads * 16), 16, 0, 0);
    }
    __device__ __forceinline__ void prologue_issue() const {
#pragma unroll
        for (int p = 0; p < kSlots - 1; ++p) issue(p);
    }
    __device__ __forceinline__ void prologue_wait() const {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kGldsPerPage * (kSlots - 2)) : "memory");   // page 0 landed (mine)
        __builtin_amdgcn_s_barrier();                                                          // ... and everyone's
    }
    // called when block b == 8 (mod 16), P = b/16: make page P+1 readable, recycle the slot of page P-1
    __device__ __forceinline__ void sync(int P) const {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kWaitInFlight) : "memory");
        __builtin_amdgcn_s_barrier();
        issue(P + kSlots - 1);
    }
    __device__ __forceinline__ void drain() const { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

    __device__ __forceinline__ bf16x8 frag(int b, int lane) const {
        const char* p = lds + ((b / kPageBlocks) % kSlots) * kPageBytes + (b % kPageBlocks) * 1024 + lane * 16;
        return *reinterpret_cast<const bf16x8*>(p);
    }
};

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

// f32x16 accumulator -> two bf16x8 B-operand k-steps (regs 0..7 -> k-step 0, regs 8..15 -> k-step 1)
__device__ __forceinline__ void pack_acc(const f32x16& a, bf16x8& lo, bf16x8& hi) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { lo[j] = (__bf16)a[j]; hi[j] = (__bf16)a[8 + j]; }
}

// A-fragment prefetch ring: block b's fragment is read from LDS kPrefetch MFMAs before it is used.  kPrefetch <= 8
// keeps every read inside the half page that the last sync made readable (see Ring::sync).
constexpr int kPrefetch = 4;
struct Prefetch {
    bf16x8 a[kPrefetch];
    template <int NBLOCKS>
    __device__ __forceinline__ void start(const Ring& ring, int lane) {
#pragma unroll
        for (int i = 0; i < kPrefetch; ++i) a[i] = ring.frag(i < NBLOCKS ? i : 0, lane);
    }
};

// One dense stage: for every out tile, acc = init(ot); acc += A(block) x in(ks) over the stage's k-steps; epi(ot, acc).
// B0 = index of the stage's first block in the stream, NBLOCKS the stream length.  All indices fold to constants
// after unrolling.
template <int B0, int NKS, int NOT, int NBLOCKS, class Init, class In, class Epi>
__device__ __forceinline__ void dense_stage(const Ring& ring, Prefetch& pf, int lane, Init&& init, In&& in, Epi&& epi) {
#pragma unroll
    for (int ot = 0; ot < NOT; ++ot) {
        f32x16 acc = init(ot);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int b = B0 + ot * NKS + ks;
            if (b % kPageBlocks == kPageBlocks / 2) ring.sync(b / kPageBlocks);
            bf16x8 cur = pf.a[b % kPrefetch];
            if (b + kPrefetch < NBLOCKS) pf.a[b % kPrefetch] = ring.frag(b + kPrefetch, lane);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur, in(ks), acc, 0, 0, 0);
        }
        epi(ot, acc);
    }
}

// saved B-operand block (layout.h saved_off): lane (h = lane>>5, s = lane&31) -> (2*(s ^ 4*(block&1)) + h) * 16
__device__ __forceinline__ void store_block(char* base, size_t block, int lane, const bf16x8& v) {
    const int s = lane & 31, h = lane >> 5;
    *reinterpret_cast<bf16x8*>(base + block * 1024 + (2 * (s ^ (((int)block & 1) << 2)) + h) * 16) = v;
}

}  // namespace knerf
