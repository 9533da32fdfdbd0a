// probe.hip (libknerf_probe.so, diagnostics only) -- tiny kernels that pin the hardware facts the fused kernels rely on (tests/test_gpu_probe.py):
// the v_mfma_f32_32x32x16_bf16 operand maps and the ds_read_b64_tr_b16 transposing LDS read.
#include <hip/hip_runtime.h>
#include "../../include/knerf_debug.h"
#include "chain.h"

namespace knerf {
typedef __attribute__((ext_vector_type(4))) short s16x4;

__global__ void probe_mfma(const bf16x8* a, const bf16x8* b, f32x16* out) {
    f32x16 acc = zero_acc();
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    out[threadIdx.x] = acc;
}
// in: 4096 B LDS image, addr: 64 byte offsets (8-aligned), out: 64 x 4 u16
__global__ void probe_tr(const unsigned short* in, const int* addr, s16x4* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = in[i];
    __syncthreads();
    auto p = (__attribute__((address_space(3))) s16x4*)((__attribute__((address_space(3))) char*)lds + addr[threadIdx.x]);
    out[threadIdx.x] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
}
// MFMA-shape rate probes (MI355X_MICROARCH.md, DVFS give-back item 7): the chain kernels' inner loop -- one 1 KiB A fragment
// read from LDS per 32768 FLOP, B operands in registers, dependent accumulation -- once with v_mfma_f32_32x32x16_bf16
// and once with v_mfma_f32_16x16x32_bf16 (two 16-sample groups share each A fragment).  in0: >= 96 KiB of random bf16,
// in1: >= 64 x 128 B of random bf16, out: one float per thread (keeps the results alive).  iters = out-tile sweeps.
typedef __attribute__((ext_vector_type(4))) float f32x4p;
template <int SHAPE>
__global__ __launch_bounds__(512, 2) void probe_rate(const char* in0, const bf16x8* in1, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 96 * 1024 / 16; i += 512) reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(in0)[i];
    bf16x8 b[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) b[k] = in1[k * 64 + lane];
    __syncthreads();
    float sink = 0.f;
    // A fragments are read 4 blocks ahead of their use, as in chain.h (Prefetch)
    for (int it = 0; it < iters; ++it) {
        unsigned bo = lane * 16;
        asm volatile("" : "+v"(bo));            // the image never changes: keep hipcc from hoisting all 96 reads out of the loop
        const char* base = lds + bo;
        bf16x8 pf0 = *reinterpret_cast<const bf16x8*>(base), pf1 = *reinterpret_cast<const bf16x8*>(base + 1024),
               pf2 = *reinterpret_cast<const bf16x8*>(base + 2048), pf3 = *reinterpret_cast<const bf16x8*>(base + 3072);
        if (SHAPE == 32) {
#pragma unroll
            for (int ot = 0; ot < 6; ++ot) {
                f32x16 acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 16; ks += 4) {
                    const int blk = ot * 16 + ks;
                    bf16x8 c0 = pf0; pf0 = *reinterpret_cast<const bf16x8*>(base + ((blk + 4) % 96) * 1024);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c0, b[ks], acc, 0, 0, 0);
                    bf16x8 c1 = pf1; pf1 = *reinterpret_cast<const bf16x8*>(base + ((blk + 5) % 96) * 1024);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c1, b[ks + 1], acc, 0, 0, 0);
                    bf16x8 c2 = pf2; pf2 = *reinterpret_cast<const bf16x8*>(base + ((blk + 6) % 96) * 1024);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c2, b[ks + 2], acc, 0, 0, 0);
                    bf16x8 c3 = pf3; pf3 = *reinterpret_cast<const bf16x8*>(base + ((blk + 7) % 96) * 1024);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c3, b[ks + 3], acc, 0, 0, 0);
                }
                sink += acc[0] + acc[15];
            }
        } else {
#pragma unroll
            for (int ot = 0; ot < 12; ++ot) {
                f32x4p a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 8; ks += 4) {
                    const int blk = ot * 8 + ks;
                    bf16x8 c0 = pf0; pf0 = *reinterpret_cast<const bf16x8*>(base + ((blk + 4) % 96) * 1024);
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c0, b[ks], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c0, b[8 + ks], a1, 0, 0, 0);
                    bf16x8 c1 = pf1; pf1 = *reinterpret_cast<const bf16x8*>(base + ((blk + 5) % 96) * 1024);
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c1, b[ks + 1], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c1, b[9 + ks], a1, 0, 0, 0);
                    bf16x8 c2 = pf2; pf2 = *reinterpret_cast<const bf16x8*>(base + ((blk + 6) % 96) * 1024);
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c2, b[ks + 2], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c2, b[10 + ks], a1, 0, 0, 0);
                    bf16x8 c3 = pf3; pf3 = *reinterpret_cast<const bf16x8*>(base + ((blk + 7) % 96) * 1024);
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c3, b[ks + 3], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c3, b[11 + ks], a1, 0, 0, 0);
                }
                sink += a0[0] + a1[3];
            }
        }
    }
    out[blockIdx.x * 512 + tid] = sink;
}
// the same loop with 64 samples per wave: 4 waves per workgroup (one per SIMD), every A fragment feeds TWO B tiles, so
// the LDS bytes per FLOP halve (candidate layout for the inference kernel)
__global__ __launch_bounds__(256) void probe_rate64(const char* in0, const bf16x8* in1, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 96 * 1024 / 16; i += 256) reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(in0)[i];
    bf16x8 b0[16], b1[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { b0[k] = in1[k * 64 + lane]; b1[k] = in1[(16 + k) * 64 + lane]; }
    __syncthreads();
    float sink = 0.f;
    for (int it = 0; it < iters; ++it) {
        unsigned bo = lane * 16;
        asm volatile("" : "+v"(bo));
        const char* base = lds + bo;
        bf16x8 pf0 = *reinterpret_cast<const bf16x8*>(base), pf1 = *reinterpret_cast<const bf16x8*>(base + 1024),
               pf2 = *reinterpret_cast<const bf16x8*>(base + 2048), pf3 = *reinterpret_cast<const bf16x8*>(base + 3072);
#pragma unroll
        for (int ot = 0; ot < 6; ++ot) {
            f32x16 a0, a1;
#pragma unroll
            for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 16; ks += 4) {
                const int blk = ot * 16 + ks;
                bf16x8 c0 = pf0; pf0 = *reinterpret_cast<const bf16x8*>(base + ((blk + 4) % 96) * 1024);
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c0, b0[ks], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c0, b1[ks], a1, 0, 0, 0);
                bf16x8 c1 = pf1; pf1 = *reinterpret_cast<const bf16x8*>(base + ((blk + 5) % 96) * 1024);
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c1, b0[ks + 1], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c1, b1[ks + 1], a1, 0, 0, 0);
                bf16x8 c2 = pf2; pf2 = *reinterpret_cast<const bf16x8*>(base + ((blk + 6) % 96) * 1024);
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c2, b0[ks + 2], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c2, b1[ks + 2], a1, 0, 0, 0);
                bf16x8 c3 = pf3; pf3 = *reinterpret_cast<const bf16x8*>(base + ((blk + 7) % 96) * 1024);
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c3, b0[ks + 3], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c3, b1[ks + 3], a1, 0, 0, 0);
            }
            sink += a0[0] + a1[15];
        }
    }
    out[blockIdx.x * 256 + tid] = sink;
}
// write-pattern probe: every wave streams `blocks` 1 KiB blocks (nt stores, as the chain kernels issue them)
__global__ __launch_bounds__(512, 2) void probe_write(char* out, int blocks, long long tile_stride, int mode, int spin) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long tile = (long long)blockIdx.x * 8 + wave;
    u32x4 v = {(unsigned)tile, (unsigned)lane, 3u, 4u};
    // mode 0: tile-major; mode g > 0: groups of g tiles stored block-major ([group][block][tile in group]), so that waves
    // that are resident together write neighbouring KiB (g = 8: the waves of one workgroup)
    const long long grp = mode > 0 ? mode : 1;
    char* base = mode == 0 ? out + tile * tile_stride : out + (tile / grp) * grp * tile_stride + (tile % grp) * 1024;
    const long long bstride = mode == 0 ? 1024 : grp * 1024;
    for (int b = 0; b < blocks; ++b) {
        for (int k = 0; k < (spin & 0xffff); ++k) asm volatile("s_nop 7");          // stands in for the MFMAs between two epilogues
        v[2] += b;
        char* p = base + b * bstride + lane * 16;
        // spin's upper bits select the cache policy (diagnostic): 0 nt (the kernels' choice), 1 plain, 2 sc1, 3 sc0 sc1
        const int pol = spin >> 16;
        if (pol == 0) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
        else if (pol == 1) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
        else if (pol == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    }
}
}  // namespace knerf

namespace knerf {
// read-pattern probe: `workgroups` x 8 waves stream contiguous ranges with 16-byte-per-lane loads (1 KiB per wave
// instruction, `depth` instructions in flight per wave), mode 0: nt LDS-DMA as wgrad issues them, mode 1: plain loads into
// registers.  Nothing is computed: this is the HBM read ceiling of the access pattern.
__global__ __launch_bounds__(512, 2) void probe_read(const char* in, long long bytes_per_wave, int mode, float* out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* p = in + ((long long)blockIdx.x * 8 + wave) * bytes_per_wave + lane * 16;
    const long long n = bytes_per_wave / 1024;
    u32x4 acc = {0u, 0u, 0u, 0u};
    if (mode == 2 || mode == 3) {
        // wgrad's rhythm: G = 4 (mode 2) or 8 (mode 3) copies per wave, wait until all but the youngest 2 G have landed,
        // workgroup barrier -- 3 "tiles" in flight, one barrier per tile
        const int G = mode == 2 ? 4 : 8;
        const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(lds + wave * 16 * 1024);
        for (long long b = 0; b < n; b += G) {
            for (int k = 0; k < G; ++k) {
                const unsigned m0v = dst + (unsigned)((b + k) & 15) * 1024;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(p + (b + k) * 1024), "s"(m0v) : "memory");
            }
            if (G == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (mode == 0) {
        const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(lds + wave * 16 * 1024);
        for (long long b = 0; b < n; ++b) {
            const unsigned m0v = dst + (unsigned)(b & 15) * 1024;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(12)"
                         : "=&s"(keep) : "v"(p + b * 1024), "s"(m0v) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        for (long long b = 0; b < n; b += 8) {
            u32x4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const u32x4*>(p + (b + k) * 1024);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc ^= v[k];
        }
    }
    if (acc[0] == 0x12345u) out[threadIdx.x] = 1.f;
}
}  // namespace knerf

extern "C" int knerf_debug_read_probe(const void* in, int workgroups, long long bytes_per_wave, int mode, void* out, void* stream) {
    using namespace knerf;
    const size_t lds = 128 * 1024;
    static bool done = false;
    if (!done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(probe_read), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return KNERF_ERR_HIP;
        done = true;
    }
    hipLaunchKernelGGL(probe_read, dim3(workgroups), dim3(512), lds, (hipStream_t)stream, (const char*)in, bytes_per_wave, mode, (float*)out);
    return hipGetLastError() == hipSuccess ? KNERF_OK : KNERF_ERR_HIP;
}

extern "C" int knerf_debug_write_probe(void* out, int workgroups, int blocks, long long tile_stride, int mode, int spin, void* stream) {
    using namespace knerf;
    // spin bit 30: reserve 100 KiB of LDS so that one workgroup fills a CU (2048 resident waves, as in the chain kernels)
    const size_t lds = (spin >> 30) & 1 ? 100 * 1024 : 0;
    static bool done = false;
    if (!done && lds) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(probe_write), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return KNERF_ERR_HIP;
        done = true;
    }
    hipLaunchKernelGGL(probe_write, dim3(workgroups), dim3(512), lds, (hipStream_t)stream, (char*)out, blocks, tile_stride, mode, spin & ~(1 << 30));
    return hipGetLastError() == hipSuccess ? KNERF_OK : KNERF_ERR_HIP;
}

extern "C" int knerf_debug_rate_probe(int shape, const void* in0, const void* in1, void* out, int blocks, int iters, void* stream) {
    using namespace knerf;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = 96 * 1024;
    static bool done = false;
    if (!done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(probe_rate<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return KNERF_ERR_HIP;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(probe_rate<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return KNERF_ERR_HIP;
        done = true;
    }
    if (shape == 64) {
        static bool done64 = false;
        if (!done64) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(probe_rate64), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return KNERF_ERR_HIP;
            done64 = true;
        }
        hipLaunchKernelGGL(probe_rate64, dim3(blocks), dim3(256), lds, s, (const char*)in0, (const bf16x8*)in1, (float*)out, iters);
        return hipGetLastError() == hipSuccess ? KNERF_OK : KNERF_ERR_HIP;
    }
    if (shape == 32) hipLaunchKernelGGL(probe_rate<32>, dim3(blocks), dim3(512), lds, s, (const char*)in0, (const bf16x8*)in1, (float*)out, iters);
    else if (shape == 16) hipLaunchKernelGGL(probe_rate<16>, dim3(blocks), dim3(512), lds, s, (const char*)in0, (const bf16x8*)in1, (float*)out, iters);
    else return KNERF_ERR_INVALID;
    return hipGetLastError() == hipSuccess ? KNERF_OK : KNERF_ERR_HIP;
}

extern "C" int knerf_debug_probe(int kind, const void* in0, const void* in1, void* out, void* stream) {
    using namespace knerf;
    hipStream_t s = (hipStream_t)stream;
    if (kind == 0) hipLaunchKernelGGL(probe_mfma, dim3(1), dim3(64), 0, s, (const bf16x8*)in0, (const bf16x8*)in1, (f32x16*)out);
    else if (kind == 1) hipLaunchKernelGGL(probe_tr, dim3(1), dim3(64), 0, s, (const unsigned short*)in0, (const int*)in1, (s16x4*)out);
    else return KNERF_ERR_INVALID;
    return hipGetLastError() == hipSuccess ? KNERF_OK : KNERF_ERR_HIP;
}
