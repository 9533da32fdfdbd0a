// raygen.hip -- pinhole ray generation on device (gfx950).
//
// Restates RaysGenerator.__call__ (reference keras_nerf/data/rays.py:69-130): pixel-corner grid, camera vector
// ((x-W/2)/f, -(y-H/2)/f, -1), d = sum(cam[...,None,:] * R, -1) normalised, o = c2w[:3,3],
// t = clip(linspace(near,far,N) + noise*interval - interval/2, near, far), noise ~ U[0,1) (injected or Philox).
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace knerf {

__device__ __forceinline__ void philox_round_rg(unsigned (&c)[4], unsigned k0, unsigned k1) {
    const unsigned long long p0 = (unsigned long long)c[0] * 0xD2511F53ull;
    const unsigned long long p1 = (unsigned long long)c[2] * 0xCD9E8D57ull;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (unsigned)p1; c[3] = (unsigned)p0; c[0] = n0; c[2] = n2;
}

// one thread per (ray, sample); the ray part is recomputed per sample (cheap) to keep the stores coalesced
__global__ void raygen_kernel(RayGenArgs a) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)a.B * a.H * a.W * a.N;
    if (idx >= total) return;
    const int n = (int)(idx % a.N);
    const long long ray = idx / a.N;
    const int xpix = (int)(ray % a.W);
    const int ypix = (int)((ray / a.W) % a.H);
    const int b = (int)(ray / ((long long)a.W * a.H));
    const float* M = a.c2w + (size_t)b * 16;
    // linspace(near, far, N)[n] as TF computes it: start + n * ((stop-start)/(N-1)); last point exact
    const float step = a.N > 1 ? (a.far_ - a.near_) / (float)(a.N - 1) : 0.f;
    const float base = (n == a.N - 1 && a.N > 1) ? a.far_ : a.near_ + (float)n * step;
    const float interval = (a.far_ - a.near_) / (float)a.N;
    float u;
    if (a.noise) u = a.noise[idx];
    else {
        unsigned c[4] = {(unsigned)(n >> 2), (unsigned)ray, (unsigned)a.stream_id, 1u};
        unsigned k0 = (unsigned)a.seed, k1 = (unsigned)(a.seed >> 32);
#pragma unroll
        for (int i = 0; i < 10; ++i) { philox_round_rg(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
        u = (float)(c[n & 3] >> 8) * 5.9604644775390625e-08f;
    }
    float tv = __fsub_rn(__fadd_rn(base, __fmul_rn(u, interval)), interval / 2.f);
    a.t[idx] = fminf(fmaxf(tv, a.near_), a.far_);
    if (n == 0) {
        const float xc = __fdiv_rn((float)xpix - (float)a.W * 0.5f, a.focal);
        const float yc = __fdiv_rn((float)ypix - (float)a.H * 0.5f, a.focal);
        const float cam[3] = {xc, -yc, -1.f};
        float dv[3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
            dv[r] = __fadd_rn(__fadd_rn(__fmul_rn(cam[0], M[r * 4 + 0]), __fmul_rn(cam[1], M[r * 4 + 1])), __fmul_rn(cam[2], M[r * 4 + 2]));
        const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dv[0], dv[0]), __fmul_rn(dv[1], dv[1])), __fmul_rn(dv[2], dv[2])));
#pragma unroll
        for (int r = 0; r < 3; ++r) { a.d[ray * 3 + r] = __fdiv_rn(dv[r], nrm); a.o[ray * 3 + r] = M[r * 4 + 3]; }
    }
}

hipError_t launch_raygen(const RayGenArgs& a, hipStream_t stream) {
    const long long total = (long long)a.B * a.H * a.W * a.N;
    hipLaunchKernelGGL(raygen_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace knerf
