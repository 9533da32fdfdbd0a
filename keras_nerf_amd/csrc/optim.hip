// optim.hip -- Keras-form Adam over the flat parameter buffer, bf16 re-packing of the MFMA weight streams, finite check.
//
// Adam restates tf.keras.optimizers Adam as used at reference keras_nerf/model/nerf/nerf.py:163-165,455-458:
//   lr_t = lr*sqrt(1-b2^t)/(1-b1^t) (host, double);  m += (g-m)(1-b1);  v += (g*g-v)(1-b2);
//   w -= lr_t * m / (sqrt(v) + eps)      -- eps OUTSIDE the bias-corrected root, eps = 1e-7.
// The gradient accumulator is zeroed in the same pass (nerf.py:464-471) and a non-finite gradient raises a flag
// (the reference asserts finiteness per chunk, nerf.py:381-382,410-411; here once per step, before the update).
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace knerf {

__global__ void adam_kernel(AdamArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const float g = a.g[i];
    if (!__builtin_isfinite(g)) { *a.nonfinite = 1; }
    float m = a.m[i], v = a.v[i];
    m = m + (g - m) * (1.f - a.b1);
    v = v + (g * g - v) * (1.f - a.b2);
    a.m[i] = m; a.v[i] = v;
    a.w[i] = a.w[i] - a.lr_t * m / (sqrtf(v) + a.eps);
    a.g[i] = 0.f;
}
hipError_t launch_adam(const AdamArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(adam_kernel, dim3((a.n + 255) / 256), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// out[i] = bf16(w[table[i]]) (round to nearest even via the hardware convert), 0 where table[i] < 0
__global__ void pack_kernel(const float* w, const int* table, unsigned short* out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int t = table[i];
    const __bf16 b = (__bf16)(t >= 0 ? w[t] : 0.f);
    out[i] = __builtin_bit_cast(unsigned short, b);
}
hipError_t launch_pack(const float* w, const int* table, unsigned short* out, size_t n, hipStream_t stream) {
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, w, table, out, n);
    return hipGetLastError();
}

__global__ void gather_f32_kernel(const float* w, const int* table, float* out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int t = table[i];
    out[i] = t >= 0 ? w[t] : 0.f;
}
hipError_t launch_gather_f32(const float* w, const int* table, float* out, size_t n, hipStream_t stream) {
    hipLaunchKernelGGL(gather_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, w, table, out, n);
    return hipGetLastError();
}

__global__ void check_finite_kernel(const float* g, int n, int* flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !__builtin_isfinite(g[i])) *flag = 1;
}
hipError_t launch_check_finite(const float* g, int n, int* flag, hipStream_t stream) {
    hipLaunchKernelGGL(check_finite_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, g, n, flag);
    return hipGetLastError();
}

}  // namespace knerf
