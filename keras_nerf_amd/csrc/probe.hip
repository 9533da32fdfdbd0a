// probe.hip -- tiny kernels that pin the hardware facts the fused kernels rely on (tests/test_gpu_probe.py):
// the v_mfma_f32_32x32x16_bf16 operand maps and the ds_read_b64_tr_b16 transposing LDS read.
#include <hip/hip_runtime.h>
#include "../../include/knerf.h"
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
}  // namespace knerf

extern "C" int knerf_debug_probe(int kind, const void* in0, const void* in1, void* out, void* stream) {
    using namespace knerf;
    hipStream_t s = (hipStream_t)stream;
    if (kind == 0) hipLaunchKernelGGL(probe_mfma, dim3(1), dim3(64), 0, s, (const bf16x8*)in0, (const bf16x8*)in1, (f32x16*)out);
    else if (kind == 1) hipLaunchKernelGGL(probe_tr, dim3(1), dim3(64), 0, s, (const unsigned short*)in0, (const int*)in1, (s16x4*)out);
    else return KNERF_ERR_INVALID;
    return hipGetLastError() == hipSuccess ? KNERF_OK : KNERF_ERR_HIP;
}
