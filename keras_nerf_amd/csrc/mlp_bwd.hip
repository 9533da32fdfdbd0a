// mlp_bwd.hip -- stand-alone launch of the fused dgrad chain (bwd_body.h): one workgroup per 256-sample tile.
#include "bwd_body.h"

namespace knerf {

// NET only names the instantiation (0 = coarse pass, 1 = fine pass) for profiler summaries
template <int NET>
__global__ __launch_bounds__(kThreads, 2) void mlp_bwd_kernel(BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    mlp_bwd_tile(a, smem, blockIdx.x);
}

hipError_t launch_mlp_bwd(const BwdArgs& a, hipStream_t stream) {
    const long long tiles = (a.n_samples + kTile - 1) / kTile;
    const int grid = (int)((tiles + kWaves - 1) / kWaves);
    const size_t lds = kRingBytes;
    static AttrOnce once;
    hipError_t ae = once([&]() -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_bwd_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        return hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_bwd_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    if (ae != hipSuccess) return ae;
    if (a.net == 0) hipLaunchKernelGGL(mlp_bwd_kernel<0>, dim3(grid), dim3(kThreads), lds, stream, a);
    else hipLaunchKernelGGL(mlp_bwd_kernel<1>, dim3(grid), dim3(kThreads), lds, stream, a);
    return hipGetLastError();
}

}  // namespace knerf
