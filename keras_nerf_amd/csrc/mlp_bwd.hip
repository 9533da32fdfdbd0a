// mlp_bwd.hip -- stand-alone launch of the fused dgrad chain (bwd_body.h): one workgroup per 256-sample tile.
#include "bwd_body.h"

namespace knerf {

// NET only names the instantiation (0 = coarse pass, 1 = fine pass) for profiler summaries; S = the trunk shape (layout.h)
template <class S, int NET>
__global__ __launch_bounds__(kThreads, 2) void mlp_bwd_kernel(BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    mlp_bwd_tile<S>(a, smem, blockIdx.x);
}

template <class S>
hipError_t launch_mlp_bwd_t(const BwdArgs& a, hipStream_t stream) {
    const long long tiles = (a.n_samples + kTile - 1) / kTile;
    const int grid = (int)((tiles + kWaves - 1) / kWaves);
    const size_t lds = kRingBytes;
    static AttrOnce once;
    hipError_t ae = once([&]() -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_bwd_kernel<S, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        return hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_bwd_kernel<S, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    if (ae != hipSuccess) return ae;
    if (a.net == 0) hipLaunchKernelGGL((mlp_bwd_kernel<S, 0>), dim3(grid), dim3(kThreads), lds, stream, a);
    else hipLaunchKernelGGL((mlp_bwd_kernel<S, 1>), dim3(grid), dim3(kThreads), lds, stream, a);
    return hipGetLastError();
}

// explicit instantiation of this translation unit's shape(s), `extern template` for the others (layout.h KNERF_FUSED_SHAPES)
#define KNERF_X(I, ...) KNERF_PICK(I, template, extern template) hipError_t launch_mlp_bwd_t<KNERF_SHAPE_T(__VA_ARGS__)>(const BwdArgs&, hipStream_t);
KNERF_FUSED_SHAPES(KNERF_X)
#undef KNERF_X

#if KNERF_HAS_DISPATCH
hipError_t launch_mlp_bwd(const BwdArgs& a, hipStream_t stream) {
    switch (a.shape) {
#define KNERF_X(I, ...) case I: return launch_mlp_bwd_t<KNERF_SHAPE_T(__VA_ARGS__)>(a, stream);
        KNERF_FUSED_SHAPES(KNERF_X)
#undef KNERF_X
        default: return hipErrorInvalidValue;
    }
}
#endif

}  // namespace knerf
