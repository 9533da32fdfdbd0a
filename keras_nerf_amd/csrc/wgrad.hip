// wgrad.hip -- stand-alone launch of the weight-gradient jobs (wgrad_body.h): one workgroup per (job, contiguous tile range).
#include "wgrad_body.h"

namespace knerf {

// NET only names the instantiation (0 = coarse pass, 1 = fine pass) so that profiler summaries list the two launch sizes
// separately (profiles/*kernel_stats*.csv against bench.py's wgrad_coarse / wgrad_fine)
template <int NET>
__global__ __launch_bounds__(kWgThreads, 2) void wgrad_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef KNERF_WGRAD_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_wgrad_stamps[blockIdx.x * 8 + 6] = stamp();
#endif
    const WgradPlan pl = reinterpret_cast<const WgradPlan*>(a.plan)[blockIdx.x];
    const ContigSeq seq{a.n_tiles * pl.split / pl.nsplit, a.n_tiles * (pl.split + 1) / pl.nsplit};
    wgrad_dispatch(a, pl.job, seq, smem);
#ifdef KNERF_WGRAD_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_wgrad_stamps[blockIdx.x * 8 + 7] = stamp() - g_wgrad_stamps[blockIdx.x * 8 + 7];
#endif
}

#ifdef KNERF_WGRAD_STAMPS
extern "C" int knerf_debug_wgrad_stamps(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wgrad_stamps), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif

hipError_t launch_wgrad(const WgradArgs& a, hipStream_t stream) {
    const size_t lds = 160 * 1024;
    static AttrOnce once;
    hipError_t ae = once([&]() -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        return hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    if (ae != hipSuccess) return ae;
    if (a.net == 0) hipLaunchKernelGGL(wgrad_kernel<0>, dim3(a.n_plan), dim3(kWgThreads), lds, stream, a);
    else hipLaunchKernelGGL(wgrad_kernel<1>, dim3(a.n_plan), dim3(kWgThreads), lds, stream, a);
    return hipGetLastError();
}

}  // namespace knerf
