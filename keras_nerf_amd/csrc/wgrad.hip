// wgrad.hip -- stand-alone launch of the weight-gradient jobs (wgrad_body.h): one workgroup per (job, contiguous tile range).
#include "wgrad_body.h"

namespace knerf {

// first index of the ascending list whose entry is >= t (uniform: scalar loads)
__device__ __forceinline__ int list_lower_bound(const int* list, int n, int t) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (list[mid] < t) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// NET only names the instantiation (0 = coarse pass, 1 = fine pass) so that profiler summaries list the two launch sizes
// separately (profiles/*kernel_stats*.csv against bench.py's wgrad_coarse / wgrad_fine).  LIST: the tiles come from the
// compacted list of live tiles (dead-tile skipping) instead of the contiguous range.
template <class S, int NET, bool LIST>
__global__ __launch_bounds__(kWgThreads, 2) void wgrad_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef KNERF_WGRAD_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_wgrad_stamps[blockIdx.x * 8 + 6] = stamp();
#endif
    const WgradPlan pl = reinterpret_cast<const WgradPlan*>(a.plan)[blockIdx.x];
    if constexpr (!LIST) {
        const ContigSeq seq{(int)(a.n_tiles * pl.split / pl.nsplit), (int)(a.n_tiles * (pl.split + 1) / pl.nsplit)};
        wgrad_dispatch<S>(a, pl.job, seq, smem);
    } else {
        // plain loads, before any LDS-DMA copy is in flight (hipcc waits for them with its own vmcnt)
        int n = __builtin_amdgcn_readfirstlane(*a.n_live);
        if (n > a.n_tiles) n = (int)a.n_tiles;                  // never more entries than the launch has tiles
        if (n < 0) n = 0;
        ListSeq seq{a.live, (int)((long long)n * pl.split / pl.nsplit), (int)((long long)n * (pl.split + 1) / pl.nsplit)};
        if (a.by_range) {
            seq.i0 = list_lower_bound(a.live, n, (int)(a.n_tiles * pl.split / pl.nsplit));
            seq.i1 = list_lower_bound(a.live, n, (int)(a.n_tiles * (pl.split + 1) / pl.nsplit));
        }
        seq.i0 = __builtin_amdgcn_readfirstlane(seq.i0); seq.i1 = __builtin_amdgcn_readfirstlane(seq.i1);
        wgrad_dispatch<S>(a, pl.job, seq, smem);
    }
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

template <class S>
hipError_t launch_wgrad_t(const WgradArgs& a, hipStream_t stream) {
    const size_t lds = 160 * 1024;
    static AttrOnce once;
    hipError_t ae = once([&]() -> hipError_t {
        const void* fns[4] = {reinterpret_cast<const void*>(wgrad_kernel<S, 0, false>), reinterpret_cast<const void*>(wgrad_kernel<S, 1, false>),
                              reinterpret_cast<const void*>(wgrad_kernel<S, 0, true>), reinterpret_cast<const void*>(wgrad_kernel<S, 1, true>)};
        for (const void* f : fns) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    });
    if (ae != hipSuccess) return ae;
    const dim3 g(a.n_plan), b(kWgThreads);
    if (a.live) {
        if (a.net == 0) hipLaunchKernelGGL((wgrad_kernel<S, 0, true>), g, b, lds, stream, a);
        else hipLaunchKernelGGL((wgrad_kernel<S, 1, true>), g, b, lds, stream, a);
    } else {
        if (a.net == 0) hipLaunchKernelGGL((wgrad_kernel<S, 0, false>), g, b, lds, stream, a);
        else hipLaunchKernelGGL((wgrad_kernel<S, 1, false>), g, b, lds, stream, a);
    }
    return hipGetLastError();
}

// explicit instantiation of this translation unit's shape(s), `extern template` for the others (layout.h KNERF_FUSED_SHAPES)
#define KNERF_X(I, ...) KNERF_PICK(I, template, extern template) hipError_t launch_wgrad_t<KNERF_SHAPE_T(__VA_ARGS__)>(const WgradArgs&, hipStream_t);
KNERF_FUSED_SHAPES(KNERF_X)
#undef KNERF_X

#if KNERF_HAS_DISPATCH
hipError_t launch_wgrad(const WgradArgs& a, hipStream_t stream) {
    switch (a.shape) {
#define KNERF_X(I, ...) case I: return launch_wgrad_t<KNERF_SHAPE_T(__VA_ARGS__)>(a, stream);
        KNERF_FUSED_SHAPES(KNERF_X)
#undef KNERF_X
        default: return hipErrorInvalidValue;
    }
}

// ---- deterministic mode: ordered second pass over the per-workgroup slabs (wgrad_body.h flush_acc / flush_bias) ------------------
size_t wgrad_partial_floats(int n_plan, int stride) { return (size_t)n_plan * (size_t)stride; }

// one thread per element of a job's destination table; the job's workgroups are the plan entries [job_wg0[j], job_wg0[j+1])
// in ascending split order.  Every destination index occurs once per launch (a weight belongs to one job), so the plain
// read-modify-write of grad races with nothing.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgradArgs a, const int* job_wg0, int stride) {
    const int job = blockIdx.y;
    const int n_elem = a.job_off[job + 1] - a.job_off[job];
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n_elem) return;
    const int d = a.dst[a.job_off[job] + e];
    if (d < 0) return;
    float s = 0.f;
    for (int wg = job_wg0[job]; wg < job_wg0[job + 1]; ++wg) s += a.partial[(size_t)wg * stride + e];
    float* p = d < a.aux_base ? a.grad + d : a.aux + (d - a.aux_base);
    *p += s;
}
hipError_t launch_wgrad_reduce(const WgradArgs& a, const int* job_wg0, int stride, hipStream_t stream) {
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((stride + 255) / 256, a.n_jobs), dim3(256), 0, stream, a, job_wg0, stride);
    return hipGetLastError();
}
#endif

}  // namespace knerf
