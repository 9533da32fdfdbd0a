// fused_bwd.hip -- dgrad chain and weight-gradient jobs in ONE launch, so that dZ is consumed while it is still on chip.
//
// 256 co-resident workgroups (one per CU) with fixed roles: workgroups [0, P) are PRODUCERS, persistent over the
// 256-sample workgroup tiles (bwd_body.h: the same dgrad chain as mlp_bwd.hip); the rest are CONSUMERS running the wgrad
// jobs of wgrad_body.h on every nsplit-th workgroup tile.  Hand-off per workgroup tile (CDNA4 guide, Guideline 16):
//   producer: every wave drains its stores (vmcnt(0)) -> workgroup barrier -> one lane: agent-scope release (writes the
//             XCD's L2 back) -> vmcnt(0) -> relaxed agent-scope store of the launch epoch into flags[tile]
//   consumer: sc1 load of the flag (pre-issued as LDS-DMA three iterations ahead, blocking poll only when the producers
//             are behind), then first-touch LDS-DMA reads of that tile's dZ blocks.
// Every poll is bounded: a time-out raises `abort` and lets the kernel finish (the host reports failure), it never hangs.
// Nothing here depends on dispatch order or XCD placement; co-residency follows from grid = number of CUs with one
// 160 KiB-LDS workgroup per CU.
#include "bwd_body.h"
#include "wgrad_body.h"

namespace knerf {

__global__ __launch_bounds__(kThreads, 2) void bwd_wgrad_kernel(FusedArgs f) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x < f.n_producers) {
        if (f.debug & 2) return;
        for (long long T = blockIdx.x; T < f.n_wg_tiles; T += f.n_producers) {
            BwdArgs b = f.bwd;                            // opaque per tile: no per-page address is hoisted out of the loop
            asm volatile("" : "+s"(b.stream), "+s"(b.dz), "+s"(b.mask), "+s"(b.raw), "+s"(b.draw));
            mlp_bwd_tile(b, smem, T);                     // ends with every wave's vmcnt(0)
            __syncthreads();                              // all 8 waves' dZ stores of this tile have completed
            if (threadIdx.x == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(f.flags + T, f.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    } else {
        if (f.debug & 1) return;
        const WgradPlan pl = reinterpret_cast<const WgradPlan*>(f.wgrad.plan)[blockIdx.x - f.n_producers];
        const FusedSeq seq{pl.split, pl.nsplit, f.n_wg_tiles, f.flags, f.epoch, f.abort_flag, (f.debug & 4) != 0};
        wgrad_dispatch(f.wgrad, pl.job, seq, smem);
    }
}

hipError_t launch_bwd_wgrad(const FusedArgs& f, hipStream_t stream) {
    const size_t lds = 160 * 1024;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bwd_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(bwd_wgrad_kernel, dim3(f.n_producers + f.wgrad.n_plan), dim3(kThreads), lds, stream, f);
    return hipGetLastError();
}

}  // namespace knerf
