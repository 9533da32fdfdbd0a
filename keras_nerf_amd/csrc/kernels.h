// kernels.h -- launch interfaces between knerf_api.hip and the kernel files (internal; the public ABI is include/knerf.h)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace knerf {

struct FwdArgs {
    const char* stream;     // packed forward A-fragments (bf16), kFwdBlocks KiB + tail pages
    const float* bias;      // kFwdBiasTiles*32 fp32 in accumulator-tile order
    const float* o;         // [R,3]
    const float* d;         // [R,3]
    const float* t;         // [R,S]
    float* raw;             // [R*S,4]  (r,g,b,sigma)
    char* act;              // SAVE: [tiles][kActBlocks][1 KiB]
    char* mask;             // SAVE: [tiles][kMaskBlocks][1 KiB]
    long long n_samples;    // R*S
    int S;
    int net;                // 0 coarse / 1 fine: selects the kernel instantiation (a name for profilers), nothing else
};
hipError_t launch_mlp_fwd(const FwdArgs& a, bool save, hipStream_t stream);

struct BwdArgs {
    const char* stream;     // packed dgrad A-fragments
    const float* raw;       // [R*S,4] forward outputs (rgb after sigmoid, sigma after relu)
    const float* draw;      // [R*S,4] dL/d(rgb,sigma) from the compositing backward
    const char* mask;       // relu masks saved by the forward
    char* dz;               // out: [tiles][kDzBlocks][1 KiB]
    long long n_samples;
    int net;                // 0 coarse / 1 fine: kernel instantiation name only
};
hipError_t launch_mlp_bwd(const BwdArgs& a, hipStream_t stream);

struct WgradArgs {
    const char* act;        // [tiles][kActBlocks][1 KiB]
    const char* dz;         // [tiles][kDzBlocks][1 KiB]
    float* grad;            // flat fp32 gradient accumulator of this net (kParamCount)
    const int* dst;         // concatenated per-job destination tables: [(32*n_it + 1) rows][32*n_ot cols] param index or -1
    const void* plan;       // device array of WgradPlan, one per workgroup
    long long n_tiles;
    int n_plan;
    int net;                // 0 coarse / 1 fine: selects the kernel instantiation (a name for profilers), nothing else
    int job_off[14];        // offset of each job's table inside dst
};
hipError_t launch_wgrad(const WgradArgs& a, hipStream_t stream);

// dgrad + wgrad in one launch (fused_bwd.hip)
struct FusedArgs {
    BwdArgs bwd;
    WgradArgs wgrad;          // plan = consumer workgroups only
    unsigned* flags;          // one word per 256-sample workgroup tile; == epoch when that tile's dZ is published
    int* abort_flag;          // raised by a consumer whose bounded poll timed out
    long long n_wg_tiles;
    unsigned epoch;           // launch counter (flags are never cleared)
    int n_producers;
    int debug;                // KNERF_FUSED_DEBUG bits: 1 consumers idle, 2 producers idle, 4 consumers ignore flags
};
hipError_t launch_bwd_wgrad(const FusedArgs& f, hipStream_t stream);

struct CompositeArgs {
    const float* raw;       // [R,S,4]
    const float* t;         // [R,S]
    const float* target;    // [R,3] or null (forward only)
    float* image;           // [R,3]
    float* depth;           // [R] or null
    float* weights;         // [R,S] or null
    float* draw;            // [R,S,4] out (training) or null
    float* loss;            // scalar accumulator (training): += mse_chunk * loss_scale
    int R, S;
    int white;
    float grad_scale;       // 2 / (3R) * inv_chunks  -> dL/dimage = grad_scale * (image - target)
    float loss_scale;       // inv_chunks / (3R)
};
hipError_t launch_composite(const CompositeArgs& a, hipStream_t stream);

struct SampleArgs {
    const float* t_coarse;  // [R,Nc]
    const float* w_coarse;  // [R,Nc]
    const float* u;         // [R,Nf] or null -> Philox
    float* t_out;           // [R,Nc+Nf] sorted ascending
    int R, Nc, Nf;
    int oob_clamp;          // 0: out-of-range gather yields 0 (tf.gather on GPU); 1: clamp
    unsigned long long seed, stream_id, ray_offset;
};
hipError_t launch_sample_fine(const SampleArgs& a, hipStream_t stream);

struct AdamArgs {
    float* w; float* m; float* v; float* g;   // kParamCount each; g is zeroed
    int n;
    float lr_t, b1, b2, eps;
    int* nonfinite;                            // set to 1 when a gradient is not finite
};
hipError_t launch_adam(const AdamArgs& a, hipStream_t stream);
hipError_t launch_pack(const float* w, const int* table, unsigned short* out, size_t n, hipStream_t stream);
hipError_t launch_gather_f32(const float* w, const int* table, float* out, size_t n, hipStream_t stream);
hipError_t launch_check_finite(const float* g, int n, int* flag, hipStream_t stream);

struct RayGenArgs {
    const float* c2w;       // [B,4,4] row-major (device)
    const float* noise;     // [B,H,W,N] in [0,1) or null -> Philox
    float* o; float* d; float* t;
    int B, H, W, N;
    float focal, near_, far_;
    unsigned long long seed, stream_id;
};
hipError_t launch_raygen(const RayGenArgs& a, hipStream_t stream);

}  // namespace knerf
