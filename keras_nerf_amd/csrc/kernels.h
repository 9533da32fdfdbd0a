// kernels.h -- launch interfaces between knerf_api.hip and the kernel files (internal; the public ABI is include/knerf.h)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

namespace knerf {

// hipFuncSetAttribute once per DEVICE: a function-local `static bool` is neither per device nor safe with two host threads
// (two racing threads may both set the attribute, which is idempotent)
struct AttrOnce {
    std::atomic<unsigned long long> done{0};
    template <class F>
    hipError_t operator()(F&& set) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        const unsigned long long bit = 1ull << (dev & 63);
        if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
        e = set();
        if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
        return e;
    }
};

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
    int shape;              // layout.h fused_shape_id: which trunk shape's instantiation
};
hipError_t launch_mlp_fwd(const FwdArgs& a, bool save, hipStream_t stream);
template <class S> hipError_t launch_mlp_fwd_t(const FwdArgs& a, bool save, hipStream_t stream);

struct BwdArgs {
    const char* stream;     // packed dgrad A-fragments
    const float* raw;       // [R*S,4] forward outputs (rgb after sigmoid, sigma after relu)
    const float* draw;      // [R*S,4] dL/d(rgb,sigma) from the compositing backward
    const char* mask;       // relu masks saved by the forward
    char* dz;               // out: [tiles][kDzBlocks][1 KiB]
    long long n_samples;
    int net;                // 0 coarse / 1 fine: kernel instantiation name only
    int shape;              // layout.h fused_shape_id
    const int* live;        // dead-tile skipping: ascending list of the pass's live 32-sample tiles and its length (device), or null
    const int* n_live;
    long long* stats;       // list mode: [0] += live tiles, [1] += tiles of the pass (knerf_tile_stats); -DKNERF_LIST_GUARD builds: [2] +=
                            //   list entries outside [0, n_tiles) seen (and clamped) by this kernel
};
hipError_t launch_mlp_bwd(const BwdArgs& a, hipStream_t stream);
template <class S> hipError_t launch_mlp_bwd_t(const BwdArgs& a, hipStream_t stream);

struct WgradArgs {
    const char* act;        // [tiles][kActBlocks][1 KiB]
    const char* dz;         // [tiles][kDzBlocks][1 KiB]
    float* grad;            // flat fp32 gradient accumulator of this net (kParamCount)
    float* aux;             // head accumulator of this net (layout.h kAuxCount): destinations >= kAuxBase land here
    const char* fwd_stream; // this net's packed forward A-fragments and bias tiles: the layer_1 job recomputes h0 from them
    const float* bias;
    const char* bwd_stream; // this net's packed dgrad A-fragments and the forward's relu masks: the layer_7 job recomputes dz7
    const char* mask;       //   = mask7 * (H dz_head) from them
    const int* dst;         // concatenated per-job destination tables: [(32*n_it + 1) rows][32*n_ot cols] index or -1
    const void* plan;       // device array of WgradPlan, one per workgroup
    long long n_tiles;
    int n_plan;
    int net;                // 0 coarse / 1 fine: selects the kernel instantiation (a name for profilers), nothing else
    int shape;              // layout.h fused_shape_id
    int n_jobs;             // Shape::kWgradJobs = n_layers + 1
    int aux_base;           // Shape::kAuxBase = the shape's parameter count: destination indices >= it address `aux`
    int job_off[18];        // offset of each job's table inside dst (n_jobs + 1 entries)
    float* partial;         // deterministic mode: [n_plan][ShapeInfo::partial_stride] per-workgroup sums instead of atomics (zero-filled by the caller), or null
    const int* live;        // dead-tile skipping: ascending list of live tiles (relative to act / dz / mask) and its length (device), or null
    const int* n_live;
    long long* stats;       // -DKNERF_LIST_GUARD builds only: [3] += list entries outside [0, n_tiles) seen (and clamped) by this kernel
    int by_range;           // list mode: 0 = the live tiles are dealt out evenly over a job's workgroups; 1 = a workgroup takes the live
                            //   tiles inside the range [n_tiles s/ns, n_tiles (s+1)/ns) it would own without skipping (same sums per
                            //   workgroup as the non-skipping launch: the deterministic mode's bit-exactness check)
};
hipError_t launch_wgrad(const WgradArgs& a, hipStream_t stream);
template <class S> hipError_t launch_wgrad_t(const WgradArgs& a, hipStream_t stream);
// deterministic mode, after launch_wgrad: grad[dst] += sum over the job's workgroups (ascending split) of their partial slabs
// stride = layout.h ShapeInfo::partial_stride (floats per workgroup slab: the shape's largest job table)
hipError_t launch_wgrad_reduce(const WgradArgs& a, const int* job_wg0 /* device: kWgradJobs+1 plan offsets */, int stride, hipStream_t stream);
size_t wgrad_partial_floats(int n_plan, int stride);

// dead-tile skipping: flags[i] (1 = some sample of tile i has a non-zero dL/d(rgb, sigma), written by the compositing kernel) ->
// ascending list of the live tile indices among i in [0, n) with (i % period) < real, and their count; stats[0] += count,
// stats[1] += number of real tiles (running totals for knerf_tile_stats), stats may be null.  One workgroup; ascending list.
hipError_t launch_compact_tiles(const int* flags, int n, int period, int real, int* list, int* count, long long* stats, hipStream_t stream);
// deterministic mode: *loss += partial[0] + partial[1] + ... (fixed order)
hipError_t launch_loss_reduce(const float* partial, int n, float* loss, hipStream_t stream);

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
    int* tile_flags;        // training, S % 32 == 0 only: [R*S/32] 1 = the 32-sample tile has a sample with non-zero draw, 0 = dead; or null
    int* tile_list;         // or (default mode) the live tiles appended to this list, their number added to *tile_count (zero on entry); and,
    int* tile_count;        //   for a coarse pass of a grouped weight-gradient launch, also to tile_list2 / tile_count2 as index + tile_off2
    int* tile_list2;
    int* tile_count2;
    int tile_off2;
    float* loss_partial;    // deterministic mode: per-workgroup loss terms [ceil(R/4)] instead of one atomic per workgroup; or null
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
    const float* lr_t;                         // device: bias-corrected learning rate of this step (optim.hip step_status_kernel)
    float b1, b2, eps;
    const int* nonfinite;                      // device flag of this step's finite check: non-zero = skip the update
};
struct AdamHyper { float lr, b1, b2; };
hipError_t launch_adam(const AdamArgs& a, hipStream_t stream);
hipError_t launch_pack(const float* w, const int* table, unsigned short* out, size_t n, hipStream_t stream);
hipError_t launch_gather_f32(const float* w, const int* table, float* out, size_t n, hipStream_t stream);
hipError_t launch_check_finite(const float* g, int n, int* flag, hipStream_t stream);
// zero-gradient diagnostics (nerf.py:430-451): non-zero counts of g = [coarse | fine] (n floats each) -> counts (device [2]) -> host
// (pinned: [0], [1] the counts, [2] += 1); add_into: dst[i] += src[i]
hipError_t launch_grad_diagnostics(const float* g, int n, unsigned long long* counts, long long* host, hipStream_t stream);
hipError_t launch_add_into(float* dst, const float* src, size_t n, hipStream_t stream);
hipError_t launch_step_status(const int* flag, int* host_status, int* step_state, float* lr_t, const AdamHyper& h, hipStream_t stream);
hipError_t launch_step_set(int step, int* step_state, float* lr_t, const AdamHyper& h, hipStream_t stream);
// collapsed head (layout.h): w = one net's extended weight buffer (kExtParamCount floats).  compose writes the head matrix
// and bias behind the parameters; expand turns the wgrad head job's aux sums into the gradients of features, rgb_features
// and rgb (added to grad, aux zeroed).
// trunk_params = layout.h Shape::kTrunkParams: the offset of the sigma kernel in the flat parameter vector; units = the trunk width
// dir_dim / dir_slots = width of the direction encoding (27) and its slots in the composed head (32); trunk_x / trunk_x_slots = the same for
// xyz_enc in the trunk's output (63 / 64 when the reference concatenates behind the last layer, else 0 / 0): layout.h ShapeInfo
hipError_t launch_head_compose(float* w0, float* w1 /* may be null */, int trunk_params, int units, int trunk_x, int trunk_x_slots, int dir_dim, int dir_slots, hipStream_t stream);   // one workgroup per net
hipError_t launch_head_expand(const float* w0, float* aux0, float* grad0, const float* w1, float* aux1, float* grad1, int trunk_params, int units, int trunk_x, int trunk_x_slots, int dir_dim, int dir_slots,
                              hipStream_t stream);

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
