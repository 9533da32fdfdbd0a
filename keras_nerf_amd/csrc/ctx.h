// ctx.h -- the context object behind the C ABI (include/knerf.h).  Internal: shared by knerf_api.hip and by the
// diagnostics library (debug_api.hip -> libknerf_probe.so), which reads workspace pointers out of it for kernel-level tests.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/knerf.h"
#include "generic.h"
#include "layout.h"

namespace knerf {

struct Net {
    float *w = nullptr, *m = nullptr, *v = nullptr;     // w: kExtParamCount floats on the fused path (parameters + composed head)
    float *g = nullptr, *aux = nullptr;                 // point into knerf_ctx::grads / knerf_ctx::aux
    char *fwd_stream = nullptr, *bwd_stream = nullptr;
    float* bias = nullptr;
};

struct Tables {
    PackTables host;
    std::vector<int32_t> wgrad;        // concatenated per-job destination tables
    std::vector<int32_t> wgrad_off;    // kWgradJobs+1 offsets
    int *d_fwd = nullptr, *d_bias = nullptr, *d_bwd = nullptr, *d_wgrad = nullptr, *d_plan = nullptr;
    int n_plan = 0;
};

// the packing / destination tables of layout.h for one trunk shape, built once per process and shape (host only)
template <class S>
inline void build_wgrad_tables(Tables& t) {
    auto tt = tensor_table<S>();
    t.wgrad.clear(); t.wgrad_off.clear();
    for (int jb = 0; jb < S::kWgradJobs; ++jb) {
        t.wgrad_off.push_back((int32_t)t.wgrad.size());
        const WgradJob J = wgrad_job<S>(jb);
        const int rows = J.n_it * 32 + 1, cols = J.n_ot * 32;     // last row = bias
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c) t.wgrad.push_back(wgrad_dst<S>(tt, jb, r == rows - 1 ? -2 : r, c));
    }
    t.wgrad_off.push_back((int32_t)t.wgrad.size());
}
template <class S>
inline const Tables& host_tables_of() {
    static const Tables t = [] { Tables x; build_fwd<S>(x.host); build_bwd<S>(x.host); build_wgrad_tables<S>(x); return x; }();
    return t;
}
inline const Tables& host_tables(int shape_id = 0) {      // layout.h fused_shape_id
    switch (shape_id) {
#define KNERF_X(I, ...) case I: return host_tables_of<KNERF_SHAPE_T(__VA_ARGS__)>();
        KNERF_FUSED_SHAPES(KNERF_X)
#undef KNERF_X
        default: return host_tables_of<DefaultShape>();
    }
}

}  // namespace knerf

struct knerf_ctx {
    knerf_config cfg;
    std::string err;
    knerf::Net net[2];
    float* grads = nullptr;            // [coarse | fine] flat gradient accumulators: the DP all-reduce operand
    float* aux = nullptr;              // [coarse | fine] head accumulators (layout.h kAuxCount each), expanded into grads per batch
    knerf::Tables tab;
    int step = 0;
    int* d_flag = nullptr;             // device: set by the finite check of the current step
    int* d_step = nullptr;             // device: optimizer steps applied so far (Adam's t - 1 of the next step)
    float* d_lr_t = nullptr;           // device: bias-corrected learning rate of the next step, derived from d_step on the device
    int* h_status = nullptr;           // pinned host: [0] number of skipped (non-finite) steps so far, written by the device
    int skipped_seen = 0;
    int n_cu = 256;
    // general-shape MLP path (generic.h): used when the fused kernels do not cover the config's MLP shape (layout.h KNERF_FUSED_SHAPES)
    bool generic = false;
    bool mlp_only = false;              // created with KNERF_FLAG_ENCODED_WIDTHS: a stand-alone NeRFMLP (weights + knerf_mlp_call)
    int shape = 0;                      // fused path: layout.h fused_shape_id of the trunk
    knerf::ShapeInfo si = knerf::shape_info(0);
    int n_params = knerf::kParamCount;
    knerf::gen::Plan gplan;
    knerf::gen::Workspace gws;
    knerf::gen::NetDev gnet[2];
    // knerf_mlp_call (NeRFMLP.__call__ on encoded inputs): own plan / packed weights / workspace, also on a fused-path context
    knerf::gen::Plan call_plan; bool call_plan_ok = false;
    knerf::gen::Workspace call_ws;
    knerf::gen::NetDev call_net;
    float* call_raw = nullptr;
    // run-time options (knerf_set_option)
    bool deterministic = false;         // per-workgroup partial sums + ordered second pass instead of fp32 atomics (wgrad, loss)
    bool skip_dead = true;              // dgrad / wgrad skip 32-sample tiles whose dL/d(rgb, sigma) is exactly zero (exact; +0.3 % when nothing is dead)
    int wgrad_group_max = 4;            // chunks per coarse weight-gradient launch of knerf_train_batch (1 = one launch per chunk)
    double wgrad_group_gb = 40.0;       // memory budget of those grouped workspaces
    int merge_rays = 4096;              // knerf_train_batch: consecutive chunks share launches of up to this many rays (0: off)
    int merge_render_rays = 65536;      // knerf_render_batch on the fused path (no saved tensors: 5 KB of workspace per ray); general-shape path: merge_rays
    int wgrad_cost[17] = {};            // workgroups per job ~ cost (build_wgrad_plan); filled from the job kinds at creation
    bool grad_diag = false;             // zero-gradient diagnostics of the last chunk (nerf.py:430-451; NeRF.compile(run_eagerly=True))
    float* diag_tmp = nullptr;          // the earlier chunks' accumulated gradient while the last chunk runs alone
    unsigned long long* d_diag = nullptr;
    long long* h_diag = nullptr;        // pinned: [0] coarse, [1] fine non-zero count of the last chunk's gradient, [2] steps published
    bool plan_dirty = false;
    int group_cache_rays = 0, group_cache_chunks = 0, group_cache = 0;   // wgrad_group_for memo (hipMemGetInfo is a driver call)
    // (ray_chunks, chunk count) whose MERGED launches did not fit in memory: later calls start from the caller's own chunks instead
    // of failing the same allocation again (train / render); how often a merge was given up (option "merge_fallbacks", read-only)
    int merge_fail_rays = 0, merge_fail_chunks = 0, rmerge_fail_rays = 0, rmerge_fail_chunks = 0, merge_fallbacks = 0;
    double ws_limit_gb = 0.0;           // option "workspace_limit_gb" (tests): a workspace request above it meets a REAL failing hipMalloc
    // workspaces (grow-only).  Inference buffers (raw, w_c, t_f, img_tmp) follow the largest chunk seen by any call; the training
    // buffers (draw, act, mask, dz, tile lists) follow the largest TRAINING chunk and group only, so that rendering with a larger
    // ray_chunks does not re-size them.
    int ws_rays = 0; bool ws_train = false;
    int ws_train_rays = 0;
    int ws_group = 1;                   // training workspaces hold this many chunks (knerf_train_batch: one wgrad launch per group)
    float *raw = nullptr, *draw = nullptr, *w_c = nullptr, *t_f = nullptr, *img_tmp = nullptr, *loss_tmp = nullptr;
    char *act = nullptr, *mask = nullptr, *dz = nullptr;
    size_t act_bytes = 0, mask_bytes = 0, dz_bytes = 0, raw_bytes = 0, draw_bytes = 0;
    // dead-tile skipping: per-tile flags of the training workspaces (composite.hip), the compacted list and its length, running
    // totals (live, all) of the tiles seen by the dgrad launches
    int *tile_flags = nullptr, *tile_list = nullptr, *tile_list_g = nullptr, *tile_count = nullptr;   // tile_list_g: the list of a group of coarse passes
    // tile_count: a ring of counters, one per list; the whole ring is zeroed by ONE memset when a call takes its first counter (and
    // again should a call need more than the ring holds: its earlier consumers are already enqueued, the stream orders the memset)
    // Sized per call (ensure_tile_counters: a train_batch of C chunks in groups of G needs 2 C + ceil(C / G)), grow-only, so that a
    // call never wraps while a group's counter is still being appended to.
    int tile_counters = 4096;
    int tile_counter_next = 0;
    long long* tile_stats = nullptr;    // [net][4]: live, total (dgrad launches of that net's passes), two guard counters
    size_t ws_tiles = 0;
    // deterministic mode: per-workgroup weight-gradient slabs, per-workgroup loss terms, plan offsets of the jobs
    float *partial = nullptr, *loss_partial = nullptr;
    int partial_plan = 0;
    int* d_job_wg0 = nullptr;
    // optional per-kernel timing with HIP events on the caller's stream (knerf_profile_*)
    bool prof_on = false;
    struct ProfRec { int id; hipEvent_t e0, e1; };
    std::vector<ProfRec> prof;
};
