// knerf_api.hip -- extern "C" entry points of libknerf_hip.so (see include/knerf.h for the contract and the reference
// lines each one replaces).  Host-side only: owns device memory, sequences the kernels on the caller's stream.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/knerf.h"
#include "chain.h"
#include "ctx.h"
#include "kernels.h"
#include "generic.h"
#include "layout.h"

using namespace knerf;

#ifdef KNERF_N_SHAPE_SLICES
static_assert(KNERF_N_SHAPE_SLICES == kNumFusedShapes, "keras_nerf_amd/build.py N_SHAPE_SLICES must equal the number of entries of layout.h KNERF_FUSED_SHAPES");
#endif

namespace {

std::string g_create_error;

// Workgroups per wgrad job in proportion to the job's measured cost per sample tile (cycles per loop iteration from the
// s_memtime stamps of a -DKNERF_WGRAD_STAMPS build, tools/kbench.py): the streaming jobs are HBM-bound, the small ones
// (layer_0, head) latency-bound, so bytes alone mis-balance them.  About one workgroup per CU in total.
// knerf_set_option(ctx, "wgrad_cost<j>", c) overrides an entry (tuning sweeps in one gpurun call: tools/tune_costs.py).
// job_wg0[j] = index of job j's first workgroup (kWgradJobs + 1 entries): the deterministic mode's second pass walks them in order.
std::vector<int32_t> build_wgrad_plan(int n_wg, const int* cost, int n_jobs, std::vector<int32_t>& job_wg0) {
    // r02 stamps (gpurun_out/r2b/stamps.json): 1180 / 2080 / 2590 / 1530 cycles per tile for layer_0 / 256x256 / layer_5 / head;
    // head swept 100..200 (1.40 / 1.20 / 1.13 / 1.12 ms per fine launch at 100 / 130 / 160 / 200)
    // layer_1 recomputes h0 (wgrad_l1_recompute, 20 KiB tiles but 22 MFMAs and an LDS exchange per tile): swept 160 / 204 / 240 /
    // 280 -> 1.27 / 1.08 / 1.056 / 1.064 ms per fine launch
    // layer_7 recomputes dz7 (wgrad_l7_recompute, 18 KiB tiles): 204 -> 240; with the copies issued behind the first half's MFMAs
    // the plain jobs gained most, tools/tune_costs.py (coordinate search on the box) moved the others up: coarse + fine launch
    // 1.439 -> 1.382 ms
    // (the defaults are set per job KIND at creation, default_wgrad_cost: 128 first layer, 264 layer_1, 204 plain, 267 concat, 240
    // last layer, 193 head -- i.e. 128, 264, 204, 204, 204, 267, 204, 240, 193 for the default shape)
    int total = 0;
    for (int j = 0; j < n_jobs; ++j) total += cost[j] > 0 ? cost[j] : 0;
    std::vector<int32_t> plan;
    job_wg0.assign(n_jobs + 1, 0);
    for (int j = 0; j < n_jobs; ++j) {
        job_wg0[j] = (int32_t)plan.size() / 4;
        if (cost[j] <= 0) continue;          // a sweep may switch a job off (its gradient is then missing: timing experiments only)
        int ns = (cost[j] * (n_wg - 4) + total / 2) / total; if (ns < 1) ns = 1;
        for (int s = 0; s < ns; ++s) { plan.push_back(j); plan.push_back(s); plan.push_back(ns); plan.push_back(0); }
    }
    job_wg0[n_jobs] = (int32_t)plan.size() / 4;
    return plan;
}
int default_wgrad_cost(int kind) { const int c[6] = {128, 264, 204, 267, 240, 193}; return c[kind >= 0 && kind < 6 ? kind : 2]; }

}  // namespace

namespace {

int fail(knerf_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}
#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(ctx, KNERF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));        \
    } while (0)

size_t stream_bytes(int blocks) { return (size_t)((blocks + kPageBlocks - 1) / kPageBlocks * kPageBlocks + kTailPages * kPageBlocks) * 1024; }

// compose = false: the caller has already composed the heads of both nets in one launch (knerf_apply_adam)
int repack(knerf_ctx* ctx, int n, hipStream_t s, bool compose = true) {
    Net& N = ctx->net[n];
    if (ctx->generic) {
        HIPCHK(gen::pack_weights(ctx->gplan, N.w, ctx->gnet[n], s));
        return KNERF_OK;
    }
    if (compose) HIPCHK(launch_head_compose(N.w, nullptr, ctx->si.trunk_params, ctx->si.units, ctx->si.trunk_x, ctx->si.trunk_x_slots, ctx->si.dir_dim, ctx->si.dir_slots, s));   // the composed head behind the parameters (layout.h), then the bf16 streams
    HIPCHK(launch_pack(N.w, ctx->tab.d_fwd, reinterpret_cast<unsigned short*>(N.fwd_stream), (size_t)ctx->si.fwd_blocks * 512, s));
    HIPCHK(launch_pack(N.w, ctx->tab.d_bwd, reinterpret_cast<unsigned short*>(N.bwd_stream), (size_t)ctx->si.bwd_blocks * 512, s));
    HIPCHK(launch_gather_f32(N.w, ctx->tab.d_bias, N.bias, (size_t)ctx->si.fwd_bias_tiles * 32, s));
    return KNERF_OK;
}

size_t tiles_for(long long n_samples) {
    size_t t = (size_t)((n_samples + kTile - 1) / kTile);
    constexpr size_t q = kWaves > kSavedGroup ? kWaves : kSavedGroup;
    return (t + q - 1) / q * q;                  // whole workgroups (and whole layout groups, layout.h)
}

// (re)build the wgrad plan from ctx->wgrad_cost and upload it with the jobs' workgroup offsets
int upload_plan(knerf_ctx* ctx) {
    std::vector<int32_t> job_wg0;
    std::vector<int32_t> plan = build_wgrad_plan(ctx->n_cu, ctx->wgrad_cost, ctx->si.n_jobs, job_wg0);
    if (ctx->tab.d_plan) { (void)hipFree(ctx->tab.d_plan); ctx->tab.d_plan = nullptr; }
    if (ctx->d_job_wg0) { (void)hipFree(ctx->d_job_wg0); ctx->d_job_wg0 = nullptr; }
    HIPCHK(hipMalloc(&ctx->tab.d_plan, plan.size() * sizeof(int32_t)));
    HIPCHK(hipMemcpy(ctx->tab.d_plan, plan.data(), plan.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&ctx->d_job_wg0, job_wg0.size() * sizeof(int32_t)));
    HIPCHK(hipMemcpy(ctx->d_job_wg0, job_wg0.data(), job_wg0.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    ctx->tab.n_plan = (int)plan.size() / 4;
    ctx->plan_dirty = false;
    return KNERF_OK;
}

template <class T> void free_dev(T*& p) { if (p) { (void)hipFree(p); p = nullptr; } }

// grow-only workspaces; the zero-fills are enqueued on the caller's stream `s`, the one the consuming kernels run on
// (hipMalloc / hipFree themselves synchronise the device).
// Inference buffers (raw, w_c, t_f, img_tmp) follow the largest chunk of ANY call; the training workspaces of the fused path
// (draw, act / mask / dz, tile flags and lists) follow the largest TRAINING chunk and group: `group` coarse-pass regions followed
// by one fine-pass region, so that one weight-gradient launch covers the coarse passes of a group of chunks
// (launch_wgrad_tiles below).  A render with a larger ray_chunks therefore never re-sizes the training regions.
// option "workspace_limit_gb": the out-of-memory paths of the callers (a merged launch or a wgrad group that does not fit) are taken
// by a REAL failed hipMalloc -- same error code, same state of the runtime's last-error word -- without filling 288 GB first
int over_limit(knerf_ctx* ctx, double bytes) {
    if (ctx->ws_limit_gb <= 0 || bytes <= ctx->ws_limit_gb * 1e9) return KNERF_OK;
    void* never = nullptr;
    HIPCHK(hipMalloc(&never, (size_t)1 << 50));
    (void)hipFree(never);
    return KNERF_OK;
}

int ensure_ws_impl(knerf_ctx* ctx, int n_rays, bool train, hipStream_t s, int group) {
    const int Na = ctx->cfg.n_coarse + ctx->cfg.n_fine;
    if (ctx->generic) {          // general-shape path: one size for everything (its activations are forward buffers too)
        if (n_rays <= ctx->ws_rays && (!train || ctx->ws_train)) return KNERF_OK;
        const int R = n_rays > ctx->ws_rays ? n_rays : ctx->ws_rays;
        train = train || ctx->ws_train;
        {
            const double mp = (double)gen::padded_rows((long long)R * Na);
            if (int r = over_limit(ctx, (double)R * Na * 20 + mp * 256 + 2.0 * ctx->gplan.act_elems_per_row * mp +
                                            (train ? 2.0 * ctx->gplan.dz_elems_per_row * mp + (double)R * Na * 16 : 0.0))) return r;
        }
        HIPCHK(hipStreamSynchronize(s));           // nothing enqueued earlier may still use the buffers that are freed below
        free_dev(ctx->raw); free_dev(ctx->draw); free_dev(ctx->w_c); free_dev(ctx->t_f); free_dev(ctx->img_tmp);
        ctx->ws_rays = 0;
        const size_t ns = (size_t)R * Na;
        ctx->raw_bytes = ns * 4 * sizeof(float);
        HIPCHK(hipMalloc(&ctx->raw, ctx->raw_bytes));
        HIPCHK(hipMalloc(&ctx->w_c, (size_t)R * ctx->cfg.n_coarse * sizeof(float)));
        HIPCHK(hipMalloc(&ctx->t_f, ns * sizeof(float)));
        HIPCHK(hipMalloc(&ctx->img_tmp, (size_t)R * 8 * sizeof(float)));
        gen::Workspace& g = ctx->gws;
        free_dev(g.act); free_dev(g.dz); free_dev(g.zs); free_dev(g.zc); free_dev(g.mask);
        g.mp = gen::padded_rows((long long)ns);
        const size_t ab = ctx->gplan.act_elems_per_row * g.mp * sizeof(unsigned short), zb = ctx->gplan.dz_elems_per_row * g.mp * sizeof(unsigned short);
        HIPCHK(hipMalloc(&g.act, ab));
        HIPCHK(hipMemsetAsync(g.act, 0, ab, s));
        HIPCHK(hipMalloc(&g.zs, g.mp * 32 * sizeof(float)));
        HIPCHK(hipMalloc(&g.zc, g.mp * 32 * sizeof(float)));
        if (train) {
            HIPCHK(hipMalloc(&g.dz, zb));
            HIPCHK(hipMemsetAsync(g.dz, 0, zb, s));
            HIPCHK(hipMalloc(&g.mask, gen::relu_bits_bytes(ctx->gplan, g.mp)));
            HIPCHK(hipMalloc(&ctx->draw, ctx->raw_bytes));
            ctx->draw_bytes = ctx->raw_bytes;
            free_dev(ctx->loss_partial);
            HIPCHK(hipMalloc(&ctx->loss_partial, ((size_t)R + 3) / 4 * sizeof(float)));
            // dead-tile skipping: per-tile flags (deterministic mode) and the list of live tiles of the current pass
            free_dev(ctx->tile_flags); free_dev(ctx->tile_list);
            const size_t tiles = tiles_for((long long)ns);
            HIPCHK(hipMalloc(&ctx->tile_flags, tiles * sizeof(int)));
            HIPCHK(hipMemsetAsync(ctx->tile_flags, 0, tiles * sizeof(int), s));
            HIPCHK(hipMalloc(&ctx->tile_list, tiles * sizeof(int)));
            ctx->ws_tiles = tiles;
        }
        ctx->ws_rays = R; ctx->ws_train = train; ctx->ws_train_rays = train ? R : 0; ctx->ws_group = 1;
        return KNERF_OK;
    }
    const bool grow_base = n_rays > ctx->ws_rays;
    const bool grow_train = train && (!ctx->ws_train || n_rays > ctx->ws_train_rays || group > ctx->ws_group);
    if (!grow_base && !grow_train) return KNERF_OK;
    if (ctx->ws_limit_gb > 0) {
        const int Rt = n_rays > ctx->ws_train_rays ? n_rays : ctx->ws_train_rays;
        const size_t tl = group == 1 ? tiles_for((long long)Rt * Na) : (size_t)group * tiles_for((long long)Rt * ctx->cfg.n_coarse) + tiles_for((long long)Rt * Na);
        const double base_b = grow_base ? (double)n_rays * Na * 20 + (double)n_rays * (ctx->cfg.n_coarse * 4 + 32) : 0.0;
        const double train_b = grow_train ? (double)Rt * Na * 16 + (double)saved_region_bytes(tl, ctx->si.act_blocks) +
                                                (double)saved_region_bytes(tl, ctx->si.mask_blocks) + (double)saved_region_bytes(tl, ctx->si.dz_blocks) : 0.0;
        if (int r = over_limit(ctx, base_b + train_b)) return r;
    }
    HIPCHK(hipStreamSynchronize(s));               // nothing enqueued earlier may still use the buffers that are freed below
    if (grow_base) {
        free_dev(ctx->raw); free_dev(ctx->w_c); free_dev(ctx->t_f); free_dev(ctx->img_tmp);
        ctx->ws_rays = 0;
        const size_t ns = (size_t)n_rays * Na;
        ctx->raw_bytes = ns * 4 * sizeof(float);
        HIPCHK(hipMalloc(&ctx->raw, ctx->raw_bytes));
        HIPCHK(hipMalloc(&ctx->w_c, (size_t)n_rays * ctx->cfg.n_coarse * sizeof(float)));
        HIPCHK(hipMalloc(&ctx->t_f, ns * sizeof(float)));
        HIPCHK(hipMalloc(&ctx->img_tmp, (size_t)n_rays * 8 * sizeof(float)));
        ctx->ws_rays = n_rays;
    }
    if (grow_train) {
        const int R = n_rays > ctx->ws_train_rays ? n_rays : ctx->ws_train_rays;
        if (group < ctx->ws_group && n_rays <= ctx->ws_train_rays) group = ctx->ws_group;   // a larger chunk size starts from the group asked for: group x size is what costs memory
        free_dev(ctx->draw); free_dev(ctx->act); free_dev(ctx->mask); free_dev(ctx->dz);
        free_dev(ctx->tile_flags); free_dev(ctx->tile_list); free_dev(ctx->tile_list_g); free_dev(ctx->loss_partial);
        ctx->ws_train = false; ctx->ws_train_rays = 0; ctx->ws_group = 1; ctx->group_cache = 0;
        const size_t ns = (size_t)R * Na;
        // group 1: the coarse and the fine pass of a chunk share one region (each pass's weight gradients follow it at once)
        const size_t tiles = group == 1 ? tiles_for((long long)ns) : (size_t)group * tiles_for((long long)R * ctx->cfg.n_coarse) + tiles_for((long long)ns);
        ctx->act_bytes = saved_region_bytes(tiles, ctx->si.act_blocks); ctx->mask_bytes = saved_region_bytes(tiles, ctx->si.mask_blocks);
        ctx->dz_bytes = saved_region_bytes(tiles, ctx->si.dz_blocks);
        ctx->draw_bytes = ns * 4 * sizeof(float);
        HIPCHK(hipMalloc(&ctx->draw, ctx->draw_bytes));
        HIPCHK(hipMalloc(&ctx->act, ctx->act_bytes));
        HIPCHK(hipMalloc(&ctx->mask, ctx->mask_bytes));
        HIPCHK(hipMalloc(&ctx->dz, ctx->dz_bytes));
        HIPCHK(hipMemsetAsync(ctx->dz, 0, ctx->dz_bytes, s));     // block kDzHead+1 is never written and must read 0
        HIPCHK(hipMemsetAsync(ctx->act, 0, ctx->act_bytes, s));
        HIPCHK(hipMalloc(&ctx->tile_flags, tiles * sizeof(int)));
        HIPCHK(hipMemsetAsync(ctx->tile_flags, 0, tiles * sizeof(int), s));
        HIPCHK(hipMalloc(&ctx->tile_list, tiles * sizeof(int)));
        if (group > 1) HIPCHK(hipMalloc(&ctx->tile_list_g, (size_t)group * tiles_for((long long)R * ctx->cfg.n_coarse) * sizeof(int)));
        HIPCHK(hipMalloc(&ctx->loss_partial, ((size_t)R + 3) / 4 * sizeof(float)));
        ctx->ws_tiles = tiles;
        ctx->ws_train = true; ctx->ws_train_rays = R; ctx->ws_group = group;
    }
    return KNERF_OK;
}

// A failed hipMalloc stays behind as the runtime's LAST ERROR: every launch helper ends in `return hipGetLastError()`, so the first
// kernel after a failed allocation would report hipErrorOutOfMemory although the caller's retry at a smaller size succeeded
// (ADVICE r05).  The word is read -- which clears it -- here, where the failure is already on its way to the caller as a status.
int ensure_ws(knerf_ctx* ctx, int n_rays, bool train, hipStream_t s, int group = 1) {
    const int r = ensure_ws_impl(ctx, n_rays, train, s, group);
    if (r) (void)hipGetLastError();
    return r;
}

// kernel classes reported by knerf_profile_read
enum ProfId { P_FWD_C = 0, P_FWD_F, P_COMPOSITE, P_SAMPLE, P_BWD_C, P_BWD_F, P_WGRAD_C, P_WGRAD_F, P_ADAM, P_COUNT };

struct ProfScope {
    knerf_ctx* c; hipStream_t s; int idx = -1;
    ProfScope(knerf_ctx* ctx, hipStream_t st, int id) : c(ctx), s(st) {
        if (!c->prof_on) return;
        knerf_ctx::ProfRec r{id, nullptr, nullptr};
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
        (void)hipEventRecord(r.e0, s);
        c->prof.push_back(r); idx = (int)c->prof.size() - 1;
    }
    ~ProfScope() { if (idx >= 0) (void)hipEventRecord(c->prof[idx].e1, s); }
};

// a context created with KNERF_FLAG_ENCODED_WIDTHS is a stand-alone NeRFMLP: it has no ray encodings, samplers or workspaces
int check_rays(knerf_ctx* ctx, const char* what) {
    if (!ctx) return KNERF_ERR_INVALID;
    if (ctx->mlp_only) return fail(ctx, KNERF_ERR_INVALID, std::string(what) + ": this context was created with KNERF_FLAG_ENCODED_WIDTHS (NeRFMLP.__call__ only)");
    return KNERF_OK;
}

int check_net(knerf_ctx* ctx, int net) {
    if (!ctx) return KNERF_ERR_INVALID;
    if (net != KNERF_COARSE && net != KNERF_FINE) return fail(ctx, KNERF_ERR_INVALID, "net must be 0 (coarse) or 1 (fine)");
    return KNERF_OK;
}

// dead-tile skipping applies when 32-sample tiles do not straddle rays in either pass (both MLP paths since round 5: the
// general-shape kernels walk the same list of live tiles, generic.hip)
bool skipping(const knerf_ctx* ctx) {
    return ctx->skip_dead && ctx->cfg.n_coarse % kTile == 0 && (ctx->cfg.n_coarse + ctx->cfg.n_fine) % kTile == 0;
}

// A fresh (zero) counter for one list of live tiles.  The ring is zeroed by one memset when a call takes its first counter, and again
// if a call runs through the whole ring (the consumers of its earlier counters are enqueued by then; the stream orders the memset).
int next_tile_counter(knerf_ctx* ctx, hipStream_t s, int** count) {
    if (ctx->tile_counter_next >= ctx->tile_counters) ctx->tile_counter_next = 0;     // not reached by the entry points: they size the ring first
    if (ctx->tile_counter_next == 0) HIPCHK(hipMemsetAsync(ctx->tile_count, 0, (size_t)ctx->tile_counters * sizeof(int), s));
    *count = ctx->tile_count + ctx->tile_counter_next++;
    return KNERF_OK;
}

// The ring holds every counter one call takes (grow-only): a wrap inside a call would zero a group's counter that later coarse
// passes still append to, and the grouped weight-gradient launch would then see a short list.
int ensure_tile_counters(knerf_ctx* ctx, hipStream_t s, long long need) {
    if (need <= ctx->tile_counters) return KNERF_OK;
    HIPCHK(hipStreamSynchronize(s));               // consumers of the old ring's counters have finished
    free_dev(ctx->tile_count);
    long long n = ctx->tile_counters;
    while (n < need) n *= 2;
    HIPCHK(hipMalloc(&ctx->tile_count, (size_t)n * sizeof(int)));
    ctx->tile_counters = (int)n; ctx->tile_counter_next = 0;
    return KNERF_OK;
}

// Deterministic mode: tile flags -> ASCENDING list (its per-workgroup ranges need the order) by the one-workgroup compaction kernel.
int compact_tiles(knerf_ctx* ctx, hipStream_t s, const int* flags, int n, int period, int real, int* list, int** count) {
    if (int r = next_tile_counter(ctx, s, count)) return r;
    HIPCHK(launch_compact_tiles(flags, n, period, real, list, *count, nullptr, s));
    return KNERF_OK;
}

// weight gradients of `net` over n_tiles sample tiles starting at tile `tile0` of the act / mask / dz workspaces.  live: null, or the
// list of live tiles (indices relative to tile0) with its device counter.
int launch_wgrad_tiles(knerf_ctx* ctx, hipStream_t s, int net, size_t tile0, size_t n_tiles, const int* live, const int* live_count) {
    WgradArgs wa{};
    wa.act = ctx->act + saved_tile_off(tile0, ctx->si.act_blocks); wa.dz = ctx->dz + saved_tile_off(tile0, ctx->si.dz_blocks);
    wa.mask = ctx->mask + saved_tile_off(tile0, ctx->si.mask_blocks);
    wa.shape = ctx->shape; wa.n_jobs = ctx->si.n_jobs; wa.aux_base = ctx->si.param_count;
    wa.grad = ctx->net[net].g; wa.aux = ctx->net[net].aux; wa.dst = ctx->tab.d_wgrad;
    wa.fwd_stream = ctx->net[net].fwd_stream; wa.bias = ctx->net[net].bias; wa.bwd_stream = ctx->net[net].bwd_stream;
    wa.n_tiles = (long long)n_tiles;
    wa.plan = ctx->tab.d_plan; wa.n_plan = ctx->tab.n_plan; wa.net = net == KNERF_COARSE ? 0 : 1;
    for (int j = 0; j <= ctx->si.n_jobs; ++j) wa.job_off[j] = ctx->tab.wgrad_off[j];
    if (live) { wa.live = live; wa.n_live = live_count; wa.by_range = ctx->deterministic ? 1 : 0; }
    wa.stats = ctx->tile_stats + 4 * (net == KNERF_COARSE ? 0 : 1);
    if (ctx->deterministic) {
        // per-workgroup slabs (zeroed: a workgroup without tiles writes nothing) + the ordered second pass
        if (!ctx->partial || ctx->partial_plan != ctx->tab.n_plan) {
            HIPCHK(hipStreamSynchronize(s));
            free_dev(ctx->partial);
            HIPCHK(hipMalloc(&ctx->partial, wgrad_partial_floats(ctx->tab.n_plan, ctx->si.partial_stride) * sizeof(float)));
            ctx->partial_plan = ctx->tab.n_plan;
        }
        HIPCHK(hipMemsetAsync(ctx->partial, 0, wgrad_partial_floats(ctx->tab.n_plan, ctx->si.partial_stride) * sizeof(float), s));
        wa.partial = ctx->partial;
    }
    ProfScope ps(ctx, s, net == KNERF_COARSE ? P_WGRAD_C : P_WGRAD_F);
    HIPCHK(launch_wgrad(wa, s));
    if (ctx->deterministic) HIPCHK(launch_wgrad_reduce(wa, ctx->d_job_wg0, ctx->si.partial_stride, s));
    return KNERF_OK;
}

// tile0: first tile of this pass in the training workspaces; wgrad_now: launch the weight-gradient kernel for just this pass
// (false: the caller launches it later over several passes, launch_wgrad_tiles)
// group_count: (default-mode skipping, coarse pass of a grouped launch) the group's counter: the pass's live tiles are appended to
// ctx->tile_list_g as well, as indices relative to the group's first region (+ tile0)
int run_pass(knerf_ctx* ctx, hipStream_t s, int net, const float* o, const float* d, const float* t, int R, int S,
             float* image, float* depth, float* weights, const float* target, float inv_chunks, float* loss,
             size_t tile0 = 0, bool wgrad_now = true, int* group_count = nullptr) {
    const bool train = target != nullptr;
    FwdArgs fa{};
    fa.stream = ctx->net[net].fwd_stream; fa.bias = ctx->net[net].bias;
    fa.o = o; fa.d = d; fa.t = t; fa.raw = ctx->raw;
    fa.act = ctx->act ? ctx->act + saved_tile_off(tile0, ctx->si.act_blocks) : nullptr;
    fa.mask = ctx->mask ? ctx->mask + saved_tile_off(tile0, ctx->si.mask_blocks) : nullptr;
    fa.n_samples = (long long)R * S; fa.S = S; fa.net = net == KNERF_COARSE ? 0 : 1; fa.shape = ctx->shape;
    if (ctx->generic) {
        ProfScope ps(ctx, s, net == KNERF_COARSE ? P_FWD_C : P_FWD_F);
        HIPCHK(gen::forward(ctx->gplan, ctx->gws, ctx->gnet[net], ctx->net[net].w, o, d, t, fa.n_samples, S, ctx->raw, s));
    } else {
        ProfScope ps(ctx, s, net == KNERF_COARSE ? P_FWD_C : P_FWD_F);
        HIPCHK(launch_mlp_fwd(fa, train, s));
    }
    CompositeArgs ca{};
    ca.raw = ctx->raw; ca.t = t; ca.target = target; ca.image = image; ca.depth = depth; ca.weights = weights;
    ca.draw = train ? ctx->draw : nullptr; ca.loss = loss; ca.R = R; ca.S = S; ca.white = ctx->cfg.white_background;
    ca.grad_scale = 2.0f / (3.0f * (float)R) * inv_chunks;
    ca.loss_scale = inv_chunks / (3.0f * (float)R);
    const bool skip = train && skipping(ctx);
    const size_t n_tiles = tiles_for(fa.n_samples);
    int* live_count = nullptr;
    if (skip && ctx->deterministic) {
        ca.tile_flags = ctx->tile_flags + tile0;                     // flags now, an ascending list from the compaction kernel below
    } else if (skip) {
        if (int r = next_tile_counter(ctx, s, &live_count)) return r;     // the compositing kernel appends the live tiles itself
        ca.tile_list = ctx->tile_list; ca.tile_count = live_count;
        if (group_count) { ca.tile_list2 = ctx->tile_list_g; ca.tile_count2 = group_count; ca.tile_off2 = (int)tile0; }
    }
    if (train && ctx->deterministic) ca.loss_partial = ctx->loss_partial;
    {
        ProfScope ps(ctx, s, P_COMPOSITE);
        HIPCHK(launch_composite(ca, s));
        if (ca.loss_partial) HIPCHK(launch_loss_reduce(ca.loss_partial, (R + 3) / 4, loss, s));
        // deterministic mode: the pass's live tiles (indices relative to tile0) in ascending order; the padding tiles behind the
        // last real one count as dead
        if (ca.tile_flags) { if (int r = compact_tiles(ctx, s, ca.tile_flags, (int)n_tiles, (int)n_tiles, (int)(fa.n_samples / kTile), ctx->tile_list, &live_count)) return r; }
    }
    if (train && ctx->generic) {
        ProfScope ps(ctx, s, net == KNERF_COARSE ? P_BWD_C : P_BWD_F);
        if (ctx->deterministic && !ctx->partial) {       // general-shape path: the slab arena of generic.hip (the fused path's lives in launch_wgrad_tiles)
            HIPCHK(hipStreamSynchronize(s));
            HIPCHK(hipMalloc(&ctx->partial, gen::wgrad_partial_floats(ctx->gplan) * sizeof(float)));
        }
        HIPCHK(gen::backward(ctx->gplan, ctx->gws, ctx->gnet[net], ctx->raw, ctx->draw, fa.n_samples, ctx->net[net].g, s,
                             ctx->deterministic ? ctx->partial : nullptr, skip ? ctx->tile_list : nullptr, skip ? live_count : nullptr,
                             skip ? ctx->tile_stats + 4 * fa.net : nullptr));
    } else if (train) {
        BwdArgs ba{};
        ba.stream = ctx->net[net].bwd_stream; ba.raw = ctx->raw; ba.draw = ctx->draw; ba.mask = fa.mask; ba.dz = ctx->dz + saved_tile_off(tile0, ctx->si.dz_blocks);
        ba.n_samples = fa.n_samples; ba.net = fa.net; ba.shape = ctx->shape;
        if (skip) { ba.live = ctx->tile_list; ba.n_live = live_count; ba.stats = ctx->tile_stats + 4 * fa.net; }
        { ProfScope ps(ctx, s, net == KNERF_COARSE ? P_BWD_C : P_BWD_F); HIPCHK(launch_mlp_bwd(ba, s)); }
        if (wgrad_now) { if (int r = launch_wgrad_tiles(ctx, s, net, tile0, n_tiles, skip ? ctx->tile_list : nullptr, live_count)) return r; }
    }
    return KNERF_OK;
}

// the wgrad head jobs leave sums in ctx->aux; this turns them into the gradients of features / rgb_features / rgb of both
// nets (optim.hip head_expand).  Linear in the sums, so once per batch of chunks is the same as once per chunk.
int expand_head_grads(knerf_ctx* ctx, hipStream_t s) {
    if (ctx->generic) {
        for (int n = 0; n < 2; ++n) HIPCHK(gen::expand_head(ctx->gplan, ctx->gnet[n], ctx->net[n].w, ctx->net[n].g, s));
        return KNERF_OK;
    }
    HIPCHK(launch_head_expand(ctx->net[0].w, ctx->net[0].aux, ctx->net[0].g, ctx->net[1].w, ctx->net[1].aux, ctx->net[1].g, ctx->si.trunk_params, ctx->si.units, ctx->si.trunk_x, ctx->si.trunk_x_slots, ctx->si.dir_dim, ctx->si.dir_slots, s));
    return KNERF_OK;
}

// One chunk through both nets.  With group > 1 the chunk's COARSE pass occupies slot `slot` of the coarse regions (tiles
// [slot tc, (slot+1) tc)) and its weight gradients are left to one launch over the whole group (knerf_train_batch); the fine pass
// uses the single fine region behind them and is followed by its own wgrad launch as before.  Why only the coarse pass: a wgrad
// launch carries ~44 us of flush, ramp and tail, 12 % of a coarse launch (8192 tiles at ray_chunks 4096) but 4 % of a fine one, and
// launches of 4 x 24576 tiles ran 4.7 % SLOWER per tile (41.4 vs 39.6 ns; power-limited clocks over a 4 ms kernel) -- measured with
// both nets grouped: coarse 2.86 -> 2.57 ms, fine 7.8 -> 8.15 ms per step.
int train_chunk_impl(knerf_ctx* ctx, hipStream_t s, const float* o, const float* d, const float* t, const float* target,
                     const float* u, uint64_t seed, uint64_t ray_offset, int n_rays, float inv_chunks, float* loss,
                     float* c_image, float* f_image, int slot = 0, int group = 1, int* group_count = nullptr) {
    const int Nc = ctx->cfg.n_coarse, Na = Nc + ctx->cfg.n_fine;
    float* ci = c_image ? c_image : ctx->img_tmp;
    float* fi = f_image ? f_image : ctx->img_tmp + (size_t)n_rays * 4;
    float* ls = loss ? loss : ctx->loss_tmp;
    const size_t tc = ctx->generic ? 0 : tiles_for((long long)n_rays * Nc);
    const size_t tile0_c = group == 1 ? 0 : slot * tc, tile0_f = group == 1 ? 0 : group * tc;
    if (int r = run_pass(ctx, s, KNERF_COARSE, o, d, t, n_rays, Nc, ci, nullptr, ctx->w_c, target, inv_chunks, ls, tile0_c, group == 1, group_count)) return r;
    if (int r = knerf_sample_fine(ctx, s, t, ctx->w_c, u, seed, 0, ray_offset, n_rays, ctx->t_f)) return r;
    return run_pass(ctx, s, KNERF_FINE, o, d, ctx->t_f, n_rays, Na, fi, nullptr, nullptr, target, inv_chunks, ls + 1, tile0_f, true);
}

// chunks per COARSE weight-gradient launch of knerf_train_batch: up to KNERF_WGRAD_GROUP_MAX (4), within the budget
// (KNERF_WGRAD_GROUP_GB, default 40; 0 = one launch per chunk) and a quarter of the free device memory
int wgrad_group_for(knerf_ctx* ctx, int n_rays, int n_chunks) {
    if (ctx->generic || n_chunks <= 1) return 1;
    if (ctx->group_cache > 0 && ctx->group_cache_rays == n_rays && ctx->group_cache_chunks == n_chunks) return ctx->group_cache;
    auto memo = [&](int g) { ctx->group_cache_rays = n_rays; ctx->group_cache_chunks = n_chunks; ctx->group_cache = g; return g; };
    const double budget = ctx->wgrad_group_gb;
    if (budget <= 0 || ctx->wgrad_group_max <= 1) return memo(1);
    const int Nc = ctx->cfg.n_coarse, Na = Nc + ctx->cfg.n_fine;
    const double per_chunk = (double)(saved_region_bytes(tiles_for((long long)n_rays * Nc), ctx->si.act_blocks) + saved_region_bytes(tiles_for((long long)n_rays * Nc), ctx->si.mask_blocks) +
                                      saved_region_bytes(tiles_for((long long)n_rays * Nc), ctx->si.dz_blocks));   // one coarse region
    (void)Na;
    size_t free_b = 0, total_b = 0;
    double avail = budget * 1e9;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const double mine = ctx->ws_train ? (double)(ctx->act_bytes + ctx->mask_bytes + ctx->dz_bytes) : 0.0;   // what a re-allocation gives back
        const double cap = 0.25 * ((double)free_b + mine);
        if (cap < avail) avail = cap;
    }
    int g = (int)(avail / per_chunk);
    const int g_max = ctx->wgrad_group_max;         // default 4: coarse launches of more than ~4 x 8192 tiles gain nothing more
    if (g > g_max) g = g_max;
    if (g > n_chunks) g = n_chunks;
    if (g <= 1) return memo(1);
    const int n_groups = (n_chunks + g - 1) / g;
    return memo((n_chunks + n_groups - 1) / n_groups);      // the smallest group that needs no more launches
}

// Chunks per LAUNCH (option "merge_chunk_rays", default 4096 rays).  `ray_chunks` is the reference's memory knob (nerf.py:100,
// 332-473: a Python loop over chunks, gradients accumulated as g / C): every ray's forward, loss term and gradient contribution is
// independent of which chunk holds it -- the per-chunk mean over R rays times 1 / C is the per-launch mean over m R rays times
// m / C, the fine sampler's random numbers are keyed by the ray's index in the batch -- so m consecutive chunks run as ONE set of
// launches: rendered outputs bit-identical, accumulated gradients equal up to the order of fp32 sums.  On this device a 4,096-ray
// launch needs 8 GB of workspace; launches of 256 / 512 / 1,024 rays (the reference's train.py defaults to 1,024) fill 256 CUs
// badly: 23.7 / 18.0 / 14.3 ms per 128 x 128 image against 13.7.  m divides the chunk count (equal-sized launches).
int merge_factor(int limit_rays, int ray_chunks, int n_chunks) {
    if (limit_rays <= 0 || n_chunks <= 1) return 1;
    for (int m = n_chunks; m > 1; --m)
        if (n_chunks % m == 0 && (long long)m * ray_chunks <= limit_rays) return m;
    return 1;
}

}  // namespace

extern "C" {

size_t knerf_param_count(void) { return (size_t)kParamCount; }

size_t knerf_param_count_for(const knerf_config* cfg) {
    if (!cfg || cfg->n_layers < 1 || cfg->dense_units < 2 || cfg->skip_layer < 1 || cfg->pos_emb_xyz < 0 || cfg->pos_emb_dir < 0) return 0;
    if (cfg->flags & KNERF_FLAG_ENCODED_WIDTHS) {
        if (cfg->pos_emb_xyz < 1 || cfg->pos_emb_dir < 1) return 0;
        return (size_t)gen::build_plan_widths(cfg->n_layers, cfg->dense_units, cfg->skip_layer, cfg->pos_emb_xyz, cfg->pos_emb_dir).n_params;
    }
    return (size_t)gen::param_count(cfg->n_layers, cfg->dense_units, cfg->skip_layer, cfg->pos_emb_xyz, cfg->pos_emb_dir);
}

const char* knerf_last_error(const knerf_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int knerf_create(const knerf_config* cfg, knerf_ctx** out) {
    knerf_ctx* ctx = nullptr;
    if (!cfg || !out) return fail(nullptr, KNERF_ERR_INVALID, "null argument");
    const bool widths = (cfg->flags & KNERF_FLAG_ENCODED_WIDTHS) != 0;      // stand-alone NeRFMLP: pos_emb_* ARE the two input widths
    if (cfg->n_layers < 1 || cfg->n_layers > 64 || cfg->dense_units < 2 || cfg->dense_units > 4096 || cfg->skip_layer < 1)
        return fail(nullptr, KNERF_ERR_INVALID, "need 1 <= n_layers <= 64, 2 <= dense_units <= 4096, skip_layer >= 1");
    if (widths && (cfg->pos_emb_xyz < 1 || cfg->pos_emb_xyz > 4096 || cfg->pos_emb_dir < 1 || cfg->pos_emb_dir > 4096))
        return fail(nullptr, KNERF_ERR_INVALID, "KNERF_FLAG_ENCODED_WIDTHS: need 1 <= input width <= 4096 in pos_emb_xyz / pos_emb_dir");
    if (!widths && (cfg->pos_emb_xyz < 0 || cfg->pos_emb_xyz > 32 || cfg->pos_emb_dir < 0 || cfg->pos_emb_dir > 32))
        return fail(nullptr, KNERF_ERR_INVALID, "need 0 <= pos_emb_* <= 32");
    // one wavefront walks a ray in compositing (a lane's run of up to 16 samples in registers) and in the sampler (tables in LDS)
    if (!widths && (cfg->n_coarse < 2 || cfg->n_coarse > 512 || cfg->n_fine < 0 || cfg->n_coarse + cfg->n_fine > 1024))
        return fail(nullptr, KNERF_ERR_INVALID, "need 2 <= n_coarse <= 512 and n_coarse + n_fine <= 1024");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(nullptr, KNERF_ERR_NODEVICE, "no HIP device visible");
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
        return fail(nullptr, KNERF_ERR_HIP, "hipGetDeviceProperties failed");
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(nullptr, KNERF_ERR_NODEVICE, std::string("libknerf_hip is built for gfx950 only; device is ") + prop.gcnArchName);
    ctx = new knerf_ctx();
    ctx->cfg = *cfg;
    // the fused kernels cover the trunk shapes of layout.h KNERF_FUSED_SHAPES (widths 256 and 128; the reference's encodings unless the build added others); everything
    // else -- and, for tests, any shape under KNERF_FLAG_FORCE_GENERIC -- runs on the general-shape kernels
    const int sid = widths ? -1 : fused_shape_id(cfg->n_layers, cfg->skip_layer, cfg->dense_units, cfg->pos_emb_xyz, cfg->pos_emb_dir);
    ctx->generic = sid < 0 || (cfg->flags & KNERF_FLAG_FORCE_GENERIC) != 0;
    ctx->mlp_only = widths;
    if (ctx->generic) {
        ctx->gplan = widths ? gen::build_plan_widths(cfg->n_layers, cfg->dense_units, cfg->skip_layer, cfg->pos_emb_xyz, cfg->pos_emb_dir)
                            : gen::build_plan(cfg->n_layers, cfg->dense_units, cfg->skip_layer, cfg->pos_emb_xyz, cfg->pos_emb_dir);
        ctx->n_params = ctx->gplan.n_params;
    } else {
        ctx->shape = sid; ctx->si = shape_info(sid); ctx->n_params = ctx->si.param_count;
    }
    for (int j = 0; j < ctx->si.n_jobs; ++j) ctx->wgrad_cost[j] = default_wgrad_cost(ctx->si.job_kind[j]);
    const size_t NP = (size_t)ctx->n_params;
    const Tables& ht = host_tables(ctx->shape);
    ctx->tab.host = ht.host; ctx->tab.wgrad = ht.wgrad; ctx->tab.wgrad_off = ht.wgrad_off;
    auto up = [&](int*& dptr, const std::vector<int32_t>& v) -> hipError_t {
        hipError_t e = hipMalloc(&dptr, v.size() * sizeof(int32_t));
        if (e != hipSuccess) return e;
        return hipMemcpy(dptr, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice);
    };
#define CREATECHK(expr)                                                                                          \
    do {                                                                                                         \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess) {                                                                                  \
            std::string m_ = std::string(#expr) + ": " + hipGetErrorString(e_);                                  \
            knerf_destroy(ctx);                                                                                  \
            return fail(nullptr, KNERF_ERR_HIP, m_);                                                             \
        }                                                                                                        \
    } while (0)
    CREATECHK(up(ctx->tab.d_fwd, ctx->tab.host.fwd));
    CREATECHK(up(ctx->tab.d_bias, ctx->tab.host.fwd_bias));
    CREATECHK(up(ctx->tab.d_bwd, ctx->tab.host.bwd));
    CREATECHK(up(ctx->tab.d_wgrad, ctx->tab.wgrad));
    ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (int r = upload_plan(ctx)) { std::string m_ = ctx->err; knerf_destroy(ctx); return fail(nullptr, r, m_); }
    CREATECHK(hipMalloc(&ctx->tile_count, (size_t)ctx->tile_counters * sizeof(int)));
    CREATECHK(hipMemset(ctx->tile_count, 0, (size_t)ctx->tile_counters * sizeof(int)));
    CREATECHK(hipMalloc(&ctx->tile_stats, 8 * sizeof(long long)));       // per net: live, total, [2], [3]: out-of-range list entries (diagnostic builds)
    CREATECHK(hipMemset(ctx->tile_stats, 0, 8 * sizeof(long long)));
    CREATECHK(hipMalloc(&ctx->d_diag, 2 * sizeof(unsigned long long)));
    CREATECHK(hipHostMalloc(&ctx->h_diag, 4 * sizeof(long long), hipHostMallocDefault));
    ctx->h_diag[0] = ctx->h_diag[1] = ctx->h_diag[2] = ctx->h_diag[3] = 0;
    CREATECHK(hipMalloc(&ctx->grads, 2 * NP * sizeof(float)));
    CREATECHK(hipMemset(ctx->grads, 0, 2 * NP * sizeof(float)));
    CREATECHK(hipMalloc(&ctx->aux, 2 * (size_t)kAuxCount * sizeof(float)));
    CREATECHK(hipMemset(ctx->aux, 0, 2 * (size_t)kAuxCount * sizeof(float)));
    CREATECHK(hipMalloc(&ctx->d_flag, sizeof(int)));
    CREATECHK(hipMemset(ctx->d_flag, 0, sizeof(int)));
    CREATECHK(hipMalloc(&ctx->d_step, sizeof(int)));
    CREATECHK(hipMalloc(&ctx->d_lr_t, sizeof(float)));
    CREATECHK(launch_step_set(0, ctx->d_step, ctx->d_lr_t, AdamHyper{cfg->lr, cfg->beta1, cfg->beta2}, nullptr));
    CREATECHK(hipHostMalloc(&ctx->h_status, 2 * sizeof(int), hipHostMallocDefault));   // written by the device (optim.hip step_status)
    ctx->h_status[0] = ctx->h_status[1] = 0;
    CREATECHK(hipMalloc(&ctx->loss_tmp, 2 * sizeof(float)));
    for (int n = 0; n < 2; ++n) {
        Net& N = ctx->net[n];
        const size_t NW = ctx->generic ? NP : (size_t)ctx->si.ext_param_count;  // fused path: parameters + composed head (layout.h)
        CREATECHK(hipMalloc(&N.w, NW * sizeof(float)));
        CREATECHK(hipMalloc(&N.m, NP * sizeof(float)));
        CREATECHK(hipMalloc(&N.v, NP * sizeof(float)));
        CREATECHK(hipMemset(N.w, 0, NW * sizeof(float)));
        CREATECHK(hipMemset(N.m, 0, NP * sizeof(float)));
        CREATECHK(hipMemset(N.v, 0, NP * sizeof(float)));
        N.g = ctx->grads + (size_t)n * NP;
        N.aux = ctx->aux + (size_t)n * kAuxCount;
        if (ctx->generic) {
            CREATECHK(hipMalloc(&ctx->gnet[n].packed, ctx->gplan.packed_elems * sizeof(unsigned short)));
            CREATECHK(hipMemset(ctx->gnet[n].packed, 0, ctx->gplan.packed_elems * sizeof(unsigned short)));
            CREATECHK(hipMalloc(&ctx->gnet[n].head, gen::head_floats(ctx->gplan) * sizeof(float)));
            CREATECHK(hipMemset(ctx->gnet[n].head, 0, gen::head_floats(ctx->gplan) * sizeof(float)));
            CREATECHK(hipMalloc(&ctx->gnet[n].gaux, gen::aux_floats(ctx->gplan) * sizeof(float)));
            CREATECHK(hipMemset(ctx->gnet[n].gaux, 0, gen::aux_floats(ctx->gplan) * sizeof(float)));
        }
        CREATECHK(hipMalloc(&N.fwd_stream, stream_bytes(ctx->si.fwd_blocks)));
        CREATECHK(hipMalloc(&N.bwd_stream, stream_bytes(ctx->si.bwd_blocks)));
        CREATECHK(hipMemset(N.fwd_stream, 0, stream_bytes(ctx->si.fwd_blocks)));
        CREATECHK(hipMemset(N.bwd_stream, 0, stream_bytes(ctx->si.bwd_blocks)));
        CREATECHK(hipMalloc(&N.bias, ctx->si.fwd_bias_tiles * 32 * sizeof(float)));
        CREATECHK(hipMemset(N.bias, 0, ctx->si.fwd_bias_tiles * 32 * sizeof(float)));
    }
#undef CREATECHK
    *out = ctx;
    return KNERF_OK;
}

int knerf_destroy(knerf_ctx* ctx) {
    if (!ctx) return KNERF_OK;
    for (int n = 0; n < 2; ++n) {
        Net& N = ctx->net[n];
        free_dev(N.w); free_dev(N.m); free_dev(N.v); free_dev(N.fwd_stream); free_dev(N.bwd_stream); free_dev(N.bias);
    }
    free_dev(ctx->grads); free_dev(ctx->aux); free_dev(ctx->d_flag); free_dev(ctx->loss_tmp); free_dev(ctx->d_step); free_dev(ctx->d_lr_t);
    if (ctx->h_status) { (void)hipHostFree(ctx->h_status); ctx->h_status = nullptr; }
    if (ctx->h_diag) { (void)hipHostFree(ctx->h_diag); ctx->h_diag = nullptr; }
    free_dev(ctx->d_diag); free_dev(ctx->diag_tmp);
    free_dev(ctx->tab.d_fwd); free_dev(ctx->tab.d_bias); free_dev(ctx->tab.d_bwd); free_dev(ctx->tab.d_wgrad); free_dev(ctx->tab.d_plan);
    free_dev(ctx->raw); free_dev(ctx->draw); free_dev(ctx->w_c); free_dev(ctx->t_f); free_dev(ctx->img_tmp);
    free_dev(ctx->act); free_dev(ctx->mask); free_dev(ctx->dz);
    free_dev(ctx->tile_flags); free_dev(ctx->tile_list); free_dev(ctx->tile_list_g); free_dev(ctx->tile_count); free_dev(ctx->tile_stats);
    free_dev(ctx->partial); free_dev(ctx->loss_partial); free_dev(ctx->d_job_wg0);
    free_dev(ctx->gws.act); free_dev(ctx->gws.dz); free_dev(ctx->gws.zs); free_dev(ctx->gws.zc); free_dev(ctx->gws.mask);
    for (int n = 0; n < 2; ++n) { free_dev(ctx->gnet[n].packed); free_dev(ctx->gnet[n].head); free_dev(ctx->gnet[n].gaux); }
    free_dev(ctx->call_net.head);
    free_dev(ctx->call_ws.act); free_dev(ctx->call_ws.zs); free_dev(ctx->call_ws.zc); free_dev(ctx->call_net.packed); free_dev(ctx->call_raw);
    delete ctx;
    return KNERF_OK;
}

int knerf_set_weights(knerf_ctx* ctx, int net, const float* host_flat, size_t n) {
    if (int r = check_net(ctx, net)) return r;
    if (!host_flat || n != (size_t)ctx->n_params) return fail(ctx, KNERF_ERR_INVALID, "weights: expected " + std::to_string(ctx->n_params) + " floats");
    HIPCHK(hipMemcpy(ctx->net[net].w, host_flat, n * sizeof(float), hipMemcpyHostToDevice));
    if (int r = repack(ctx, net, nullptr)) return r;
    HIPCHK(hipStreamSynchronize(nullptr));
    return KNERF_OK;
}

int knerf_get_weights(knerf_ctx* ctx, int net, float* host_flat, size_t n) {
    if (int r = check_net(ctx, net)) return r;
    if (!host_flat || n != (size_t)ctx->n_params) return fail(ctx, KNERF_ERR_INVALID, "weights: expected " + std::to_string(ctx->n_params) + " floats");
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(host_flat, ctx->net[net].w, n * sizeof(float), hipMemcpyDeviceToHost));
    return KNERF_OK;
}

int knerf_weights_device(knerf_ctx* ctx, int net, float** dev, size_t* n) {
    if (int r = check_net(ctx, net)) return r;
    if (dev) *dev = ctx->net[net].w;
    if (n) *n = (size_t)ctx->n_params;
    return KNERF_OK;
}

int knerf_grads_device(knerf_ctx* ctx, float** dev, size_t* n) {
    if (!ctx) return KNERF_ERR_INVALID;
    if (dev) *dev = ctx->grads;
    if (n) *n = 2 * (size_t)ctx->n_params;
    return KNERF_OK;
}

int knerf_refresh_weights(knerf_ctx* ctx, void* stream) {
    if (!ctx) return KNERF_ERR_INVALID;
    for (int n = 0; n < 2; ++n) if (int r = repack(ctx, n, (hipStream_t)stream)) return r;
    return KNERF_OK;
}

int knerf_forward_chunk(knerf_ctx* ctx, void* stream, int net, const float* o, const float* d, const float* t,
                        int n_rays, int n_samples, float* image, float* depth, float* weights) {
    if (int r = check_net(ctx, net)) return r;
    if (int r = check_rays(ctx, "forward_chunk")) return r;
    if (!o || !d || !t || !image || n_rays <= 0) return fail(ctx, KNERF_ERR_INVALID, "forward_chunk: null/empty argument");
    if (n_samples < 1 || n_samples > ctx->cfg.n_coarse + ctx->cfg.n_fine) return fail(ctx, KNERF_ERR_INVALID, "forward_chunk: n_samples out of range");
    if (int r = ensure_ws(ctx, n_rays, false, (hipStream_t)stream)) return r;
    return run_pass(ctx, (hipStream_t)stream, net, o, d, t, n_rays, n_samples, image, depth, weights, nullptr, 1.f, nullptr);
}

int knerf_mlp_call(knerf_ctx* ctx, void* stream, int net, const float* xyz_enc, const float* dir_enc, uint64_t n, float* raw) {
    if (int r = check_net(ctx, net)) return r;
    if (!xyz_enc || !dir_enc || !raw || n == 0) return fail(ctx, KNERF_ERR_INVALID, "mlp_call: null/empty argument");
    hipStream_t s = (hipStream_t)stream;
    if (!ctx->call_plan_ok) {
        const knerf_config& c = ctx->cfg;
        ctx->call_plan = ctx->mlp_only ? gen::build_plan_widths(c.n_layers, c.dense_units, c.skip_layer, c.pos_emb_xyz, c.pos_emb_dir)
                                       : gen::build_plan(c.n_layers, c.dense_units, c.skip_layer, c.pos_emb_xyz, c.pos_emb_dir);
        HIPCHK(hipMalloc(&ctx->call_net.packed, ctx->call_plan.packed_elems * sizeof(unsigned short)));
        HIPCHK(hipMalloc(&ctx->call_net.head, gen::head_floats(ctx->call_plan) * sizeof(float)));
        ctx->call_plan_ok = true;
    }
    const gen::Plan& p = ctx->call_plan;
    const size_t mp = gen::padded_rows((long long)n);
    gen::Workspace& w = ctx->call_ws;
    if (mp > w.mp) {
        HIPCHK(hipStreamSynchronize(s));
        free_dev(w.act); free_dev(w.zs); free_dev(w.zc); free_dev(ctx->call_raw);
        w.mp = 0;
        const size_t ab = p.act_elems_per_row * mp * sizeof(unsigned short);
        HIPCHK(hipMalloc(&w.act, ab));
        HIPCHK(hipMemsetAsync(w.act, 0, ab, s));        // on the stream the consuming kernels run on
        HIPCHK(hipMalloc(&w.zs, mp * 32 * sizeof(float)));
        HIPCHK(hipMalloc(&w.zc, mp * 32 * sizeof(float)));
        w.mp = mp;
    }
    HIPCHK(gen::pack_weights(p, ctx->net[net].w, ctx->call_net, s));       // the weights may have changed since the last call
    HIPCHK(gen::forward_encoded(p, w, ctx->call_net, ctx->net[net].w, xyz_enc, dir_enc, (long long)n, raw, s));
    return KNERF_OK;
}

int knerf_sample_fine(knerf_ctx* ctx, void* stream, const float* t_coarse, const float* w_coarse, const float* u,
                      uint64_t seed, uint64_t stream_id, uint64_t ray_offset, int n_rays, float* t_out) {
    if (int r = check_rays(ctx, "sample_fine")) return r;
    if (!t_coarse || !w_coarse || !t_out || n_rays <= 0) return fail(ctx, KNERF_ERR_INVALID, "sample_fine: null/empty argument");
    SampleArgs sa{};
    sa.t_coarse = t_coarse; sa.w_coarse = w_coarse; sa.u = u; sa.t_out = t_out; sa.R = n_rays;
    sa.Nc = ctx->cfg.n_coarse; sa.Nf = ctx->cfg.n_fine; sa.oob_clamp = ctx->cfg.oob_clamp;
    sa.seed = seed; sa.stream_id = stream_id; sa.ray_offset = ray_offset;
    { ProfScope ps(ctx, (hipStream_t)stream, P_SAMPLE); HIPCHK(launch_sample_fine(sa, (hipStream_t)stream)); }
    return KNERF_OK;
}

int knerf_render_chunk(knerf_ctx* ctx, void* stream, const float* o, const float* d, const float* t, const float* u,
                       uint64_t seed, uint64_t ray_offset, int n_rays, float* c_image, float* c_depth, float* c_weights,
                       float* f_image, float* f_depth, float* f_weights, float* t_fine) {
    if (int r = check_rays(ctx, "render_chunk")) return r;
    if (!o || !d || !t || !c_image || !f_image || n_rays <= 0) return fail(ctx, KNERF_ERR_INVALID, "render_chunk: null/empty argument");
    hipStream_t s = (hipStream_t)stream;
    if (int r = ensure_ws(ctx, n_rays, false, s)) return r;
    const int Nc = ctx->cfg.n_coarse, Na = Nc + ctx->cfg.n_fine;
    float* wc = c_weights ? c_weights : ctx->w_c;
    float* tf = t_fine ? t_fine : ctx->t_f;
    if (int r = run_pass(ctx, s, KNERF_COARSE, o, d, t, n_rays, Nc, c_image, c_depth, wc, nullptr, 1.f, nullptr)) return r;
    if (int r = knerf_sample_fine(ctx, stream, t, wc, u, seed, 0, ray_offset, n_rays, tf)) return r;
    return run_pass(ctx, s, KNERF_FINE, o, d, tf, n_rays, Na, f_image, f_depth, f_weights, nullptr, 1.f, nullptr);
}

int knerf_render_batch(knerf_ctx* ctx, void* stream, const float* o, const float* d, const float* t, const float* u, uint64_t seed,
                       int n_rays, int ray_chunks, float* c_image, float* c_depth, float* c_weights, float* f_image, float* f_depth,
                       float* f_weights, float* t_fine) {
    if (int r = check_rays(ctx, "render_batch")) return r;            // (before any workspace is sized: a stand-alone-MLP context has none)
    if (ray_chunks <= 0 || n_rays <= 0 || n_rays % ray_chunks != 0)
        return fail(ctx, KNERF_ERR_INVALID, "render_batch: ray_chunks must be a divisor of the number of rays");   // nerf.py:100
    const size_t Nc = (size_t)ctx->cfg.n_coarse, Nf = (size_t)ctx->cfg.n_fine, Na = Nc + Nf;
    // per-ray work only: bit-identical outputs.  Rendering keeps no saved tensors (5 KB of workspace per ray on the fused path), so its
    // launches may be far larger than a training step's: 65,536 rays by default ("merge_render_rays"; a 256 x 256 frame in ONE set
    // of launches is 4 % faster than in sixteen); the general-shape path's activations stay in memory, it keeps the training limit
    // ... and `ray_chunks` stays the reference's memory knob: merged launches that do not fit fall back to the caller's own chunks
    // (remembered per (ray_chunks, chunk count), so that later frames do not fail the same allocation again)
    const int user_chunks = ray_chunks, user_n = n_rays / ray_chunks;
    int m = merge_factor(ctx->generic ? ctx->merge_rays : ctx->merge_render_rays, ray_chunks, user_n);
    if (m > 1 && ctx->rmerge_fail_rays == user_chunks && ctx->rmerge_fail_chunks == user_n) m = 1;
    if (m > 1 && ensure_ws(ctx, user_chunks * m, false, (hipStream_t)stream)) {
        ctx->rmerge_fail_rays = user_chunks; ctx->rmerge_fail_chunks = user_n; ++ctx->merge_fallbacks;
        m = 1;
    }
    ray_chunks = user_chunks * m;
    for (int i = 0; i < n_rays / ray_chunks; ++i) {
        const size_t r0 = (size_t)i * ray_chunks;
        if (int r = knerf_render_chunk(ctx, stream, o + r0 * 3, d + r0 * 3, t + r0 * Nc, u ? u + r0 * Nf : nullptr, seed, (uint64_t)r0,
                                       ray_chunks, c_image ? c_image + r0 * 3 : nullptr, c_depth ? c_depth + r0 : nullptr,
                                       c_weights ? c_weights + r0 * Nc : nullptr, f_image ? f_image + r0 * 3 : nullptr,
                                       f_depth ? f_depth + r0 : nullptr, f_weights ? f_weights + r0 * Na : nullptr,
                                       t_fine ? t_fine + r0 * Na : nullptr))
            return r;
    }
    return KNERF_OK;
}

int knerf_train_chunk(knerf_ctx* ctx, void* stream, const float* o, const float* d, const float* t, const float* target,
                      const float* u, uint64_t seed, uint64_t ray_offset, int n_rays, float inv_chunks, float* loss,
                      float* c_image, float* f_image) {
    if (int r = check_rays(ctx, "train_chunk")) return r;
    if (!o || !d || !t || !target || n_rays <= 0) return fail(ctx, KNERF_ERR_INVALID, "train_chunk: null/empty argument");
    hipStream_t s = (hipStream_t)stream;
    if (ctx->plan_dirty) { if (int r = upload_plan(ctx)) return r; }
    ctx->tile_counter_next = 0;
    if (int r = ensure_ws(ctx, n_rays, true, s)) return r;
    if (int r = train_chunk_impl(ctx, s, o, d, t, target, u, seed, ray_offset, n_rays, inv_chunks, loss, c_image, f_image)) return r;
    return expand_head_grads(ctx, s);
}

int knerf_train_batch(knerf_ctx* ctx, void* stream, const float* o, const float* d, const float* t, const float* target,
                      const float* u, uint64_t seed, int n_rays, int ray_chunks, float* loss, float* c_image, float* f_image) {
    if (int r = check_rays(ctx, "train_batch")) return r;
    if (!o || !d || !t || !target) return fail(ctx, KNERF_ERR_INVALID, "train_batch: null argument");
    if (ray_chunks <= 0 || n_rays <= 0 || n_rays % ray_chunks != 0)
        return fail(ctx, KNERF_ERR_INVALID, "train_batch: ray_chunks must be a divisor of the number of rays");   // nerf.py:100
    hipStream_t s = (hipStream_t)stream;
    const int Nc = ctx->cfg.n_coarse, Nf = ctx->cfg.n_fine;
    if (ctx->plan_dirty) { if (int r = upload_plan(ctx)) return r; }
    ctx->tile_counter_next = 0;
    // zero-gradient diagnostics count the LAST chunk's gradient (nerf.py:430-451): that chunk's launches must be its own
    int merge = ctx->grad_diag ? 1 : merge_factor(ctx->merge_rays, ray_chunks, n_rays / ray_chunks);
    const int user_chunks = ray_chunks;
    if (merge > 1 && ctx->merge_fail_rays == user_chunks && ctx->merge_fail_chunks == n_rays / user_chunks) merge = 1;   // did not fit last time
    int C = 0, G = 1;
    for (;;) {
        ray_chunks = user_chunks * merge;
        C = n_rays / ray_chunks;
        G = ctx->grad_diag ? 1 : wgrad_group_for(ctx, ray_chunks, C);
        int r = ensure_ws(ctx, ray_chunks, true, s, G);
        if (r && G > 1) {                                           // the group did not fit: one chunk per wgrad launch
            G = 1;
            ctx->group_cache_rays = ray_chunks; ctx->group_cache_chunks = C; ctx->group_cache = 1;
            r = ensure_ws(ctx, ray_chunks, true, s, 1);
        }
        if (!r) break;
        if (merge == 1) return r;
        merge = 1;                                                  // the merged launches did not fit either: the caller's own chunks
        ctx->merge_fail_rays = user_chunks; ctx->merge_fail_chunks = n_rays / user_chunks; ++ctx->merge_fallbacks;
    }
    const size_t tc = tiles_for((long long)ray_chunks * Nc);
    const bool skip = skipping(ctx);
    if (skip) { if (int r = ensure_tile_counters(ctx, s, 2LL * C + (C + G - 1) / G + 8)) return r; }   // two passes per chunk + one per group
    int* group_count = nullptr;
    for (int i = 0; i < C; ++i) {
        const size_t r0 = (size_t)i * ray_chunks;
        const int slot = i % G;
        const bool last = slot == G - 1 || i == C - 1;
        if (ctx->grad_diag && i == C - 1 && C > 1) {
            // set the sum of the earlier chunks aside (head sums expanded first: the expansion is linear), so that the accumulator
            // holds the last chunk's gradient alone when it is counted
            if (int r = expand_head_grads(ctx, s)) return r;
            const size_t nb = 2 * (size_t)ctx->n_params * sizeof(float);
            if (!ctx->diag_tmp) HIPCHK(hipMalloc(&ctx->diag_tmp, nb));
            HIPCHK(hipMemcpyAsync(ctx->diag_tmp, ctx->grads, nb, hipMemcpyDeviceToDevice, s));
            HIPCHK(hipMemsetAsync(ctx->grads, 0, nb, s));
        }
        // default-mode skipping: the coarse passes of a group append their live tiles to the group's list under one counter
        if (G > 1 && skip && !ctx->deterministic && slot == 0) { if (int r = next_tile_counter(ctx, s, &group_count)) return r; }
        if (int r = train_chunk_impl(ctx, s, o + r0 * 3, d + r0 * 3, t + r0 * Nc, target + r0 * 3, u ? u + r0 * Nf : nullptr, seed,
                                     (uint64_t)r0, ray_chunks, 1.0f / (float)C, loss, c_image ? c_image + r0 * 3 : nullptr,
                                     f_image ? f_image + r0 * 3 : nullptr, slot, G, G > 1 && skip && !ctx->deterministic ? group_count : nullptr))
            return r;
        if (G > 1 && last) {                                        // the group's coarse weight gradients in one launch
            const int* live = nullptr; int* live_count = nullptr;
            if (skip && ctx->deterministic) {                       // ascending list of the group's live tiles from its flags
                if (int r = compact_tiles(ctx, s, ctx->tile_flags, (int)((slot + 1) * tc), (int)tc, ray_chunks * Nc / kTile, ctx->tile_list, &live_count)) return r;
                live = ctx->tile_list;
            } else if (skip) {
                live = ctx->tile_list_g; live_count = group_count;
            }
            if (int r = launch_wgrad_tiles(ctx, s, KNERF_COARSE, 0, (size_t)(slot + 1) * tc, live, live_count)) return r;
        }
    }
    if (int r = expand_head_grads(ctx, s)) return r;
    if (ctx->grad_diag) {
        HIPCHK(launch_grad_diagnostics(ctx->grads, ctx->n_params, ctx->d_diag, ctx->h_diag, s));
        if (C > 1) HIPCHK(launch_add_into(ctx->grads, ctx->diag_tmp, 2 * (size_t)ctx->n_params, s));
    }
    return KNERF_OK;
}

int knerf_apply_adam(knerf_ctx* ctx, void* stream) {
    if (!ctx) return KNERF_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    // Finite check first, on the device; the Adam kernels read its flag and leave weights and slots untouched when it is
    // set (the reference aborts fit at nerf.py:381-382).  Nothing here waits for the GPU: the outcome reaches the host through
    // a pinned status word (knerf_poll_nonfinite).
    HIPCHK(hipMemsetAsync(ctx->d_flag, 0, sizeof(int), s));
    HIPCHK(launch_check_finite(ctx->grads, 2 * ctx->n_params, ctx->d_flag, s));
    ctx->step += 1;           // host mirror: steps enqueued minus the skipped ones that have been polled; Adam's t lives on the device
    ProfScope ps(ctx, s, P_ADAM);
    for (int n = 0; n < 2; ++n) {
        AdamArgs a{};
        a.w = ctx->net[n].w; a.m = ctx->net[n].m; a.v = ctx->net[n].v; a.g = ctx->net[n].g; a.n = ctx->n_params;
        a.lr_t = ctx->d_lr_t; a.b1 = ctx->cfg.beta1; a.b2 = ctx->cfg.beta2; a.eps = ctx->cfg.epsilon; a.nonfinite = ctx->d_flag;
        HIPCHK(launch_adam(a, s));
    }
    if (!ctx->generic) HIPCHK(launch_head_compose(ctx->net[0].w, ctx->net[1].w, ctx->si.trunk_params, ctx->si.units, ctx->si.trunk_x, ctx->si.trunk_x_slots, ctx->si.dir_dim, ctx->si.dir_slots, s));      // both nets' heads in one launch
    for (int n = 0; n < 2; ++n)
        if (int r = repack(ctx, n, s, false)) return r;
    HIPCHK(launch_step_status(ctx->d_flag, ctx->h_status, ctx->d_step, ctx->d_lr_t, AdamHyper{ctx->cfg.lr, ctx->cfg.beta1, ctx->cfg.beta2}, s));
    return KNERF_OK;
}

int knerf_poll_nonfinite(knerf_ctx* ctx, void* stream, int wait) {
    if (!ctx) return KNERF_ERR_INVALID;
    if (wait) HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    const int skipped = *reinterpret_cast<volatile int*>(ctx->h_status);
    if (skipped != ctx->skipped_seen) {
        ctx->step -= skipped - ctx->skipped_seen;          // those updates did not happen: the step counter (Adam's t) goes back
        ctx->skipped_seen = skipped;
        return fail(ctx, KNERF_ERR_NONFINITE, "Gradient is not finite");
    }
    return KNERF_OK;
}

int knerf_zero_grads(knerf_ctx* ctx, void* stream) {
    if (!ctx) return KNERF_ERR_INVALID;
    HIPCHK(hipMemsetAsync(ctx->grads, 0, 2 * (size_t)ctx->n_params * sizeof(float), (hipStream_t)stream));
    HIPCHK(hipMemsetAsync(ctx->aux, 0, 2 * (size_t)kAuxCount * sizeof(float), (hipStream_t)stream));
    if (ctx->generic)
        for (int n = 0; n < 2; ++n) HIPCHK(hipMemsetAsync(ctx->gnet[n].gaux, 0, gen::aux_floats(ctx->gplan) * sizeof(float), (hipStream_t)stream));
    return KNERF_OK;
}

int knerf_set_option(knerf_ctx* ctx, const char* name, double value) {
    if (!ctx || !name) return KNERF_ERR_INVALID;
    const std::string n(name);
    if (n == "deterministic") {
        ctx->deterministic = value != 0;         // fused path: wgrad slabs + wgrad_reduce_kernel; general-shape path: generic.hip's slabs (round 5)
    } else if (n == "grad_diagnostics") {
        ctx->grad_diag = value != 0;
    } else if (n == "skip_dead_tiles") {
        ctx->skip_dead = value != 0;               // ignored where it does not apply (general-shape path, sample counts not multiples of 32)
    } else if (n == "merge_chunk_rays") {
        if (value < 0 || value > 1048576) return fail(ctx, KNERF_ERR_INVALID, "merge_chunk_rays: 0..1048576");
        ctx->merge_rays = (int)value; ctx->merge_fail_rays = 0; if (ctx->generic) ctx->rmerge_fail_rays = 0;
    } else if (n == "merge_render_rays") {
        if (value < 0 || value > 1048576) return fail(ctx, KNERF_ERR_INVALID, "merge_render_rays: 0..1048576");
        ctx->merge_render_rays = (int)value; ctx->rmerge_fail_rays = 0;
    } else if (n == "workspace_limit_gb") {      // tests: workspace requests above it fail like an exhausted device (0 = off)
        if (value < 0) return fail(ctx, KNERF_ERR_INVALID, "workspace_limit_gb: >= 0");
        ctx->ws_limit_gb = value; ctx->merge_fail_rays = ctx->rmerge_fail_rays = 0; ctx->group_cache = 0;
    } else if (n == "wgrad_group_max") {
        if (value < 1 || value > 64) return fail(ctx, KNERF_ERR_INVALID, "wgrad_group_max: 1..64");
        ctx->wgrad_group_max = (int)value; ctx->group_cache = 0;
    } else if (n == "wgrad_group_gb") {
        if (value < 0) return fail(ctx, KNERF_ERR_INVALID, "wgrad_group_gb: >= 0");
        ctx->wgrad_group_gb = value; ctx->group_cache = 0;
    } else if (n.rfind("wgrad_cost", 0) == 0 && n.size() >= 11 && n.size() <= 12 && std::atoi(n.c_str() + 10) < ctx->si.n_jobs && n[10] >= '0' && n[10] <= '9') {
        if (value < 0 || value > 1e6) return fail(ctx, KNERF_ERR_INVALID, "wgrad_cost: 0..1e6");
        ctx->wgrad_cost[std::atoi(n.c_str() + 10)] = (int)value; ctx->plan_dirty = true;
    } else {
        return fail(ctx, KNERF_ERR_INVALID, "unknown option '" + n + "'");
    }
    return KNERF_OK;
}

int knerf_get_option(knerf_ctx* ctx, const char* name, double* value) {
    if (!ctx || !name || !value) return KNERF_ERR_INVALID;
    const std::string n(name);
    if (n == "deterministic") *value = ctx->deterministic;
    else if (n == "skip_dead_tiles") *value = ctx->skip_dead;
    else if (n == "grad_diagnostics") *value = ctx->grad_diag;
    else if (n == "skip_dead_tiles_active") *value = skipping(ctx);
    else if (n == "wgrad_group_max") *value = ctx->wgrad_group_max;
    else if (n == "merge_chunk_rays") *value = ctx->merge_rays;
    else if (n == "merge_render_rays") *value = ctx->merge_render_rays;
    else if (n == "workspace_limit_gb") *value = ctx->ws_limit_gb;
    else if (n == "merge_fallbacks") *value = ctx->merge_fallbacks;
    else if (n == "wgrad_group_gb") *value = ctx->wgrad_group_gb;
    else if (n == "wgrad_group") *value = ctx->ws_train ? ctx->ws_group : 0;            // chunks per coarse wgrad launch of the current workspaces
    else if (n == "general_shape_path") *value = ctx->generic;
    else if (n.rfind("wgrad_cost", 0) == 0 && n.size() >= 11 && n.size() <= 12 && n[10] >= '0' && n[10] <= '9' && std::atoi(n.c_str() + 10) < ctx->si.n_jobs)
        *value = ctx->wgrad_cost[std::atoi(n.c_str() + 10)];
    else return fail(ctx, KNERF_ERR_INVALID, "unknown option '" + n + "'");
    return KNERF_OK;
}

int knerf_tile_stats_net(knerf_ctx* ctx, void* stream, int64_t* live, int64_t* total, int reset) {
    if (!ctx || !live || !total) return KNERF_ERR_INVALID;
    long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    HIPCHK(hipMemcpy(h, ctx->tile_stats, sizeof(h), hipMemcpyDeviceToHost));
    if (reset) HIPCHK(hipMemset(ctx->tile_stats, 0, sizeof(h)));
    for (int n = 0; n < 2; ++n) { live[n] = h[4 * n]; total[n] = h[4 * n + 1]; }
    if (h[2] || h[3] || h[6] || h[7])      // only a -DKNERF_LIST_GUARD build counts these
        return fail(ctx, KNERF_ERR_HIP, "tile list held " + std::to_string(h[2] + h[6]) + " (dgrad) / " + std::to_string(h[3] + h[7]) + " (wgrad) entries outside their pass");
    return KNERF_OK;
}

int knerf_tile_stats(knerf_ctx* ctx, void* stream, int64_t* live, int64_t* total, int reset) {
    if (!ctx || !live || !total) return KNERF_ERR_INVALID;
    int64_t l[2], t[2];
    const int r = knerf_tile_stats_net(ctx, stream, l, t, reset);
    *live = l[0] + l[1]; *total = t[0] + t[1];
    return r;
}

int knerf_grad_diagnostics(knerf_ctx* ctx, void* stream, int wait, int64_t* out) {
    if (!ctx || !out) return KNERF_ERR_INVALID;
    if (wait) HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    // seqlock reader (optim.hip diag_publish_kernel): [2] = published, [3] = begun.  Equal on both sides of the reads <=> the two
    // counts belong to step [2]; a publication in flight makes them differ for a few microseconds.
    volatile long long* h = ctx->h_diag;
    for (int attempt = 0; attempt < 4096; ++attempt) {
        const long long done = h[2];
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        const long long c = h[0], f = h[1];
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        const long long begun = h[3];
        if (begun == done) { out[0] = c; out[1] = f; out[2] = done; return KNERF_OK; }
    }
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));       // never seen in practice: settle it the slow way
    out[2] = h[2]; out[0] = h[0]; out[1] = h[1];
    return KNERF_OK;
}

int knerf_step_count(const knerf_ctx* ctx) { return ctx ? ctx->step : KNERF_ERR_INVALID; }
int knerf_set_step_count(knerf_ctx* ctx, int step) {
    if (!ctx || step < 0) return KNERF_ERR_INVALID;
    HIPCHK(hipDeviceSynchronize());        // steps still in flight count from the old value
    ctx->step = step;
    HIPCHK(launch_step_set(step, ctx->d_step, ctx->d_lr_t, AdamHyper{ctx->cfg.lr, ctx->cfg.beta1, ctx->cfg.beta2}, nullptr));
    HIPCHK(hipDeviceSynchronize());
    return KNERF_OK;
}

int knerf_generate_rays(knerf_ctx* ctx, void* stream, const float* c2w, const float* noise, uint64_t seed, uint64_t stream_id,
                        int batch, int height, int width, int n_samples, float focal, float near_plane, float far_plane,
                        float* o, float* d, float* t) {
    if (!c2w || !o || !d || !t || batch <= 0 || height <= 0 || width <= 0 || n_samples <= 0)
        return fail(ctx, KNERF_ERR_INVALID, "generate_rays: null/empty argument");
    RayGenArgs a{};
    a.c2w = c2w; a.noise = noise; a.o = o; a.d = d; a.t = t; a.B = batch; a.H = height; a.W = width; a.N = n_samples;
    a.focal = focal; a.near_ = near_plane; a.far_ = far_plane; a.seed = seed; a.stream_id = stream_id;
    if (launch_raygen(a, (hipStream_t)stream) != hipSuccess) return fail(ctx, KNERF_ERR_HIP, "generate_rays: launch failed");
    return KNERF_OK;
}

int knerf_profile_enable(knerf_ctx* ctx, int on) {
    if (!ctx) return KNERF_ERR_INVALID;
    ctx->prof_on = on != 0;
    return KNERF_OK;
}

int knerf_profile_read(knerf_ctx* ctx, double* total_ms, int64_t* launches, int n) {
    if (!ctx || !total_ms || !launches || n < P_COUNT) return KNERF_ERR_INVALID;
    for (int i = 0; i < n; ++i) { total_ms[i] = 0.0; launches[i] = 0; }
    HIPCHK(hipDeviceSynchronize());
    for (auto& r : ctx->prof) {
        float ms = 0.f;
        if (r.e0 && r.e1 && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) { total_ms[r.id] += ms; launches[r.id] += 1; }
        if (r.e0) (void)hipEventDestroy(r.e0);
        if (r.e1) (void)hipEventDestroy(r.e1);
    }
    ctx->prof.clear();
    return KNERF_OK;
}

}  // extern "C"
