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

namespace {

std::string g_create_error;

// Workgroups per wgrad job in proportion to the job's measured cost per sample tile (cycles per loop iteration from the
// s_memtime stamps of a -DKNERF_WGRAD_STAMPS build, tools/kbench.py): the streaming jobs are HBM-bound, the small ones
// (layer_0, head) latency-bound, so bytes alone mis-balance them.  About one workgroup per CU in total.
// KNERF_WGRAD_COSTS="c0,c1,...,c8" overrides the table (tuning sweeps in one gpurun call).
std::vector<int32_t> build_wgrad_plan(int n_wg) {
    // r02 stamps (gpurun_out/r2b/stamps.json): 1180 / 2080 / 2590 / 1530 cycles per tile for layer_0 / 256x256 / layer_5 / head;
    // head swept 100..200 (1.40 / 1.20 / 1.13 / 1.12 ms per fine launch at 100 / 130 / 160 / 200)
    // layer_1 recomputes h0 (wgrad_l1_recompute, 20 KiB tiles but 22 MFMAs and an LDS exchange per tile): swept 160 / 204 / 240 /
    // 280 -> 1.27 / 1.08 / 1.056 / 1.064 ms per fine launch
    // layer_7 recomputes dz7 (wgrad_l7_recompute, 18 KiB tiles): 204 -> 240; with the copies issued behind the first half's MFMAs
    // the plain jobs gained most, tools/tune_costs.py (coordinate search on the box) moved the others up: coarse + fine launch
    // 1.439 -> 1.382 ms
    int cost[kWgradJobs] = {128, 264, 204, 204, 204, 267, 204, 240, 193};
    if (const char* e = std::getenv("KNERF_WGRAD_COSTS")) {
        int j = 0;
        for (const char* p = e; *p && j < kWgradJobs; ++j) { cost[j] = std::atoi(p); while (*p && *p != ',') ++p; if (*p == ',') ++p; }
    }
    int total = 0;
    for (int j = 0; j < kWgradJobs; ++j) total += cost[j];
    std::vector<int32_t> plan;
    for (int j = 0; j < kWgradJobs; ++j) {
        if (cost[j] <= 0) continue;
        int ns = (cost[j] * (n_wg - 4) + total / 2) / total; if (ns < 1) ns = 1;
        for (int s = 0; s < ns; ++s) { plan.push_back(j); plan.push_back(s); plan.push_back(ns); plan.push_back(0); }
    }
    return plan;
}

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

constexpr size_t kFwdStreamBytes = (size_t)((kFwdBlocks + kPageBlocks - 1) / kPageBlocks * kPageBlocks + kTailPages * kPageBlocks) * 1024;
constexpr size_t kBwdStreamBytes = (size_t)((kBwdBlocks + kPageBlocks - 1) / kPageBlocks * kPageBlocks + kTailPages * kPageBlocks) * 1024;

// compose = false: the caller has already composed the heads of both nets in one launch (knerf_apply_adam)
int repack(knerf_ctx* ctx, int n, hipStream_t s, bool compose = true) {
    Net& N = ctx->net[n];
    if (ctx->generic) {
        HIPCHK(gen::pack_weights(ctx->gplan, N.w, ctx->gnet[n], s));
        return KNERF_OK;
    }
    if (compose) HIPCHK(launch_head_compose(N.w, nullptr, s));   // the composed head behind the parameters (layout.h), then the bf16 streams
    HIPCHK(launch_pack(N.w, ctx->tab.d_fwd, reinterpret_cast<unsigned short*>(N.fwd_stream), (size_t)kFwdBlocks * 512, s));
    HIPCHK(launch_pack(N.w, ctx->tab.d_bwd, reinterpret_cast<unsigned short*>(N.bwd_stream), (size_t)kBwdBlocks * 512, s));
    HIPCHK(launch_gather_f32(N.w, ctx->tab.d_bias, N.bias, (size_t)kFwdBiasTiles * 32, s));
    return KNERF_OK;
}

size_t tiles_for(long long n_samples) {
    size_t t = (size_t)((n_samples + kTile - 1) / kTile);
    return (t + kWaves - 1) / kWaves * kWaves;   // whole workgroups
}

template <class T> void free_dev(T*& p) { if (p) { (void)hipFree(p); p = nullptr; } }

// grow-only workspaces; the zero-fills are enqueued on the caller's stream `s`, the one the consuming kernels run on
// (hipMalloc / hipFree themselves synchronise the device)
// Training workspaces of the fused path (act / mask / dz): `group` coarse-pass regions followed by one fine-pass region, so that
// one weight-gradient launch covers the coarse passes of a group of chunks (launch_wgrad_tiles below).
int ensure_ws(knerf_ctx* ctx, int n_rays, bool train, hipStream_t s, int group = 1) {
    if (n_rays <= ctx->ws_rays && (!train || ctx->ws_train) && (!train || group <= ctx->ws_group)) return KNERF_OK;
    const int R = n_rays > ctx->ws_rays ? n_rays : ctx->ws_rays;
    const int Na = ctx->cfg.n_coarse + ctx->cfg.n_fine;
    train = train || ctx->ws_train;
    if (group < ctx->ws_group && n_rays <= ctx->ws_rays) group = ctx->ws_group;      // a larger chunk size starts from the group asked for: group x size is what costs memory
    HIPCHK(hipStreamSynchronize(s));           // nothing enqueued earlier may still use the buffers that are freed below
    free_dev(ctx->raw); free_dev(ctx->draw); free_dev(ctx->w_c); free_dev(ctx->t_f); free_dev(ctx->img_tmp);
    free_dev(ctx->act); free_dev(ctx->mask); free_dev(ctx->dz);
    ctx->ws_rays = 0;
    const size_t ns = (size_t)R * Na;
    ctx->raw_bytes = ns * 4 * sizeof(float);
    HIPCHK(hipMalloc(&ctx->raw, ctx->raw_bytes));
    HIPCHK(hipMalloc(&ctx->w_c, (size_t)R * ctx->cfg.n_coarse * sizeof(float)));
    HIPCHK(hipMalloc(&ctx->t_f, ns * sizeof(float)));
    HIPCHK(hipMalloc(&ctx->img_tmp, (size_t)R * 8 * sizeof(float)));
    if (ctx->generic) {
        gen::Workspace& g = ctx->gws;
        free_dev(g.act); free_dev(g.dz); free_dev(g.zs); free_dev(g.zc);
        g.mp = gen::padded_rows((long long)ns);
        const size_t ab = ctx->gplan.act_elems_per_row * g.mp * sizeof(unsigned short), zb = ctx->gplan.dz_elems_per_row * g.mp * sizeof(unsigned short);
        HIPCHK(hipMalloc(&g.act, ab));
        HIPCHK(hipMemsetAsync(g.act, 0, ab, s));
        HIPCHK(hipMalloc(&g.zs, g.mp * 32 * sizeof(float)));
        HIPCHK(hipMalloc(&g.zc, g.mp * 32 * sizeof(float)));
        if (train) {
            HIPCHK(hipMalloc(&g.dz, zb));
            HIPCHK(hipMemsetAsync(g.dz, 0, zb, s));
            HIPCHK(hipMalloc(&ctx->draw, ctx->raw_bytes));
        }
    } else if (train) {
        // group 1: the coarse and the fine pass of a chunk share one region (each pass's weight gradients follow it at once)
        const size_t tiles = group == 1 ? tiles_for((long long)ns) : (size_t)group * tiles_for((long long)R * ctx->cfg.n_coarse) + tiles_for((long long)ns);
        ctx->act_bytes = tiles * kActTileBytes; ctx->mask_bytes = tiles * kMaskTileBytes; ctx->dz_bytes = tiles * kDzTileBytes;
        HIPCHK(hipMalloc(&ctx->draw, ctx->raw_bytes));
        HIPCHK(hipMalloc(&ctx->act, ctx->act_bytes));
        HIPCHK(hipMalloc(&ctx->mask, ctx->mask_bytes));
        HIPCHK(hipMalloc(&ctx->dz, ctx->dz_bytes));
        HIPCHK(hipMemsetAsync(ctx->dz, 0, ctx->dz_bytes, s));     // block kDzHead+1 is never written and must read 0
        HIPCHK(hipMemsetAsync(ctx->act, 0, ctx->act_bytes, s));
    }
    ctx->ws_rays = R; ctx->ws_train = train; ctx->ws_group = train ? group : ctx->ws_group;
    return KNERF_OK;
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

int check_net(knerf_ctx* ctx, int net) {
    if (!ctx) return KNERF_ERR_INVALID;
    if (net != KNERF_COARSE && net != KNERF_FINE) return fail(ctx, KNERF_ERR_INVALID, "net must be 0 (coarse) or 1 (fine)");
    return KNERF_OK;
}

// forward (+ optional training half) of one net on given t-values; leaves raw/draw/act/dz in the workspace
// weight gradients of `net` over n_tiles sample tiles starting at tile `tile0` of the act / mask / dz workspaces
int launch_wgrad_tiles(knerf_ctx* ctx, hipStream_t s, int net, size_t tile0, size_t n_tiles) {
    WgradArgs wa{};
    wa.act = ctx->act + tile0 * kActTileBytes; wa.dz = ctx->dz + tile0 * kDzTileBytes; wa.mask = ctx->mask + tile0 * kMaskTileBytes;
    wa.grad = ctx->net[net].g; wa.aux = ctx->net[net].aux; wa.dst = ctx->tab.d_wgrad;
    wa.fwd_stream = ctx->net[net].fwd_stream; wa.bias = ctx->net[net].bias; wa.bwd_stream = ctx->net[net].bwd_stream;
    wa.n_tiles = (long long)n_tiles;
    wa.plan = ctx->tab.d_plan; wa.n_plan = ctx->tab.n_plan; wa.net = net == KNERF_COARSE ? 0 : 1;
    for (int j = 0; j <= kWgradJobs; ++j) wa.job_off[j] = ctx->tab.wgrad_off[j];
    ProfScope ps(ctx, s, net == KNERF_COARSE ? P_WGRAD_C : P_WGRAD_F);
    HIPCHK(launch_wgrad(wa, s));
    return KNERF_OK;
}

// tile0: first tile of this pass in the training workspaces; wgrad_now: launch the weight-gradient kernel for just this pass
// (false: the caller launches it later over several passes, launch_wgrad_tiles)
int run_pass(knerf_ctx* ctx, hipStream_t s, int net, const float* o, const float* d, const float* t, int R, int S,
             float* image, float* depth, float* weights, const float* target, float inv_chunks, float* loss,
             size_t tile0 = 0, bool wgrad_now = true) {
    const bool train = target != nullptr;
    FwdArgs fa{};
    fa.stream = ctx->net[net].fwd_stream; fa.bias = ctx->net[net].bias;
    fa.o = o; fa.d = d; fa.t = t; fa.raw = ctx->raw;
    fa.act = ctx->act ? ctx->act + tile0 * kActTileBytes : nullptr; fa.mask = ctx->mask ? ctx->mask + tile0 * kMaskTileBytes : nullptr;
    fa.n_samples = (long long)R * S; fa.S = S; fa.net = net == KNERF_COARSE ? 0 : 1;
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
    { ProfScope ps(ctx, s, P_COMPOSITE); HIPCHK(launch_composite(ca, s)); }
    if (train && ctx->generic) {
        ProfScope ps(ctx, s, net == KNERF_COARSE ? P_BWD_C : P_BWD_F);
        HIPCHK(gen::backward(ctx->gplan, ctx->gws, ctx->gnet[net], ctx->raw, ctx->draw, fa.n_samples, ctx->net[net].g, s));
    } else if (train) {
        BwdArgs ba{};
        ba.stream = ctx->net[net].bwd_stream; ba.raw = ctx->raw; ba.draw = ctx->draw; ba.mask = fa.mask; ba.dz = ctx->dz + tile0 * kDzTileBytes;
        ba.n_samples = fa.n_samples; ba.net = fa.net;
        { ProfScope ps(ctx, s, net == KNERF_COARSE ? P_BWD_C : P_BWD_F); HIPCHK(launch_mlp_bwd(ba, s)); }
        if (wgrad_now) { if (int r = launch_wgrad_tiles(ctx, s, net, tile0, tiles_for(fa.n_samples))) return r; }
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
    HIPCHK(launch_head_expand(ctx->net[0].w, ctx->net[0].aux, ctx->net[0].g, ctx->net[1].w, ctx->net[1].aux, ctx->net[1].g, s));
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
                     float* c_image, float* f_image, int slot = 0, int group = 1) {
    const int Nc = ctx->cfg.n_coarse, Na = Nc + ctx->cfg.n_fine;
    float* ci = c_image ? c_image : ctx->img_tmp;
    float* fi = f_image ? f_image : ctx->img_tmp + (size_t)n_rays * 4;
    float* ls = loss ? loss : ctx->loss_tmp;
    const size_t tc = ctx->generic ? 0 : tiles_for((long long)n_rays * Nc);
    const size_t tile0_c = group == 1 ? 0 : slot * tc, tile0_f = group == 1 ? 0 : group * tc;
    if (int r = run_pass(ctx, s, KNERF_COARSE, o, d, t, n_rays, Nc, ci, nullptr, ctx->w_c, target, inv_chunks, ls, tile0_c, group == 1)) return r;
    if (int r = knerf_sample_fine(ctx, s, t, ctx->w_c, u, seed, 0, ray_offset, n_rays, ctx->t_f)) return r;
    return run_pass(ctx, s, KNERF_FINE, o, d, ctx->t_f, n_rays, Na, fi, nullptr, nullptr, target, inv_chunks, ls + 1, tile0_f, true);
}

// chunks per COARSE weight-gradient launch of knerf_train_batch: up to KNERF_WGRAD_GROUP_MAX (4), within the budget
// (KNERF_WGRAD_GROUP_GB, default 40; 0 = one launch per chunk) and a quarter of the free device memory
int wgrad_group_for(knerf_ctx* ctx, int n_rays, int n_chunks) {
    if (ctx->generic || n_chunks <= 1) return 1;
    double budget = 40.0;
    if (const char* e = std::getenv("KNERF_WGRAD_GROUP_GB")) budget = std::atof(e);
    if (budget <= 0) return 1;
    const int Nc = ctx->cfg.n_coarse, Na = Nc + ctx->cfg.n_fine;
    const double per_chunk = (double)tiles_for((long long)n_rays * Nc) * (kActTileBytes + kMaskTileBytes + kDzTileBytes);   // one coarse region
    (void)Na;
    size_t free_b = 0, total_b = 0;
    double avail = budget * 1e9;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const double mine = ctx->ws_train ? (double)(ctx->act_bytes + ctx->mask_bytes + ctx->dz_bytes) : 0.0;   // what a re-allocation gives back
        const double cap = 0.25 * ((double)free_b + mine);
        if (cap < avail) avail = cap;
    }
    int g = (int)(avail / per_chunk);
    int g_max = 4;                                  // coarse launches of more than ~4 x 8192 tiles gain nothing more
    if (const char* e = std::getenv("KNERF_WGRAD_GROUP_MAX")) { const int v = std::atoi(e); if (v > 0) g_max = v; }
    if (g > g_max) g = g_max;
    if (g > n_chunks) g = n_chunks;
    if (g <= 1) return 1;
    const int n_groups = (n_chunks + g - 1) / g;
    return (n_chunks + n_groups - 1) / n_groups;      // the smallest group that needs no more launches
}

}  // namespace

extern "C" {

size_t knerf_param_count(void) { return (size_t)kParamCount; }

size_t knerf_param_count_for(const knerf_config* cfg) {
    if (!cfg || cfg->n_layers < 1 || cfg->dense_units < 2 || cfg->skip_layer < 1 || cfg->pos_emb_xyz < 0 || cfg->pos_emb_dir < 0) return 0;
    return (size_t)gen::param_count(cfg->n_layers, cfg->dense_units, cfg->skip_layer, cfg->pos_emb_xyz, cfg->pos_emb_dir);
}

const char* knerf_last_error(const knerf_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int knerf_create(const knerf_config* cfg, knerf_ctx** out) {
    knerf_ctx* ctx = nullptr;
    if (!cfg || !out) return fail(nullptr, KNERF_ERR_INVALID, "null argument");
    if (cfg->n_layers < 1 || cfg->n_layers > 64 || cfg->dense_units < 2 || cfg->dense_units > 4096 || cfg->skip_layer < 1 ||
        cfg->pos_emb_xyz < 0 || cfg->pos_emb_xyz > 32 || cfg->pos_emb_dir < 0 || cfg->pos_emb_dir > 32)
        return fail(nullptr, KNERF_ERR_INVALID, "need 1 <= n_layers <= 64, 2 <= dense_units <= 4096, skip_layer >= 1, 0 <= pos_emb_* <= 32");
    if (cfg->n_coarse < 2 || cfg->n_coarse > 256 || cfg->n_fine < 0 || cfg->n_coarse + cfg->n_fine > 512)
        return fail(nullptr, KNERF_ERR_INVALID, "need 2 <= n_coarse <= 256 and n_coarse + n_fine <= 512");
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
    ctx->generic = !gen::is_default_shape(cfg->n_layers, cfg->dense_units, cfg->skip_layer, cfg->pos_emb_xyz, cfg->pos_emb_dir) ||
                   std::getenv("KNERF_FORCE_GENERIC") != nullptr;      // the override lets tests run the default shape through both paths
    if (ctx->generic) {
        ctx->gplan = gen::build_plan(cfg->n_layers, cfg->dense_units, cfg->skip_layer, cfg->pos_emb_xyz, cfg->pos_emb_dir);
        ctx->n_params = ctx->gplan.n_params;
    }
    const size_t NP = (size_t)ctx->n_params;
    const Tables& ht = host_tables();
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
    {
        std::vector<int32_t> plan = build_wgrad_plan(prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256);
        ctx->tab.n_plan = (int)plan.size() / 4;
        CREATECHK(up(ctx->tab.d_plan, plan));
    }
    CREATECHK(hipMalloc(&ctx->grads, 2 * NP * sizeof(float)));
    CREATECHK(hipMemset(ctx->grads, 0, 2 * NP * sizeof(float)));
    CREATECHK(hipMalloc(&ctx->aux, 2 * (size_t)kAuxCount * sizeof(float)));
    CREATECHK(hipMemset(ctx->aux, 0, 2 * (size_t)kAuxCount * sizeof(float)));
    CREATECHK(hipMalloc(&ctx->d_flag, sizeof(int)));
    CREATECHK(hipMemset(ctx->d_flag, 0, sizeof(int)));
    CREATECHK(hipHostMalloc(&ctx->h_status, 2 * sizeof(int), hipHostMallocDefault));   // written by the device (optim.hip step_status)
    ctx->h_status[0] = ctx->h_status[1] = 0;
    ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    CREATECHK(hipMalloc(&ctx->loss_tmp, 2 * sizeof(float)));
    for (int n = 0; n < 2; ++n) {
        Net& N = ctx->net[n];
        const size_t NW = ctx->generic ? NP : (size_t)kExtParamCount;          // fused path: parameters + composed head (layout.h)
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
        CREATECHK(hipMalloc(&N.fwd_stream, kFwdStreamBytes));
        CREATECHK(hipMalloc(&N.bwd_stream, kBwdStreamBytes));
        CREATECHK(hipMemset(N.fwd_stream, 0, kFwdStreamBytes));
        CREATECHK(hipMemset(N.bwd_stream, 0, kBwdStreamBytes));
        CREATECHK(hipMalloc(&N.bias, kFwdBiasTiles * 32 * sizeof(float)));
        CREATECHK(hipMemset(N.bias, 0, kFwdBiasTiles * 32 * sizeof(float)));
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
    free_dev(ctx->grads); free_dev(ctx->aux); free_dev(ctx->d_flag); free_dev(ctx->loss_tmp);
    if (ctx->h_status) { (void)hipHostFree(ctx->h_status); ctx->h_status = nullptr; }
    free_dev(ctx->tab.d_fwd); free_dev(ctx->tab.d_bias); free_dev(ctx->tab.d_bwd); free_dev(ctx->tab.d_wgrad); free_dev(ctx->tab.d_plan);
    free_dev(ctx->raw); free_dev(ctx->draw); free_dev(ctx->w_c); free_dev(ctx->t_f); free_dev(ctx->img_tmp);
    free_dev(ctx->act); free_dev(ctx->mask); free_dev(ctx->dz);
    free_dev(ctx->gws.act); free_dev(ctx->gws.dz); free_dev(ctx->gws.zs); free_dev(ctx->gws.zc);
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
        ctx->call_plan = gen::build_plan(c.n_layers, c.dense_units, c.skip_layer, c.pos_emb_xyz, c.pos_emb_dir);
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
    if (!ctx) return KNERF_ERR_INVALID;
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
    if (!ctx) return KNERF_ERR_INVALID;
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
    if (!ctx) return KNERF_ERR_INVALID;
    if (ray_chunks <= 0 || n_rays <= 0 || n_rays % ray_chunks != 0)
        return fail(ctx, KNERF_ERR_INVALID, "render_batch: ray_chunks must be a divisor of the number of rays");   // nerf.py:100
    const size_t Nc = (size_t)ctx->cfg.n_coarse, Nf = (size_t)ctx->cfg.n_fine, Na = Nc + Nf;
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
    if (!ctx) return KNERF_ERR_INVALID;
    if (!o || !d || !t || !target || n_rays <= 0) return fail(ctx, KNERF_ERR_INVALID, "train_chunk: null/empty argument");
    hipStream_t s = (hipStream_t)stream;
    if (int r = ensure_ws(ctx, n_rays, true, s)) return r;
    if (int r = train_chunk_impl(ctx, s, o, d, t, target, u, seed, ray_offset, n_rays, inv_chunks, loss, c_image, f_image)) return r;
    return expand_head_grads(ctx, s);
}

int knerf_train_batch(knerf_ctx* ctx, void* stream, const float* o, const float* d, const float* t, const float* target,
                      const float* u, uint64_t seed, int n_rays, int ray_chunks, float* loss, float* c_image, float* f_image) {
    if (!ctx) return KNERF_ERR_INVALID;
    if (!o || !d || !t || !target) return fail(ctx, KNERF_ERR_INVALID, "train_batch: null argument");
    if (ray_chunks <= 0 || n_rays <= 0 || n_rays % ray_chunks != 0)
        return fail(ctx, KNERF_ERR_INVALID, "train_batch: ray_chunks must be a divisor of the number of rays");   // nerf.py:100
    hipStream_t s = (hipStream_t)stream;
    const int C = n_rays / ray_chunks, Nc = ctx->cfg.n_coarse, Nf = ctx->cfg.n_fine;
    int G = ctx->ws_train && ctx->ws_rays >= ray_chunks && ctx->ws_group > 1 ? (ctx->ws_group < C ? ctx->ws_group : C) : wgrad_group_for(ctx, ray_chunks, C);
    if (int r = ensure_ws(ctx, ray_chunks, true, s, G)) {
        if (G == 1) return r;
        G = 1;                                                      // the group did not fit: one chunk per wgrad launch
        if (int r1 = ensure_ws(ctx, ray_chunks, true, s, 1)) return r1;
    }
    const size_t tc = tiles_for((long long)ray_chunks * Nc);
    for (int i = 0; i < C; ++i) {
        const size_t r0 = (size_t)i * ray_chunks;
        const int slot = i % G;
        const bool last = slot == G - 1 || i == C - 1;
        if (int r = train_chunk_impl(ctx, s, o + r0 * 3, d + r0 * 3, t + r0 * Nc, target + r0 * 3, u ? u + r0 * Nf : nullptr, seed,
                                     (uint64_t)r0, ray_chunks, 1.0f / (float)C, loss, c_image ? c_image + r0 * 3 : nullptr,
                                     f_image ? f_image + r0 * 3 : nullptr, slot, G))
            return r;
        if (G > 1 && last) {                                        // the group's coarse weight gradients in one launch
            if (int r = launch_wgrad_tiles(ctx, s, KNERF_COARSE, 0, (size_t)(slot + 1) * tc)) return r;
        }
    }
    return expand_head_grads(ctx, s);
}

int knerf_apply_adam(knerf_ctx* ctx, void* stream) {
    if (!ctx) return KNERF_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    // Finite check first, on the device; the Adam kernels read its flag and leave weights and slots untouched when it is
    // set (the reference aborts fit at nerf.py:381-382).  Nothing here waits for the GPU: the outcome reaches the host through
    // a pinned status word (knerf_poll_nonfinite).
    HIPCHK(hipMemsetAsync(ctx->d_flag, 0, sizeof(int), s));
    HIPCHK(launch_check_finite(ctx->grads, 2 * ctx->n_params, ctx->d_flag, s));
    ctx->step += 1;
    const double b1 = ctx->cfg.beta1, b2 = ctx->cfg.beta2;
    const float lr_t = (float)((double)ctx->cfg.lr * std::sqrt(1.0 - std::pow(b2, ctx->step)) / (1.0 - std::pow(b1, ctx->step)));
    ProfScope ps(ctx, s, P_ADAM);
    for (int n = 0; n < 2; ++n) {
        AdamArgs a{};
        a.w = ctx->net[n].w; a.m = ctx->net[n].m; a.v = ctx->net[n].v; a.g = ctx->net[n].g; a.n = ctx->n_params;
        a.lr_t = lr_t; a.b1 = ctx->cfg.beta1; a.b2 = ctx->cfg.beta2; a.eps = ctx->cfg.epsilon; a.nonfinite = ctx->d_flag;
        HIPCHK(launch_adam(a, s));
    }
    if (!ctx->generic) HIPCHK(launch_head_compose(ctx->net[0].w, ctx->net[1].w, s));      // both nets' heads in one launch
    for (int n = 0; n < 2; ++n)
        if (int r = repack(ctx, n, s, false)) return r;
    HIPCHK(launch_step_status(ctx->d_flag, ctx->h_status, s));
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

int knerf_step_count(const knerf_ctx* ctx) { return ctx ? ctx->step : KNERF_ERR_INVALID; }
int knerf_set_step_count(knerf_ctx* ctx, int step) {
    if (!ctx || step < 0) return KNERF_ERR_INVALID;
    ctx->step = step;
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
