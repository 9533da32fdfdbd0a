// debug_api.hip -- diagnostics behind include/knerf_debug.h (libknerf_probe.so; tests/ and tools/ only, never the product).
#include <cstring>
#include <vector>

#include "../../include/knerf_debug.h"
#include "ctx.h"

using namespace knerf;

extern "C" {

int knerf_debug_generic_plan(const knerf_config* cfg, int32_t* out, size_t* n) {
    if (!cfg || !n || knerf_param_count_for(cfg) == 0) return KNERF_ERR_INVALID;
    const gen::Plan p = gen::build_plan(cfg->n_layers, cfg->dense_units, cfg->skip_layer, cfg->pos_emb_xyz, cfg->pos_emb_dir);
    std::vector<int32_t> v;
    auto r32 = [](int v) { return (v + 31) / 32 * 32; };
    for (size_t li = 0; li < p.layers.size(); ++li) {
        const gen::Layer& L = p.layers[li];
        // padded input width: the buffer's ld for the trunk layers; the four head layers are evaluated together on the composed
        // matrix (generic.h) and own no buffer, so theirs is the width a stand-alone layer would have
        const int in_ld = (int)li < p.n_layers ? p.buf_ld[L.in_buf] : r32(L.seg[0].width) + (L.n_seg == 2 ? r32(L.seg[1].width) : 0);
        const int32_t row[16] = {L.w_off, L.b_off, L.k_real, L.n_real, in_ld, L.np, L.n_seg, L.seg[0].col0, L.seg[0].width,
                                 L.seg[0].wrow0, L.seg[1].col0, L.seg[1].width, L.seg[1].wrow0, L.relu, L.head, L.out_buf < 0 ? -1 : p.buf_ld[L.out_buf]};
        v.insert(v.end(), row, row + 16);
    }
    if (out) {
        if (*n < v.size()) return KNERF_ERR_INVALID;
        memcpy(out, v.data(), v.size() * sizeof(int32_t));
    }
    *n = v.size();
    return KNERF_OK;
}

int knerf_debug_table(int kind, int32_t* out, size_t* n) {
    if (!n || kind < 0) return KNERF_ERR_INVALID;
    const int shape = kind >> 4;        // kind + 16 * (index in csrc/layout.h KNERF_FUSED_SHAPES); kind 5: {n_layers, skip_layer, dense_units, param_count, pos_emb_xyz, pos_emb_dir} of that shape
    kind &= 15;
    if (shape >= kNumFusedShapes) return KNERF_ERR_INVALID;
    const Tables& t = host_tables(shape);
    const ShapeInfo& si = shape_info(shape);
    const std::vector<int32_t> info = {si.n_layers, si.skip, si.units, si.param_count, si.lx, si.ld};
    const std::vector<int32_t>* v = nullptr;
    switch (kind) {
        case 5: v = &info; break;
        case 0: v = &t.host.fwd; break;
        case 1: v = &t.host.fwd_bias; break;
        case 2: v = &t.host.bwd; break;
        case 3: v = &t.wgrad; break;
        case 4: v = &t.wgrad_off; break;
        default: return KNERF_ERR_INVALID;
    }
    if (out) {
        if (*n < v->size()) return KNERF_ERR_INVALID;
        memcpy(out, v->data(), v->size() * sizeof(int32_t));
    }
    *n = v->size();
    return KNERF_OK;
}

int knerf_debug_buffer(knerf_ctx* ctx, int net, int which, void** dev, size_t* bytes) {
    if (!ctx || !dev || !bytes) return KNERF_ERR_INVALID;
    switch (which) {
        case 0: *dev = ctx->act; *bytes = ctx->act_bytes; break;
        case 1: *dev = ctx->mask; *bytes = ctx->mask_bytes; break;
        case 2: *dev = ctx->dz; *bytes = ctx->dz_bytes; break;
        case 3: *dev = ctx->raw; *bytes = ctx->raw_bytes; break;
        case 4: *dev = ctx->draw; *bytes = ctx->draw_bytes; break;      // sized by the largest TRAINING chunk (raw follows renders too)
        case 5: *dev = ctx->t_f; *bytes = (size_t)ctx->ws_rays * (ctx->cfg.n_coarse + ctx->cfg.n_fine) * sizeof(float); break;
        case 6: *dev = ctx->w_c; *bytes = (size_t)ctx->ws_rays * ctx->cfg.n_coarse * sizeof(float); break;
        case 7:
            if (net != 0 && net != 1) return KNERF_ERR_INVALID;
            *dev = ctx->net[net].w; *bytes = (size_t)(ctx->generic ? ctx->n_params : ctx->si.ext_param_count) * sizeof(float); break;
        // general-shape path (generic.h): every activation buffer [Mp][ld] bf16 / every dZ buffer of the last pass
        case 8: *dev = ctx->gws.act; *bytes = ctx->gplan.act_elems_per_row * ctx->gws.mp * sizeof(unsigned short); break;
        case 9: *dev = ctx->gws.dz; *bytes = ctx->gplan.dz_elems_per_row * ctx->gws.mp * sizeof(unsigned short); break;
        default: return KNERF_ERR_INVALID;
    }
    // not allocated: no pass has run yet, or the buffer belongs to the fused path and this context runs the general-shape kernels
    return *dev ? KNERF_OK : KNERF_ERR_INVALID;
}

}  // extern "C"
