// composite.hip -- alpha-compositing volume quadrature, its loss and its backward, one wavefront per ray (gfx950).
//
// Forward restates NeRFUtils.render_image_depth_chunk (reference keras_nerf/model/nerf/utils.py:16-58):
//   delta_i = t_{i+1}-t_i, last delta = 1e-10; alpha = 1-exp(-sigma*delta); e = 1-alpha;
//   T = cumprod(e + 1e-10, exclusive); w = alpha*T; image = sum w*rgb (+ 1 - sum w on white); depth = sum w*t; clip.
// Training adds the MSE of train_single.py:127 / train.py:130-136 and the gradient wrt (rgb, sigma):
//   clip passes gradient on [0,1] inclusive; dL/de_k = (sum_{i>k} dw_i w_i) / (e_k + 1e-10)  (TF cumprod gradient).
// The transmittance is a wave-level exclusive product scan (lane-local run of C samples, then 6 shuffle steps);
// the backward suffix sum is the mirrored scan.
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace knerf {

typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int C>
__global__ __launch_bounds__(256) void composite_kernel(CompositeArgs a) {
    __shared__ float s_loss[4];
    __shared__ int s_cnt[4], s_base[2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ray = blockIdx.x * 4 + wv;
    const bool training = a.draw != nullptr;
    if (training && lane == 0) { s_loss[wv] = 0.f; s_cnt[wv] = 0; }
    if (ray >= a.R) {         // whole wave leaves together; in training it still joins the block's loss reduction and list append
        if (training) { __syncthreads(); if (a.tile_list) { __syncthreads(); __syncthreads(); } }
        return;
    }
    const int S = a.S;
    const float eps = 1e-10f;
    const f32x4* raw = reinterpret_cast<const f32x4*>(a.raw) + (size_t)ray * S;
    const float* t = a.t + (size_t)ray * S;

    float r[C], g[C], b[C], sg[C], tt[C], dl[C], ex[C], al[C], x[C], T[C], w[C];
    float run = 1.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = lane * C + c;
        const bool ok = i < S;
        f32x4 v = ok ? raw[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        r[c] = v[0]; g[c] = v[1]; b[c] = v[2]; sg[c] = v[3];
        tt[c] = ok ? t[i] : 0.f;
        const float tn = (i + 1 < S) ? t[i + 1] : 0.f;
        dl[c] = (i + 1 < S) ? __fsub_rn(tn, tt[c]) : eps;
        ex[c] = expf(-__fmul_rn(sg[c], dl[c]));
        al[c] = ok ? __fsub_rn(1.f, ex[c]) : 0.f;
        x[c] = ok ? __fadd_rn(__fsub_rn(1.f, al[c]), eps) : 1.f;
        T[c] = run;               // lane-local exclusive product
        run *= x[c];
    }
    // wave exclusive product scan of the lane totals
    float inc = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float u = __shfl_up(inc, o, 64);
        if (lane >= o) inc *= u;
    }
    float excl = __shfl_up(inc, 1, 64);
    if (lane == 0) excl = 1.f;
    float sr = 0.f, sgc = 0.f, sb = 0.f, sd = 0.f, sw = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        T[c] *= excl;
        w[c] = al[c] * T[c];
        sr += w[c] * r[c]; sgc += w[c] * g[c]; sb += w[c] * b[c];
        sd += w[c] * tt[c]; sw += w[c];
    }
    sr = wave_sum(sr); sgc = wave_sum(sgc); sb = wave_sum(sb); sd = wave_sum(sd); sw = wave_sum(sw);
    float pre[3] = {sr, sgc, sb};
    if (a.white & 1) { const float bg = 1.f - sw; pre[0] += bg; pre[1] += bg; pre[2] += bg; }
    float img[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) img[k] = (a.white & 2) ? pre[k] : fminf(fmaxf(pre[k], 0.f), 1.f);   // bit 1: the non-chunk twin (utils.py:99-134) does not clip
    if (lane < 3) a.image[(size_t)ray * 3 + lane] = lane == 0 ? img[0] : (lane == 1 ? img[1] : img[2]);
    if (a.depth && lane == 0) a.depth[ray] = sd;
    if (a.weights) {
#pragma unroll
        for (int c = 0; c < C; ++c) { const int i = lane * C + c; if (i < S) a.weights[(size_t)ray * S + i] = w[c]; }
    }
    if (!training) return;

    // ---- loss + backward
    float gi[3], l2 = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float df = img[k] - a.target[(size_t)ray * 3 + k];
        l2 += df * df;
        gi[k] = (pre[k] >= 0.f && pre[k] <= 1.f) ? a.grad_scale * df : 0.f;
    }
    // one atomic per block (4 rays) instead of one per ray: 4096 same-address atomics cost more than the kernel's work
    if (lane == 0) s_loss[wv] = l2 * a.loss_scale;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float l4 = (s_loss[0] + s_loss[1]) + (s_loss[2] + s_loss[3]);
        if (a.loss_partial) a.loss_partial[blockIdx.x] = l4;       // deterministic mode: summed in order by loss_reduce_kernel
        else atomicAdd(a.loss, l4);
    }
    const float gsum = (a.white & 1) ? (gi[0] + gi[1] + gi[2]) : 0.f;
    float dw[C], pr[C];
    float suffix = 0.f;            // lane-local exclusive suffix sums of dw*w, built right to left
    float Ql[C];
#pragma unroll
    for (int c = C - 1; c >= 0; --c) {
        dw[c] = gi[0] * r[c] + gi[1] * g[c] + gi[2] * b[c] - gsum;
        pr[c] = dw[c] * w[c];
        Ql[c] = suffix;
        suffix += pr[c];
    }
    float incs = suffix;           // wave reverse inclusive scan of lane totals
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float u = __shfl_down(incs, o, 64);
        if (lane + o < 64) incs += u;
    }
    float excls = __shfl_down(incs, 1, 64);
    if (lane == 63) excls = 0.f;
    f32x4* draw = reinterpret_cast<f32x4*>(a.draw) + (size_t)ray * S;
    unsigned live = 0;             // bit k: this lane holds a sample of the ray's k-th 32-sample tile that the backward needs
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = lane * C + c;
        if (i < S) {
            const float Q = Ql[c] + excls;
            const float dalpha = dw[c] * T[c] - Q / x[c];
            const float dsig = dalpha * dl[c] * ex[c];
            const f32x4 v = f32x4{w[c] * gi[0], w[c] * gi[1], w[c] * gi[2], dsig};
            draw[i] = v;
            // A sample is DEAD when the dgrad chain's input is exactly zero for it: dz_rgb = drgb * rgb (1 - rgb) with drgb == 0, and
            // dz_sigma = [sigma > 0] dsigma with dsigma == 0 or the ReLU gate closed (sigma == 0).  Every dZ of the sample is then
            // +-0 and it adds exactly nothing to any weight or bias gradient (mlp_bwd / wgrad skip whole tiles of such samples).
            const bool dead = v[0] == 0.f && v[1] == 0.f && v[2] == 0.f && (v[3] == 0.f || sg[c] == 0.f);
            if (!dead) live |= 1u << (i >> 5);
        }
    }
    if (a.tile_flags || a.tile_list) {      // S % 32 == 0 (checked by the caller): tiles do not straddle rays, S / 32 <= 32 of them per ray (the bits of `live`)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) live |= __shfl_xor(live, o, 64);
    }
    const int nt = S >> 5;
    if (a.tile_flags && lane < nt) a.tile_flags[(size_t)ray * nt + lane] = (live >> lane) & 1u;     // deterministic mode: compact_tiles_kernel sorts them
    if (a.tile_list) {
        // default mode: the live tiles are appended to the pass's list right here (no compaction launch): the block's four rays
        // reserve their entries with ONE atomic on the pass's counter (zeroed by the caller), the list comes out in block order,
        // i.e. nearly ascending.  A coarse pass of a grouped launch appends to the group's list (indices + tile_off2) as well.
        if (lane == 0) s_cnt[wv] = __popc(live);
        __syncthreads();
        if (threadIdx.x == 0) {
            const int tot = (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
            s_base[0] = tot ? atomicAdd(a.tile_count, tot) : 0;
            s_base[1] = (tot && a.tile_list2) ? atomicAdd(a.tile_count2, tot) : 0;
        }
        __syncthreads();
        int off = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) off += k < wv ? s_cnt[k] : 0;
        if (lane < nt && ((live >> lane) & 1u)) {
            const int r = off + __popc(live & ((1u << lane) - 1u)), id = ray * nt + lane;
            a.tile_list[s_base[0] + r] = id;
            if (a.tile_list2) a.tile_list2[s_base[1] + r] = id + a.tile_off2;
        }
    }
}

// ---- dead-tile skipping: flags -> ASCENDING list of live tiles and its length.  One workgroup walks the flags in chunks of 1024
// (eight chunks' loads in flight per thread), ranks the live ones with a ballot per wavefront and a 16-entry prefix through LDS, and
// keeps the running total in a register: no atomics, nothing to zero beforehand, the same list every time (the deterministic
// mode's per-workgroup ranges need the order; the default mode gets a launch-independent tile order for free).  ~10 us for the
// 24,576 tiles of a fine pass.
__global__ __launch_bounds__(1024) void compact_tiles_kernel(const int* flags, int n, int period, int real, int* list, int* count, long long* stats) {
    __shared__ int s_wave[2][16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    int base = 0, par = 0;
    for (int c0 = 0; c0 < n; c0 += 8 * 1024) {
        int f[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = c0 + k * 1024 + tid;
            f[k] = (i < n && (i % period) < real) ? flags[i] : 0;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (c0 + k * 1024 >= n) break;                       // uniform
            const bool live = f[k] != 0;
            const unsigned long long m = __ballot(live);
            if (lane == 0) s_wave[par][wv] = __popcll(m);
            __syncthreads();
            int off = 0, tot = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) { const int v = s_wave[par][w]; off += w < wv ? v : 0; tot += v; }
            if (live) list[base + off + __popcll(m & below)] = c0 + k * 1024 + tid;
            base += tot;
            par ^= 1;                                            // the other buffer next time: one barrier per chunk is enough
        }
    }
    if (tid == 0) {
        *count = base;
        if (stats) { stats[0] += base; stats[1] += (long long)(n / period) * real + ((n % period) < real ? (n % period) : real); }
    }
}
hipError_t launch_compact_tiles(const int* flags, int n, int period, int real, int* list, int* count, long long* stats, hipStream_t stream) {
    hipLaunchKernelGGL(compact_tiles_kernel, dim3(1), dim3(1024), 0, stream, flags, n, period, real, list, count, stats);
    return hipGetLastError();
}

// ---- deterministic mode: the chunk's loss terms added in a fixed order (one wavefront)
__global__ __launch_bounds__(64) void loss_reduce_kernel(const float* partial, int n, float* loss) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) s += partial[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) *loss += s;
}
hipError_t launch_loss_reduce(const float* partial, int n, float* loss, hipStream_t stream) {
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(64), 0, stream, partial, n, loss);
    return hipGetLastError();
}

hipError_t launch_composite(const CompositeArgs& a, hipStream_t stream) {
    const int grid = (a.R + 3) / 4;
    const int C = (a.S + 63) / 64;
    switch (C) {
        case 1: hipLaunchKernelGGL(composite_kernel<1>, dim3(grid), dim3(256), 0, stream, a); break;
        case 2: hipLaunchKernelGGL(composite_kernel<2>, dim3(grid), dim3(256), 0, stream, a); break;
        case 3: hipLaunchKernelGGL(composite_kernel<3>, dim3(grid), dim3(256), 0, stream, a); break;
        case 4: hipLaunchKernelGGL(composite_kernel<4>, dim3(grid), dim3(256), 0, stream, a); break;
        case 5: case 6: case 7: case 8: hipLaunchKernelGGL(composite_kernel<8>, dim3(grid), dim3(256), 0, stream, a); break;
        // up to 1024 samples per ray (--num_coarse_samples 512 --num_fine_samples 512): a lane's run of 12 / 16 samples stays in registers
        case 9: case 10: case 11: case 12: hipLaunchKernelGGL(composite_kernel<12>, dim3(grid), dim3(256), 0, stream, a); break;
        case 13: case 14: case 15: case 16: hipLaunchKernelGGL(composite_kernel<16>, dim3(grid), dim3(256), 0, stream, a); break;
        default: return hipErrorInvalidValue;   // more than 1024 samples per ray
    }
    return hipGetLastError();
}

}  // namespace knerf
