// utils_ops.hip -- the NeRFUtils methods as stand-alone device ops (gfx950), for callers that use the reference's
// utility class directly (reference keras_nerf/model/nerf/utils.py).  The train/render path does not use these: there
// the same arithmetic is fused into mlp_fwd.hip / sampler.hip.
#include <hip/hip_runtime.h>
#include <cmath>
#include "../../include/knerf.h"
#include "kernels.h"

namespace knerf {

// NeRFUtils.positional_encoding (utils.py:176-186): [n,3] -> [n, 3+6L] = [x, sin(2^0 x), cos(2^0 x), ...]
__global__ void posenc_kernel(const float* x, float* out, long long n, int L) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int width = 3 + 6 * L;
    if (i >= n * width) return;
    const long long row = i / width;
    const int f = (int)(i % width);
    if (f < 3) { out[i] = x[row * 3 + f]; return; }
    const int k = (f - 3) / 6, r = (f - 3) % 6, c = r % 3;
    const float a = __fmul_rn((float)(1 << k), x[row * 3 + c]);
    out[i] = r < 3 ? sinf(a) : cosf(a);
}

// NeRFUtils.fine_hierarchical_sampling_chunk (utils.py:60-97) on caller-provided mid-points: unsorted samples.
// One thread per ray builds the cdf (left to right, like sampler.hip); Nw <= 256 weights, M mid-points.
__global__ void inverse_cdf_kernel(const float* mids, const float* w, const float* u, float* out, int R, int M, int Nw, int Nf,
                                   int oob_clamp) {
    extern __shared__ float sm[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ray = blockIdx.x * 4 + wv;
    if (ray >= R) return;
    float* cdf = sm + wv * (Nw + 1);
    const float* wr = w + (size_t)ray * Nw;
    float tot = 0.f;
    for (int i = 0; i < Nw; ++i) tot = __fadd_rn(tot, __fadd_rn(wr[i], 1e-5f));
    float acc = 0.f;
    for (int i = 0; i < Nw; ++i) {
        acc = __fadd_rn(acc, __fdiv_rn(__fadd_rn(wr[i], 1e-5f), tot));
        if (lane == 0) cdf[i + 1] = acc;
    }
    if (lane == 0) cdf[0] = 0.f;
    __builtin_amdgcn_wave_barrier();
    const float* mr = mids + (size_t)ray * M;
    for (int j = lane; j < Nf; j += 64) {
        const float uu = u[(size_t)ray * Nf + j];
        int lo = 0, hi = Nw + 1;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] <= uu) lo = mid + 1; else hi = mid; }
        const int below = max(0, lo - 1), above = min(Nw, lo);
        const float cb = cdf[below], ca = cdf[above];
        float mb, ma;
        if (oob_clamp) { mb = mr[min(below, M - 1)]; ma = mr[min(above, M - 1)]; }
        else { mb = below < M ? mr[below] : 0.f; ma = above < M ? mr[above] : 0.f; }
        float denom = __fsub_rn(ca, cb);
        if (denom < 1e-5f) denom = 1.f;
        const float q = __fdiv_rn(__fsub_rn(uu, cb), denom);
        out[(size_t)ray * Nf + j] = __fadd_rn(mb, __fmul_rn(q, __fsub_rn(ma, mb)));
    }
}

// ray(t) = o + t d for every sample (utils.py:193-194), separate multiply and add like the reference's two ops
__global__ __launch_bounds__(256) void ray_points_kernel(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ t,
                                                        long long n, int S, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * 3) return;
    const long long m = i / 3; const int c = (int)(i % 3);
    const long long ray = m / S;
    out[i] = __fadd_rn(o[ray * 3 + c], __fmul_rn(d[ray * 3 + c], t[m]));
}

// tf.image.psnr / tf.image.ssim with their defaults as logged by NeRF.update_and_return_metrics (nerf.py:306-330):
// SSIM = mean over VALID 11x11 Gaussian (sigma 1.5) windows and channels of luminance x contrast-structure, k1 0.01,
// k2 0.03, max_val 1.  One thread per window position; out[b] = {sum of window-channel SSIM terms, sum of squared
// differences}; the two divisions and the log10 happen on the host side of the call.
struct MetricArgs { const float* a; const float* b; float* out; int H, W, C; float g[11]; };
__global__ __launch_bounds__(256) void image_metrics_kernel(MetricArgs m) {
    const int img = blockIdx.y;
    const int wh = m.H - 10, ww = m.W - 10;
    const float* A = m.a + (size_t)img * m.H * m.W * m.C;
    const float* B = m.b + (size_t)img * m.H * m.W * m.C;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    float ssum = 0.f, sq = 0.f;
    if (idx < wh * ww) {
        const int wi = idx / ww, wj = idx % ww;
        const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
        for (int c = 0; c < m.C; ++c) {
            float mx = 0.f, my = 0.f, xx = 0.f, yy = 0.f, xy = 0.f;
            for (int di = 0; di < 11; ++di)
                for (int dj = 0; dj < 11; ++dj) {
                    const float w = m.g[di] * m.g[dj];
                    const size_t p = ((size_t)(wi + di) * m.W + (wj + dj)) * m.C + c;
                    const float x = A[p], y = B[p];
                    mx += w * x; my += w * y; xx += w * x * x; yy += w * y * y; xy += w * x * y;
                }
            const float sxx = xx - mx * mx, syy = yy - my * my, sxy = xy - mx * my;
            ssum += (2.f * mx * my + c1) / (mx * mx + my * my + c1) * ((2.f * sxy + c2) / (sxx + syy + c2));
        }
    }
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < (size_t)m.H * m.W * m.C; p += (size_t)gridDim.x * 256) {
        const float d = A[p] - B[p];
        sq += d * d;
    }
    __shared__ float red[2][4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ssum += __shfl_xor(ssum, o, 64); sq += __shfl_xor(sq, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ssum; red[1][threadIdx.x >> 6] = sq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(m.out + 2 * img, (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
        atomicAdd(m.out + 2 * img + 1, (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
    }
}

// NeRF.update_and_return_metrics (nerf.py:306-330) on the device: six running means (tf.keras.metrics.Mean: total and count)
// in the order coarse_loss, coarse_psnr, coarse_ssim, fine_loss, fine_psnr, fine_ssim; one thread, enqueued behind the two
// image_metrics launches -- nothing returns to the host until somebody reads a result.
struct MetricUpdateArgs { const float* sc; const float* sf; const float* loss; double* state; int n_images, nwin, npix; };
__global__ void metrics_update_kernel(MetricUpdateArgs m) {
    double ps[2] = {0.0, 0.0}, ss[2] = {0.0, 0.0};
    float sq[2] = {0.f, 0.f};
    for (int b = 0; b < m.n_images; ++b) {
        const float* s[2] = {m.sc + 2 * b, m.sf + 2 * b};
        for (int k = 0; k < 2; ++k) {
            ss[k] += (double)(s[k][0] / (float)m.nwin);                              // tf.image.ssim: mean over windows and channels
            ps[k] += (double)(-10.0f * log10f(s[k][1] / (float)m.npix));            // tf.image.psnr, max_val 1
            sq[k] += s[k][1];
        }
    }
    for (int k = 0; k < 2; ++k) {
        double* st = m.state + 6 * k;
        // loss == null: test_step's whole-image mean squared error (nerf.py:484-487) from the same squared-difference sums
        st[0] += m.loss ? (double)m.loss[k] : (double)(sq[k] / ((float)m.npix * (float)m.n_images)); st[1] += 1.0;
        st[2] += ps[k]; st[3] += (double)m.n_images;
        st[4] += ss[k]; st[5] += (double)m.n_images;
    }
}

}  // namespace knerf

using namespace knerf;

extern "C" int knerf_metrics_update(void* stream, const float* sums_coarse, const float* sums_fine, const float* loss, int n_images,
                                    int height, int width, int channels, double* state) {
    if (!sums_coarse || !sums_fine || !state || n_images <= 0 || height < 11 || width < 11 || channels <= 0) return KNERF_ERR_INVALID;
    MetricUpdateArgs m{sums_coarse, sums_fine, loss, state, n_images, (height - 10) * (width - 10) * channels, height * width * channels};
    hipLaunchKernelGGL(metrics_update_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, m);
    return hipGetLastError() == hipSuccess ? KNERF_OK : KNERF_ERR_HIP;
}

extern "C" int knerf_ray_points(void* stream, const float* o, const float* d, const float* t, int n_rays, int n_samples, float* out) {
    if (!o || !d || !t || !out || n_rays <= 0 || n_samples <= 0) return KNERF_ERR_INVALID;
    const long long n = (long long)n_rays * n_samples;
    hipLaunchKernelGGL(ray_points_kernel, dim3((unsigned)((n * 3 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, o, d, t, n, n_samples, out);
    return hipGetLastError() == hipSuccess ? KNERF_OK : KNERF_ERR_HIP;
}

extern "C" int knerf_image_metrics(void* stream, const float* a, const float* b, int n_images, int height, int width, int channels,
                                   float* sums) {
    if (!a || !b || !sums || n_images <= 0 || channels <= 0 || height < 11 || width < 11) return KNERF_ERR_INVALID;
    MetricArgs m{};
    m.a = a; m.b = b; m.out = sums; m.H = height; m.W = width; m.C = channels;
    double g[11], tot = 0.0;
    for (int k = 0; k < 11; ++k) { g[k] = std::exp(-(double)((k - 5) * (k - 5)) / (2.0 * 1.5 * 1.5)); tot += g[k]; }
    for (int k = 0; k < 11; ++k) m.g[k] = (float)(g[k] / tot);
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(sums, 0, (size_t)n_images * 2 * sizeof(float), s) != hipSuccess) return KNERF_ERR_HIP;
    const int windows = (height - 10) * (width - 10);
    hipLaunchKernelGGL(image_metrics_kernel, dim3((windows + 255) / 256, n_images), dim3(256), 0, s, m);
    return hipGetLastError() == hipSuccess ? KNERF_OK : KNERF_ERR_HIP;
}

extern "C" int knerf_positional_encoding(void* stream, const float* x, long long n_rows, int L, float* out) {
    if (!x || !out || n_rows <= 0 || L < 0 || L > 30) return KNERF_ERR_INVALID;
    const long long total = n_rows * (3 + 6 * L);
    hipLaunchKernelGGL(posenc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, out, n_rows, L);
    return hipGetLastError() == hipSuccess ? KNERF_OK : KNERF_ERR_HIP;
}

extern "C" int knerf_composite(void* stream, const float* raw, const float* t, int n_rays, int n_samples, int white_background,
                               float* image, float* depth, float* weights) {
    if (!raw || !t || !image || n_rays <= 0 || n_samples <= 0) return KNERF_ERR_INVALID;
    CompositeArgs ca{};
    ca.raw = raw; ca.t = t; ca.image = image; ca.depth = depth; ca.weights = weights; ca.R = n_rays; ca.S = n_samples;
    ca.white = white_background;
    return launch_composite(ca, (hipStream_t)stream) == hipSuccess ? KNERF_OK : KNERF_ERR_HIP;
}

extern "C" int knerf_inverse_cdf(void* stream, const float* mid_points, const float* weights, const float* u, int n_rays,
                                 int n_mid, int n_weights, int n_samples, int oob_clamp, float* out) {
    if (!mid_points || !weights || !u || !out || n_rays <= 0 || n_mid < 1 || n_weights < 1 || n_weights > 4096 || n_samples < 1)
        return KNERF_ERR_INVALID;
    const size_t lds = 4 * (size_t)(n_weights + 1) * sizeof(float);
    hipLaunchKernelGGL(inverse_cdf_kernel, dim3((n_rays + 3) / 4), dim3(256), lds, (hipStream_t)stream, mid_points, weights, u,
                       out, n_rays, n_mid, n_weights, n_samples, oob_clamp);
    return hipGetLastError() == hipSuccess ? KNERF_OK : KNERF_ERR_HIP;
}
