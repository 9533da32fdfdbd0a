// sampler.hip -- inverse-CDF hierarchical sampling + merge with the coarse t-values, one wavefront per ray (gfx950).
//
// Restates NeRF._predict_and_render_chunk's fine branch (reference keras_nerf/model/nerf/nerf.py:182-191) and
// NeRFUtils.fine_hierarchical_sampling_chunk (utils.py:60-97):
//   mids = 0.5*(t[1:]+t[:-1]);  w += 1e-5; pdf = w/sum(w); cdf = [0, cumsum(pdf)];
//   idx = searchsorted(cdf, u, 'right'); below = max(0, idx-1); above = min(Nc, idx);
//   gather cdf and mids (mids has only Nc-1 entries: an out-of-range gather yields 0 -- tf.gather on GPU -- or is
//   clamped); denom<1e-5 -> 1; sample = m_b + (u-cdf_b)/denom*(m_a-m_b);
//   t_all = sort(concat(t_coarse, samples)).
// u is either caller-provided (parity tests) or Philox4x32-10 keyed by (seed; block, ray, stream).
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace knerf {

__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
    const unsigned long long p0 = (unsigned long long)c[0] * 0xD2511F53ull;
    const unsigned long long p1 = (unsigned long long)c[2] * 0xCD9E8D57ull;
    const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0, hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
    const unsigned n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
}
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int i = 0; i < 10; ++i) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
}
// u[ray][j]: counter (j/4, ray, stream, 0), key (seed_lo, seed_hi), word j%4, u = (x>>8) * 2^-24
__device__ __forceinline__ float philox_u(unsigned long long seed, unsigned stream_id, unsigned ray, int j) {
    unsigned c[4] = {(unsigned)(j >> 2), ray, stream_id, 0u};
    philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
    const unsigned x = c[j & 3];
    return (float)(x >> 8) * 5.9604644775390625e-08f;
}

// Any sample count the reference's CLI can ask for (train_single.py:28-29: --num_coarse_samples / --num_fine_samples are free
// integers) up to kMaxCoarse coarse and kMaxAll samples per ray: the per-wave tables live in DYNAMIC LDS sized by the launch
// (t: Nc, cdf: Nc + 1, all: max(Nc + Nf, 2 Nc) floats -- the fine half of `all` doubles as scratch for the pdf), 20 KB per
// workgroup at 64 + 128, 32.8 KB at the limits -- the limits of knerf_create and include/knerf.h (the constants used to
// advertise 1024 / 4096 and an opt-in to 160 KB of LDS that no context could reach).
constexpr int kMaxCoarse = 512, kMaxAll = 1024;
__host__ __device__ constexpr int sampler_wave_floats(int Nc, int Nf) { const int na = Nc + Nf > 2 * Nc ? Nc + Nf : 2 * Nc; return Nc + (Nc + 1) + na + 3; }
static_assert(4 * sampler_wave_floats(kMaxCoarse, kMaxAll - kMaxCoarse) * sizeof(float) <= 48 * 1024, "the sampler's tables fit the default dynamic-LDS limit");

__global__ __launch_bounds__(256) void sample_fine_kernel(SampleArgs a) {
    extern __shared__ float s_dyn[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ray = blockIdx.x * 4 + wv;
    if (ray >= a.R) return;
    const int Nc = a.Nc, Nf = a.Nf, Na = Nc + Nf;
    float* tt = s_dyn + (size_t)wv * sampler_wave_floats(Nc, Nf); float* cdf = tt + Nc; float* all = cdf + Nc + 1;
    const float* tc = a.t_coarse + (size_t)ray * Nc;
    const float* wc = a.w_coarse + (size_t)ray * Nc;

    // stage t and w + 1e-5 in LDS
    float* wl = all + Nc;            // scratch: the fine half of `all` is not written until the cdf exists
    for (int i = lane; i < Nc; i += 64) { const float tv = tc[i]; tt[i] = tv; all[i] = tv; wl[i] = __fadd_rn(wc[i], 1e-5f); }
    __builtin_amdgcn_wave_barrier();
    // total and cdf = [0, cumsum(pdf)] strictly left to right (the oracle's declared order): the knot positions decide
    // searchsorted and the denom<1e-5 branch bit for bit.  The two running sums are serial by definition (every lane carries
    // the same chain); the Nc divisions are not -- lane i forms pdf_i = w_i / total once, in parallel.
    float tot = 0.f;
    for (int i = 0; i < Nc; ++i) tot = __fadd_rn(tot, wl[i]);
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < Nc; i += 64) wl[i] = __fdiv_rn(wl[i], tot);
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    float acc = 0.f;
    for (int i = 0; i < Nc; ++i) {
        acc = __fadd_rn(acc, wl[i]);
        if (lane == 0) cdf[i + 1] = acc;
    }
    if (lane == 0) cdf[0] = 0.f;
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();

    const int nm = Nc - 1;   // number of mid-points
    for (int j = lane; j < Nf; j += 64) {
        const float u = a.u ? a.u[(size_t)ray * Nf + j]
                            : philox_u(a.seed, (unsigned)a.stream_id, (unsigned)(a.ray_offset + ray), j);
        // searchsorted(side='right'): number of cdf entries (Nc+1 of them) that are <= u
        int lo = 0, hi = Nc + 1;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] <= u) lo = mid + 1; else hi = mid; }
        const int idx = lo;
        const int below = max(0, idx - 1), above = min(Nc, idx);
        const float cb = cdf[below], ca = cdf[above];
        float mb, ma;
        if (a.oob_clamp) {
            const int ib = min(below, nm - 1), ia = min(above, nm - 1);
            mb = 0.5f * (tt[ib + 1] + tt[ib]); ma = 0.5f * (tt[ia + 1] + tt[ia]);
        } else {
            mb = below < nm ? 0.5f * (tt[below + 1] + tt[below]) : 0.f;
            ma = above < nm ? 0.5f * (tt[above + 1] + tt[above]) : 0.f;
        }
        float denom = __fsub_rn(ca, cb);
        if (denom < 1e-5f) denom = 1.f;
        const float q = __fdiv_rn(__fsub_rn(u, cb), denom);
        all[Nc + j] = __fadd_rn(mb, __fmul_rn(q, __fsub_rn(ma, mb)));
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();

    // merge: the output is the ascending sequence of the Na values (ties are indistinguishable in it).  The coarse half is
    // already ascending (rays.py:116-127: jitter below half a bin), so an element's rank is its rank inside its own half plus
    // the number of elements of the other half that precede it: a binary search over the coarse t for a fine sample, one pass
    // over the Nf fine samples for everybody.  Ties: coarse before fine, fine by index (the stable order of concat).
    // Falls back to ranking against all Na values when the coarse t of this ray are not sorted.
    float* out = a.t_out + (size_t)ray * Na;
    bool sorted = true;
    for (int i = lane; i + 1 < Nc; i += 64) sorted = sorted && (all[i] <= all[i + 1]);
    sorted = __builtin_amdgcn_ballot_w64(!sorted) == 0;
    const float* fine = all + Nc;
    for (int e = lane; e < Na; e += 64) {
        const float v = all[e];
        int rank = 0;
        if (!sorted) {
            for (int k = 0; k < Na; ++k) {
                const float o = all[k];
                rank += (o < v || (o == v && k < e)) ? 1 : 0;
            }
        } else if (e < Nc) {
            // equal coarse values keep their index order, so e itself counts the coarse elements in front; fine samples precede only when smaller
            rank = e;
            for (int k = 0; k < Nf; ++k) rank += fine[k] < v ? 1 : 0;
        } else {
            int lo = 0, hi = Nc;                             // coarse elements <= v come first
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (all[mid] <= v) lo = mid + 1; else hi = mid; }
            rank = lo;
            const int je = e - Nc;
            for (int k = 0; k < Nf; ++k) { const float o = fine[k]; rank += (o < v || (o == v && k < je)) ? 1 : 0; }
        }
        out[rank] = v;
    }
}

hipError_t launch_sample_fine(const SampleArgs& a, hipStream_t stream) {
    if (a.Nc > kMaxCoarse || a.Nc + a.Nf > kMaxAll || a.Nc < 2) return hipErrorInvalidValue;
    const size_t lds = 4 * (size_t)sampler_wave_floats(a.Nc, a.Nf) * sizeof(float);       // <= 48 KB by the static_assert above
    hipLaunchKernelGGL(sample_fine_kernel, dim3((a.R + 3) / 4), dim3(256), lds, stream, a);
    return hipGetLastError();
}

}  // namespace knerf
