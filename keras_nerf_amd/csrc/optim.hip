// optim.hip -- Keras-form Adam over the flat parameter buffer, bf16 re-packing of the MFMA weight streams, finite check.
//
// Adam restates tf.keras.optimizers Adam as used at reference keras_nerf/model/nerf/nerf.py:163-165,455-458:
//   lr_t = lr*sqrt(1-b2^t)/(1-b1^t) (double; computed ON THE DEVICE from the device-side count of APPLIED steps, so a step that
//   follows a skipped one uses the right t without the host knowing yet);  m += (g-m)(1-b1);  v += (g*g-v)(1-b2);
//   w -= lr_t * m / (sqrt(v) + eps)      -- eps OUTSIDE the bias-corrected root, eps = 1e-7.
// The gradient accumulator is zeroed in the same pass (nerf.py:464-471).  Finiteness (the reference asserts it per chunk,
// nerf.py:381-382,410-411) is checked once per step BEFORE the update by check_finite_kernel; the Adam kernels read that
// flag and skip the update when it is set, so a bad step leaves weights and slots untouched without a host round trip, and
// step_status_kernel counts the skipped step in a pinned host word (knerf_poll_nonfinite).
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "layout.h"

namespace knerf {

__global__ void adam_kernel(AdamArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const float g = a.g[i];
    a.g[i] = 0.f;                                   // also after a skipped step: the next batch must not add onto NaN
    if (*a.nonfinite) return;
    float m = a.m[i], v = a.v[i];
    m = m + (g - m) * (1.f - a.b1);
    v = v + (g * g - v) * (1.f - a.b2);
    a.m[i] = m; a.v[i] = v;
    a.w[i] = a.w[i] - *a.lr_t * m / (sqrtf(v) + a.eps);
}
hipError_t launch_adam(const AdamArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(adam_kernel, dim3((a.n + 255) / 256), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// out[i] = bf16(w[table[i]]) (round to nearest even via the hardware convert), 0 where table[i] < 0
__global__ void pack_kernel(const float* w, const int* table, unsigned short* out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int t = table[i];
    const __bf16 b = (__bf16)(t >= 0 ? w[t] : 0.f);
    out[i] = __builtin_bit_cast(unsigned short, b);
}
hipError_t launch_pack(const float* w, const int* table, unsigned short* out, size_t n, hipStream_t stream) {
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, w, table, out, n);
    return hipGetLastError();
}

__global__ void gather_f32_kernel(const float* w, const int* table, float* out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int t = table[i];
    out[i] = t >= 0 ? w[t] : 0.f;
}
hipError_t launch_gather_f32(const float* w, const int* table, float* out, size_t n, hipStream_t stream) {
    hipLaunchKernelGGL(gather_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, w, table, out, n);
    return hipGetLastError();
}

__global__ void check_finite_kernel(const float* g, int n, int* flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !__builtin_isfinite(g[i])) *flag = 1;
}
hipError_t launch_check_finite(const float* g, int n, int* flag, hipStream_t stream) {
    hipLaunchKernelGGL(check_finite_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, g, n, flag);
    return hipGetLastError();
}

// ---- zero-gradient diagnostics (reference nerf.py:430-451, eager mode): tf.math.count_nonzero summed over the 24 gradient tensors
// of each net -- of the LAST chunk, as there (knerf_train_batch keeps the earlier chunks' sum aside while the last chunk runs).
// g = [coarse | fine], n floats per net; counts: device [2], zeroed by the caller.
__global__ __launch_bounds__(256) void count_nonzero_kernel(const float* g, int n, unsigned long long* counts) {
    const int net = blockIdx.y;
    unsigned c = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) c += g[(size_t)net * n + i] != 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(counts + net, (unsigned long long)c);
}
// host (pinned): [0] coarse count, [1] fine count, [2] number of steps published so far, [3] number of steps whose publication has
// BEGUN -- a seqlock: [3] is raised before the counts are written, [2] after, so a reader that finds [2] == [3] on both
// sides of its reads of [0] / [1] holds the two counts of ONE step (knerf_grad_diagnostics; the asynchronous `fit` path reads while
// the next step's kernels run)
__global__ void diag_publish_kernel(const unsigned long long* counts, long long* host) {
    const long long seq = host[2] + 1;
    host[3] = seq;
    __threadfence_system();
    host[0] = (long long)counts[0]; host[1] = (long long)counts[1];
    __threadfence_system();
    host[2] = seq;
    __threadfence_system();
}
hipError_t launch_grad_diagnostics(const float* g, int n, unsigned long long* counts, long long* host, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(counts, 0, 2 * sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(count_nonzero_kernel, dim3(64, 2), dim3(256), 0, stream, g, n, counts);
    hipLaunchKernelGGL(diag_publish_kernel, dim3(1), dim3(1), 0, stream, counts, host);
    return hipGetLastError();
}
__global__ void add_into_kernel(float* dst, const float* src, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}
hipError_t launch_add_into(float* dst, const float* src, size_t n, hipStream_t stream) {
    hipLaunchKernelGGL(add_into_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dst, src, n);
    return hipGetLastError();
}

// End of a step (one thread).  host_status (pinned, device-visible): [0] += 1 when this step's finite check failed, [1] = steps
// processed.  step_state (device): [0] = optimizer steps APPLIED so far; lr_t = the bias-corrected learning rate of the NEXT
// step, t = applied + 1 (Keras form, see the header) -- a skipped step leaves both as they were.
__device__ __forceinline__ float keras_lr_t(AdamHyper h, int t) {
    return (float)((double)h.lr * sqrt(1.0 - pow((double)h.b2, (double)t)) / (1.0 - pow((double)h.b1, (double)t)));
}
__global__ void step_status_kernel(const int* flag, int* host_status, int* step_state, float* lr_t, AdamHyper h) {
    if (*flag) {
        host_status[0] = host_status[0] + 1;
    } else {
        step_state[0] = step_state[0] + 1;
        *lr_t = keras_lr_t(h, step_state[0] + 1);
    }
    host_status[1] = host_status[1] + 1;
    __threadfence_system();
}
hipError_t launch_step_status(const int* flag, int* host_status, int* step_state, float* lr_t, const AdamHyper& h, hipStream_t stream) {
    hipLaunchKernelGGL(step_status_kernel, dim3(1), dim3(1), 0, stream, flag, host_status, step_state, lr_t, h);
    return hipGetLastError();
}
// the applied-step counter set from the host (knerf_create, knerf_set_step_count)
__global__ void step_set_kernel(int step, int* step_state, float* lr_t, AdamHyper h) {
    step_state[0] = step;
    *lr_t = keras_lr_t(h, step + 1);
}
hipError_t launch_step_set(int step, int* step_state, float* lr_t, const AdamHyper& h, hipStream_t stream) {
    hipLaunchKernelGGL(step_set_kernel, dim3(1), dim3(1), 0, stream, step, step_state, lr_t, h);
    return hipGetLastError();
}

// ---- collapsed head (layout.h) --------------------------------------------------------------------------------------
// Offsets of the six tensors behind the trunk in the flat parameter vector (Keras order: sigma, features, rgb_features, rgb)
namespace {
// the six tensors behind the trunk, from the offset of the first float behind the last trunk layer (layout.h Shape::kTrunkParams)
// the trunk width U, the width X of xyz_enc in the trunk's OUTPUT (63 when the reference concatenates behind the last layer too,
// mlp.py:36-38; else 0) and the width D of the direction encoding (27): sigma [U+X,1], features [U+X,U], rgb_features [U+D, U/2],
// rgb [U/2, 3]; XS, DS = slots of the two encodings in the composed head (16 per k-step: 64 / 32)
struct HeadOff {
    int ws, bs, wf, bf, wr, br, wc, bc, head, head_bias;
    __device__ HeadOff(int off_l, int U, int X, int XS, int D, int DS) {
        ws = off_l; bs = ws + U + X;
        wf = bs + 1; bf = wf + (U + X) * U;
        wr = bf + U; br = wr + (U + D) * (U / 2);
        wc = br + U / 2; bc = wc + (U / 2) * 3;
        head = bc + 3; head_bias = head + (U + XS + DS) * 4;     // = Shape::kHeadOff (the parameter count), kHeadBiasOff
    }
};
static_assert(DefaultShape::kTrunkParams + 257 + 256 * 257 + 283 * 128 + 128 + 128 * 3 + 3 == DefaultShape::kParamCount && DefaultShape::kHeadRows == 288, "head tensor offsets");
constexpr int kMaxU = 256;
}  // namespace

// H[i][0..2] = (W_f (W_r1 W_c))[i], H[i][3] = w_s[i]  (i < Tr = U + X: the rows of the features / sigma kernels);
// H[Tr+m][0..2] = (W_r2 W_c)[m], H[Tr+m][3] = 0 (m < D; 0 up to DS);
// bias = ((b_f W_r1 + b_r) W_c + b_c, b_s).  fp32, one workgroup of 256 threads; ~0.6 MFLOP at U = 256.
__global__ __launch_bounds__(256) void head_compose_kernel(float* w0, float* w1, int off_l, int U, int X, int XS, int D, int DS) {
    float* w = blockIdx.x == 0 ? w0 : w1;       // one workgroup per net
    const HeadOff o(off_l, U, X, XS, D, DS);
    const int U2 = U / 2, R = U + D, Tr = U + X;
    __shared__ float P[kMaxU + kMaxDirDim][3];   // W_r W_c : rows 0..U-1 = W_r1 W_c, U..U+26 = W_r2 W_c
    __shared__ float wc[kMaxU / 2][3];
    const int tid = threadIdx.x;
    for (int i = tid; i < U2 * 3; i += 256) wc[i / 3][i % 3] = w[o.wc + i];
    __syncthreads();
    for (int r = tid; r < R; r += 256) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        const float* wr = w + o.wr + (size_t)r * U2;
        for (int k = 0; k < U2; ++k) { const float v = wr[k]; a0 += v * wc[k][0]; a1 += v * wc[k][1]; a2 += v * wc[k][2]; }
        P[r][0] = a0; P[r][1] = a1; P[r][2] = a2;
    }
    __syncthreads();
    float* H = w + o.head;
    for (int i = tid; i < Tr; i += 256) {   // row i of A = W_f P1
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        const float* wf = w + o.wf + (size_t)i * U;
        for (int j = 0; j < U; ++j) { const float v = wf[j]; a0 += v * P[j][0]; a1 += v * P[j][1]; a2 += v * P[j][2]; }
        H[i * 4 + 0] = a0; H[i * 4 + 1] = a1; H[i * 4 + 2] = a2; H[i * 4 + 3] = w[o.ws + i];
    }
    if (tid < DS) {
        const int r = Tr + tid, pr = U + tid;             // row of H, row of P
        const bool real = tid < D;
        H[r * 4 + 0] = real ? P[pr][0] : 0.f; H[r * 4 + 1] = real ? P[pr][1] : 0.f; H[r * 4 + 2] = real ? P[pr][2] : 0.f; H[r * 4 + 3] = 0.f;
    }
    if (tid < 3) {
        float c = w[o.bc + tid];
        for (int j = 0; j < U; ++j) c += w[o.bf + j] * P[j][tid];
        for (int k = 0; k < U2; ++k) c += w[o.br + k] * wc[k][tid];
        w[o.head_bias + tid] = c;
    }
    if (tid == 3) w[o.head_bias + 3] = w[o.bs];
}
hipError_t launch_head_compose(float* w0, float* w1, int trunk_params, int units, int trunk_x, int trunk_x_slots, int dir_dim, int dir_slots, hipStream_t stream) {
    if (units > kMaxU || dir_dim > kMaxDirDim || dir_slots > 64 || trunk_x > kMaxXyzDim) return hipErrorInvalidValue;
    hipLaunchKernelGGL(head_compose_kernel, dim3(w1 ? 2 : 1), dim3(256), 0, stream, w0, w1, trunk_params, units, trunk_x, trunk_x_slots, dir_dim, dir_slots);
    return hipGetLastError();
}

// aux: M[row][c] = sum_s [h ; (xyz ;) dir][s][row] dz_rgb[s][c] (row < Tr + D, Tr = U + X), s[c] = sum_s dz_rgb[s][c].  With M1 = rows
// 0..Tr-1 (the trunk's output), M2 = rows Tr..Tr+26, P1 = W_r1 W_c and Q = W_f^T M1 + b_f (x) s  (= sum_s features[s]^T dz_rgb[s]):
//   d rgb/kernel          = W_r1^T Q + W_r2^T M2 + b_r (x) s        d rgb/bias          = s
//   d rgb_features/kernel = [Q ; M2] W_c^T                          d rgb_features/bias = s W_c^T
//   d features/kernel     = M1 P1^T                                 d features/bias     = s P1^T
// -- the chain rule through the three linear layers (what the tape yields at nerf.py:376-377 for those six tensors),
// evaluated on sums over samples instead of per sample.  Added to grad; aux is zeroed.  One workgroup of 1024 threads.
struct HeadExpandArgs { const float* w[2]; float* aux[2]; float* grad[2]; int off_l, U, X, XS, D, DS; };
__global__ __launch_bounds__(1024) void head_expand_kernel(HeadExpandArgs a) {
    const float* w = a.w[blockIdx.x]; float* aux = a.aux[blockIdx.x]; float* grad = a.grad[blockIdx.x];      // one workgroup per net
    const int U = a.U, U2 = U / 2, R = U + a.D, Tr = U + a.X;
    const HeadOff o(a.off_l, U, a.X, a.XS, a.D, a.DS);
    __shared__ float M[kMaxU + kMaxXyzDim + kMaxDirDim][3], s_[3], P1[kMaxU][3], Q[kMaxU][3], wc[kMaxU / 2][3];
    const int tid = threadIdx.x;
    for (int i = tid; i < (Tr + a.D) * 3; i += 1024) M[i / 3][i % 3] = aux[kAuxM + i];
    if (tid < 3) s_[tid] = aux[kAuxS + tid];
    for (int i = tid; i < U2 * 3; i += 1024) wc[i / 3][i % 3] = w[o.wc + i];
    __syncthreads();
    for (int i = tid; i < kAuxCount; i += 1024) aux[i] = 0.f;
    if (tid < U) {              // P1 row tid
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        const float* wr = w + o.wr + (size_t)tid * U2;
        for (int k = 0; k < U2; ++k) { const float v = wr[k]; a0 += v * wc[k][0]; a1 += v * wc[k][1]; a2 += v * wc[k][2]; }
        P1[tid][0] = a0; P1[tid][1] = a1; P1[tid][2] = a2;
    } else if (tid >= 256 && tid < 256 + U) {     // Q row j: column j of W_f against M1 (coalesced over j)
        const int j = tid - 256;
        const float bf = w[o.bf + j];
        float a0 = bf * s_[0], a1 = bf * s_[1], a2 = bf * s_[2];
        for (int i = 0; i < Tr; ++i) { const float v = w[o.wf + (size_t)i * U + j]; a0 += v * M[i][0]; a1 += v * M[i][1]; a2 += v * M[i][2]; }
        Q[j][0] = a0; Q[j][1] = a1; Q[j][2] = a2;
    }
    __syncthreads();
    // features: kernel [Tr,U] += M1 P1^T, bias += s P1^T
    for (int e = tid; e < Tr * U; e += 1024) {
        const int i = e / U, j = e - i * U;
        grad[o.wf + e] += M[i][0] * P1[j][0] + M[i][1] * P1[j][1] + M[i][2] * P1[j][2];
    }
    if (tid < U) grad[o.bf + tid] += s_[0] * P1[tid][0] + s_[1] * P1[tid][1] + s_[2] * P1[tid][2];
    // rgb_features: kernel [U+27, U/2] += [Q ; M2] W_c^T, bias += s W_c^T
    for (int e = tid; e < R * U2; e += 1024) {
        const int r = e / U2, k = e - r * U2;
        const float* v = r < U ? Q[r] : M[Tr + (r - U)];
        grad[o.wr + e] += v[0] * wc[k][0] + v[1] * wc[k][1] + v[2] * wc[k][2];
    }
    if (tid < U2) grad[o.br + tid] += s_[0] * wc[tid][0] + s_[1] * wc[tid][1] + s_[2] * wc[tid][2];
    // rgb: kernel [U/2,3] += W_r1^T Q + W_r2^T M2 + b_r (x) s, bias += s
    if (tid < U2 * 3) {
        const int k = tid / 3, c = tid % 3;
        float a = w[o.br + k] * s_[c];
        for (int j = 0; j < U; ++j) a += w[o.wr + (size_t)j * U2 + k] * Q[j][c];
        for (int m = U; m < R; ++m) a += w[o.wr + (size_t)m * U2 + k] * M[Tr + (m - U)][c];
        grad[o.wc + tid] += a;
    }
    if (tid < 3) grad[o.bc + tid] += s_[tid];
}
hipError_t launch_head_expand(const float* w0, float* aux0, float* grad0, const float* w1, float* aux1, float* grad1, int trunk_params, int units, int trunk_x,
                              int trunk_x_slots, int dir_dim, int dir_slots, hipStream_t stream) {
    if (units > kMaxU || dir_dim > kMaxDirDim || trunk_x > kMaxXyzDim) return hipErrorInvalidValue;
    HeadExpandArgs a{{w0, w1}, {aux0, aux1}, {grad0, grad1}, trunk_params, units, trunk_x, trunk_x_slots, dir_dim, dir_slots};
    hipLaunchKernelGGL(head_expand_kernel, dim3(2), dim3(1024), 0, stream, a);
    return hipGetLastError();
}

}  // namespace knerf
