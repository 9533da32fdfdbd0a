// generic.h -- the general-shape MLP path (any n_layers / dense_units / skip_layer / pos_emb_*), gfx950.
//
// The fused chain kernels (mlp_fwd/mlp_bwd/wgrad) are specialised for the reference's default NeRFMLP (8 x 256, skip 4,
// L = 10/4; reference mlp.py:4-50 with the constructor defaults of nerf.py:11-14).  Every other shape the reference's
// constructor and CLI accept (train_single.py:30-36) runs here: one MFMA GEMM launch per Dense layer over row-major bf16
// activations in HBM, same numerics contract as the fused path (bf16 matmul operands, fp32 accumulate, fp32 bias and
// activation), same compositing / sampling / Adam kernels around it.  This path is about coverage, not peak speed.
//
// Head (round 2): sigma, features, rgb_features and rgb are evaluated as ONE GEMM on a composed [K][4] matrix, exactly as
// on the fused path (layout.h "collapsed head") -- the identity holds for every shape because the
// reference never puts an activation on features / rgb_features (mlp.py:21-24).  The four Keras layers stay in Plan::layers
// for their parameter offsets; they own no buffer and no packed weights any more.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace knerf {
namespace gen {

struct Seg { int col0, width, wrow0; };      // input-buffer columns [col0, col0+width) <- Keras kernel rows [wrow0, ...)

struct Layer {
    int w_off, b_off;          // offsets of kernel / bias in the net's flat fp32 parameter vector (Keras order)
    int k_real, n_real;        // Keras kernel shape
    int n_seg; Seg seg[2];     // where the kernel's rows sit in the (padded, possibly concatenated) input buffer
    int in_buf;                // activation buffer read by this layer; its ld is the GEMM K
    int out_buf, out_col0;     // bf16 output buffer (+ first column), or out_buf < 0: fp32 head output `head`
    int head;                  // 0 sigma, 1 rgb (fp32 [M,32] outputs), -1 otherwise
    int np;                    // n_real rounded up to 32
    int relu;
    size_t wt_off;             // packed Wt [np][K]      (forward:  out = in . Wt^T)
    size_t wd_off; int wd_ld, wd_col0;   // packed Wd [K][wd_ld] (dgrad: d_in = d_out . Wd^T), this layer's columns at wd_col0
    int dz_buf, dz_col0;       // where this layer's dZ lives
};

struct Plan {
    int n_layers, units, skip, lx, ld;
    int xyz_dim, dir_dim, kxp, kdp, up, u2p;
    int n_params;
    std::vector<Layer> layers;           // Keras order: layer_0..n-1, sigma, features, rgb_features, rgb
    std::vector<int> buf_ld;             // activation buffers (bf16, [Mp][ld])
    std::vector<int> dz_ld;              // dZ buffers (bf16, [Mp][ld])
    std::vector<int> concat_after;       // per trunk layer: 1 if [h ; xyz_enc] follows it
    int buf_encx, buf_encd, buf_trunk;
    // collapsed head: the trunk buffer carries the dir encoding behind its own columns ([h ; (xyz_enc) ; dir_enc]) and is
    // the head GEMM's input; its ld is the head's K
    int head_K, head_dir_col0;           // columns of the trunk buffer / first dir column
    int trunk_real;                      // rows of the sigma / features kernels: units (+ xyz_dim when a concat follows the last layer)
    size_t head_wt_off, head_wd_off;     // packed bf16 [32][head_K] (forward) and [head_K][32] (dgrad)
    int dz_head;                         // dZ buffer [Mp][32]: columns r, g, b, sigma
    size_t packed_elems;                 // bf16 elements of the packed-weight arena
    size_t act_elems_per_row, dz_elems_per_row;
};

bool is_default_shape(int n_layers, int units, int skip, int lx, int ld);
int param_count(int n_layers, int units, int skip, int lx, int ld);
Plan build_plan(int n_layers, int units, int skip, int lx, int ld);
// the same from the two encoded input WIDTHS (any positive integers; knerf.h KNERF_FLAG_ENCODED_WIDTHS): a plan for
// forward_encoded only
Plan build_plan_widths(int n_layers, int units, int skip, int xyz_dim, int dir_dim);

struct Workspace {             // per context, grow-only; Mp = padded sample count
    size_t mp = 0;
    unsigned short* act = nullptr;     // all activation buffers, buffer b at act + off_b * mp
    unsigned short* dz = nullptr;
    unsigned char* mask = nullptr;     // training workspaces only: the relu bits of every trunk layer (relu_bits_bytes; generic.hip GemmArgs)
    float* zs = nullptr;               // (unused since the head collapse; kept so that callers' allocation code stays valid)
    float* zc = nullptr;               // [Mp][32] head pre-activations: columns r, g, b, sigma
};

struct NetDev {                // per net
    unsigned short* packed = nullptr;  // packed bf16 weights (Wt and Wd of every trunk layer and of the composed head)
    float* head = nullptr;             // head_floats(plan) fp32: composed head matrix [head_K][4] in buffer-column order, bias [4],
                                       //   then scratch P [units + dir_dim][3] and Q [units][3]
    float* gaux = nullptr;             // aux_floats(plan) fp32: head sums M [head_K][4], s [4] accumulated by the head wgrad
};
size_t head_floats(const Plan& p);
size_t aux_floats(const Plan& p);

// All functions enqueue on `s` and return the first HIP error.
// composes the head (net.head) and re-packs every bf16 weight matrix (net.packed) from the flat fp32 parameters
hipError_t pack_weights(const Plan& p, const float* w_flat, const NetDev& net, hipStream_t s);
// net.gaux (head sums of the backward passes since the last call) -> += gradients of sigma / features / rgb_features / rgb
// in grad_flat; gaux is zeroed
hipError_t expand_head(const Plan& p, const NetDev& net, const float* w_flat, float* grad_flat, hipStream_t s);
// forward over n = R*S samples: fills raw [n][4] (rgb after sigmoid, sigma after relu); keeps activations for backward
hipError_t forward(const Plan& p, const Workspace& ws, const NetDev& net, const float* w_flat, const float* o, const float* d,
                   const float* t, long long n, int S, float* raw, hipStream_t s);
// the same on inputs that are already encoded: xyz_enc [n][p.xyz_dim], dir_enc [n][p.dir_dim] (fp32)
hipError_t forward_encoded(const Plan& p, const Workspace& ws, const NetDev& net, const float* w_flat, const float* xyz_enc,
                           const float* dir_enc, long long n, float* raw, hipStream_t s);
// backward from draw [n][4] (dL/d rgb, dL/d sigma): trunk gradients accumulate into grad_flat (fp32 atomics), the head's
// sums into net.gaux (expand_head turns them into gradients)
// partial: null (fp32 atomics), or wgrad_partial_floats(p) floats of scratch for the deterministic mode -- per-unit slabs and an
// ordered second pass per weight-gradient launch, bit-identical between runs
// live / n_live: null, or the device list of live 32-row tiles (composite.hip) and its device-side length: dead-tile skipping --
// the dgrad GEMMs and the weight-gradient products walk the list; stats (optional): [0] += live, [1] += all tiles of the pass
hipError_t backward(const Plan& p, const Workspace& ws, const NetDev& net, const float* raw, const float* draw, long long n,
                    float* grad_flat, hipStream_t s, float* partial = nullptr, const int* live = nullptr, const int* n_live = nullptr,
                    long long* stats = nullptr);
size_t wgrad_partial_floats(const Plan& p);
size_t padded_rows(long long n);
size_t relu_bits_bytes_per_row(const Plan& p);          // one bit per padded trunk feature
size_t relu_bits_bytes(const Plan& p, size_t mp);       // Workspace::mask for mp padded rows

}  // namespace gen
}  // namespace knerf
