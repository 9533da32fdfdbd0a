/* knerf.h -- C ABI of libknerf_hip.so: the MI355X (gfx950) NeRF train/render hot path.
 *
 * The reference (naufalso/keras_nerf) has no FFI layer: its hot path is the Python class surface
 * keras_nerf/model/nerf/{nerf,utils,mlp}.py on stock TensorFlow ops.  Each entry point below names the reference
 * code it replaces; the modules under keras_nerf_amd/model/nerf bind them with ctypes behind the reference's class names.
 *
 * Conventions: every function returns 0 on success or a negative knerf_status; knerf_last_error() gives the text.
 * All array arguments are DEVICE pointers to contiguous row-major fp32 unless marked host.  `stream` is a
 * hipStream_t (may be NULL).  Nothing is retained past a call; the context owns weights, gradients, Adam slots
 * and workspaces.  One context per GPU, not thread safe.
 */
#ifndef KNERF_H
#define KNERF_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct knerf_ctx knerf_ctx;

typedef enum knerf_status {
    KNERF_OK = 0,
    KNERF_ERR_INVALID = -1,       /* bad argument / unsupported architecture                      */
    KNERF_ERR_HIP = -2,           /* a HIP runtime call failed                                    */
    KNERF_ERR_NONFINITE = -3,     /* a gradient is not finite (reference: assert_all_finite)      */
    KNERF_ERR_NODEVICE = -4       /* no gfx950 device visible                                     */
} knerf_status;

/* NeRF(...) constructor arguments (reference nerf.py:11-14) + compile() arguments (nerf.py:78) + Adam defaults
 * of tf.keras.optimizers.get('adam') (nerf.py:163-165). */
typedef struct knerf_config {
    int32_t n_coarse, n_fine;          /* 64, 128 */
    int32_t pos_emb_xyz, pos_emb_dir;  /* 10, 4   */
    int32_t n_layers, dense_units, skip_layer; /* 8, 256, 4 : the shape of the fused kernels; others use csrc/generic.hip */
    int32_t white_background;          /* compile(white_background=...) */
    int32_t oob_clamp;                 /* 0: out-of-range mid-point gather yields 0 (tf.gather on GPU); 1: clamp */
    float lr, beta1, beta2, epsilon;   /* 1e-3, 0.9, 0.999, 1e-7 */
} knerf_config;

enum { KNERF_COARSE = 0, KNERF_FINE = 1 };

/* number of trainable scalars per MLP (595,844 for the default shape; reference mlp.py:11-27) */
size_t knerf_param_count(void);
/* the same for any NeRFMLP(n_layers, dense_units, skip_layer) with pos_emb_xyz / pos_emb_dir (mlp.py:11-27); 0 = invalid.
 * Shapes other than the default run on the general-shape kernels (csrc/generic.hip): same API, same numerics contract. */
size_t knerf_param_count_for(const knerf_config* cfg);

int knerf_create(const knerf_config* cfg, knerf_ctx** out);
int knerf_destroy(knerf_ctx* ctx);
const char* knerf_last_error(const knerf_ctx* ctx);   /* ctx may be NULL: error of the last failed create */

/* Weights of one MLP as ONE flat fp32 vector in Keras trainable_variables order
 * (layer_0/kernel[in,out], layer_0/bias, ..., sigma, features, rgb_features, rgb); HOST pointers.
 * Replaces NeRFMLP weight creation / load_weights / save_weights (nerf.py:116-136, 45-64). */
int knerf_set_weights(knerf_ctx* ctx, int net, const float* host_flat, size_t n);
int knerf_get_weights(knerf_ctx* ctx, int net, float* host_flat, size_t n);
/* device views for collectives: weights of one net; the gradient accumulators of BOTH nets as one buffer
 * [coarse | fine] (2*param_count floats) so that a single all-reduce covers the step (train.py:75). */
int knerf_weights_device(knerf_ctx* ctx, int net, float** dev, size_t* n);
int knerf_grads_device(knerf_ctx* ctx, float** dev, size_t* n);
/* re-derive the bf16 MFMA weight streams from the fp32 master weights (after an external write to them) */
int knerf_refresh_weights(knerf_ctx* ctx, void* stream);

/* NeRF._predict_and_render_chunk (nerf.py:175-216) for one net on given t-values:
 * encode -> MLP -> composite.  o,d [R,3]; t [R,S]; outputs image [R,3], depth [R], weights [R,S]. */
int knerf_forward_chunk(knerf_ctx* ctx, void* stream, int net, const float* o, const float* d, const float* t,
                        int n_rays, int n_samples, float* image, float* depth, float* weights);

/* NeRFMLP.__call__((xyz_enc, dir_enc)) (mlp.py:29-50) on inputs that are ALREADY positional encodings:
 * xyz_enc [n, 3+6*pos_emb_xyz], dir_enc [n, 3+6*pos_emb_dir] fp32 device pointers; raw [n,4] = (rgb after sigmoid, sigma
 * after relu).  The reference calls its MLPs this way only to create weights and in a shape test; it runs on the
 * general-shape kernels (bf16 operands, fp32 accumulate) for every shape, including the default one. */
int knerf_mlp_call(knerf_ctx* ctx, void* stream, int net, const float* xyz_enc, const float* dir_enc, uint64_t n, float* raw);

/* the fine branch's sampling (nerf.py:182-191, utils.py:60-97): t_out [R, n_coarse+n_fine] sorted.
 * u [R,n_fine] in [0,1) or NULL for the built-in Philox stream keyed by (seed, stream_id, ray_offset + ray). */
int knerf_sample_fine(knerf_ctx* ctx, void* stream, const float* t_coarse, const float* w_coarse, const float* u,
                      uint64_t seed, uint64_t stream_id, uint64_t ray_offset, int n_rays, float* t_out);

/* NeRF.predict_and_render_chunk (nerf.py:218-227): coarse pass, sampling, fine pass. Any output may be NULL
 * except the two images.  t_fine [R, n_coarse+n_fine] receives the merged t-values. */
int knerf_render_chunk(knerf_ctx* ctx, void* stream, const float* o, const float* d, const float* t, const float* u,
                       uint64_t seed, uint64_t ray_offset, int n_rays,
                       float* c_image, float* c_depth, float* c_weights,
                       float* f_image, float* f_depth, float* f_weights, float* t_fine);

/* predict_and_render_images (nerf.py:229-304): the chunk loop of knerf_render_chunk over n_rays = C * ray_chunks rays in one
 * host call; outputs are whole-batch arrays ([n_rays, ...], images mandatory, the rest optional; t_fine [n_rays,
 * n_coarse+n_fine] receives the merged t-values of nerf.py:190-191).  ray_offset of chunk i is i * ray_chunks, so results
 * equal C separate knerf_render_chunk calls. */
int knerf_render_batch(knerf_ctx* ctx, void* stream, const float* o, const float* d, const float* t, const float* u, uint64_t seed,
                       int n_rays, int ray_chunks, float* c_image, float* c_depth, float* c_weights, float* f_image, float* f_depth,
                       float* f_weights, float* t_fine);

/* One iteration of train_step's chunk loop (nerf.py:351-421): coarse forward+backward, fine forward+backward,
 * gradients accumulated as acc += g * inv_chunks, chunk losses accumulated into loss[0] (coarse) / loss[1] (fine)
 * (device, 2 floats, caller zeroes them per step).  target [R,3]. */
int knerf_train_chunk(knerf_ctx* ctx, void* stream, const float* o, const float* d, const float* t,
                      const float* target, const float* u, uint64_t seed, uint64_t ray_offset, int n_rays,
                      float inv_chunks, float* loss, float* c_image, float* f_image);

/* The whole chunk loop of train_step (nerf.py:351-421) in one call: n_rays must be a multiple of ray_chunks (the
 * reference's assert, nerf.py:100); chunk i covers rays [i*ray_chunks, (i+1)*ray_chunks) with weight 1/C.  Same arguments
 * as knerf_train_chunk, arrays sized for all n_rays. */
int knerf_train_batch(knerf_ctx* ctx, void* stream, const float* o, const float* d, const float* t, const float* target,
                      const float* u, uint64_t seed, int n_rays, int ray_chunks, float* loss, float* c_image, float* f_image);

/* coarse_optimizer.apply_gradients + fine_optimizer.apply_gradients + accumulator reset (nerf.py:455-471).
 * Call after the optional all-reduce of knerf_grads_device().  Returns KNERF_ERR_NONFINITE (weights untouched)
 * when a gradient is not finite (nerf.py:381-382).  Synchronises the stream. */
int knerf_apply_adam(knerf_ctx* ctx, void* stream);
int knerf_zero_grads(knerf_ctx* ctx, void* stream);
int knerf_step_count(const knerf_ctx* ctx);
int knerf_set_step_count(knerf_ctx* ctx, int step);

/* RaysGenerator.__call__ (keras_nerf/data/rays.py:69-130) on device: c2w [B,4,4], noise [B,H,W,N] in [0,1) or
 * NULL for Philox; writes o,d [B,H,W,3] and t [B,H,W,N].  ctx may be NULL (stand-alone op). */
int knerf_generate_rays(knerf_ctx* ctx, void* stream, const float* c2w, const float* noise, uint64_t seed,
                        uint64_t stream_id, int batch, int height, int width, int n_samples, float focal,
                        float near_plane, float far_plane, float* o, float* d, float* t);

/* ---- NeRFUtils as stand-alone ops (no context needed; the train/render path fuses the same arithmetic) ----
 * positional_encoding (utils.py:176-186): x [n_rows,3] -> out [n_rows, 3+6L].
 * composite = render_image_depth_chunk (utils.py:16-58): raw [R,S,4] = (r,g,b,sigma), t [R,S] -> image [R,3], depth [R]
 *   or NULL, weights [R,S] or NULL.
 * inverse_cdf = fine_hierarchical_sampling_chunk (utils.py:60-97) with u injected: mid_points [R,n_mid],
 *   weights [R,n_weights], u [R,n_samples] -> out [R,n_samples] (unsorted). */
int knerf_positional_encoding(void* stream, const float* x, long long n_rows, int L, float* out);
/* ray(t) = o + t d (utils.py:193-194): o, d [R,3], t [R,S] -> out [R,S,3] */
int knerf_ray_points(void* stream, const float* o, const float* d, const float* t, int n_rays, int n_samples, float* out);
/* the two image metrics NeRF.update_and_return_metrics logs (nerf.py:306-330; tf.image.psnr / tf.image.ssim defaults):
 * a, b [n_images, H, W, C] device; sums [n_images][2] = {sum of the SSIM terms over the (H-10)(W-10) VALID windows and C
 * channels, sum of squared differences}: ssim = sums[0] / ((H-10)(W-10)C), psnr = -10 log10(sums[1] / (HWC)). */
int knerf_image_metrics(void* stream, const float* a, const float* b, int n_images, int height, int width, int channels,
                        float* sums);
int knerf_composite(void* stream, const float* raw, const float* t, int n_rays, int n_samples, int white_background /* bit 0: white background, bit 1: no clip (utils.py:99-134) */,
                    float* image, float* depth, float* weights);
int knerf_inverse_cdf(void* stream, const float* mid_points, const float* weights, const float* u, int n_rays, int n_mid,
                      int n_weights, int n_samples, int oob_clamp, float* out);

/* Per-kernel timing with HIP events recorded on the caller's stream around every launch (bench.py's roofline leg).
 * Classes: 0 mlp_fwd coarse, 1 mlp_fwd fine, 2 composite, 3 sample_fine, 4 mlp_bwd coarse, 5 mlp_bwd fine,
 * 6 wgrad coarse, 7 wgrad fine, 8 adam+repack.  read() synchronises the device, returns summed milliseconds and launch
 * counts since enable/the previous read (n >= 9). */
/* Backward schedule.  producers = 0 (default): dgrad and wgrad are separate launches.  producers = P > 0: one launch of
 * P persistent dgrad workgroups + (CUs - P) wgrad workgroups that consume dZ as it is published (fused_bwd.hip).  Results
 * are the same up to fp32 summation order.  A poll time-out inside the fused launch is reported by knerf_apply_adam
 * (KNERF_ERR_HIP); it cannot hang.  No reference counterpart (scheduling only). */
int knerf_set_fused_backward(knerf_ctx* ctx, int producers);

int knerf_profile_enable(knerf_ctx* ctx, int on);
int knerf_profile_read(knerf_ctx* ctx, double* total_ms, int64_t* launches, int n);

/* ---- introspection used by the CPU-side layout tests (no device work) ---- */
/* kind 0: forward A-fragment table, 1: forward bias table, 2: dgrad A-fragment table, 3: wgrad destination table.
 * Entries are indices into the flat parameter vector or -1.  Pass out=NULL to query the length. */
int knerf_debug_table(int kind, int32_t* out, size_t* n);
/* device buffers of the last knerf_train_chunk for kernel-level tests: 0 act, 1 mask, 2 dz, 3 raw, 4 draw,
 * 5 merged fine t-values, 6 coarse weights */
int knerf_debug_buffer(knerf_ctx* ctx, int net, int which, void** dev, size_t* bytes);
/* hardware-fact probes for tests: kind 0 = one v_mfma_f32_32x32x16_bf16 (in0 = A fragments [64][8] bf16, in1 = B
 * fragments, out = [64][16] f32); kind 1 = one ds_read_b64_tr_b16 (in0 = 4 KiB LDS image, in1 = [64] int32 byte
 * offsets, out = [64][4] u16).  All device pointers. */
/* the general-shape path's layer program for a config (no device needed): 16 int32 per Dense layer in Keras order =
 * {kernel offset, bias offset, fan_in, fan_out, padded input width, padded output width, n_seg, seg0 (buffer col0, width,
 * kernel row0), seg1 (...), relu, head (-1 | 0 sigma | 1 rgb), padded width of the output buffer or -1}. */
int knerf_debug_generic_plan(const knerf_config* cfg, int32_t* out, size_t* n);
int knerf_debug_probe(int kind, const void* in0, const void* in1, void* out, void* stream);
/* MFMA-shape rate probe (shape 32: v_mfma_f32_32x32x16_bf16, 16: v_mfma_f32_16x16x32_bf16) with the chain kernels' operand
 * traffic; `blocks` workgroups of 512 threads, 96 * 2^15 * 16 FLOP per wave and iteration.  Diagnostic only. */
/* HBM write-pattern probe (diagnostic): `workgroups` x 8 waves each store `blocks` 1 KiB blocks into tiles `tile_stride`
 * bytes apart; mode 0 = the chain kernels' pattern, 1 = the 8 waves of a workgroup interleaved. */
int knerf_debug_write_probe(void* out, int workgroups, int blocks, long long tile_stride, int mode, int spin, void* stream);
/* HBM read-pattern probe (diagnostic): each of workgroups x 8 waves streams bytes_per_wave contiguous bytes in 1 KiB
 * instructions; mode 0 = nt LDS-DMA (wgrad's loads), 1 = plain register loads. */
int knerf_debug_read_probe(const void* in, int workgroups, long long bytes_per_wave, int mode, void* out, void* stream);
int knerf_debug_rate_probe(int shape, const void* in0, const void* in1, void* out, int blocks, int iters, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KNERF_H */
