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
    int32_t n_coarse, n_fine;          /* 64, 128; 2 <= n_coarse <= 512, n_coarse + n_fine <= 1024 */
    int32_t pos_emb_xyz, pos_emb_dir;  /* 10, 4   */
    int32_t n_layers, dense_units, skip_layer; /* 8, 256, 4; fused kernels: the triples of csrc/layout.h KNERF_FUSED_SHAPES (widths 256, 128; 64 at build time); others: csrc/generic.hip */
    int32_t white_background;          /* compile(white_background=...) */
    int32_t oob_clamp;                 /* 0: out-of-range mid-point gather yields 0 (tf.gather on GPU); 1: clamp */
    float lr, beta1, beta2, epsilon;   /* 1e-3, 0.9, 0.999, 1e-7 */
    int32_t flags;                     /* KNERF_FLAG_* */
} knerf_config;

/* KNERF_FLAG_FORCE_GENERIC: run a shape the fused kernels cover through the general-shape kernels too (tests compare the two paths).
 * KNERF_FLAG_ENCODED_WIDTHS: a stand-alone NeRFMLP (reference mlp.py:4-59; tests/model/nerf/test_nerf_mlp.py:6-45 feeds 99-wide
 *   tensors to BOTH inputs).  Keras Dense takes its input size from the last dimension of the first call (mlp.py:11-27 names
 *   none), so the two widths are free: with this flag pos_emb_xyz / pos_emb_dir hold the encoded input WIDTHS themselves
 *   (1..4096 each, not the number of frequencies) and n_coarse / n_fine are ignored.  Such a context serves knerf_set_weights /
 *   knerf_get_weights / knerf_weights_device / knerf_mlp_call only; the ray entry points return KNERF_ERR_INVALID. */
enum { KNERF_FLAG_FORCE_GENERIC = 1, KNERF_FLAG_ENCODED_WIDTHS = 2 };

enum { KNERF_COARSE = 0, KNERF_FINE = 1 };

/* number of trainable scalars per MLP (595,844 for the default shape; reference mlp.py:11-27) */
size_t knerf_param_count(void);
/* the same for any NeRFMLP(n_layers, dense_units, skip_layer) with pos_emb_xyz / pos_emb_dir (mlp.py:11-27); 0 = invalid.
 * Shapes outside csrc/layout.h KNERF_FUSED_SHAPES run on the general-shape kernels (csrc/generic.hip): same API, same numerics contract. */
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
 * xyz_enc [n, 3+6*pos_emb_xyz], dir_enc [n, 3+6*pos_emb_dir] (or [n, pos_emb_xyz], [n, pos_emb_dir] on a context created with
 * KNERF_FLAG_ENCODED_WIDTHS) fp32 device pointers; raw [n,4] = (rgb after sigmoid, sigma after relu).  The reference calls its MLPs this way only to create weights and in a shape test; it runs on the
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
 * Call after the optional all-reduce of knerf_grads_device().  ASYNCHRONOUS: the finite check of nerf.py:381-382 runs on the
 * device in front of the update, and a step whose gradient is not finite leaves weights and Adam slots untouched (the
 * accumulators are zeroed either way).  The host learns of it from knerf_poll_nonfinite. */
int knerf_apply_adam(knerf_ctx* ctx, void* stream);
/* Returns KNERF_ERR_NONFINITE once for every batch of skipped steps since the previous poll (and takes them back out of the
 * step counter), KNERF_OK otherwise.  wait != 0 synchronises `stream` first, i.e. covers every knerf_apply_adam issued so far;
 * wait == 0 reads the device-written status word as it stands (steps that have completed). */
int knerf_poll_nonfinite(knerf_ctx* ctx, void* stream, int wait);
int knerf_zero_grads(knerf_ctx* ctx, void* stream);

/* Run-time options of a context (no environment variables are read by the library).  Names:
 *   "deterministic"     0/1  weight gradients and losses without floating-point atomics: every workgroup writes its partial sums
 *                            to its own slab and an ordered second pass adds them, so two runs of the same step are bit-identical
 *                            (slower; the reference's TF ops give no such guarantee either -- this is a test/diagnosis mode).  Both MLP paths: the
 *                            fused kernels and, since round 5, the general-shape kernels (csrc/generic.hip).
 *   "skip_dead_tiles"   0/1  (default 1) the backward kernels skip every 32-sample tile whose dL/d(rgb, sigma) is EXACTLY zero for all samples
 *                            (empty space with a closed ReLU gate on sigma, rays whose pixel error is exactly 0): such samples add
 *                            exactly nothing to any of the 48 gradient tensors (utils.py:36-45, mlp.py:40), so the result is the
 *                            same; applies when n_coarse and n_coarse + n_fine are multiples of 32 (every MLP shape: the fused kernels
 *                            and, since round 5, the general-shape kernels walk the list of live tiles).
 *   "grad_diagnostics"  0/1  (default 0) knerf_train_batch counts the non-zero entries of the last chunk's gradient of each net
 *                            (knerf_grad_diagnostics; the reference does this when run_eagerly, nerf.py:430-451); one launch of the
 *                            coarse weight-gradient kernel per chunk while it is on.
 *   "merge_render_rays" 0..1048576 (default 65536) the same for knerf_render_batch on the fused kernels (rendering keeps no saved
 *                            tensors: 5 KB of workspace per ray; outputs bit-identical for every value); the general-shape kernels
 *                            render under "merge_chunk_rays".
 *   "merge_chunk_rays"  0..1048576 (default 4096) knerf_train_batch runs m consecutive chunks as one set of launches,
 *                            m the largest divisor of the chunk count with m * ray_chunks <= this value (0: every chunk its own
 *                            launches).  `ray_chunks` is the reference's memory knob (nerf.py:100, 332-473): every ray's forward, loss
 *                            term and gradient contribution is independent of the chunk that holds it (mean over R rays times 1 / C
 *                            = mean over m R rays times m / C; the fine sampler's random numbers are keyed by the ray's index in the
 *                            batch), so rendered outputs are bit-identical and accumulated gradients equal up to the order of fp32
 *                            sums.  4,096 rays need 8 GB of workspace here; if that cannot be allocated the caller's own chunk
 *                            size is used -- by knerf_render_batch too -- and remembered for that (ray_chunks, chunk count), so later
 *                            calls do not fail the same allocation again ("merge_fallbacks", read-only: how often that happened).
 *                            Off while "grad_diagnostics" is on (it counts the LAST chunk's gradient).
 *   "workspace_limit_gb" >= 0 (default 0 = off; tests) a workspace request above it is answered like an exhausted device -- by a real
 *                            failing hipMalloc -- so that the two fall-backs above can be exercised without filling the memory.
 *   "wgrad_group_max"   1..64, "wgrad_group_gb" >= 0: chunks per coarse weight-gradient launch of knerf_train_batch and the memory
 *                            budget of the workspaces that takes (defaults 4 and 40 GB; 1 or 0 = one launch per chunk).
 *   "wgrad_cost0".."wgrad_cost<n_layers>": relative cost per sample tile of the n_layers + 1 weight-gradient jobs (nine for the default shape) (workgroups are dealt out in that
 *                            proportion; tuning sweeps).
 * knerf_get_option also answers "skip_dead_tiles_active", "wgrad_group" (of the current workspaces), "general_shape_path" and
 * "merge_fallbacks". */
int knerf_set_option(knerf_ctx* ctx, const char* name, double value);
int knerf_get_option(knerf_ctx* ctx, const char* name, double* value);
/* running totals since the last reset: 32-sample tiles the dgrad launches found live / all tiles they covered (skip_dead_tiles
 * on; both 0 otherwise).  Synchronises `stream`. */
int knerf_tile_stats(knerf_ctx* ctx, void* stream, int64_t* live, int64_t* total, int reset);
/* the same per net: live[2], total[2] = {coarse passes, fine passes} */
int knerf_tile_stats_net(knerf_ctx* ctx, void* stream, int64_t* live, int64_t* total, int reset);
/* The reference's eager-mode check "is the gradient zero" (nerf.py:430-451: tf.math.count_nonzero summed over the 24 gradient
 * tensors of each net, taken from the LAST chunk of the step).  With option "grad_diagnostics" on, knerf_train_batch keeps the sum of
 * the earlier chunks aside while the last chunk runs, counts on the device and publishes to pinned memory; this call reads
 * out[0] = coarse count, out[1] = fine count, out[2] = number of steps published so far (wait != 0: synchronises `stream` first,
 * i.e. the counts of the step just enqueued; wait == 0: of the newest step that has completed).  out: 3 values. */
int knerf_grad_diagnostics(knerf_ctx* ctx, void* stream, int wait, int64_t* out);
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
/* NeRF.update_and_return_metrics (nerf.py:306-330) without a host round trip: the six tf.keras.metrics.Mean objects as device
 * state [6][2] doubles = {total, count} in the order coarse_loss, coarse_psnr, coarse_ssim, fine_loss, fine_psnr, fine_ssim.
 * sums_coarse / sums_fine = knerf_image_metrics(images, coarse images) / (images, fine images) [n_images][2]; loss [2] = the
 * step's coarse and fine loss (device), or NULL for test_step's whole-image mean squared error (nerf.py:484-487) taken from
 * the same sums.  PSNR and SSIM add one value per image, the losses one per step, as Keras' Mean does. */
int knerf_metrics_update(void* stream, const float* sums_coarse, const float* sums_fine, const float* loss, int n_images,
                         int height, int width, int channels, double* state);
int knerf_composite(void* stream, const float* raw, const float* t, int n_rays, int n_samples, int white_background /* bit 0: white background, bit 1: no clip (utils.py:99-134) */,
                    float* image, float* depth, float* weights);
int knerf_inverse_cdf(void* stream, const float* mid_points, const float* weights, const float* u, int n_rays, int n_mid,
                      int n_weights, int n_samples, int oob_clamp, float* out);

/* Per-kernel timing with HIP events recorded on the caller's stream around every launch (bench.py's roofline leg).
 * Classes: 0 mlp_fwd coarse, 1 mlp_fwd fine, 2 composite, 3 sample_fine, 4 mlp_bwd coarse, 5 mlp_bwd fine,
 * 6 wgrad coarse, 7 wgrad fine, 8 adam+repack.  read() synchronises the device, returns summed milliseconds and launch
 * counts since enable/the previous read (n >= 9). */
int knerf_profile_enable(knerf_ctx* ctx, int on);
int knerf_profile_read(knerf_ctx* ctx, double* total_ms, int64_t* launches, int n);

/* Diagnostics (layout tables, workspace views, hardware-fact and bandwidth probes) are NOT part of this library: they are
 * declared in include/knerf_debug.h and built into libknerf_probe.so for tests/ and tools/ only. */

#ifdef __cplusplus
}
#endif
#endif /* KNERF_H */
