/* knerf_debug.h -- diagnostics for tests/ and tools/ only, built into libknerf_probe.so (keras_nerf_amd/build.py).
 *
 * Nothing here is part of the product ABI (include/knerf.h) and the product library does not contain it: layout-table
 * introspection for the CPU-side lane simulator (tests/test_layout_sim.py), views of a context's workspaces for kernel-level
 * parity tests, the hardware-fact probes of tests/test_gpu_probe.py and the bandwidth / MFMA-rate probes of tools/.
 * libknerf_probe.so links against libknerf_hip.so (same directory) and shares its internal context layout (csrc/ctx.h).
 */
#ifndef KNERF_DEBUG_H
#define KNERF_DEBUG_H
#include "knerf.h"

#ifdef __cplusplus
extern "C" {
#endif

/* kind 0: forward A-fragment table, 1: forward bias table, 2: dgrad A-fragment table, 3: wgrad destination tables
 * (concatenated), 4: offsets of the jobs inside kind 3.  Entries of 0..2 index the extended weight buffer (parameters, then the
 * composed head matrix [288][4] and its bias [4]; csrc/layout.h) or are -1; entries of 3 index the flat gradient, or
 * param_count + i for element i of the head accumulator, or are -1.  kind + 16 k: the same for entry k of the built-in trunk shapes
 * (csrc/layout.h KNERF_FUSED_SHAPES; composed head [dense_units + 32][4]); kind 5 + 16 k: {n_layers, skip_layer, dense_units,
 * param_count, pos_emb_xyz, pos_emb_dir} of entry k; an index behind the last entry fails.  Pass out=NULL to query the length.  No device needed. */
int knerf_debug_table(int kind, int32_t* out, size_t* n);
/* the general-shape path's layer program for a config (no device needed): 16 int32 per Dense layer in Keras order =
 * {kernel offset, bias offset, fan_in, fan_out, padded input width, padded output width, n_seg, seg0 (buffer col0, width,
 * kernel row0), seg1 (...), relu, head (-1 | 0 sigma | 1 rgb), padded width of the output buffer or -1}. */
int knerf_debug_generic_plan(const knerf_config* cfg, int32_t* out, size_t* n);
/* device buffers of the last knerf_train_chunk for kernel-level tests: 0 act, 1 mask, 2 dz, 3 raw, 4 draw,
 * 5 merged fine t-values, 6 coarse weights, 7 extended weight buffer of `net` (parameters + composed head); on a context of the
 * general-shape path (csrc/generic.h): 8 all activation buffers, 9 all dZ buffers of the last pass (0-2 belong to the fused path) */
int knerf_debug_buffer(knerf_ctx* ctx, int net, int which, void** dev, size_t* bytes);
/* hardware-fact probes: kind 0 = one v_mfma_f32_32x32x16_bf16 (in0 = A fragments [64][8] bf16, in1 = B fragments,
 * out = [64][16] f32); kind 1 = one ds_read_b64_tr_b16 (in0 = 4 KiB LDS image, in1 = [64] int32 byte offsets,
 * out = [64][4] u16).  All device pointers. */
int knerf_debug_probe(int kind, const void* in0, const void* in1, void* out, void* stream);
/* HBM write-pattern probe: `workgroups` x 8 waves each store `blocks` 1 KiB blocks into tiles `tile_stride` bytes apart;
 * mode 0 = the chain kernels' pattern, 1 = the 8 waves of a workgroup interleaved. */
int knerf_debug_write_probe(void* out, int workgroups, int blocks, long long tile_stride, int mode, int spin, void* stream);
/* HBM read-pattern probe: each of workgroups x 8 waves streams bytes_per_wave contiguous bytes in 1 KiB instructions;
 * mode 0 = nt LDS-DMA (wgrad's loads), 1 = plain register loads. */
int knerf_debug_read_probe(const void* in, int workgroups, long long bytes_per_wave, int mode, void* out, void* stream);
/* MFMA-shape rate probe (shape 32: v_mfma_f32_32x32x16_bf16, 16: v_mfma_f32_16x16x32_bf16) with the chain kernels' operand
 * traffic; `blocks` workgroups of 512 threads, 96 * 2^15 * 16 FLOP per wave and iteration. */
int knerf_debug_rate_probe(int shape, const void* in0, const void* in1, void* out, int blocks, int iters, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KNERF_DEBUG_H */
