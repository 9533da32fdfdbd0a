#!/usr/bin/env python3
"""bench.py -- NeRF train-step throughput on MI355X, in BASELINE.json's metric (rays*samples/s).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one full NeRF.train_step (reference keras_nerf/model/nerf/nerf.py:332-473) on one synthetic batch that is
already resident in HBM: chunk loop (coarse fwd+bwd, inverse-CDF sampling, fine fwd+bwd, gradient accumulation), the
data-parallel all-reduce(SUM) of the accumulated gradients when N > 1, both Adam updates and the bf16 weight re-pack.
PSNR/SSIM bookkeeping is excluded.  Per-GPU work is fixed as N grows ("weak" scaling, the
reference's train.py semantics: global batch = batch_size x replicas).

    python bench.py --mode fit [--config cfg2|cfg4]     what NeRF.fit delivers (loader -> train_step with metrics -> monitor), see bench_fit

One JSON line is printed by rank 0; see the task contract for the fields.  `roofline` is measured live with HIP events
around the dominant kernel (knerf_profile_*), `cpu_baseline` times the op-for-op torch-CPU restatement of the reference
path (oracle/torch_ref.py, kind "port") on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FWD_FLOP = 2 * 593408                      # per ray*sample, SURVEY.md section 2.2 / 8d: the reference's 12 Dense layers as written
TRAIN_FLOP = 3 * FWD_FLOP                  # the 3x-forward figure (the unit the step-level fractions are quoted in)
# what the fused kernels EXECUTE per ray*sample: the trunk as written; features -> rgb_features -> rgb (all linear in the
# reference) and sigma as one composed 283x4 head, i.e. 17 % fewer MACs for the same function
TRUNK_MAC = 63 * 256 + 4 * 256 * 256 + 319 * 256 + 2 * 256 * 256
FWD_FLOP_EXEC = 2 * (TRUNK_MAC + 283 * 4)
DGRAD_FLOP_EXEC = 2 * (256 * 4 + 7 * 256 * 256)
WGRAD_FLOP_EXEC = 2 * (TRUNK_MAC + 283 * 4)
TRAIN_FLOP_EXEC = FWD_FLOP_EXEC + DGRAD_FLOP_EXEC + WGRAD_FLOP_EXEC
MFMA_PEAK_TFLOPS = 2500.0                  # bf16 dense, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0

CONFIGS = {
    # name: (img_wh, batch per GPU, ray_chunks, description)
    "cfg2": (128, 2, 4096, "lego-shaped 128x128, batch 2 per GPU, ray_chunks 4096, coarse64+fine128"),
    "cfg3": (400, 1, 16000, "400x400, batch 1, ray_chunks 16000 (16384 does not divide 160000), coarse64+fine128"),
    "cfg4": (128, 1, 4096, "chair-shaped 128x128, 1 image per GPU, ray_chunks 4096, coarse64+fine128"),
    # forward only: one "step" = one frame of the 360-degree sweep of inference.py (pose -> rays -> render -> D2H)
    "cfg5": (256, 1, 4096, "inference.py-shaped 360-degree render, 256x256, ray_chunks 4096, coarse64+fine128, forward only"),
    # the command lines the reference's source comments quote a wall-clock time for: the only published numbers
    # of this path, 1x / 2x Tesla V100.  Parity-of-configuration lines, not the headline (BASELINE.json's metric is quoted on cfg2).
    "ref1": (128, 1, 2048, "train_single.py:16-17: --img_wh 128 --ray_chunks 2048, batch 1, coarse64+fine128 (reference: 3 s/step on 1x V100)"),
    "ref2": (128, 1, 4096, "train_single.py:18: --img_wh 128 --ray_chunks 4096, batch 1 (reference, eager: 2 s/step on 1x V100)"),
    "ref3": (128, 4, 4096, "train_single.py:19: --img_wh 128 --ray_chunks 4096 --batch_size 4 (reference, eager: 11 s/step on 1x V100)"),
    "ref4": (128, 1, 2048, "train.py:16-17: --img_wh 128 --ray_chunks 2048, one image per GPU (reference: 5-6 s/step on 2x V100; run with --gpus 2)"),
}
# seconds per step the reference's comments state for those command lines -> vs_baseline of THAT line only
PUBLISHED_S_PER_STEP = {"ref1": (3.0, "1x Tesla V100 32 GB, train_single.py:16-17"), "ref2": (2.0, "1x Tesla V100 32 GB, train_single.py:18"),
                        "ref3": (11.0, "1x Tesla V100 32 GB, train_single.py:19"), "ref4": (5.5, "2x Tesla V100 32 GB, train.py:16-17 (5-6 s)")}


def make_batch(nerf, wh, batch, rank, seed=42):
    """Synthetic nerf_synthetic-shaped batch generated on the device: pose_spherical poses,
    fov 0.6911112070083618, jittered coarse t, uniform random targets."""
    from keras_nerf_amd.data.utils import get_focal_from_fov, pose_spherical
    focal = get_focal_from_fov(0.6911112070083618, wh)
    poses = np.stack([pose_spherical(360.0 * (rank * batch + i) / 97.0, -30.0, 4.0) for i in range(batch)])
    o, d, t = nerf._ctx.generate_rays(poses, focal, wh, wh, 2.0, 6.0, nerf.n_coarse, None, seed=seed, stream_id=rank)
    g = torch.Generator(device="cuda"); g.manual_seed(seed + rank)
    images = torch.rand((batch, wh, wh, 3), device="cuda", generator=g)
    return images, (o, d, t)


def _cpu_problem(wh, n_images, n_rays=None):
    from oracle import nerf_oracle as O
    rng = np.random.default_rng(42)
    focal = O.get_focal_from_fov(0.6911112070083618, wh)
    os_, ds_, ts_ = [], [], []
    for i in range(n_images):
        o, d, t = O.generate_rays(O.pose_spherical(20.0 + 40.0 * i, -30.0, 4.0), focal, wh, wh, 2.0, 6.0, 64, rng.random((wh, wh, 64)))
        os_.append(o.reshape(-1, 3)); ds_.append(d.reshape(-1, 3)); ts_.append(t.reshape(-1, 64))
    o, d, t = [torch.tensor(np.concatenate(a)[:n_rays]) for a in (os_, ds_, ts_)]
    n = o.shape[0]
    img = torch.tensor(rng.random((n, 3), dtype=np.float32))
    u = torch.tensor(np.random.default_rng(7).random((n, 128), dtype=np.float32))
    return o, d, t, img, u


def _time_cpu_steps(step, want, budget_s, at_least=3):
    """1 warm-up, then up to `want` timed steps while the wall-clock budget lasts -- and `at_least` of them whatever they cost"""
    t0 = time.perf_counter(); step(); first = time.perf_counter() - t0
    times, spent = [], first
    while len(times) < want and (len(times) < at_least or spent + (times[-1] if times else first) <= budget_s):
        t0 = time.perf_counter(); step(); times.append(time.perf_counter() - t0); spent += times[-1]
    return times


def cpu_baseline(budget_s=70.0):
    """The reference's TF-CPU path cannot run here (no TensorFlow): its op-for-op torch-CPU restatement
    (oracle/torch_ref.py, autograd backward, Keras-form Adam; kind "port") is timed on this box's host cores, as SURVEY.md
    section 8d prescribes: (1) BASELINE configs[0] exactly -- 64x64 image, 4 chunks of 1024 rays, coarse net only, 64 samples;
    (2) a bounded sample of the benched workload's shape (cfg2: coarse64 + fine128, one chunk).  Median of 3 to 5 steps
    after 1 warm-up (5 while the wall-clock budget lasts, never fewer than 3: cfg1 costs ~13 s per step on the GPU box's host, so
    its leg is ~50 s); `cores` = torch threads actually used."""
    from oracle import nerf_oracle as O
    from oracle import torch_ref as T
    cfg = O.NerfConfig()
    threads = torch.get_num_threads()

    def fresh():
        cp = [torch.tensor(p, requires_grad=True) for p in O.init_params(cfg, 0)]
        fp = [torch.tensor(p, requires_grad=True) for p in O.init_params(cfg, 1)]
        return cp, fp, T.TorchKerasAdam(cp), T.TorchKerasAdam(fp)
    # (2) cfg2-shaped sample: 256 rays x (64 + 192) samples, one chunk
    o, d, t, img, u = _cpu_problem(16, 1)
    cp, fp, oc, of_ = fresh()
    t2 = _time_cpu_steps(lambda: T.train_step(cp, fp, oc, of_, img, o, d, t, u, cfg, 256, False), 5, 0.35 * budget_s)
    med2 = statistics.median(t2)
    # (1) cfg1 exactly
    o, d, t, img, u = _cpu_problem(64, 1)
    cp, fp, oc, of_ = fresh()
    t1 = _time_cpu_steps(lambda: T.train_step(cp, fp, oc, of_, img, o, d, t, u, cfg, 1024, False, coarse_only=True), 5, 0.65 * budget_s)
    med1 = statistics.median(t1)
    return {"value": 256 * 256 / med2, "unit": "rays*samples/s", "cores": threads, "kind": "port",
            "sample": f"cfg2-shaped: 256 rays x 256 samples (coarse64 + fine128), one chunk, full train step; median of {len(t2)} after "
                      f"1 warm-up; torch-CPU fp32 restatement of the reference's TF-CPU path, {threads} threads on {os.cpu_count()} host cpus",
            "s_per_step": med2, "steps_timed": len(t2),
            "cfg1": {"value": 4096 * 64 / med1, "unit": "rays*samples/s", "s_per_step": med1, "steps_timed": len(t1),
                     "sample": "BASELINE configs[0] exactly: 64x64 image, batch 1, ray_chunks 1024 (4 chunks), coarse net only "
                               f"(64 samples), forward + backward + Adam; median of {len(t1)} steps after 1 warm-up"}}


_REAL_STDOUT = None


def own_stdout():
    """The contract is ONE JSON line on stdout.  Libraries write there too -- RCCL prints a five-line version banner with plain
    printf when NCCL_DEBUG is set (found by the one-rank rehearsal, round 5), gloo its "[Gloo] Rank r is connected" lines -- so the
    process's file descriptor 1 is pointed at stderr for the whole run and the line goes to a private copy of the real stdout."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(line):
    sys.stdout.flush()
    if _REAL_STDOUT is None:
        print(line, flush=True)
    else:
        os.write(_REAL_STDOUT, (line + "\n").encode())


def dist_on(world):
    """N > 1 -- or the one-rank rehearsal of the N > 1 path (KNERF_DIST_SINGLE=1, keras_nerf_amd/parallel.py): every branch below
    that holds a collective, times it or reports it is taken, over the real backend, on a one-GPU box"""
    from keras_nerf_amd import parallel
    return world > 1 or parallel.single_rank_rehearsal()


def sync(world):
    torch.cuda.synchronize()
    if dist_on(world):
        torch.distributed.barrier()
        torch.cuda.synchronize()


RANK_ELAPSED = []      # every rank's elapsed time of the last timed region, in rank order (filled by max_over_ranks)


def max_over_ranks(elapsed, world):
    RANK_ELAPSED[:] = [elapsed]
    if dist_on(world):          # all_reduce is the one collective both RCCL and gloo run on device tensors: rank r fills slot r, SUM
        every = torch.zeros(world, device="cuda", dtype=torch.float64)
        every[torch.distributed.get_rank()] = elapsed
        torch.distributed.all_reduce(every)
        RANK_ELAPSED[:] = [float(v) for v in every.cpu()]
        elapsed = max(RANK_ELAPSED)
    return elapsed


def dist_fields(world, backend, steps):
    from keras_nerf_amd import parallel
    per = [e / steps * 1e3 for e in RANK_ELAPSED]
    # launch_attempts 2 = the launcher's first set of ranks never got a working process group and a second, fresh set under the
    # other HSA_ENABLE_IPC_MODE_LEGACY setting produced this line (keras_nerf_amd/parallel.py _launch_attempts)
    return {**parallel.launch_fields(), "rccl_ranks": torch.distributed.get_world_size() if dist_on(world) and backend == "nccl" else (1 if world == 1 else 0),
            "dist_backend": backend if dist_on(world) else None,
            "ms_per_step_rank_min": min(per), "ms_per_step_rank_max": max(per), "ms_per_step_by_rank": [round(v, 4) for v in per]}


def bench_render(args, world, rank, wh, chunks, desc, backend="nccl"):
    """cfg5: frames/s of the 360-degree render loop of the reference's inference.py:62-114 (theta sweep at phi=-30,
    radius 4; rays generated on the device; fine image + depth copied to the host per frame as the reference does -- pipelined: pinned
    double buffers on a side stream, the host collects frame i-1 while frame i renders)."""
    from keras_nerf_amd.data.rays import RaysGenerator
    from keras_nerf_amd.data.utils import get_focal_from_fov, pose_spherical
    from keras_nerf_amd.model.nerf.nerf import NeRF
    nerf = NeRF(seed=0)
    nerf.compile(optimizer="adam", loss="mse", batch_size=1, image_height=wh, image_width=wh, ray_chunks=chunks,
                 white_background=True, is_training=False)
    # rays per set of launches: the largest whole number of chunks of a frame within the library's limit (knerf.h "merge_render_rays")
    lim, n_chunks = int(nerf._ctx.get_option("merge_render_rays")), wh * wh // chunks
    launch_rays = chunks * max([m for m in range(1, n_chunks + 1) if n_chunks % m == 0 and m * chunks <= lim] or [1])
    rg = RaysGenerator(get_focal_from_fov(0.6911112070083618, wh), wh, wh, 2.0, 6.0, nerf.n_coarse, seed=rank)
    n_frames = args.steps
    poses = [pose_spherical(360.0 * i / max(n_frames, 1), -30.0, 4.0) for i in range(n_frames + args.warmup)]

    # inference.py:108-114 reads the fine image and depth of every frame back (`.numpy()`).  Here: only those two outputs are
    # computed (outputs=...: the four [N, S] weight arrays are neither allocated nor written), and the copy of frame i runs on a side
    # stream into pinned, double-buffered host memory while frame i+1 renders; the host waits for frame i-1's copy, not for the GPU.
    copy_stream = torch.cuda.Stream()
    pinned = [(torch.empty((1, wh, wh, 3), pin_memory=True), torch.empty((1, wh, wh), pin_memory=True)) for _ in range(2)]
    copied = [torch.cuda.Event(), torch.cuda.Event()]
    frames_out = []

    def frame(i, k):
        o, d, t = rg(poses[i])
        _, fine = nerf.predict_and_render_images((o[None], d[None], t[None]), outputs=("image", "depth"))
        rendered = torch.cuda.Event(); rendered.record()
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(rendered)
            pinned[k][0].copy_(fine["image"], non_blocking=True); pinned[k][1].copy_(fine["depth"], non_blocking=True)
            fine["image"].record_stream(copy_stream); fine["depth"].record_stream(copy_stream)
            copied[k].record(copy_stream)

    def collect(k):          # the host-side end of a frame: its pixels are in pageable memory, the pinned pair is free again
        copied[k].synchronize()
        frames_out.append((pinned[k][0].numpy().copy(), pinned[k][1].numpy().copy()))
    for i in range(args.warmup):
        frame(i, i & 1); collect(i & 1)
    frames_out.clear()
    sync(world)
    t0 = time.perf_counter()
    for i in range(n_frames):
        frame(args.warmup + i, i & 1)
        if i:
            collect((i - 1) & 1)
    collect((n_frames - 1) & 1)
    sync(world)
    elapsed = max_over_ranks(time.perf_counter() - t0, world)
    assert len(frames_out) == n_frames
    fps = world * n_frames / elapsed
    rs = fps * wh * wh * 256
    roofline = None
    if rank == 0:
        nerf._ctx.profile_enable(True); nerf._ctx.profile_read()
        frame(0, 0); collect(0)
        prof = nerf._ctx.profile_read(); nerf._ctx.profile_enable(False)
        ms, cnt = prof["mlp_fwd_fine"]
        avg = ms / max(cnt, 1)
        # executed FLOPs (collapsed head, not the 12-layer count) of ONE launch: the launches of the profiled frame cover wh * wh rays
        # between them (merge_render_rays: normally one launch per frame; round 5 left `chunks` here and under-reported 16x)
        flop = FWD_FLOP_EXEC * (wh * wh / max(cnt, 1)) * (nerf.n_coarse + nerf.n_fine)
        roofline = {"bound": "mfma", "kernel": "mlp_fwd_fine", "achieved": flop / (avg * 1e-3) / 1e12, "peak": MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": flop / (avg * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, "traffic": None, "avg_launch_ms": avg,
                    "launches": cnt, "kernel_ms_per_frame": {k: round(v[0], 4) for k, v in prof.items() if v[1]},
                    "frame_tflops": rs * FWD_FLOP / 1e12, "frame_frac_of_mfma_peak": rs * FWD_FLOP / 1e12 / MFMA_PEAK_TFLOPS,
                    "frame_executed_tflops": rs * FWD_FLOP_EXEC / 1e12}
        emit(json.dumps({"metric": "frames/sec (360-degree render 256^2, coarse64+fine128, forward only)", "value": fps,
                          "unit": "frames/s", "n_gpus": world, "steps": n_frames, "warmup": args.warmup,
                          "ms_per_step": elapsed / n_frames * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "bf16", "data": "synthetic", "rays_samples_per_s": rs,
                          "config": {"workload": f"cfg5: {desc}", "frames": n_frames, "parallelism": f"dp{world}",
                                     "launch_rays": launch_rays, "launch_note": "consecutive chunks of a frame share launches (option merge_render_rays; per-ray work only: outputs bit-identical for every value)", "readback": "pinned double buffers on a side stream (tools/cfg5_readback_probe.py: A/B against blocking copies)"},
                          "roofline": roofline, "cpu_baseline": None, **dist_fields(world, backend, n_frames), **provenance()}))


# per 32-sample tile, KiB (csrc/layout.h): saved activations (h0 is recomputed, not saved), dZ WRITTEN (the run has 130 blocks;
# the 16 of dz7 are never written: wgrad recomputes them from the dz_head block and the layer-7 mask block), relu masks.
# wgrad reads the 4 enc blocks three times (layer_0, layer_1's h0 recomputation, layer_5), the dz_head block twice (head job,
# layer_7's recomputation) and one mask block.
ACT_KIB, DZ_KIB, MASK_KIB, WGRAD_REREAD_KIB = 118, 114, 8, 10
LAYOUT_TAG = f"act{ACT_KIB}_dz{DZ_KIB}"
WGRAD_KIB_PER_TILE = ACT_KIB + DZ_KIB + WGRAD_REREAD_KIB
# whole step, per 32-sample tile: fwd writes act + masks, dgrad reads masks + raw/draw and writes dZ, wgrad reads act + dZ (+ its
# re-reads); raw/draw/t: 32 samples x (16 B written + 16 B read) x 2 + t
STEP_KIB_PER_TILE = 2 * (ACT_KIB + DZ_KIB + MASK_KIB) + WGRAD_REREAD_KIB + 2.5


def pmc_traffic(kernel, skip_dead_tiles, kernel_digest="unchecked", profiles_dir=None):
    """(HBM bytes per launch of `kernel`, source file) from the newest committed PMC summary of THE KERNELS THIS PROCESS LOADED
    (`_kernel_digest` of the summary == kernel_digest of the loaded library, keras_nerf_amd/_lib.py build_info: the digest of the three
    big kernels' sources and flags, tied to the binary by the library's own hash; a summary without a digest, one of other kernels,
    or a library without a valid record gives (None, None) -- no counters are better than stale ones), of this data layout, that profiled the
    INSTANTIATION the bench ran (profiles/*pmc_traffic*.json, produced by tools/pmc.sh + tools/pmc_report.py on the GPU box: separate
    rocprofv3 --pmc passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950, WRITE_SIZE as is,
    both in KiB).  Counters cannot be read from inside the benched process, so the line names its source.  The weight-gradient kernel
    has a list-mode instantiation (`wgrad_kernel<Shape, NET, true>`: skip_dead_tiles on, the default) and a contiguous one
    (`..., false>`); the dgrad kernel takes its list at run time, so for it the summary's recorded `_options` must match.  A summary
    that holds only the other instantiation is NOT quoted: (None, None)."""
    import glob
    net = "1" if kernel.endswith("fine") else "0"
    lst = "true" if skip_dead_tiles else "false"
    # (name prefix, the end of the template argument list that must match)
    want = {"wgrad": ("wgrad_kernel<", f", {net}, {lst}>"), "mlp_fwd": ("mlp_fwd_kernel<", f", true, {net}>"),
            "mlp_bwd": ("mlp_bwd_kernel<", f", {net}>")}.get(kernel.rsplit("_", 1)[0])
    if not want:
        return None, None
    if kernel_digest is None:
        return None, None
    for f in sorted(glob.glob(os.path.join(profiles_dir or os.path.join(ROOT, "profiles"), "*pmc_traffic*.json")), reverse=True):
        rep = json.load(open(f))
        if rep.get("_layout", "act158_dz156") != LAYOUT_TAG:
            continue
        if kernel_digest != "unchecked" and rep.get("_kernel_digest") != kernel_digest:
            continue
        opt = rep.get("_options", {}).get("skip_dead_tiles")          # absent in summaries older than round 4: they profiled skipping off
        if kernel.startswith("mlp_bwd") and bool(opt) != bool(skip_dead_tiles):
            continue
        rows = [v for k, v in rep.items() if k.startswith(want[0]) and k.split(" grid=")[0].endswith(want[1]) and isinstance(v, dict)
                and "hbm_bytes_per_launch" in v]
        if not rows:
            continue
        return max(r["hbm_bytes_per_launch"] for r in rows), os.path.relpath(f, ROOT) if f.startswith(ROOT) else f
    return None, None


def provenance():
    """what produced this line: the library that was loaded (path, hash of the file, digest of its three big kernels' sources, the
    commit it was built at), this script's hash and -- where the tree is a repository -- its commit"""
    from keras_nerf_amd import _lib
    from keras_nerf_amd.build import file_sha16, git_head
    info = _lib.build_info()
    return {"lib_sha16": info["lib_sha16"], "lib_path": info["lib_path"], "kernel_digest": info["kernel_digest"],
            "git_head": git_head(), "git_head_at_build": info["git_head_at_build"], "bench_py_sha16": file_sha16(os.path.abspath(__file__))}


def write_synthetic_dataset(root, wh, n=(100, 4, 4)):
    """nerf_synthetic-layout directory (transforms_{train,val,test}.json + RGBA PNGs, reference loader.py:55-113) with procedural
    images: 100 training views like the real scenes; the pixel content is irrelevant to throughput"""
    from PIL import Image
    from keras_nerf_amd.data.utils import pose_spherical
    yy, xx = np.mgrid[0:wh, 0:wh].astype(np.float32) / wh
    for subset, cnt in zip(("train", "val", "test"), n):
        os.makedirs(os.path.join(root, subset), exist_ok=True)
        frames = []
        for i in range(cnt):
            r = np.hypot(xx - 0.5 + 0.1 * np.sin(i), yy - 0.5) * 3
            alpha = (r < 1).astype(np.float32)
            rgb = np.stack([1 - r, 0.5 + 0.5 * np.cos(i + 3 * r), r], -1).clip(0, 1) * alpha[..., None]
            Image.fromarray((np.concatenate([rgb, alpha[..., None]], -1) * 255).astype(np.uint8), "RGBA").save(os.path.join(root, subset, f"r_{i}.png"))
            frames.append({"file_path": f"./{subset}/r_{i}", "transform_matrix": np.asarray(pose_spherical(360.0 * i / cnt, -30.0, 4.0)).tolist()})
        json.dump({"camera_angle_x": 0.6911112070083618, "frames": frames}, open(os.path.join(root, f"transforms_{subset}.json"), "w"))
    return root


def bench_fit(args, world, rank, wh, batch, chunks, desc, backend):
    """`--mode fit`: the rate of the thing train_single.py:137-143 calls -- DatasetLoader (PNG decode, device-resident views,
    on-device rays) -> NeRF.fit -> train_step WITH its six metrics -> NeRFTrainMonitor (panels off; CSV, checkpoint and the two
    monitor renders at every epoch end) -- against the plain `train_step(with_metrics=False)` loop on one resident batch (the
    default mode's loop), both with dead-tile skipping off so that the two loops do the same work.  One warm-up epoch (decodes the PNGs, fills the device cache), then `--epochs` timed epochs of 100
    training views.  value = rays*samples of the training batches / time of the train loops (from an epoch's first batch to the
    read-back of its logs, GPU drained); `fit_wall_*` adds validation and the monitor.  One further epoch with skipping on (the
    library default) is timed and reported with its dead-tile share."""
    import tempfile
    from keras_nerf_amd.data.loader import DatasetLoader
    from keras_nerf_amd.model.nerf.callback import NeRFTrainMonitor
    from keras_nerf_amd.model.nerf.nerf import NeRF
    # ONE directory for all ranks (rank 0 makes it and writes the dataset, the others learn its name: every rank reads the same
    # transforms_*.json and derives the same shuffled order, data/loader.py)
    box = [tempfile.mkdtemp(prefix="knerf_fit_") if rank == 0 else None]
    if dist_on(world):
        torch.distributed.broadcast_object_list(box, src=0)
    root = box[0]
    if rank == 0:
        write_synthetic_dataset(os.path.join(root, "data"), wh)
    if dist_on(world):
        torch.distributed.barrier()
    # the GLOBAL batch, as train.py:84-93 passes it: every rank yields its slice of `batch` images
    train, val, test = DatasetLoader(os.path.join(root, "data"), white_background=True).load_dataset(batch * world, wh, wh, 2.0, 6.0, 64)
    nerf = NeRF(seed=100 + rank if dist_on(world) else 0)     # N > 1: own initial weights per rank; compile() mirrors rank 0's
    try:
        nerf.compile(optimizer="adam", loss="mse", batch_size=batch, image_height=wh, image_width=wh, ray_chunks=chunks, white_background=True)
    except Exception as e:                     # noqa: BLE001
        if dist_on(world):
            rank_fail("NeRF.compile (weight broadcast)", e)
        raise
    monitor = NeRFTrainMonitor(test, os.path.join(root, "log"), batch, update_freq=1, plots=False)
    marks = []

    class Clock:
        def on_epoch_begin(self, epoch, logs=None):
            torch.cuda.synchronize()
            nerf._ctx.tile_stats(reset=True)          # dead-tile share per epoch (read after the skipping epoch)
            marks.append(["begin", time.perf_counter()])

        def on_test_begin(self, logs=None):          # fit has read the epoch's logs back: the train loop is over and the GPU drained
            marks.append(["train_end", time.perf_counter()])

        def on_epoch_end(self, epoch, logs=None):
            torch.cuda.synchronize(); marks.append(["end", time.perf_counter()])
    # The overhead comparison runs with dead-tile skipping OFF in both loops: with it on, the work of a step depends on how far the
    # scene has trained (the disc images' exact white background is dead within an epoch), which says nothing about loader, metrics
    # or monitor.  One more epoch with the option on follows and is reported beside it.
    skip_default = bool(nerf._ctx.get_option("skip_dead_tiles"))
    nerf._ctx.set_option("skip_dead_tiles", 0)
    epochs = 1 + args.epochs
    nerf.fit(train, epochs=epochs, validation_data=val, callbacks=[Clock(), monitor], verbose=0)
    torch.cuda.synchronize()
    t = {k: [m[1] for m in marks if m[0] == k] for k in ("begin", "train_end", "end")}
    steps = len(train)
    loop = sum(t["train_end"][e] - t["begin"][e] for e in range(1, epochs))
    wall = t["end"][-1] - t["begin"][1]
    per_step = wh * wh * batch * 256
    value = world * per_step * steps * args.epochs / loop
    # the plain train_step loop in the same process on one resident batch of the same dataset: no loader, no metrics, no callbacks
    data = next(iter(train))
    for _ in range(3):
        nerf.train_step(data, with_metrics=False)
    sync(world); t0 = time.perf_counter()
    for _ in range(20):
        nerf.train_step(data, with_metrics=False)
    sync(world); ts = (time.perf_counter() - t0) / 20
    for _ in range(3):
        nerf.train_step(data, sync=False)
    nerf._metric_state.events = []               # HIP events around the three metric launches of every step (metrics.py)
    sync(world); t0 = time.perf_counter()
    for _ in range(20):
        nerf.train_step(data, sync=False)
    sync(world); tm = (time.perf_counter() - t0) / 20
    ev, nerf._metric_state.events = nerf._metric_state.events, None
    metrics_ms = sum(a_.elapsed_time(b_) for a_, b_ in ev) / max(len(ev), 1)
    # ... and what fit delivers with skipping on (the default), at this point of the training
    nerf._ctx.set_option("skip_dead_tiles", 1)
    skip_ms = dead_fit = None
    if nerf._ctx.get_option("skip_dead_tiles_active"):
        marks.clear()
        nerf.fit(train, epochs=epochs + 1, initial_epoch=epochs, validation_data=val, callbacks=[Clock(), monitor], verbose=0)
        torch.cuda.synchronize()
        b = [m[1] for m in marks if m[0] == "begin"]; e = [m[1] for m in marks if m[0] == "train_end"]
        skip_ms = (e[0] - b[0]) / steps * 1e3
        live, total = nerf._ctx.tile_stats(reset=True)
        dead_fit = 1.0 - live / max(total, 1) if total else None
    nerf._ctx.set_option("skip_dead_tiles", int(skip_default))
    drift = {}
    if args.check_replicas:
        drift["replica_drift"], drift["weight_checksum"] = replica_drift(nerf, world)
    if rank == 0:
        emit(json.dumps({**drift, "metric": f"rays*samples/sec (NeRF.fit train loop with metrics, loader and monitor), {args.config}", "value": value,
                          "unit": "rays*samples/s", "n_gpus": world, "steps": steps * args.epochs, "warmup": steps, "ms_per_step": loop / (steps * args.epochs) * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                          "config": {"workload": f"{args.config} through NeRF.fit: {desc}; 100 procedural training views in nerf_synthetic layout, "
                                                 f"{steps} steps per epoch, {args.epochs} timed epochs after one warm-up epoch", "parallelism": f"dp{world}"},
                          "train_step_ms": ts * 1e3, "train_step_with_metrics_ms": tm * 1e3, "metrics_ms_per_step": metrics_ms,
                          "metrics_clock": "hip events around the 3 metric launches, mean of 20 steps",
                          "fit_vs_train_step": (loop / (steps * args.epochs)) and ts / (loop / (steps * args.epochs)),
                          "options_during_comparison": {"skip_dead_tiles": False, "deterministic": False},
                          "fit_ms_per_step_with_skip_dead_tiles": skip_ms, "dead_tile_frac_that_epoch": dead_fit,
                          "fit_wall_s": wall, "fit_wall_rays_samples_per_s": world * per_step * steps * args.epochs / wall,
                          "epoch_end_s": [t["end"][e] - t["train_end"][e] for e in range(1, epochs)],
                          "roofline": None, "cpu_baseline": None, **dist_fields(world, backend, 1), **provenance()}))
    if dist_on(world):
        torch.distributed.barrier()            # nobody is still reading when rank 0 removes the directory
    if rank == 0:
        import shutil
        shutil.rmtree(root, ignore_errors=True)


def rank_fail(what, exc):
    """keras_nerf_amd/parallel.py rank_fail under this script's tag: the rank names itself, prints the tail of its RCCL warnings
    and leaves with exit code 3; the launcher stops the others"""
    from keras_nerf_amd import parallel
    parallel.rank_fail(what, exc, tag="bench")


def inject(stage):
    """fault injection for the fail-fast test (tests/test_gpu_api.py): KNERF_BENCH_INJECT_FAILURE="<rank>:<stage>" makes that rank
    raise at that stage (init | first_all_reduce | compile | warmup)"""
    from keras_nerf_amd import parallel
    parallel.inject(stage)


def replica_drift(nerf, world):
    """`--check-replicas`: mirrored variables must be IDENTICAL on every rank after identical Adam updates on the all-reduced
    gradient (train.py:110-148).  An exact 64-bit checksum of both nets' fp32 master weights (their bit patterns summed as
    integers) is reduced with MAX and MIN; the line carries max - min, which must be 0 under SUM and under mean."""
    w = torch.cat([nerf._ctx.weights_view(0), nerf._ctx.weights_view(1)])
    bits = w.view(torch.int32).to(torch.int64)
    chk = (bits * (torch.arange(bits.numel(), device=bits.device, dtype=torch.int64) % 8191 + 1)).sum().reshape(1)
    hi, lo = chk.clone(), chk.clone()
    if dist_on(world):
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
    return float(int(hi[0]) - int(lo[0])), int(chk[0])


def allreduce_selftest(nerf, world, backend, n=20):
    """N > 1, before the timed region: `n` stand-alone all-reduces of the REAL operand (the library-owned 4.77 MB gradient buffer;
    all zeros at this point, so the sums change nothing) with HIP events on the stream the collective runs on -- the first RCCL run
    with N > 1 happens on the driver's node with nobody to debug it, so the line itself must say whether the collective is healthy:
    min / median microseconds, the implied bus bandwidth of a ring (2 (N-1)/N x bytes / time; one xGMI link is ~153 GB/s peak per
    direction), RCCL's version.  Under gloo (rehearsal) the collective is synchronous on the host: wall clock is reported beside it."""
    g = nerf._ctx.grads_view()
    byts = int(g.numel()) * 4
    for _ in range(3):
        torch.distributed.all_reduce(g)
    torch.cuda.synchronize()
    ev, wall = [], []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(); torch.distributed.all_reduce(g); e1.record()
        if backend != "nccl":
            torch.cuda.synchronize()
        wall.append((time.perf_counter() - t0) * 1e6); ev.append((e0, e1))
    torch.cuda.synchronize()
    us = sorted(a.elapsed_time(b_) * 1e3 for a, b_ in ev)
    was_zero = float(g.abs().max()) == 0.0      # the accumulators were zero (Adam resets them) and sums of zeros still are;
    nerf._ctx.zero_grads()                      # whatever happened, the timed steps start from zero accumulators
    med = statistics.median(us if backend == "nccl" else wall)
    try:
        ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:                            # noqa: BLE001
        ver = None
    return {"allreduce_us_standalone": {"min": us[0], "median": statistics.median(us), "max": us[-1], "n": n, "clock": "hip events on the collective's stream",
                                        "host_wall_us_median": statistics.median(wall)},
            "allreduce_busbw_GBps": byts * 2 * (world - 1) / world / (med * 1e-6) / 1e9, "rccl_version": ver,
            "allreduce_selftest_operand_stayed_zero": was_zero}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--ray-chunks", type=int, default=None, help="override the configuration's ray_chunks (sensitivity sweeps; the line's workload says so)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--mode", default="step", choices=["step", "fit"], help="step: the train_step loop on one resident batch (the contract's line); "
                                                                            "fit: NeRF.fit through the loader, the metrics and the monitor")
    ap.add_argument("--epochs", type=int, default=2, help="--mode fit: timed epochs")
    ap.add_argument("--skip-dead-tiles", type=int, default=None, help="override the library default of the skip_dead_tiles option (0/1)")
    ap.add_argument("--deterministic", type=int, default=None, help="set the deterministic option (0/1): gradient sums without atomics")
    ap.add_argument("--ignore-nonfinite", action="store_true", help="TIMING EXPERIMENTS with ablation builds only (their gradients are garbage): "
                    "a step skipped for a non-finite gradient is not reported -- it costs the same launches, the Adam kernels are predicated off")
    ap.add_argument("--check-replicas", type=int, default=None, help="after the timed region: exact weight checksum of every rank reduced "
                    "with MAX and MIN; `replica_drift` (must be 0.0) goes on the line.  Default: on for N > 1 (two one-word collectives, "
                    "outside every timed region), off for N = 1")
    args = ap.parse_args()
    from keras_nerf_amd import parallel
    if args.check_replicas is None:
        args.check_replicas = int(args.gpus > 1 or parallel.single_rank_rehearsal())
    # the same environment for self-spawned ranks and for ranks of an external launcher (MASTER_ADDR, dmabuf IPC; N > 1: RCCL's
    # warnings to one file per rank, printed by a rank that fails)
    parallel.dist_env(args.gpus)

    # KNERF_DIST_BACKEND=gloo rehearses the N>1 control flow on a box with fewer GPUs than ranks (ranks then share devices)
    backend = os.environ.get("KNERF_DIST_BACKEND", "nccl")
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: keras_nerf_amd.parallel.launch re-runs this script as N rank processes
        # (children started before this process touches the GPU, never an exec), waits, and exits with their code
        parallel.launch(None, args.gpus, backend=backend)
    own_stdout()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start bench.py either plainly (it spawns its own ranks) or "
                         f"under torch.distributed.run with --nproc-per-node {args.gpus}")
    # joins the process group (300 s time-out) and proves it with a one-word all-reduce; every failure names rank, device and stage
    rank, world, device_index = parallel.init_rank(backend, tag="bench")
    try:
        run(args, world, rank, device_index, backend)
    except Exception as e:                     # noqa: BLE001 -- N > 1: ANY rank that raises says who it is and ends the job (exit 3);
        if dist_on(world):                          # its peers would otherwise meet it as "connection closed by peer" in their next collective
            rank_fail("the benchmark body", e)
        raise
    if dist_on(world):
        torch.distributed.destroy_process_group()


def run(args, world, rank, device_index, backend):
    """everything behind the process group's set-up: one of the three benchmark bodies"""
    from keras_nerf_amd.model.nerf.nerf import NeRF
    wh, batch, chunks, desc = CONFIGS[args.config]
    if args.ray_chunks:
        chunks, desc = args.ray_chunks, desc + f" [ray_chunks overridden: {args.ray_chunks}]"
    if args.config == "cfg5":
        return bench_render(args, world, rank, wh, chunks, desc, backend)
    if args.mode == "fit":
        RANK_ELAPSED[:] = [0.0]
        return bench_fit(args, world, rank, wh, batch, chunks, desc, backend)
    nerf = NeRF(seed=100 + rank if dist_on(world) else 0)     # N > 1: every rank draws its OWN initial weights; compile() must mirror rank 0's
    try:
        inject("compile")
        nerf.compile(optimizer="adam", loss="mse", batch_size=batch, image_height=wh, image_width=wh, ray_chunks=chunks,
                     white_background=True)
    except Exception as e:                     # noqa: BLE001 -- N > 1: the weight broadcast of compile() is the second collective
        if dist_on(world):
            rank_fail("NeRF.compile (weight broadcast)", e)
        raise
    if args.ignore_nonfinite:
        nerf._ctx.poll_nonfinite = lambda wait=False: None
    if args.skip_dead_tiles is not None:
        nerf._ctx.set_option("skip_dead_tiles", args.skip_dead_tiles)
    if args.deterministic is not None:
        nerf._ctx.set_option("deterministic", args.deterministic)
    data = make_batch(nerf, wh, batch, rank)
    n_rays = batch * wh * wh
    lim, n_chunks = int(nerf._ctx.get_option("merge_chunk_rays")), n_rays // chunks
    launch_rays = chunks * max([m for m in range(1, n_chunks + 1) if n_chunks % m == 0 and m * chunks <= lim] or [1])
    samples_per_ray = nerf.n_coarse + (nerf.n_coarse + nerf.n_fine)          # 64 + 192 = 256 MLP evaluations per ray

    try:
        inject("warmup")
        for _ in range(args.warmup):
            nerf.train_step(data, with_metrics=False)
        sync(world)
    except Exception as e:                     # noqa: BLE001 -- N > 1: the first gradient all-reduce (4.77 MB) runs in here
        if dist_on(world):
            rank_fail("the warm-up steps (first gradient all-reduce)", e)
        raise
    selftest = {}
    if dist_on(world):
        selftest = allreduce_selftest(nerf, world, backend)
        sync(world)
        nerf._allreduce_events = []            # HIP events on the compute stream around the gradient all-reduce of every timed step
    t0 = time.perf_counter()
    for _ in range(args.steps):
        nerf.train_step(data, with_metrics=False)
    sync(world)
    local_elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(local_elapsed, world)
    value = world * n_rays * samples_per_ray * args.steps / elapsed
    comm = {}
    if args.check_replicas:
        comm["replica_drift"], comm["weight_checksum"] = replica_drift(nerf, world)
    if dist_on(world):
        ev, nerf._allreduce_events = nerf._allreduce_events, None
        ms = [a.elapsed_time(b_) for a, b_ in ev]
        # from the moment this rank's last chunk has finished to the moment the reduced gradients are usable: the collective
        # itself plus the wait for the slowest rank
        comm.update({"allreduce_ms_per_step": sum(ms) / max(len(ms), 1), "allreduce_ms_per_step_max": max(ms) if ms else None,
                     "grad_bytes": int(nerf._ctx.grads_view().numel()) * 4})
    if dist_on(world):          # every rank: its share of the device (ranks sharing a GPU in a rehearsal must all fit; wgrad_group follows free memory)
        free_b, total_b = torch.cuda.mem_get_info()
        print(f"[bench rank {rank}/{world}] device memory free {free_b / 2**30:.1f} GiB of {total_b / 2**30:.1f} GiB; "
              f"wgrad_group {int(nerf._ctx.get_option('wgrad_group'))} (budget {nerf._ctx.get_option('wgrad_group_gb'):g} GB)", file=sys.stderr, flush=True)
    # what the six logged metrics add to a step (SURVEY.md 8d reports them separately): HIP events on the stream around the two
    # image-metric launches and the update of the device-side means (MetricState.update), in NeRF.fit's asynchronous form.  (Rounds
    # 3-4 took the difference of two ~3 s loops, which does not resolve three small launches: -0.14 ms on the r04 line.)
    # Every rank runs the same number of steps (each holds an all-reduce).
    nerf._metric_state.events = []
    for _ in range(6):
        nerf.train_step(data, sync=False)
    sync(world)
    ev, nerf._metric_state.events = nerf._metric_state.events[1:], None      # the first step's launches include one-time set-up
    metrics_ms = sum(a_.elapsed_time(b_) for a_, b_ in ev) / max(len(ev), 1)

    roofline = None
    if not args.no_profile:
        # every rank takes the two extra steps (train_step all-reduces); only rank 0 records HIP events around its kernels
        if rank == 0:
            nerf._ctx.profile_enable(True)
            nerf._ctx.profile_read()
        for _ in range(2):
            nerf.train_step(data, with_metrics=False)
        sync(world)
    if rank == 0 and not args.no_profile:
        prof = nerf._ctx.profile_read()
        nerf._ctx.profile_enable(False)
        total = sum(ms for ms, _ in prof.values())
        dom = max(prof, key=lambda k: prof[k][0])
        ms, cnt = prof[dom]
        avg_ms = ms / max(cnt, 1)
        # samples per launch of the dominant kernel: the profiled region is two steps; the chain kernels run once per chunk, the
        # weight-gradient kernel once per GROUP of chunks (knerf_train_batch defers it: up to the whole step in one launch)
        per_ray = nerf.n_coarse if dom.endswith("coarse") else nerf.n_coarse + nerf.n_fine
        per_launch_samples = 2 * n_rays * per_ray / max(cnt, 1)
        chunk_samples = chunks * per_ray
        flop = {"mlp_fwd": FWD_FLOP_EXEC, "mlp_bwd": DGRAD_FLOP_EXEC, "wgrad": WGRAD_FLOP_EXEC}.get(dom.rsplit("_", 1)[0], 0) * per_launch_samples
        # wgrad streams the saved activations and dZ once (+ the re-read of enc for layer_5)
        if dom.startswith("wgrad"):
            byts = per_launch_samples / 32 * WGRAD_KIB_PER_TILE * 1024
            roofline = {"bound": "hbm", "kernel": dom, "achieved": byts / (avg_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": byts / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None}
        else:
            roofline = {"bound": "mfma", "kernel": dom, "achieved": flop / (avg_ms * 1e-3) / 1e12, "peak": MFMA_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": flop / (avg_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, "traffic": None}
        roofline["traffic"], roofline["traffic_source"] = pmc_traffic(dom, bool(nerf._ctx.get_option("skip_dead_tiles_active")), provenance()["kernel_digest"])
        if roofline["traffic"] is not None and per_launch_samples != chunk_samples:
            # the committed PMC summary profiles one-chunk launches (tools/kbench.py); this launch covers several chunks of the same tiles
            roofline["traffic"] *= per_launch_samples / chunk_samples
            roofline["traffic_source"] += f" x {per_launch_samples / chunk_samples:g} (chunks per launch)"
        roofline["avg_launch_ms"] = avg_ms
        roofline["launches"] = cnt
        roofline["kernel_ms_per_step"] = {k: round(v[0] / 2, 4) for k, v in prof.items()}
        # the other big kernels against the same roof, algorithmic bytes per 32-sample tile: the training forward
        # WRITES act + masks, dgrad writes dZ and reads masks + raw/draw, wgrad reads act + dZ (+ its re-reads).  The chain kernels are
        # write-bound (the write path of this chip reaches ~5.4 TB/s in a bare probe), wgrad read-bound.
        per_tile_kib = {"mlp_fwd": ACT_KIB + MASK_KIB + 0.5, "mlp_bwd": DZ_KIB + MASK_KIB + 1.0, "wgrad": WGRAD_KIB_PER_TILE}
        roofline["kernels"] = {}
        for k, (ms_k, cnt_k) in prof.items():
            base = k.rsplit("_", 1)[0]
            if base in per_tile_kib and cnt_k:
                tiles_k = 2 * n_rays * (nerf.n_coarse if k.endswith("coarse") else nerf.n_coarse + nerf.n_fine) / 32 / cnt_k
                gbs = tiles_k * per_tile_kib[base] * 1024 / (ms_k / cnt_k * 1e-3) / 1e9
                roofline["kernels"][k] = {"avg_launch_ms": round(ms_k / cnt_k, 4), "achieved_GBps": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 3)}
        roofline["step_train_tflops"] = n_rays * samples_per_ray * TRAIN_FLOP / (elapsed / args.steps) / 1e12
        roofline["step_frac_of_mfma_peak"] = roofline["step_train_tflops"] / MFMA_PEAK_TFLOPS
        roofline["step_executed_tflops"] = n_rays * samples_per_ray * TRAIN_FLOP_EXEC / (elapsed / args.steps) / 1e12
        roofline["step_executed_frac_of_mfma_peak"] = roofline["step_executed_tflops"] / MFMA_PEAK_TFLOPS
        # the GOVERNING roofline of the path is MFMA on the algorithmic 3x-forward FLOP count of the whole step; the
        # object around it describes the dominant KERNEL, which in this dataflow (saved bf16 activations) is bound by HBM
        roofline = {"governing": {"bound": "mfma", "achieved": roofline["step_train_tflops"], "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": roofline["step_frac_of_mfma_peak"], "executed_frac": roofline["step_executed_frac_of_mfma_peak"],
                                  "scope": "whole train step, algorithmic FLOPs (3 x forward of the 12 Dense layers as written) / step time"},
                    **roofline}
        roofline["executed_note"] = ("all 12 Dense layers of both MLPs are trained (24 gradient tensors each); the three activation-free "
                                     "layers behind the trunk are evaluated as one composed 283x4 stage, an exact identity of the "
                                     "reference network (DESIGN.md 2.0), hence executed < algorithmic FLOPs")
        # whole-step HBM view: saved activations, relu masks and dZ written once and read back once
        step_bytes = n_rays * samples_per_ray / 32 * STEP_KIB_PER_TILE * 1024
        roofline["bytes_per_ray_sample"] = STEP_KIB_PER_TILE * 1024 / 32
        roofline["step_algorithmic_gbytes"] = step_bytes / 1e9
        roofline["step_frac_of_hbm_peak"] = step_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS

    # how much of this workload's backward is exactly-zero work (skip_dead_tiles): one extra step with the option
    # on, OUTSIDE every timed region.  Random initial weights: ~0 (sigma is positive almost everywhere), i.e. the line above is
    # not a measurement of skipped work.
    opts = {k: bool(nerf._ctx.get_option(k)) for k in ("skip_dead_tiles", "deterministic")}
    dead_frac = None
    if nerf._ctx.get_option("general_shape_path") == 0:
        nerf._ctx.set_option("skip_dead_tiles", 1)
        if nerf._ctx.get_option("skip_dead_tiles_active"):
            nerf._ctx.tile_stats(reset=True)
            nerf.train_step(data, with_metrics=False)
            live, total = nerf._ctx.tile_stats(reset=True)
            dead_frac = 1.0 - live / max(total, 1)
        nerf._ctx.set_option("skip_dead_tiles", int(opts["skip_dead_tiles"]))
    sync(world)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    # vs_baseline: null on the headline (nothing is published for cfg2); on a ref* line the reference's own stated seconds per step
    # for that command line, on its hardware -- for ref4 only at the GPU count the comment was written for
    vs_baseline, baseline_note = None, None
    if args.config in PUBLISHED_S_PER_STEP and (args.config != "ref4" or world == 2) and (args.config == "ref4" or world == 1):
        ref_s, where = PUBLISHED_S_PER_STEP[args.config]
        vs_baseline = value / (n_rays * world * samples_per_ray / ref_s)
        baseline_note = f"{ref_s:g} s/step = {n_rays * world * samples_per_ray / ref_s / 1e6:.2f} M rays*samples/s, {where}"
    if rank == 0:
        out = {
            "metric": "rays*samples/sec (train step), lego 128^2 coarse64+fine128" if args.config == "cfg2"
                      else f"rays*samples/sec (train step), {args.config}",
            "value": value, "unit": "rays*samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": vs_baseline,
            **({"baseline": baseline_note} if baseline_note else {}),
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.config}: {desc}", "rays_per_step_per_gpu": n_rays, "samples_per_ray": samples_per_ray,
                       "ray_chunks": chunks, "launch_rays": launch_rays,      # consecutive chunks share launches up to merge_chunk_rays (knerf.h)
                       "parallelism": f"dp{world}", "global_batch_images": batch * world},
            "roofline": roofline, "cpu_baseline": cpu, **dist_fields(world, backend, args.steps),
            "metrics_ms_per_step": metrics_ms, "metrics_clock": "hip events around the 3 metric launches, mean of 5 steps",
            "options": opts, "dead_tile_frac": dead_frac, **comm, **selftest, **provenance(),
        }
        emit(json.dumps(out))


if __name__ == "__main__":
    main()
