#!/usr/bin/env python3
"""bench.py -- NeRF train-step throughput on MI355X, in BASELINE.json's metric (rays*samples/s).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one full NeRF.train_step (reference keras_nerf/model/nerf/nerf.py:332-473) on one synthetic batch that is
already resident in HBM: chunk loop (coarse fwd+bwd, inverse-CDF sampling, fine fwd+bwd, gradient accumulation), the
data-parallel all-reduce(SUM) of the accumulated gradients when N > 1, both Adam updates and the bf16 weight re-pack.
PSNR/SSIM bookkeeping is excluded.  Per-GPU work is fixed as N grows ("weak" scaling, the
reference's train.py semantics: global batch = batch_size x replicas).

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

FWD_FLOP = 2 * 593408                      # per ray*sample, SURVEY.md section 2.2 / 8d
DGRAD_FLOP = 2 * (128 * 3 + 256 * 128 + 256 * 257 + 7 * 256 * 256)
WGRAD_FLOP = 2 * 593408
TRAIN_FLOP = 3 * FWD_FLOP                  # the 3x-forward figure
MFMA_PEAK_TFLOPS = 2500.0                  # bf16 dense, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0

CONFIGS = {
    # name: (img_wh, batch per GPU, ray_chunks, description)
    "cfg2": (128, 2, 4096, "lego-shaped 128x128, batch 2 per GPU, ray_chunks 4096, coarse64+fine128"),
    "cfg3": (400, 1, 16000, "400x400, batch 1, ray_chunks 16000 (16384 does not divide 160000), coarse64+fine128"),
    "cfg4": (128, 1, 4096, "chair-shaped 128x128, 1 image per GPU, ray_chunks 4096, coarse64+fine128"),
    # forward only: one "step" = one frame of the 360-degree sweep of inference.py (pose -> rays -> render -> D2H)
    "cfg5": (256, 1, 4096, "inference.py-shaped 360-degree render, 256x256, ray_chunks 4096, coarse64+fine128, forward only"),
}


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


def cpu_baseline(n_rays=512, chunk=256, repeats=3):
    """op-for-op torch-CPU restatement of the reference train step (oracle/torch_ref.py) on a bounded sample"""
    from oracle import nerf_oracle as O
    from oracle import torch_ref as T
    cfg = O.NerfConfig()
    rng = np.random.default_rng(42)
    wh = 32
    c2w = O.pose_spherical(20.0, -30.0, 4.0)
    o, d, t = O.generate_rays(c2w, O.get_focal_from_fov(0.6911112070083618, wh), wh, wh, 2.0, 6.0, 64, rng.random((wh, wh, 64)))
    o, d, t = [torch.tensor(a.reshape(wh * wh, -1)[:n_rays]) for a in (o, d, t)]
    img = torch.tensor(rng.random((n_rays, 3), dtype=np.float32))
    u = torch.tensor(np.random.default_rng(7).random((n_rays, 128), dtype=np.float32))
    cp = [torch.tensor(p, requires_grad=True) for p in O.init_params(cfg, 0)]
    fp = [torch.tensor(p, requires_grad=True) for p in O.init_params(cfg, 1)]
    oc, of_ = T.TorchKerasAdam(cp), T.TorchKerasAdam(fp)
    times = []
    for i in range(repeats + 1):
        t0 = time.perf_counter()
        T.train_step(cp, fp, oc, of_, img, o, d, t, u, cfg, chunk, False)
        times.append(time.perf_counter() - t0)
    med = statistics.median(times[1:])
    return {"value": n_rays * 256 / med, "unit": "rays*samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n_rays} rays x 256 samples, ray_chunks {chunk}, coarse64+fine128 train step, median of {repeats} "
                      f"after 1 warm-up; torch-CPU fp32 restatement of the reference's TF-CPU path ({os.cpu_count()} host cpus)",
            "s_per_step": med}


def sync(world):
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
        torch.cuda.synchronize()


def max_over_ranks(elapsed, world):
    if world > 1:
        tt = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt[0])
    return elapsed


def bench_render(args, world, rank, wh, chunks, desc):
    """cfg5: frames/s of the 360-degree render loop of the reference's inference.py:62-114 (theta sweep at phi=-30,
    radius 4; rays generated on the device; fine image + depth copied to the host per frame as the reference does)."""
    from keras_nerf_amd.data.rays import RaysGenerator
    from keras_nerf_amd.data.utils import get_focal_from_fov, pose_spherical
    from keras_nerf_amd.model.nerf.nerf import NeRF
    nerf = NeRF(seed=0)
    nerf.compile(optimizer="adam", loss="mse", batch_size=1, image_height=wh, image_width=wh, ray_chunks=chunks,
                 white_background=True, is_training=False)
    rg = RaysGenerator(get_focal_from_fov(0.6911112070083618, wh), wh, wh, 2.0, 6.0, nerf.n_coarse, seed=rank)
    n_frames = args.steps
    poses = [pose_spherical(360.0 * i / max(n_frames, 1), -30.0, 4.0) for i in range(n_frames + args.warmup)]

    def frame(i):
        o, d, t = rg(poses[i])
        _, fine = nerf.predict_and_render_images((o[None], d[None], t[None]))
        return fine["image"].cpu().numpy(), fine["depth"].cpu().numpy()
    for i in range(args.warmup):
        frame(i)
    sync(world)
    t0 = time.perf_counter()
    for i in range(n_frames):
        img, dep = frame(args.warmup + i)
    sync(world)
    elapsed = max_over_ranks(time.perf_counter() - t0, world)
    fps = world * n_frames / elapsed
    rs = fps * wh * wh * 256
    roofline = None
    if rank == 0:
        nerf._ctx.profile_enable(True); nerf._ctx.profile_read()
        frame(0)
        prof = nerf._ctx.profile_read(); nerf._ctx.profile_enable(False)
        ms, cnt = prof["mlp_fwd_fine"]
        avg = ms / max(cnt, 1)
        flop = FWD_FLOP * chunks * (nerf.n_coarse + nerf.n_fine)
        roofline = {"bound": "mfma", "kernel": "mlp_fwd_fine", "achieved": flop / (avg * 1e-3) / 1e12, "peak": MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": flop / (avg * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, "traffic": None, "avg_launch_ms": avg,
                    "launches": cnt, "kernel_ms_per_frame": {k: round(v[0], 4) for k, v in prof.items() if v[1]},
                    "frame_tflops": rs * FWD_FLOP / 1e12, "frame_frac_of_mfma_peak": rs * FWD_FLOP / 1e12 / MFMA_PEAK_TFLOPS}
        print(json.dumps({"metric": "frames/sec (360-degree render 256^2, coarse64+fine128, forward only)", "value": fps,
                          "unit": "frames/s", "n_gpus": world, "steps": n_frames, "warmup": args.warmup,
                          "ms_per_step": elapsed / n_frames * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "bf16", "data": "synthetic", "rays_samples_per_s": rs,
                          "config": {"workload": f"cfg5: {desc}", "frames": n_frames, "parallelism": f"dp{world}"},
                          "roofline": roofline, "cpu_baseline": None}), flush=True)


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC summary (profiles/*pmc*.json, produced by tools/pmc.sh +
    tools/pmc_report.py: separate --pmc passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide coalesced
    reads on gfx950, WRITE_SIZE as is, both in KiB).  None when no summary is committed for this kernel."""
    import glob
    name = {"wgrad_fine": ("wgrad_kernel", max), "wgrad_coarse": ("wgrad_kernel", min), "mlp_fwd_fine": ("mlp_fwd_kernel<true", max),
            "mlp_fwd_coarse": ("mlp_fwd_kernel<true", min), "mlp_bwd_fine": ("mlp_bwd_kernel", max),
            "mlp_bwd_coarse": ("mlp_bwd_kernel", min)}.get(kernel)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")))
    if not name or not files:
        return None
    rep = json.load(open(files[-1]))
    rows = [v for k, v in rep.items() if k.startswith(name[0]) and "hbm_bytes_per_launch" in v]
    if not rows:
        return None
    if name[1] is max:
        return max(r["hbm_bytes_per_launch"] for r in rows)
    return min(r["hbm_bytes_per_launch_min"] for r in rows)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # KNERF_DIST_BACKEND=gloo rehearses the N>1 control flow on a box with fewer GPUs than ranks (ranks then share devices)
    backend = os.environ.get("KNERF_DIST_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if world > 1 and backend == "nccl" and local_rank >= n_dev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {n_dev} GPUs visible (RCCL needs one GPU per rank)")
    device_index = local_rank % max(n_dev, 1)
    torch.cuda.set_device(device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            torch.distributed.init_process_group(backend)

    from keras_nerf_amd.model.nerf.nerf import NeRF
    wh, batch, chunks, desc = CONFIGS[args.config]
    if args.config == "cfg5":
        bench_render(args, world, rank, wh, chunks, desc)
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    nerf = NeRF(seed=0)
    nerf.compile(optimizer="adam", loss="mse", batch_size=batch, image_height=wh, image_width=wh, ray_chunks=chunks,
                 white_background=True)
    data = make_batch(nerf, wh, batch, rank)
    n_rays = batch * wh * wh
    samples_per_ray = nerf.n_coarse + (nerf.n_coarse + nerf.n_fine)          # 64 + 192 = 256 MLP evaluations per ray

    for _ in range(args.warmup):
        nerf.train_step(data, with_metrics=False)
    sync(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        nerf.train_step(data, with_metrics=False)
    sync(world)
    elapsed = max_over_ranks(time.perf_counter() - t0, world)
    value = world * n_rays * samples_per_ray * args.steps / elapsed

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
        s_fine, s_coarse = chunks * (nerf.n_coarse + nerf.n_fine), chunks * nerf.n_coarse
        per_launch_samples = s_coarse if dom.endswith("coarse") else s_fine
        flop = {"mlp_fwd": FWD_FLOP, "mlp_bwd": DGRAD_FLOP, "wgrad": WGRAD_FLOP}.get(dom.rsplit("_", 1)[0], 0) * per_launch_samples
        # wgrad streams the saved activations and dZ once: (158 + 156) KiB per 32-sample tile + 4 KiB re-read of enc
        if dom.startswith("wgrad"):
            byts = per_launch_samples / 32 * 318 * 1024
            roofline = {"bound": "hbm", "kernel": dom, "achieved": byts / (avg_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": byts / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None}
        else:
            roofline = {"bound": "mfma", "kernel": dom, "achieved": flop / (avg_ms * 1e-3) / 1e12, "peak": MFMA_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": flop / (avg_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, "traffic": None}
        roofline["traffic"] = pmc_traffic(dom)
        roofline["avg_launch_ms"] = avg_ms
        roofline["launches"] = cnt
        roofline["kernel_ms_per_step"] = {k: round(v[0] / 2, 4) for k, v in prof.items()}
        roofline["step_train_tflops"] = n_rays * samples_per_ray * TRAIN_FLOP / (elapsed / args.steps) / 1e12
        roofline["step_frac_of_mfma_peak"] = roofline["step_train_tflops"] / MFMA_PEAK_TFLOPS
        # whole-step HBM view: per 32-sample tile the step writes 158 KiB of activations, 8 KiB of relu
        # masks and 156 KiB of dZ and reads 162 + 8 + 156 KiB of them back (wgrad, dgrad), plus raw/draw: 650.5 KiB
        step_bytes = n_rays * samples_per_ray / 32 * 650.5 * 1024
        roofline["step_algorithmic_gbytes"] = step_bytes / 1e9
        roofline["step_frac_of_hbm_peak"] = step_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        out = {
            "metric": "rays*samples/sec (train step), lego 128^2 coarse64+fine128" if args.config == "cfg2"
                      else f"rays*samples/sec (train step), {args.config}",
            "value": value, "unit": "rays*samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.config}: {desc}", "rays_per_step_per_gpu": n_rays, "samples_per_ray": samples_per_ray,
                       "parallelism": f"dp{world}", "global_batch_images": batch * world},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
