#!/usr/bin/env python3
"""Convergence parity experiment: train the same procedural scene from the same initial weights,
the same batches and the same `u` with (a) the torch-CPU fp32 port of the reference path (oracle/torch_ref.py) and
(b) the HIP path, and compare validation PSNR.  Not part of the test suite (the CPU side takes ~1 h on 8 cores).

    python tools/convergence.py --backend cpu --steps 300 --out gpurun_out/conv_cpu.json     (anywhere)
    python tools/convergence.py --backend gpu --steps 300 --out gpurun_out/conv_gpu.json     (GPU box)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import nerf_oracle as O   # noqa: E402  (experiment script: the oracle is the CPU leg and the scene renderer)

WH, NTRAIN, NVAL = 16, 24, 4


def field(p):
    """analytic scene: two soft blobs with position-dependent colour; p [...,3] -> sigma [...], rgb [...,3]"""
    c1, c2 = np.array([0.5, 0.0, 0.2]), np.array([-0.6, 0.3, -0.3])
    d1, d2 = np.sum((p - c1) ** 2, -1), np.sum((p - c2) ** 2, -1)
    sigma = 12.0 * np.exp(-d1 / 0.18) + 9.0 * np.exp(-d2 / 0.3)
    rgb = np.stack([0.5 + 0.5 * np.sin(3 * p[..., 0]), 0.5 + 0.5 * np.cos(2 * p[..., 1] + 1), 0.3 + 0.6 * (d1 < d2)], -1)
    return sigma, np.clip(rgb, 0, 1)


def make_scene():
    rng = np.random.default_rng(123)
    focal = O.get_focal_from_fov(0.6911112070083618, WH)
    views = []
    for i in range(NTRAIN + NVAL):
        c2w = O.pose_spherical(360.0 * i / (NTRAIN + NVAL), -30.0 + 10 * np.sin(i), 4.0)
        o, d, t = O.generate_rays(c2w, focal, WH, WH, 2.0, 6.0, 64, rng.random((WH, WH, 64)))
        tt = np.linspace(2.0, 6.0, 512)[None, None, :] * np.ones((WH, WH, 1))
        p = o[..., None, :].astype(np.float64) + d[..., None, :].astype(np.float64) * tt[..., None]
        sg, col = field(p)
        img, _, w = O.render_image_depth_chunk(col.reshape(-1, 512, 3), sg.reshape(-1, 512, 1), tt.reshape(-1, 512), True)
        views.append(dict(o=o, d=d, t=t, img=img.reshape(WH, WH, 3).astype(np.float32)))
    return views[:NTRAIN], views[NTRAIN:]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", choices=["cpu", "gpu"], required=True)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--eval-every", type=int, default=50)
    ap.add_argument("--out", required=True)
    ap.add_argument("--threads", type=int, default=0)
    args = ap.parse_args()
    train, val = make_scene()
    cfg = O.NerfConfig()
    cp, fp = O.init_params(cfg, 0), O.init_params(cfg, 1)
    urng = np.random.default_rng(99)
    order = np.random.default_rng(5).integers(0, NTRAIN, args.steps)
    us = [urng.random((WH * WH, 128), dtype=np.float32) for _ in range(args.steps)]
    uval = np.random.default_rng(77).random((NVAL, WH * WH, 128), dtype=np.float32)
    log = []
    t0 = time.time()
    if args.backend == "cpu":
        import torch
        from oracle import torch_ref as T
        if args.threads:
            torch.set_num_threads(args.threads)
        tc = [torch.tensor(p, requires_grad=True) for p in cp]; tf_ = [torch.tensor(p, requires_grad=True) for p in fp]
        oc, of_ = T.TorchKerasAdam(tc), T.TorchKerasAdam(tf_)

        def evaluate():
            ps = []
            with torch.no_grad():
                for k, v in enumerate(val):
                    o, d, t = [torch.tensor(v[x].reshape(WH * WH, -1)) for x in ("o", "d", "t")]
                    _, _, w = T.chunk_forward(tc, o, d, t, cfg, True)
                    tfine = torch.sort(torch.cat([t, T.fine_sampling(0.5 * (t[:, 1:] + t[:, :-1]), w, torch.tensor(uval[k]))], -1), -1).values
                    img, _, _ = T.chunk_forward(tf_, o, d, tfine, cfg, True)
                    ps.append(float(O.psnr(img.numpy().reshape(1, WH, WH, 3), v["img"][None])[0]))
            return float(np.mean(ps))
        for s in range(args.steps):
            v = train[order[s]]
            o, d, t = [torch.tensor(v[x].reshape(WH * WH, -1)) for x in ("o", "d", "t")]
            lc, lf, _, _ = T.train_step(tc, tf_, oc, of_, torch.tensor(v["img"].reshape(-1, 3)), o, d, t, torch.tensor(us[s]), cfg, WH * WH, True)
            if (s + 1) % args.eval_every == 0 or s == 0:
                log.append(dict(step=s + 1, coarse_loss=lc, fine_loss=lf, val_psnr=evaluate(), wall_s=time.time() - t0))
                print(log[-1], flush=True)
                json.dump(log, open(args.out, "w"))
    else:
        import torch
        from keras_nerf_amd.model.nerf.nerf import NeRF
        nerf = NeRF()
        nerf.compile("adam", "mse", batch_size=1, image_height=WH, image_width=WH, ray_chunks=WH * WH, white_background=True)
        nerf.coarse.set_flat_weights(O.flatten_params(cp)); nerf.fine.set_flat_weights(O.flatten_params(fp))

        def evaluate():
            ps = []
            for k, v in enumerate(val):
                _, fine = nerf.predict_and_render_images((v["o"][None], v["d"][None], v["t"][None]), u=uval[k])
                ps.append(float(O.psnr(fine["image"].cpu().numpy(), v["img"][None])[0]))
            return float(np.mean(ps))
        for s in range(args.steps):
            v = train[order[s]]
            logs = nerf.train_step((v["img"][None], (v["o"][None], v["d"][None], v["t"][None])), u=us[s], with_metrics=False)
            if (s + 1) % args.eval_every == 0 or s == 0:
                log.append(dict(step=s + 1, coarse_loss=float(logs["coarse_loss"]), fine_loss=float(logs["fine_loss"]), val_psnr=evaluate(),
                                wall_s=time.time() - t0))
                print(log[-1], flush=True)
                json.dump(log, open(args.out, "w"))


if __name__ == "__main__":
    main()
