#!/usr/bin/env python3
"""Convergence parity at the metric's size (BASELINE.json: lego 128^2, coarse64 + fine128).

The dataset is absent (no network), so the scene is procedural but nerf_synthetic-SHAPED: an analytic density / colour field
inside the unit-ish sphere, cameras on the radius-4 sphere (`pose_spherical`, the synthetic datasets' convention), fov
0.6911112070083618, white background, 100 train / 4 validation views of 128 x 128 rendered at 512 samples per ray in fp64.
Both legs start from the same weights and see the same batches (2 images per step = cfg2), the same jittered t-values and
the same inverse-CDF `u`:

  --backend hip    the product path: NeRF.train_step (HIP kernels, bf16 MFMA operands)
  --backend fp32   the reference's arithmetic: oracle/torch_ref.py (op-for-op restatement, fp32, autograd, Keras-form Adam)
                   run with torch tensors on the GPU -- EXPERIMENT SCRIPT ONLY: torch_ref is test infrastructure, never the
                   product path and never the benched path

Validation PSNR of the fine image every --eval-every steps -> JSON.  --state FILE checkpoints the fp32 leg so that it can be
continued in a later call (a gpurun call is limited to 20 minutes).

    python tools/convergence128.py --backend hip  --steps 2000 --out gpurun_out/conv128_hip.json
    python tools/convergence128.py --backend fp32 --steps 2000 --out gpurun_out/conv128_fp32.json --state gpurun_out/conv128_state.pt --budget-s 1000
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import nerf_oracle as O   # noqa: E402  (experiment script: the oracle supplies poses, initial weights and the fp32 leg)

WH, NTRAIN, NVAL, BATCH, CHUNK = 128, 100, 4, 2, 4096
FOV = 0.6911112070083618


SCALE = 1.0        # --scale: object size (1.0: the blobs span about +-0.8 of the +-1.44 half-width the cameras see)


def field(p):
    """analytic scene (torch, fp64): three soft blobs and a ring, position-dependent colour; p [...,3] -> sigma [...], rgb [...,3]"""
    p = p / SCALE
    c = torch.tensor([[0.45, 0.0, 0.15], [-0.55, 0.3, -0.25], [0.0, -0.5, 0.35]], dtype=p.dtype, device=p.device)
    d = [((p - ci) ** 2).sum(-1) for ci in c]
    ring = (torch.sqrt(p[..., 0] ** 2 + p[..., 1] ** 2) - 0.8) ** 2 + (p[..., 2] + 0.1) ** 2
    sigma = 14.0 * torch.exp(-d[0] / 0.12) + 10.0 * torch.exp(-d[1] / 0.2) + 12.0 * torch.exp(-d[2] / 0.08) + 9.0 * torch.exp(-ring / 0.015)
    rgb = torch.stack([0.5 + 0.5 * torch.sin(4 * p[..., 0] + 1.0), 0.5 + 0.5 * torch.cos(3 * p[..., 1] + 0.5),
                       0.25 + 0.7 * (d[0] < d[1]).to(p.dtype) * (0.5 + 0.5 * torch.sin(6 * p[..., 2]))], -1)
    return sigma, rgb.clamp(0, 1)


def make_scene(ctx):
    """views: o, d [V,H,W,3], t [V,H,W,64] (fp32, on the GPU, jitter fixed per view), img [V,H,W,3]"""
    from keras_nerf_amd.data.utils import get_focal_from_fov, pose_spherical
    V = NTRAIN + NVAL
    poses = np.stack([pose_spherical(360.0 * i / V * 7 % 360.0, -30.0 + 20.0 * np.sin(0.7 * i), 4.0) for i in range(V)])
    o, d, t = ctx.generate_rays(poses, get_focal_from_fov(FOV, WH), WH, WH, 2.0, 6.0, 64, None, seed=2026)
    imgs = []
    tt = torch.linspace(2.0, 6.0, 512, device="cuda", dtype=torch.float64)
    for v in range(V):
        p = o[v].double()[..., None, :] + d[v].double()[..., None, :] * tt[:, None]          # [H,W,512,3]
        sg, col = field(p)
        delta = torch.cat([tt[1:] - tt[:-1], tt.new_full((1,), 1e-10)])
        alpha = 1.0 - torch.exp(-sg * delta)
        T = torch.cumprod(torch.cat([torch.ones_like(alpha[..., :1]), 1.0 - alpha[..., :-1] + 1e-10], -1), -1)
        w = alpha * T
        img = (w[..., None] * col).sum(-2) + (1.0 - w.sum(-1))[..., None]                     # white background (utils.py:52-53)
        imgs.append(img.clamp(0, 1).float())
    return o, d, t, torch.stack(imgs)


def psnr(a, b):
    return float(-10.0 * torch.log10(((a - b) ** 2).mean()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", choices=["hip", "fp32"], required=True)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--eval-every", type=int, default=250)
    ap.add_argument("--out", required=True)
    ap.add_argument("--state", default=None)
    ap.add_argument("--budget-s", type=float, default=1e9)
    ap.add_argument("--lr", type=float, default=1e-3, help="Adam learning rate of both legs (Keras default 1e-3, nerf.py:163-165)")
    ap.add_argument("--seeds", default="0,1", help="glorot seeds of the coarse and fine MLP")
    ap.add_argument("--scale", type=float, default=1.0)
    args = ap.parse_args()
    global SCALE
    SCALE = args.scale
    t_start = time.time()
    from keras_nerf_amd.model.nerf.nerf import NeRF
    from keras_nerf_amd.runtime import KnerfContext
    ctx0 = KnerfContext(white_background=True)
    o, d, t, img = make_scene(ctx0)
    ctx0.close()
    cfg = O.NerfConfig()
    sc, sf = (int(x) for x in args.seeds.split(","))
    cp, fp = O.init_params(cfg, sc), O.init_params(cfg, sf)
    order = np.random.default_rng(5).integers(0, NTRAIN, (args.steps, BATCH))
    gen_u = lambda s: torch.rand((BATCH, WH, WH, 128), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1000 + s))
    u_val = torch.rand((NVAL, WH, WH, 128), device="cuda", generator=torch.Generator(device="cuda").manual_seed(77))
    log, start = [], 0
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)

    def batch(s):
        idx = torch.as_tensor(order[s], device="cuda")
        return img[idx], o[idx], d[idx], t[idx], gen_u(s)

    if args.backend == "hip":
        nerf = NeRF()
        nerf.compile({"learning_rate": args.lr}, "mse", batch_size=BATCH, image_height=WH, image_width=WH, ray_chunks=CHUNK, white_background=True)
        nerf.coarse.set_flat_weights(O.flatten_params(cp)); nerf.fine.set_flat_weights(O.flatten_params(fp))
        ev = NeRF()
        ev.compile("adam", "mse", batch_size=1, image_height=WH, image_width=WH, ray_chunks=CHUNK, white_background=True, is_training=False)

        def evaluate():
            ev.coarse.set_flat_weights(nerf.coarse.get_flat_weights()); ev.fine.set_flat_weights(nerf.fine.get_flat_weights())
            ps = []
            for k in range(NVAL):
                v = NTRAIN + k
                _, fine = ev.predict_and_render_images((o[v:v + 1], d[v:v + 1], t[v:v + 1]), u=u_val[k:k + 1])
                ps.append(psnr(fine["image"][0], img[v]))
            return float(np.mean(ps))

        def step(s):
            im, oo, dd, tt_, uu = batch(s)
            lg = nerf.train_step((im, (oo, dd, tt_)), u=uu, with_metrics=False)
            return lg["coarse_loss"], lg["fine_loss"]
    else:
        from oracle import torch_ref as T
        tc = [torch.tensor(p, device="cuda", requires_grad=True) for p in cp]
        tf_ = [torch.tensor(p, device="cuda", requires_grad=True) for p in fp]
        oc, of_ = T.TorchKerasAdam(tc, lr=args.lr), T.TorchKerasAdam(tf_, lr=args.lr)
        if args.state and os.path.exists(args.state):
            st = torch.load(args.state, map_location="cuda")
            with torch.no_grad():
                for dst, src in zip(tc + tf_ + oc.m + oc.v + of_.m + of_.v, st["tensors"]):
                    dst.copy_(src)
            oc.t = of_.t = st["t"]; start = st["step"]; log = st["log"]
            print(f"resumed at step {start}", flush=True)

        def evaluate():
            ps = []
            with torch.no_grad():
                for k in range(NVAL):
                    v = NTRAIN + k
                    fo, fd, ft, fu = o[v].reshape(-1, 3), d[v].reshape(-1, 3), t[v].reshape(-1, 64), u_val[k].reshape(-1, 128)
                    out = []
                    for c in range(WH * WH // CHUNK):
                        sl = slice(c * CHUNK, (c + 1) * CHUNK)
                        _, _, w = T.chunk_forward(tc, fo[sl], fd[sl], ft[sl], cfg, True)
                        tf2 = torch.sort(torch.cat([ft[sl], T.fine_sampling(0.5 * (ft[sl][:, 1:] + ft[sl][:, :-1]), w, fu[sl])], -1), -1).values
                        out.append(T.chunk_forward(tf_, fo[sl], fd[sl], tf2, cfg, True)[0])
                    ps.append(psnr(torch.cat(out).reshape(WH, WH, 3), img[v]))
            return float(np.mean(ps))

        def step(s):
            im, oo, dd, tt_, uu = batch(s)
            lc, lf, _, _ = T.train_step(tc, tf_, oc, of_, im, oo, dd, tt_, uu, cfg, CHUNK, True)
            return lc, lf

    def save_state(s):
        if args.backend == "fp32" and args.state:
            torch.save({"tensors": [x.detach().clone() for x in tc + tf_ + oc.m + oc.v + of_.m + of_.v], "t": oc.t, "step": s, "log": log}, args.state)

    if start == 0:
        log.append(dict(step=0, val_psnr=evaluate(), wall_s=0.0)); print(log[-1], flush=True)
    t0 = time.time()
    s = start
    while s < args.steps:
        lc, lf = step(s)
        s += 1
        if s % args.eval_every == 0 or s == args.steps:
            log.append(dict(step=s, coarse_loss=float(lc), fine_loss=float(lf), val_psnr=evaluate(), wall_s=time.time() - t0))
            print(log[-1], flush=True)
            json.dump(log, open(args.out, "w"))
            save_state(s)
        elif s % 25 == 0:
            print(f"step {s} {float(lf):.5f} {time.time() - t0:.0f}s", flush=True)
            if time.time() - t_start > args.budget_s:
                save_state(s); json.dump(log, open(args.out, "w"))
                print(f"budget reached at step {s}: state saved", flush=True)
                return
    json.dump(log, open(args.out, "w"))


if __name__ == "__main__":
    main()
