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


from tests.procedural_scene import make_scene as _make_scene, psnr   # noqa: E402  (the scene lives with the tests that share it)

SCALE = 1.0        # --scale: object size (1.0: the blobs span about +-0.8 of the +-1.44 half-width the cameras see)


COMPACT = False    # --scene compact: density cut off to exactly zero outside the objects (tests/procedural_scene.py)


def make_scene(ctx):
    return _make_scene(ctx, WH, NTRAIN + NVAL, SCALE, compact=COMPACT)


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
    ap.add_argument("--deterministic", action="store_true", help="hip leg: bit-reproducible gradient sums (knerf_set_option deterministic)")
    ap.add_argument("--skip-dead", action="store_true", help="hip leg: skip dead 32-sample tiles in the backward (exact); logs the dead fraction "
                                                            "and the ms per step between checkpoints")
    ap.add_argument("--save-weights", default=None, help="hip leg: write coarse/fine flat weights at every checkpoint to this .npz prefix")
    ap.add_argument("--scene", default="soft", choices=["soft", "compact"], help="soft: Gaussian blobs (density > 0 everywhere, the r02 experiments); "
                                                                                 "compact: the same shapes with the density cut off to exactly 0 outside them")
    ap.add_argument("--perturb", type=int, default=0,
                    help="fp32 leg only: visit each step's chunks in a permutation drawn from this seed (0 = reference order). Same "
                         "sums, another floating-point order: measures the fp32 arithmetic's own trajectory spread")
    args = ap.parse_args()
    global SCALE, COMPACT
    SCALE = args.scale
    COMPACT = args.scene == "compact"
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
        nerf.compile({"learning_rate": args.lr}, "mse", batch_size=BATCH, image_height=WH, image_width=WH, ray_chunks=CHUNK, white_background=True,
                     deterministic=args.deterministic, skip_dead_tiles=args.skip_dead)
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
            co = None if not args.perturb else [int(i) for i in np.random.default_rng([args.perturb, s]).permutation(BATCH * WH * WH // CHUNK)]
            lc, lf, _, _ = T.train_step(tc, tf_, oc, of_, im, oo, dd, tt_, uu, cfg, CHUNK, True, chunk_order=co)
            return lc, lf

    def save_state(s):
        if args.backend == "fp32" and args.state:
            torch.save({"tensors": [x.detach().clone() for x in tc + tf_ + oc.m + oc.v + of_.m + of_.v], "t": oc.t, "step": s, "log": log}, args.state)

    if start == 0:
        log.append(dict(step=0, val_psnr=evaluate(), wall_s=0.0)); print(log[-1], flush=True)
    t0 = time.time()
    s = start
    t_seg, s_seg = time.time(), start
    while s < args.steps:
        lc, lf = step(s)
        s += 1
        if s % args.eval_every == 0 or s == args.steps:
            extra = {}
            if args.backend == "hip":
                torch.cuda.synchronize()
                extra["ms_per_step"] = (time.time() - t_seg) / (s - s_seg) * 1e3
                if args.skip_dead:
                    live, total = nerf._ctx.tile_stats(reset=True)
                    extra["dead_tile_frac"] = 1.0 - live / max(total, 1)
                if args.save_weights:
                    np.savez(f"{args.save_weights}_step{s}.npz", coarse=nerf.coarse.get_flat_weights(), fine=nerf.fine.get_flat_weights())
            log.append(dict(step=s, coarse_loss=float(lc), fine_loss=float(lf), val_psnr=evaluate(), wall_s=time.time() - t0, **extra))
            t_seg, s_seg = time.time(), s
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
