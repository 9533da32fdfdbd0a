#!/usr/bin/env python3
"""Fused dgrad+wgrad launch vs. the separate kernels: gradient agreement and time per train chunk for several
producer counts.  GPU box:  python tools/fused_check.py --producers 64,96,128 [--rays 4096]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--rays", type=int, default=4096)
ap.add_argument("--iters", type=int, default=6)
ap.add_argument("--producers", default="96")
args = ap.parse_args()

import torch
from keras_nerf_amd.runtime import KnerfContext
from keras_nerf_amd.data.utils import get_focal_from_fov, pose_spherical
from keras_nerf_amd.model.nerf.mlp import NeRFMLP

ctx = KnerfContext(white_background=True)
for net in (0, 1):
    m = NeRFMLP(seed=net); m.build(); ctx.set_weights(net, m.get_flat_weights())
wh = 128
o, d, t = ctx.generate_rays(pose_spherical(20.0, -30.0, 4.0)[None], get_focal_from_fov(0.6911112070083618, wh), wh, wh, 2.0, 6.0, 64, None, seed=1)
R = args.rays
o, d, t = o.reshape(-1, 3)[:R].contiguous(), d.reshape(-1, 3)[:R].contiguous(), t.reshape(-1, 64)[:R].contiguous()
tgt = torch.rand((R, 3), device="cuda")
loss = torch.zeros(2, device="cuda")


def run(P):
    ctx.set_fused_backward(P)
    for _ in range(2):
        ctx.train_chunk(o, d, t, tgt, None, seed=1, loss=loss)
    ctx.zero_grads(); loss.zero_()
    ctx.train_chunk(o, d, t, tgt, None, seed=1, loss=loss)
    torch.cuda.synchronize()
    g = ctx.grads_view().clone()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        ctx.train_chunk(o, d, t, tgt, None, seed=1, loss=loss)
    torch.cuda.synchronize()
    return g, (time.perf_counter() - t0) / args.iters * 1e3


g0, ms0 = run(0)
print(json.dumps({"producers": 0, "ms_per_chunk": round(ms0, 3)}), flush=True)
for P in [int(x) for x in args.producers.split(",")]:
    g, ms = run(P)
    err = (g - g0).abs().max().item() / g0.abs().max().item()
    rel = ((g - g0).norm() / g0.norm()).item()
    print(json.dumps({"producers": P, "ms_per_chunk": round(ms, 3), "max_err_over_max": err, "rel_l2": rel}), flush=True)
ctx.zero_grads()
ctx.train_chunk(o, d, t, tgt, None, seed=1, loss=loss)
ctx.apply_adam()       # raises if a consumer poll timed out at any point
print("abort flag clean")
