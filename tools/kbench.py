#!/usr/bin/env python3
"""Per-kernel timings at a bench configuration (HIP events via knerf_profile_*).  Usage on the GPU box:
    python tools/kbench.py [--rays 4096] [--iters 6] [--lib path/to/variant.so]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--rays", type=int, default=4096)
ap.add_argument("--iters", type=int, default=6)
ap.add_argument("--lib", default=None)
ap.add_argument("--tag", default="")
ap.add_argument("--save-grads", default=None, help="write the gradient of one train chunk (fixed inputs and Philox seed) to this .npy")
ap.add_argument("--check-grads", default=None, help="compare that gradient with a saved one (A/B builds must compute the same thing)")
ap.add_argument("--skip-dead-tiles", type=int, default=0, help="kernel timings are quoted with dead-tile skipping OFF (every tile goes through dgrad and wgrad); 1 = the library default")
ap.add_argument("--shape", default="8,256,4,10,4", help="n_layers,dense_units,skip_layer,pos_emb_xyz,pos_emb_dir (non-default: general-shape path)")
args = ap.parse_args()
if args.lib:
    os.environ["KNERF_LIB"] = os.path.abspath(args.lib)

import numpy as np
import torch
from keras_nerf_amd.runtime import KnerfContext
from keras_nerf_amd.data.utils import get_focal_from_fov, pose_spherical
from keras_nerf_amd.model.nerf.mlp import NeRFMLP

NL, NU, SK, LX, LD = (int(v) for v in args.shape.split(","))
ctx = KnerfContext(white_background=True, n_layers=NL, dense_units=NU, skip_layer=SK, pos_emb_xyz=LX, pos_emb_dir=LD,
                   options=dict(skip_dead_tiles=args.skip_dead_tiles))
for net in (0, 1):
    m = NeRFMLP(NL, NU, SK, seed=net, xyz_dim=3 + 6 * LX, dir_dim=3 + 6 * LD); m.build(); ctx.set_weights(net, m.get_flat_weights())
wh = 128
o, d, t = ctx.generate_rays(pose_spherical(20.0, -30.0, 4.0)[None], get_focal_from_fov(0.6911112070083618, wh), wh, wh, 2.0, 6.0, 64, None, seed=1)
R = args.rays
o, d, t = o.reshape(-1, 3)[:R].contiguous(), d.reshape(-1, 3)[:R].contiguous(), t.reshape(-1, 64)[:R].contiguous()
tgt = torch.rand((R, 3), device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
loss = torch.zeros(2, device="cuda")
TRUNK = 63 * 256 + 4 * 256 * 256 + 319 * 256 + 2 * 256 * 256       # executed MACs per sample: trunk + the composed 283x4 head
FWD, DG, WG = 2 * (TRUNK + 283 * 4), 2 * (256 * 4 + 7 * 256 * 256), 2 * (TRUNK + 283 * 4)
if args.shape != "8,256,4,10,4":       # FLOP per sample of an arbitrary shape: 2 x MACs of every Dense layer
    from keras_nerf_amd.model.nerf.mlp import layer_shapes
    macs = sum(i * o for _, i, o in layer_shapes(NL, NU, SK, 3 + 6 * LX, 3 + 6 * LD))
    FWD, DG, WG = 2 * macs, 2 * macs, 2 * macs
for _ in range(2):
    ctx.train_chunk(o, d, t, tgt, None, seed=1, loss=loss); ctx.render_chunk(o, d, t, None, seed=1)
torch.cuda.synchronize()
res = {}
ctx.profile_enable(True); ctx.profile_read()
for _ in range(args.iters):
    ctx.render_chunk(o, d, t, None, seed=1)
pr = ctx.profile_read()
res["infer_fwd_coarse"] = pr["mlp_fwd_coarse"]; res["infer_fwd_fine"] = pr["mlp_fwd_fine"]
for _ in range(args.iters):
    ctx.train_chunk(o, d, t, tgt, None, seed=1, loss=loss)
pt = ctx.profile_read()
for k, v in pt.items():
    res["train_" + k] = v
if args.save_grads or args.check_grads:
    ctx.zero_grads(); loss.zero_()
    ctx.train_chunk(o, d, t, tgt, None, seed=1, loss=loss)
    torch.cuda.synchronize()
    g = ctx.grads_view().cpu().numpy().copy()
    if args.save_grads:
        np.save(args.save_grads, g)
    if args.check_grads:
        ref = np.load(args.check_grads)
        n = g.size // 2
        errs = [float(np.abs(g[s] - ref[s]).max() / np.abs(ref[s]).max()) for s in (slice(0, n), slice(n, 2 * n))]
        print(json.dumps({"tag": args.tag, "grad_check_rel_err": errs, "ok": max(errs) < 1e-4}))
        assert max(errs) < 1e-4, errs
ctx.apply_adam()
if hasattr(ctx.lib, "knerf_debug_wgrad_stamps"):
    import ctypes as C
    buf = (C.c_ulonglong * (1024 * 8))()
    ctx.lib.knerf_debug_wgrad_stamps(buf, 1024 * 8)
    a = np.array(buf[:]).reshape(1024, 8)
    a = a[a[:, 4] > 0]
    per = {}
    for j in sorted(set(a[:, 5])):
        r = a[a[:, 5] == j]
        tiles = r[:, 4].astype(float)
        per[int(j)] = {k: round(float((r[:, i] / tiles).mean()), 1) for i, k in enumerate(["wait", "barrier", "issue", "compute"])}
        per[int(j)]["tiles_per_wg"] = float(tiles.mean())
        per[int(j)]["entry_to_loop"] = float(r[:, 6].astype(float).mean()); per[int(j)]["loop_end_to_exit"] = float(r[:, 7].astype(float).mean())
        per[int(j)]["loop_total"] = float(r[:, 0:4].sum(1).astype(float).mean())
    print(json.dumps({"wgrad_cycles_per_tile_by_job": per}))
out = {}
for k, (ms, n) in res.items():
    if not n:
        continue
    avg = ms / n
    S = R * (64 if k.endswith("coarse") else 192)
    fl = FWD if "fwd" in k else DG if "bwd" in k else WG if "wgrad" in k else 0
    out[k] = {"ms": round(avg, 4), "TFLOPs": round(fl * S / (avg * 1e-3) / 1e12, 1) if fl else None}
    if "wgrad" in k and args.shape == "8,256,4,10,4":       # 242 KiB per tile is the default shape's wgrad read
        out[k]["TBs"] = round(S / 32 * 242 * 1024 / (avg * 1e-3) / 1e12, 2)
tot = sum(v["ms"] for k, v in out.items() if k.startswith("train_"))
print(json.dumps({"tag": args.tag, "rays": R, "train_chunk_ms": round(tot, 3), "Mrs_per_s": round(R * 256 / tot / 1e3, 1), "kernels": out}))
