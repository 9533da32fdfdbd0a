#!/usr/bin/env python3
"""Per-tensor gradient error of a fused trunk shape (and of the general-shape kernels) against the oracle: python tools/shape_diag.py NL SK"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import nerf_oracle as O
from tests.problem import make_problem
from tests.test_gpu_train import flat
from keras_nerf_amd.debug import debug_buffer
from keras_nerf_amd.runtime import KnerfContext

nl, sk = int(sys.argv[1]), int(sys.argv[2])
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.5
cfg = O.NerfConfig(n_layers=nl, dense_units=256, skip_layer=sk)
P = make_problem(n_images=1, wh=16, weight_scale=scale, bias_std=0.05, cfg=cfg)
o, d, t, u, img = flat(P)
res = {}
for force in (False, True):
    ctx = KnerfContext(n_layers=nl, dense_units=256, skip_layer=sk, white_background=True, force_generic=force)
    ctx.set_weights(0, O.flatten_params(P["cp"])); ctx.set_weights(1, O.flatten_params(P["fp"]))
    ctx.train_chunk(o, d, t, img, u, inv_chunks=1.0)
    torch.cuda.synchronize()
    tf = debug_buffer(ctx, 5).view(torch.float32).cpu().numpy()[:P["N"] * 192].reshape(P["N"], 192).copy()
    res[force] = (ctx.grads_view().cpu().numpy().copy(), tf)
    ctx.close()
n = res[False][0].size // 2
_, _, gc = O.chunk_loss_and_grads(P["cp"], o, d, t, img, cfg, True, emulate_bf16=O.FUSED)
for force in (False, True):
    g, tf = res[force]
    _, _, gf = O.chunk_loss_and_grads(P["fp"], o, d, tf, img, cfg, True, emulate_bf16=O.FUSED)
    for net, gg, ref in (("coarse", g[:n], O.flatten_params(gc)), ("fine", g[n:], O.flatten_params(gf))):
        off = 0; rows = []
        for name, fi, fo in O.layer_shapes(cfg):
            for kind, m in (("k", fi * fo), ("b", fo)):
                a, b = gg[off:off + m], ref[off:off + m]
                rows.append(f"{name}/{kind} {np.abs(a - b).max() / max(np.abs(b).max(), 1e-12):.1e}")
                off += m
        print("generic" if force else "fused  ", net, " ".join(rows), flush=True)
