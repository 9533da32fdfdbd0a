#!/usr/bin/env python3
"""Fused kernels vs general-shape kernels vs the oracle (kernel arithmetic and fp32) on deep / wide-encoding trunk shapes: per-tensor
worst gradient error of the coarse net on the 256-ray problem of the small parity tests.  Answers "is a 2-5 % difference from the
oracle on a 16-layer shape a kernel error or the problem's conditioning?": the oracle's OWN bf16-vs-fp32 gap is printed beside it
(0.56 of the largest gradient at 16 layers with pos_emb_xyz 16: two correct bf16 implementations cannot agree to 1 % there).
Needs a library that has the shapes on the fused kernels: KNERF_LIB / KNERF_PROBE_LIB = a `build.py --variant=... --add-shape=...` build
(round 5: 16,4,256,16,4 and 16,4,256,15,8; profiles/r05_corner_shapes.json)."""
import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from keras_nerf_amd.runtime import KnerfContext
from keras_nerf_amd.debug import debug_buffer
from oracle import nerf_oracle as O
from tests.problem import make_problem
from tests.test_gpu_train import flat, per_tensor_err
for shape in ((16, 4, 256, 16, 4), (16, 4, 256, 15, 8), (8, 4, 256, 10, 4)):
    nl, sk, units, lx, ld = shape
    cfg = O.NerfConfig(n_layers=nl, dense_units=units, skip_layer=sk, pos_emb_xyz=lx, pos_emb_dir=ld)
    P = make_problem(n_images=1, wh=16, weight_scale=1.5, bias_std=0.05, cfg=cfg)
    o, d, t, u, img = flat(P)
    res = {}
    for name, fg in (("fused", False), ("generic", True)):
        ctx = KnerfContext(n_layers=nl, dense_units=units, skip_layer=sk, pos_emb_xyz=lx, pos_emb_dir=ld, white_background=True, force_generic=fg)
        ctx.set_weights(0, O.flatten_params(P["cp"])); ctx.set_weights(1, O.flatten_params(P["fp"]))
        ctx.train_chunk(o, d, t, img, u, inv_chunks=1.0)
        torch.cuda.synchronize()
        S = cfg.n_coarse + cfg.n_fine
        tf = debug_buffer(ctx, 5).view(torch.float32).cpu().numpy()[:P["N"] * S].reshape(P["N"], S).copy()
        res[name] = (ctx.grads_view().cpu().numpy().copy(), tf, ctx.get_option("general_shape_path"))
        ctx.close()
    n = res["fused"][0].size // 2
    out = {"shape": shape, "paths": [res["fused"][2], res["generic"][2]], "t_fine_equal": bool(np.array_equal(res["fused"][1], res["generic"][1]))}
    for emu, tag in ((O.FUSED, "emu"), (False, "fp32")):
        _, _, gc = O.chunk_loss_and_grads(P["cp"], o, d, t, img, cfg, True, emulate_bf16=emu)
        gcf = O.flatten_params(gc)
        out[f"coarse_fused_vs_oracle_{tag}"] = float(per_tensor_err(res["fused"][0][:n], gcf, cfg)[0])
        out[f"coarse_generic_vs_oracle_{tag}"] = float(per_tensor_err(res["generic"][0][:n], gcf, cfg)[0])
        if tag == "emu":
            ge = gcf
        else:
            out["coarse_oracle_emu_vs_fp32"] = float(per_tensor_err(ge, gcf, cfg)[0])
    out["coarse_fused_vs_generic"] = float(per_tensor_err(res["fused"][0][:n], res["generic"][0][:n], cfg)[0])
    print(json.dumps(out), flush=True)
