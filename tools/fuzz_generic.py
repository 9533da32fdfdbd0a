#!/usr/bin/env python3
"""One-off fuzz of the GENERAL-SHAPE path with its round-5 options (GPU box: python tools/fuzz_generic.py [--cases 40]): random MLP shapes
the fused kernels do not cover (widths that are multiples of nothing, 1-10 layers, any skip_layer -- concat behind the last layer
included --, pos_emb 0..12 / 0..6), random sample counts (half of them multiples of 32 so that dead-tile skipping is active), both
backgrounds; per case: coarse image and every gradient tensor of both nets against the oracle in kernel arithmetic, deterministic mode
bit-repeatable and within 2e-5 of the atomic mode, skipping on / off bit-identical in deterministic mode.  Prints one line per case and
a summary; exit code 1 on any failure."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from keras_nerf_amd.debug import debug_buffer  # noqa: E402
from keras_nerf_amd.runtime import KnerfContext  # noqa: E402
from oracle import nerf_oracle as O  # noqa: E402
from tests.problem import make_problem  # noqa: E402
from tests.test_gpu_train import per_tensor_err  # noqa: E402


def one(case):
    cfg = O.NerfConfig(n_coarse=case["nc"], n_fine=case["nf"], pos_emb_xyz=case["lx"], pos_emb_dir=case["ld"], n_layers=case["nl"],
                       dense_units=case["units"], skip_layer=case["skip"])
    P = make_problem(n_images=1, wh=8, seed=case["seed"], weight_scale=1.5, bias_std=0.05, cfg=cfg)
    N = case["rays"]
    o, d, t, u, img = (torch.as_tensor(P[k].reshape(P["N"], -1)[:N].copy(), device="cuda") for k in ("o", "d", "t", "u", "img"))
    res = {}
    for det, skip in ((0, 1), (1, 1), (1, 0)):
        ctx = KnerfContext(n_coarse=case["nc"], n_fine=case["nf"], pos_emb_xyz=case["lx"], pos_emb_dir=case["ld"], n_layers=case["nl"],
                           dense_units=case["units"], skip_layer=case["skip"], white_background=case["white"], force_generic=True,
                           options=dict(deterministic=det, skip_dead_tiles=skip))
        ctx.set_weights(0, O.flatten_params(P["cp"])); ctx.set_weights(1, O.flatten_params(P["fp"]))
        if "tgt" not in res:      # a third of the rays: the coarse net's own pixel (those rays' coarse tiles are dead)
            ren = ctx.render_chunk(o, d, t, u if case["nf"] else None)
            tgt = img.clone(); tgt[::3] = ren["c_image"][::3]
            res["tgt"] = tgt
        loss = torch.zeros(2, device="cuda"); ci = torch.empty((N, 3), device="cuda")
        ctx.zero_grads()
        ctx.train_chunk(o, d, t, res["tgt"], u if case["nf"] else None, loss=loss, c_image=ci)
        torch.cuda.synchronize()
        g = ctx.grads_view().cpu().numpy().copy()
        rep = None
        if det:
            ctx.zero_grads(); ctx.train_chunk(o, d, t, res["tgt"], u if case["nf"] else None)
            torch.cuda.synchronize()
            rep = bool(np.array_equal(g, ctx.grads_view().cpu().numpy()))
        S = case["nc"] + case["nf"]
        tf = debug_buffer(ctx, 5).view(torch.float32).cpu().numpy()[:N * S].reshape(N, S).copy()
        res[det, skip] = dict(g=g, rep=rep, ci=ci.cpu().numpy(), loss=loss.cpu().numpy(), tf=tf, active=ctx.get_option("skip_dead_tiles_active"), stats=ctx.tile_stats())
        ctx.close()
    tgt = res["tgt"].cpu().numpy()
    on, dn, tn = o.cpu().numpy(), d.cpu().numpy(), t.cpu().numpy()
    rc, lc, gc = O.chunk_loss_and_grads(P["cp"], on, dn, tn, tgt, cfg, case["white"], emulate_bf16=O.FUSED)
    rf, lf, gf = O.chunk_loss_and_grads(P["fp"], on, dn, res[0, 1]["tf"], tgt, cfg, case["white"], emulate_bf16=O.FUSED)
    g = res[0, 1]["g"]; n = g.size // 2

    def err(a, ref):
        """worst per-tensor error of the tensor's max |g|; a net whose whole gradient is (next to) zero -- a dead coarse net, or one
        whose every ray was given its own pixel as target -- is compared absolutely (the ratio of two roundings of zero means nothing)"""
        scale = float(np.abs(ref).max())
        if scale < 1e-6:
            return float(np.abs(a - ref).max() / 1e-6) * 1e-2
        return per_tensor_err(a, ref, cfg)[0]
    out = {"coarse": err(g[:n], O.flatten_params(gc)), "fine": err(g[n:], O.flatten_params(gf)),
           "oracle_gmax": [float(np.abs(O.flatten_params(gc)).max()), float(np.abs(O.flatten_params(gf)).max())],
           "img": float(np.abs(res[0, 1]["ci"] - rc["image"]).max()), "loss": float(max(abs(res[0, 1]["loss"][0] - lc), abs(res[0, 1]["loss"][1] - lf))),
           "det_rep": res[1, 1]["rep"] and res[1, 0]["rep"], "skip_exact": bool(np.array_equal(res[1, 1]["g"], res[1, 0]["g"])),
           "det_vs_atomic": float(np.abs(res[1, 1]["g"] - g).max() / max(np.abs(g).max(), 1e-30)),
           "skip_active": res[0, 1]["active"], "stats": res[0, 1]["stats"]}
    gmax = float(np.abs(g).max())
    out["ok"] = bool(out["coarse"] < 5e-2 and out["fine"] < 5e-2 and out["img"] < 1.5e-2 and out["loss"] < 3e-3 and out["det_rep"] and out["skip_exact"]
                     and (out["det_vs_atomic"] < 5e-5 or gmax == 0.0) and np.isfinite(g).all())
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=2026)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    bad = 0
    for i in range(args.cases):
        nl = int(rng.integers(1, 11))
        case = dict(nl=nl, units=int(rng.choice([8, 24, 50, 70, 96, 130, 200, 256, 300, 520])), skip=int(rng.integers(1, nl + 2)),
                    lx=int(rng.integers(0, 13)), ld=int(rng.integers(0, 7)), white=bool(rng.integers(0, 2)), seed=int(rng.integers(0, 1 << 30)),
                    rays=int(rng.integers(1, 65)))
        if rng.integers(0, 2):
            case["nc"] = 32 * int(rng.integers(1, 4)); case["nf"] = 32 * int(rng.integers(0, 4))
        else:
            case["nc"] = int(rng.integers(2, 80)); case["nf"] = int(rng.integers(0, 100))
        try:
            r = one(case)
        except Exception as e:      # noqa: BLE001
            r = {"ok": False, "error": repr(e)[:300]}
        bad += not r["ok"]
        print(json.dumps({"case": case, **{k: (round(v, 5) if isinstance(v, float) else v) for k, v in r.items()}}), flush=True)
    print(f"fuzz_generic: {args.cases} cases, {bad} failures", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
