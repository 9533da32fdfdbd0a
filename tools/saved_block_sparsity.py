#!/usr/bin/env python3
"""Exact-zero sparsity of the saved tensors, measured before building any block skipping.

The 1-bit ReLU masks are already saved; h_l = relu(z_l) and dz_l = mask_l * (...) are EXACTLY zero wherever mask_l is closed.  A saved
block is 16 features x 32 consecutive samples of one ray (1 KiB; csrc/layout.h "saved tensors"): the unit the weight-gradient kernel
copies by LDS-DMA and feeds to two MFMA k-steps.  If whole blocks are zero, wgrad could skip their DMA and MFMAs and the chain kernels
their stores, with bit-identical gradients.  This tool reads the act / dz workspaces of one 4,096-ray train chunk
(include/knerf_debug.h knerf_debug_buffer 0 / 2) and reports, per saved tensor (h1..h7, dz0..dz6) of the fine pass and of a
coarse-only pass:

  elem      fraction of elements that are exactly zero (+0 or -0)
  block     fraction of 16-feature x 32-sample blocks that are all zero              <- what block skipping could use
  row       fraction of (feature, tile) rows -- one feature over a tile's 32 samples -- that are all zero (not addressable in this layout;
            reported to show at which granularity the zeros sit)
  pair      fraction of 32-feature x 32-sample block PAIRS (one wgrad input tile / one MFMA operand tile) that are all zero

each over ALL tiles and over the LIVE tiles only (tiles whose dz_head is not all zero: dead tiles are skipped by dgrad and wgrad
already, option skip_dead_tiles).  States: random initialisation (glorot, seeds 0 / 1) and weights trained for --train-steps steps on
the procedural soft and compact scenes (tools/convergence128.py checkpoints).  Go / no-go bar: >= 25 % of blocks all-zero in BOTH the
random and a trained state.  Prints one JSON line; --out writes it too."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

KS, ENCQ, DIRQ, NL = 16, 4, 2, 8          # default shape (csrc/layout.h): 16 blocks per 256-wide tensor, 4 enc + 2 dir blocks
ACT_BLOCKS, DZ_BLOCKS, SKEW = 118, 130, 256


def act_block0(l):                          # first block of h_l in a tile's act run: h1 h2 h3 h4 enc h5 h6 h7 dir
    return KS * (l - 1) + (ENCQ if l >= 5 else 0)


def analyse(buf, n_tiles, blocks, runs):
    """buf: uint8 view of a saved region; runs: {name: first block}.  Returns {name: stats}, the live-tile mask needs dz_head."""
    import torch
    stride = blocks * 1024 + SKEW
    v = buf[: n_tiles * stride].view(n_tiles, stride)[:, : blocks * 1024].contiguous().view(torch.int16).view(n_tiles, blocks, 512)
    out = {}
    for name, b0 in runs.items():
        x = (v[:, b0:b0 + KS] & 0x7FFF) == 0                     # [tiles, 16 blocks, 512]: True = exactly zero
        # lane (h, s) of a block stores 8 bf16 at byte (2 * (s ^ 4 * (b & 1)) + h) * 16: index = s' * 16 + h * 8 + j
        rows = x.view(n_tiles, KS, 32, 16).all(2)                  # all 32 samples of one feature
        blk = x.all(2)
        pair = blk.view(n_tiles, KS // 2, 2).all(2)
        out[name] = dict(x=x, rows=rows, blk=blk, pair=pair)
    return out


def summarise(st, live):
    import torch
    res = {}
    for name, s in st.items():
        r = {}
        for key, sel in (("all", slice(None)), ("live", live)):
            x, rows, blk, pair = s["x"][sel], s["rows"][sel], s["blk"][sel], s["pair"][sel]
            if x.shape[0] == 0:
                continue
            r[key] = {"elem": round(float(x.float().mean()), 4), "block": round(float(blk.float().mean()), 4),
                      "row": round(float(rows.float().mean()), 4), "pair": round(float(pair.float().mean()), 4)}
        res[name] = r
    return res


def one_pass(weights, net, n_fine, o, d, t, tgt, u):
    """one train chunk through a context with dead-tile skipping OFF (every tile's dz is written); the workspaces then hold the LAST pass"""
    import torch
    from keras_nerf_amd.debug import debug_buffer
    from keras_nerf_amd.runtime import KnerfContext
    ctx = KnerfContext(n_fine=n_fine, white_background=True, options=dict(skip_dead_tiles=0))
    ctx.set_weights(0, weights["coarse"]); ctx.set_weights(1, weights["fine"] if n_fine else weights["coarse"])
    loss = torch.zeros(2, device="cuda")
    ctx.train_chunk(o, d, t, tgt, u if n_fine else None, seed=3, loss=loss)
    torch.cuda.synchronize()
    S = 64 + n_fine
    n_tiles = o.shape[0] * S // 32
    act = analyse(debug_buffer(ctx, 0), n_tiles, ACT_BLOCKS, {f"h{l}": act_block0(l) for l in range(1, NL)})
    dz = analyse(debug_buffer(ctx, 2), n_tiles, DZ_BLOCKS, {f"dz{l}": KS * l for l in range(NL - 1)})
    stride = DZ_BLOCKS * 1024 + SKEW
    head = debug_buffer(ctx, 2)[: n_tiles * stride].view(n_tiles, stride)[:, KS * NL * 1024:(KS * NL + 1) * 1024]
    live = (head.contiguous().view(torch.int16) & 0x7FFF).ne(0).any(1)
    res = {"tiles": int(n_tiles), "live_tile_frac": round(float(live.float().mean()), 4), "loss": [round(float(x), 5) for x in loss.cpu()]}
    res.update(summarise({**act, **dz}, live))
    hs = [res[f"h{l}"] for l in range(1, NL)]; ds = [res[f"dz{l}"] for l in range(NL - 1)]
    for key in ("all", "live"):
        if all(key in r for r in hs + ds):
            res[f"mean_{key}"] = {k: round(sum(r[key][k] for r in hs + ds) / len(hs + ds), 4) for k in ("elem", "block", "row", "pair")}
    ctx.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train-steps", type=int, default=600)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--state-dir", default=os.path.join(ROOT, "gpurun_out", "sparsity"))
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    os.makedirs(args.state_dir, exist_ok=True)
    import numpy as np
    import torch
    from oracle import nerf_oracle as O       # experiment script: initial weights as every other tool draws them
    from keras_nerf_amd.runtime import KnerfContext
    from tests.procedural_scene import make_scene

    states = {"random_init": {"coarse": O.flatten_params(O.init_params(O.NerfConfig(), 0)), "fine": O.flatten_params(O.init_params(O.NerfConfig(), 1))}}
    for scene in ("soft", "compact"):
        pre = os.path.join(args.state_dir, f"w_{scene}")
        f = f"{pre}_step{args.train_steps}.npz"
        if not os.path.exists(f):
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "convergence128.py"), "--backend", "hip", "--scene", scene, "--skip-dead", "--lr", "5e-4",
                                   "--scale", "1.6", "--steps", str(args.train_steps), "--eval-every", str(args.train_steps), "--out", pre + "_log.json",
                                   "--save-weights", pre])
        z = np.load(f)
        states[f"trained_{scene}_step{args.train_steps}"] = {"coarse": z["coarse"], "fine": z["fine"], "_scene": scene,
                                                            "_log": json.load(open(pre + "_log.json"))[-1]}
    out = {"rays": args.rays, "block": "16 features x 32 samples (1 KiB bf16)", "bar": ">= 0.25 of blocks all-zero in the random AND a trained state"}
    ctx0 = KnerfContext(white_background=True)
    scenes = {sc: make_scene(ctx0, 128, 8, 1.6, compact=(sc == "compact")) for sc in ("soft", "compact")}
    ctx0.close()
    for name, w in states.items():
        sc = w.get("_scene", "soft")
        o, d, t, img = scenes[sc]
        # a chunk of 4,096 consecutive rays from the MIDDLE of a training view (rows 48..79 of 128: object and background both in view)
        R, r0 = args.rays, 48 * 128
        oo, dd, tt, tg = (x[1].reshape(128 * 128, -1)[r0:r0 + R].contiguous() for x in (o, d, t, img))
        if name == "random_init":             # the bench's regime: random targets
            tg = torch.rand((R, 3), device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
        u = torch.rand((R, 128), device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
        ent = {"fine_pass": one_pass(w, 1, 128, oo, dd, tt, tg, u), "coarse_pass": one_pass(w, 0, 0, oo, dd, tt, tg, None)}
        if "_log" in w:
            ent["checkpoint"] = {k: w["_log"].get(k) for k in ("step", "val_psnr", "dead_tile_frac")}
        out[name] = ent
        print(name, {p: ent[p].get("mean_live") for p in ("fine_pass", "coarse_pass")}, flush=True)
    blocks = [out[k]["fine_pass"]["mean_live"]["block"] for k in out if isinstance(out[k], dict) and "fine_pass" in out[k]]
    out["verdict"] = {"min_block_zero_frac_over_states_live_tiles_fine": min(blocks), "go": bool(min(blocks) >= 0.25)}
    line = json.dumps(out)
    print(line)
    if args.out:
        with open(args.out, "w") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
