#!/usr/bin/env python3
"""Upper bound of "a training forward that does not SAVE what the backward will skip" (DESIGN.md 5.4).

Trains the compact-support procedural scene (tools/convergence128.py --scene compact: density exactly 0 outside the objects, i.e.
hard-surface content like nerf_synthetic) with dead-tile skipping on until a third of the backward's tiles are dead, then times the
SAME step (same weights, same batches) twice in this process:

  default library             forward writes act + masks for every tile (126 KiB per 32-sample tile)
  --lib-nostores <variant>    `build.py --variant=nofwdstores -DKNERF_ABLATE_FWD_STORES`: the forward writes NOTHING -- what the forward
                              would cost if 100 % of its tiles were predicted dead and never mispredicted

The bound for a speculative forward at dead fraction f is  f x (step_default - step_nostores): it saves only the dead tiles' stores and
must still compute every tile (the forward decides what is dead).  The backward of the ablated leg reads stale activations: its
RESULTS are garbage, its tile lists (which come from the compositing kernel's own arithmetic on correct forward outputs) and hence
its timing are not.  Prints one JSON line."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def leg(args):
    import numpy as np
    import torch
    from keras_nerf_amd.model.nerf.nerf import NeRF
    from keras_nerf_amd.runtime import KnerfContext
    from tests.procedural_scene import make_scene
    WH, NTRAIN, BATCH, CHUNK = 128, 100, 2, 4096
    ctx0 = KnerfContext(white_background=True)
    o, d, t, img = make_scene(ctx0, WH, NTRAIN + 4, 1.6, compact=True)     # --scale 1.6: the setting of every round-2/3 convergence experiment
    ctx0.close()
    z = np.load(args.weights)
    nerf = NeRF()
    nerf.compile({"learning_rate": 5e-4}, "mse", batch_size=BATCH, image_height=WH, image_width=WH, ray_chunks=CHUNK, white_background=True, skip_dead_tiles=True)
    order = np.random.default_rng(11).integers(0, NTRAIN, (args.steps + 5, BATCH))
    out = {}
    for rep in range(args.reps):
        nerf.coarse.set_flat_weights(z["coarse"]); nerf.fine.set_flat_weights(z["fine"])      # every repetition times the same weights
        for k in range(5 + args.steps):
            if k == 5:
                torch.cuda.synchronize(); nerf._ctx.tile_stats(reset=True); nerf._ctx.profile_enable(True); nerf._ctx.profile_read(); t0 = time.perf_counter()
            idx = torch.as_tensor(order[k], device="cuda")
            u = torch.rand((BATCH, WH, WH, 128), device="cuda", generator=torch.Generator(device="cuda").manual_seed(k))
            nerf.train_step((img[idx], (o[idx], d[idx], t[idx])), u=u, with_metrics=False)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        prof = nerf._ctx.profile_read(); nerf._ctx.profile_enable(False)
        live, total = nerf._ctx.tile_stats(reset=True)
        out[f"rep{rep}"] = {"ms_per_step": ms, "dead_tile_frac": 1.0 - live / max(total, 1),
                            "kernel_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in prof.items()}}
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--weights", default="gpurun_out/fwd_bound_w_step600.npz")
    ap.add_argument("--train-steps", type=int, default=600)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--lib-nostores", default=os.path.join(ROOT, "keras_nerf_amd", "libknerf_hip_nofwdstores.so"))
    ap.add_argument("--leg", default=None)
    args = ap.parse_args()
    if args.leg:
        return leg(args)
    if not os.path.exists(args.weights):      # the checkpoint: 600 steps on the compact scene (dead tiles ~33 % from step 500 on)
        pre = args.weights[:-len(f"_step{args.train_steps}.npz")]
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "convergence128.py"), "--backend", "hip", "--scene", "compact", "--skip-dead", "--lr", "5e-4", "--scale", "1.6",
                               "--steps", str(args.train_steps), "--eval-every", str(args.train_steps), "--out", pre + "_log.json", "--save-weights", pre])
    res = {}
    for name, lib in (("default_a", None), ("nostores_a", args.lib_nostores), ("default_b", None), ("nostores_b", args.lib_nostores)):
        env = dict(os.environ)
        env.pop("KNERF_LIB", None)
        if lib:
            env["KNERF_LIB"] = lib
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--leg", name, "--weights", args.weights, "--steps", str(args.steps), "--reps", str(args.reps)],
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[name] = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    d = [res[k][f"rep{args.reps - 1}"]["ms_per_step"] for k in ("default_a", "default_b")]
    n = [res[k][f"rep{args.reps - 1}"]["ms_per_step"] for k in ("nostores_a", "nostores_b")]
    f = res["default_a"][f"rep{args.reps - 1}"]["dead_tile_frac"]
    delta = sum(d) / 2 - sum(n) / 2
    res["summary"] = {"step_ms_default": d, "step_ms_forward_without_stores": n, "dead_tile_frac": f, "all_stores_removed_ms": delta,
                      "bound_ms_at_this_dead_fraction": f * delta, "bound_frac_of_step": f * delta / (sum(d) / 2)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
