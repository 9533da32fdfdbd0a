#!/usr/bin/env python3
"""Coordinate search over the wgrad plan's per-job costs (KNERF_WGRAD_COSTS) on the GPU box: each trial is one tools/kbench.py run.
    python tools/tune_costs.py [--rounds 2] [--start 116,240,204,204,204,254,204,240,175]
    python tools/tune_costs.py --shape 8,128,4,10,4 --start 128,204,204,204,204,267,204,204,193 --jobs 0,5,8     (another fused shape)
"""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--start", default="116,240,204,204,204,254,204,240,175")
ap.add_argument("--iters", type=int, default=12)
ap.add_argument("--shape", default="8,256,4,10,4", help="tools/kbench.py --shape")
ap.add_argument("--jobs", default="7,1,5,8,0", help="the jobs whose cost is varied (the others keep their start value)")
args = ap.parse_args()


def trial(c):
    env = dict(os.environ, KNERF_WGRAD_COSTS=",".join(str(int(x)) for x in c))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kbench.py"), "--iters", str(args.iters), "--shape", args.shape], env=env, capture_output=True, text=True)
    r = json.loads(out.stdout.strip().splitlines()[-1])["kernels"]
    return r["train_wgrad_fine"]["ms"] + r["train_wgrad_coarse"]["ms"]


c = [float(x) for x in args.start.split(",")]
best = min(trial(c), trial(c))
print("start", c, round(best, 4), flush=True)
for rnd in range(args.rounds):
    step = 0.10 if rnd == 0 else 0.05
    for j in [int(v) for v in args.jobs.split(",")]:          # default: the jobs whose bodies differ from the plain 8x8 job (the plain ones share one cost)
        for f in (1 + step, 1 - step):
            t = list(c); t[j] = round(c[j] * f)
            v = min(trial(t), trial(t))
            print(" job", j, t[j], round(v, 4), flush=True)
            if v < best * 0.997:
                best, c = v, t
                break
    print("round", rnd, [int(x) for x in c], round(best, 4), flush=True)
print(json.dumps({"costs": [int(x) for x in c], "wgrad_ms": best}))
