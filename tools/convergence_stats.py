#!/usr/bin/env python3
"""Mean +- standard error of (fp32 leg - HIP leg) validation PSNR from tools/convergence128.py logs.

    python tools/convergence_stats.py --hip profiles/archive/r02_convergence128_lr5e-4_{hip*,final*}.json profiles/archive/r03_conv128/r03_conv128_lr5e-4_hip*.json \\
                                      --fp32 profiles/archive/r02_convergence128_lr5e-4_fp32.json profiles/archive/r03_conv128/r03_conv128_lr5e-4_fp32_p*.json \\
                                      --out profiles/r03_convergence_stats.json

Per checkpoint s: D(s) = mean_j F_j(s) - mean_i H_i(s), SE(s) = sqrt(var F / n_F + var H / n_H) (sample variances over the runs of each
leg; the runs of a leg differ only in the order of floating-point sums: fp32 atomics for HIP, the chunk order for the fp32 leg).
Over a range of checkpoints (600 ... 2000) the per-run mean over the range is the statistic.  "decided" = |D| + 2 SE <= 0.1 dB
(the claim holds at ~95 %), or |D| - 2 SE > 0.1 dB (it fails); otherwise the data do not decide."""
import argparse
import json

import numpy as np


def load(paths):
    runs = []
    for p in paths:
        log = json.load(open(p))
        runs.append({int(e["step"]): float(e["val_psnr"]) for e in log})
    return runs


def stat(f, h):
    f, h = np.asarray(f, float), np.asarray(h, float)
    d = f.mean() - h.mean()
    vf = f.var(ddof=1) / len(f) if len(f) > 1 else float("nan")
    vh = h.var(ddof=1) / len(h) if len(h) > 1 else float("nan")
    se = float(np.sqrt(vf + vh))
    verdict = "within 0.1 dB" if abs(d) + 2 * se <= 0.1 else ("outside 0.1 dB" if abs(d) - 2 * se > 0.1 else "undecided at 2 SE")
    return dict(fp32_mean=float(f.mean()), hip_mean=float(h.mean()), delta=float(d), se=se, n_fp32=len(f), n_hip=len(h),
                sd_fp32=float(f.std(ddof=1)) if len(f) > 1 else None, sd_hip=float(h.std(ddof=1)) if len(h) > 1 else None, verdict=verdict)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hip", nargs="+", required=True)
    ap.add_argument("--fp32", nargs="+", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--range", default="600,2000")
    args = ap.parse_args()
    H, F = load(args.hip), load(args.fp32)
    steps = sorted(set.intersection(*[set(r) for r in H + F]) - {0})
    out = {"hip_files": args.hip, "fp32_files": args.fp32, "per_checkpoint": {}, "ranges": {}}
    for s in steps:
        out["per_checkpoint"][s] = stat([r[s] for r in F], [r[s] for r in H])
    lo, hi = (int(x) for x in args.range.split(","))
    for name, sel in ((f"{lo}-{hi}", [s for s in steps if lo <= s <= hi]), ("100-500", [s for s in steps if s <= 500])):
        if sel:
            out["ranges"][name] = dict(checkpoints=sel, **stat([np.mean([r[s] for s in sel]) for r in F], [np.mean([r[s] for s in sel]) for r in H]))
    json.dump(out, open(args.out, "w"), indent=1)
    print("| step | fp32 mean (n) | HIP mean (n) | fp32 - HIP | SE | sd fp32 | sd HIP | |")
    print("|---|---|---|---|---|---|---|---|")
    rows = [(str(s), out["per_checkpoint"][s]) for s in steps if s <= 500] + [(k, v) for k, v in out["ranges"].items()]
    for k, v in rows:
        f = lambda x: "-" if x is None else f"{x:.3f}"
        print(f"| {k} | {v['fp32_mean']:.3f} ({v['n_fp32']}) | {v['hip_mean']:.3f} ({v['n_hip']}) | {v['delta']:+.3f} | {v['se']:.3f} | {f(v['sd_fp32'])} | {f(v['sd_hip'])} | {v['verdict']} |")


if __name__ == "__main__":
    main()
