#!/usr/bin/env python3
"""Is there anything to gain from overlapping the kernels of consecutive chunks (tails and ramps of one launch filled by the next)?

Upper-bound probe that needs no change to the library: TWO contexts with the same weights train their own half of a cfg2 step
(one 128x128 image each, ray_chunks 4096)
  seq   both on one stream, one after the other (what knerf_train_batch does with the chunks of a step today)
  conc  each on its own stream, enqueued alternately, so that the GPU may co-schedule workgroups of the two whenever CUs are free
If conc is not faster than seq, a two-stream software pipeline inside knerf_train_batch cannot be either.

    python tools/overlap_probe.py [--pairs 12] [--out gpurun_out/overlap_probe.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keras_nerf_amd.runtime import KnerfContext  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=12)
    ap.add_argument("--rays", type=int, default=16384)
    ap.add_argument("--chunk", type=int, default=4096)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    rng = np.random.default_rng(0)
    R = args.rays
    dev = "cuda"
    o = torch.tensor(rng.normal(0, 0.1, (R, 3)).astype(np.float32) + np.array([0, 0, 4], np.float32), device=dev)
    d = rng.normal(0, 1, (R, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    d = torch.tensor(d, device=dev)
    t = torch.tensor(np.sort(rng.uniform(2, 6, (R, 64)).astype(np.float32), -1), device=dev)
    img = torch.tensor(rng.random((R, 3), dtype=np.float32), device=dev)
    ctxs = [KnerfContext(white_background=True) for _ in range(2)]
    n = ctxs[0].param_count
    w = [(rng.uniform(-1, 1, n) * 0.05).astype(np.float32) for _ in range(2)]
    for c in ctxs:
        c.set_weights(0, w[0]); c.set_weights(1, w[1])
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    loss = [torch.zeros(2, device=dev) for _ in range(2)]

    def run(conc, pairs):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(pairs):
            for k, c in enumerate(ctxs):
                with torch.cuda.stream(streams[k if conc else 0]):
                    c.train_batch(o, d, t, img, seed=1, ray_chunks=args.chunk, loss=loss[k])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / pairs * 1e3

    res = {"rays_per_ctx": R, "ray_chunks": args.chunk, "pairs": args.pairs, "runs": []}
    run(False, 3); run(True, 3)
    for rep in range(3):        # alternate so that clock drift does not pass for a difference
        s = run(False, args.pairs); c = run(True, args.pairs)
        res["runs"].append({"seq_ms_per_pair": round(s, 3), "conc_ms_per_pair": round(c, 3), "conc_over_seq": round(c / s, 4)})
        print(res["runs"][-1], flush=True)
    res["conc_over_seq_mean"] = float(np.mean([r["conc_over_seq"] for r in res["runs"]]))
    print(json.dumps(res))
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
