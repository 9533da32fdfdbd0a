#!/usr/bin/env python3
"""One coarse + fine train chunk at the BENCH's chunk size (4,096 rays: 8,192 + 24,576 sample tiles, the launch sizes bench.py
times) against the oracle, per gradient tensor -- the one-off, minutes-long big brother of tests/test_gpu_wgrad_regime.py (1,024
rays).  The NumPy oracle (kernel arithmetic, oracle/nerf_oracle.py FUSED) runs in sub-batches of --sub rays to bound its memory:
the chunk's loss is the mean over its rays, so its gradient is the ray-weighted mean of the sub-batches' gradients.

    python tools/oracle_check_chunk.py [--rays 4096] [--sub 512]  ->  one JSON line (profiles/r04_oracle_check_4096_rays.json)"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "--lib" in sys.argv:          # before keras_nerf_amd._lib reads it
    os.environ["KNERF_LIB"] = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
import numpy as np  # noqa: E402
import torch  # noqa: E402

from keras_nerf_amd.debug import debug_buffer  # noqa: E402
from keras_nerf_amd.runtime import KnerfContext  # noqa: E402
from oracle import nerf_oracle as O  # noqa: E402  (a checking tool: the oracle is the checker, never the product)
from tests.problem import make_problem  # noqa: E402
from tests.test_gpu_train import per_tensor_err  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--sub", type=int, default=512)
    ap.add_argument("--threads", type=int, default=min(6, os.cpu_count() or 1), help="host threads evaluating oracle sub-batches side by side (~0.7 GB each at --sub 512)")
    ap.add_argument("--batch", default=None, help="IMAGES,WH,RAY_CHUNKS: a whole train_batch instead of one chunk, e.g. 2,128,4096 = BASELINE cfg2's step "
                                                  "(8 chunks, grouped coarse weight-gradient launches); the oracle then needs ~12 min of host time")
    ap.add_argument("--shape", default=None, help="NL,SK,U[,LX,LD]: another trunk shape (with --lib: a library built with --add-shape for it; "
                                                  "round 6: 9,4,256 = a trunk that ends in a concat, eleven input tiles in the head's weight-gradient job)")
    ap.add_argument("--lib", default=None, help="library to load instead of the product (KNERF_LIB)")
    args = ap.parse_args()
    n_images, ray_chunks = 1, None
    if args.batch:
        n_images, wh, ray_chunks = (int(v) for v in args.batch.split(","))
        args.rays = n_images * wh * wh
    else:
        wh = int(round(args.rays ** 0.5))
    assert n_images * wh * wh == args.rays and args.rays % args.sub == 0
    shape_kw = {}
    if args.shape:
        v = [int(x) for x in args.shape.split(",")]
        shape_kw = dict(n_layers=v[0], skip_layer=v[1], dense_units=v[2], **(dict(pos_emb_xyz=v[3], pos_emb_dir=v[4]) if len(v) == 5 else {}))
    P = make_problem(n_images=n_images, wh=wh, weight_scale=1.5, bias_std=0.05, cfg=O.NerfConfig(**shape_kw) if shape_kw else None)
    cfg, N = P["cfg"], P["N"]
    S = cfg.n_coarse + cfg.n_fine
    o, d, t, u, img = P["o"].reshape(N, 3), P["d"].reshape(N, 3), P["t"].reshape(N, -1), P["u"].reshape(N, -1), P["img"].reshape(N, 3)
    out = {"rays": N, "tiles": {"coarse": N * cfg.n_coarse // 32, "fine": N * S // 32}, "shape": args.shape or "8,4,256"}
    got = {}
    for skip in (1, 0):
        ctx = KnerfContext(white_background=True, options=dict(skip_dead_tiles=skip), **shape_kw)
        out["general_shape_path"] = ctx.get_option("general_shape_path")
        ctx.set_weights(0, O.flatten_params(P["cp"])); ctx.set_weights(1, O.flatten_params(P["fp"]))
        loss = torch.zeros(2, device="cuda")
        ci = torch.empty((N, 3), device="cuda"); fi = torch.empty_like(ci)
        if ray_chunks:           # the whole chunk loop of a train step (knerf_train_batch: gradients accumulated with weight 1 / C)
            ctx.train_batch(o, d, t, img, u, ray_chunks=ray_chunks, loss=loss, c_image=ci, f_image=fi)
            torch.cuda.synchronize()
            g_now = ctx.grads_view().cpu().numpy().copy()
            # the merged t-values of every chunk: the sampler is bit-exact and the weights have not moved, so a render of the same
            # rays with the same u reproduces them (the last chunk's are still in the training workspace: compared below)
            tf_all = ctx.render_batch(o, d, t, u, ray_chunks=ray_chunks, out=dict(c_image=torch.empty_like(ci), f_image=torch.empty_like(fi),
                                      t_fine=torch.empty((N, S), device="cuda")))["t_fine"].cpu().numpy()
            last = debug_buffer(ctx, 5).view(torch.float32).cpu().numpy()[:ray_chunks * S].reshape(ray_chunks, S)
            out["t_fine_of_render_equals_training"] = bool(np.array_equal(tf_all[-ray_chunks:], last))
            out["wgrad_group"] = ctx.get_option("wgrad_group")
        else:
            ctx.train_chunk(o, d, t, img, u, inv_chunks=1.0, loss=loss, c_image=ci, f_image=fi)
            torch.cuda.synchronize()
            g_now = ctx.grads_view().cpu().numpy().copy()
            tf_all = debug_buffer(ctx, 5).view(torch.float32).cpu().numpy()[:N * S].reshape(N, S).copy()
        got[skip] = dict(g=g_now, loss=loss.cpu().numpy().copy(), ci=ci.cpu().numpy(), fi=fi.cpu().numpy(), t_fine=tf_all)
        n = ctx.param_count
        ctx.close()
    t0 = time.time()
    ref = {}
    # the 2 x N / sub oracle evaluations are independent: a few host threads (NumPy's matmuls and element-wise loops release the GIL;
    # round 5 -- with the oracle's bf16 rounding no longer 70 % of its time, this is what keeps the test under the suite's budget)
    from concurrent.futures import ThreadPoolExecutor
    nets = ((P["cp"], t), (P["fp"], got[1]["t_fine"]))

    def one(task):
        net, s0 = task
        params, tt = nets[net]
        sl = slice(s0, s0 + args.sub)
        r, l, gr = O.chunk_loss_and_grads(params, o[sl], d[sl], tt[sl], img[sl], cfg, True, emulate_bf16=O.FUSED)
        err = float(np.abs((got[1]["ci"] if net == 0 else got[1]["fi"])[sl] - r["image"]).max())
        print(f"oracle net {net} rays {s0 + args.sub}/{N} {time.time() - t0:.0f}s", file=sys.stderr, flush=True)
        return net, O.flatten_params(gr).astype(np.float64) * (args.sub / N), float(l) * args.sub / N, err
    tasks = [(net, s0) for net in (1, 0) for s0 in range(0, N, args.sub)]          # the three-times larger fine-net tasks first
    with ThreadPoolExecutor(max_workers=args.threads) as pool:
        done = list(pool.map(one, tasks))
    for net in (0, 1):
        mine = [x for x in done if x[0] == net]                                   # pool.map keeps the task order: a fixed summation order
        gsum = np.sum([x[1] for x in mine], axis=0)
        ref[net] = (gsum.astype(np.float32), sum(x[2] for x in mine), max(x[3] for x in mine))
    out["oracle_seconds"] = round(time.time() - t0, 1)
    for skip in (1, 0):
        g = got[skip]["g"]
        ec, ef = per_tensor_err(g[:n], ref[0][0], cfg), per_tensor_err(g[n:], ref[1][0], cfg)
        out[f"skip_dead_tiles_{skip}"] = {"coarse_worst": ec[0], "coarse_where": ec[1], "fine_worst": ef[0], "fine_where": ef[1],
                                          "loss_err": [abs(float(got[skip]["loss"][k]) - ref[k][1]) for k in (0, 1)]}
    out["image_max_abs_err"] = [ref[0][2], ref[1][2]]
    out["list_vs_contiguous_rel"] = float(np.abs(got[0]["g"] - got[1]["g"]).max() / np.abs(got[1]["g"]).max())
    out["ok"] = all(out[f"skip_dead_tiles_{s}"][k] < 1.5e-2 for s in (0, 1) for k in ("coarse_worst", "fine_worst"))
    print(json.dumps(out), flush=True)
    sys.exit(0 if out["ok"] else 1)


if __name__ == "__main__":
    main()
