"""Helper of tests/test_gpu_variants.py::test_build_time_extra_shapes...: run with KNERF_LIB / KNERF_PROBE_LIB pointing at a library built
with `build.py --add-shape=...`; for every entry behind the built-in ones: the host tables are checked (tests/test_shape_tables.py) and
one JSON line is printed -- the entry, whether NeRF(...) with those arguments runs on the fused kernels, and its images / losses /
gradients against the oracle (kernel arithmetic)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    from keras_nerf_amd import debug as D
    from keras_nerf_amd.debug import debug_buffer
    from keras_nerf_amd.runtime import KnerfContext
    from oracle import nerf_oracle as O
    from tests.problem import make_problem
    from tests.test_gpu_train import flat, per_tensor_err
    from keras_nerf_amd import _lib
    from tests.test_shape_tables import _check_tables
    from keras_nerf_amd import build as B
    k = B.N_BUILTIN_SHAPES          # the first build-time entry
    while True:
        try:
            info = [int(v) for v in D.debug_table(5, k)]
        except _lib.KnerfError:
            break
        _check_tables(k)
        k += 1
        nl, sk, units, _, lx, ld = info
        cfg = O.NerfConfig(n_layers=nl, dense_units=units, skip_layer=sk, pos_emb_xyz=lx, pos_emb_dir=ld)
        P = make_problem(n_images=1, wh=16, weight_scale=1.5, bias_std=0.05, cfg=cfg)
        o, d, t, u, img = flat(P)
        ctx = KnerfContext(n_layers=nl, dense_units=units, skip_layer=sk, pos_emb_xyz=lx, pos_emb_dir=ld, white_background=True)
        ctx.set_weights(0, O.flatten_params(P["cp"])); ctx.set_weights(1, O.flatten_params(P["fp"]))
        loss = torch.zeros(2, device="cuda")
        ci = torch.empty((P["N"], 3), device="cuda"); fi = torch.empty_like(ci)
        ctx.train_chunk(o, d, t, img, u, inv_chunks=1.0, loss=loss, c_image=ci, f_image=fi)
        torch.cuda.synchronize()
        S = cfg.n_coarse + cfg.n_fine
        t_fine = debug_buffer(ctx, 5).view(torch.float32).cpu().numpy()[:P["N"] * S].reshape(P["N"], S).copy()
        g = ctx.grads_view().cpu().numpy(); n = g.size // 2
        # deterministic mode on the same chunk (per-workgroup slabs of ShapeInfo::partial_stride floats + ordered second pass): the
        # slab must hold the shape's LARGEST job table -- 11 or 12 input tiles in the concat job at pos_emb_xyz >= 11
        ctx.zero_grads(); ctx.set_option("deterministic", 1)
        ctx.train_chunk(o, d, t, img, u, inv_chunks=1.0)
        gd = ctx.grads_view().cpu().numpy()
        ctx.zero_grads(); ctx.train_chunk(o, d, t, img, u, inv_chunks=1.0)
        gd2 = ctx.grads_view().cpu().numpy()
        ctx.set_option("deterministic", 0)
        det_err = float(np.abs(gd - g).max() / np.abs(g).max())
        rc, lc, gc = O.chunk_loss_and_grads(P["cp"], o, d, t, img, cfg, True, emulate_bf16=O.FUSED)
        rf, lf, gf = O.chunk_loss_and_grads(P["fp"], o, d, t_fine, img, cfg, True, emulate_bf16=O.FUSED)
        print(json.dumps({"shape": [nl, sk, units, lx, ld], "info": info, "general_shape_path": ctx.get_option("general_shape_path"),
                          "coarse_worst": float(per_tensor_err(g[:n], O.flatten_params(gc), cfg)[0]),
                          "fine_worst": float(per_tensor_err(g[n:], O.flatten_params(gf), cfg)[0]),
                          "det_vs_atomic": det_err, "det_repeatable": bool(np.array_equal(gd, gd2)),
                          "det_coarse_worst": float(per_tensor_err(gd[:n], O.flatten_params(gc), cfg)[0]),
                          "det_fine_worst": float(per_tensor_err(gd[n:], O.flatten_params(gf), cfg)[0]),
                          "loss_err": max(abs(float(loss[0]) - float(lc)), abs(float(loss[1]) - float(lf))),
                          "img_err": float(max(np.abs(ci.cpu().numpy() - rc["image"]).max(), np.abs(fi.cpu().numpy() - rf["image"]).max()))}), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
