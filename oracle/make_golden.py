"""Regenerates tests/golden/*.npz from the NumPy oracle (fp32 unless noted).  TEST INFRASTRUCTURE.

The reference cannot run here (TensorFlow absent) and ships no numeric fixtures for the hot path,
so these vectors are OUTPUTS OF THE ORACLE ITSELF: they pin the oracle against regressions and give the GPU tests fixed
full-size expectations; they do not pin it to TensorFlow ("parity unpinned", see oracle/nerf_oracle.py).

    python -m oracle.make_golden
"""
import os

import numpy as np

from oracle import nerf_oracle as O

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def small():
    """R=16, S=8/24, units=32: every stage incl. gradients and two Adam steps, in fp32 and fp64"""
    cfg = O.NerfConfig(n_coarse=8, n_fine=16, pos_emb_xyz=4, pos_emb_dir=2, n_layers=8, dense_units=32, skip_layer=4)
    rng = np.random.default_rng(2024)
    c2w = O.pose_spherical(40.0, -30.0, 4.0)
    o, d, t = O.generate_rays(c2w, 8.0, 4, 4, 2.0, 6.0, cfg.n_coarse, rng.random((4, 4, cfg.n_coarse)))
    u = rng.random((4, 4, cfg.n_fine)).astype(np.float32)
    img = rng.random((4, 4, 3)).astype(np.float32)
    out = dict(o=o, d=d, t=t, u=u, img=img, c2w=c2w)
    for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        cp = [(p * 3).astype(dt) for p in O.init_params(cfg, 1)]
        fp = [(p * 3).astype(dt) for p in O.init_params(cfg, 2)]
        a = [x.astype(dt) for x in (o[None], d[None], t[None], u[None], img[None])]
        for oob in ("zero", "clamp"):
            c, f = O.predict_and_render_images(cp, fp, a[0], a[1], a[2], a[3], cfg, 8, True, oob)
            out[f"{tag}_{oob}_c_image"] = c["image"]; out[f"{tag}_{oob}_c_weights"] = c["weights"]
            out[f"{tag}_{oob}_f_image"] = f["image"]; out[f"{tag}_{oob}_f_depth"] = f["depth"]; out[f"{tag}_{oob}_t_fine"] = f["t"]
        oc, of_ = O.KerasAdam(cp), O.KerasAdam(fp)
        for step in range(2):
            m, ci, fi, (gc, gf) = O.train_step(cp, fp, oc, of_, a[4], a[0], a[1], a[2], a[3], cfg, 8, True)
            out[f"{tag}_step{step}_losses"] = np.array([m["coarse_loss"], m["fine_loss"]])
            if step == 0:
                out[f"{tag}_grad_c"] = O.flatten_params(gc); out[f"{tag}_grad_f"] = O.flatten_params(gf)
        out[f"{tag}_w_c_after"] = O.flatten_params(cp); out[f"{tag}_w_f_after"] = O.flatten_params(fp)
    np.savez_compressed(os.path.join(OUT, "small_r16.npz"), **out)


def fullsize():
    """the kernels' architecture (8x256, L=10/4, 64+128 samples), 64 rays: inputs + fp32 and bf16-emulated outputs;
    weights are regenerated from seeds (init_params(cfg, 0/1) * 1.5 + seeded biases) rather than stored"""
    from tests.problem import make_problem
    P = make_problem(n_images=1, wh=8, seed=42, weight_scale=1.5, bias_std=0.05)
    N, cfg = P["N"], P["cfg"]
    o, d, t, u, img = P["o"].reshape(N, 3), P["d"].reshape(N, 3), P["t"].reshape(N, -1), P["u"].reshape(N, -1), P["img"].reshape(N, 3)
    out = dict(o=o, d=d, t=t, u=u, img=img, w_c_checksum=np.array([O.flatten_params(P["cp"]).astype(np.float64).sum()]),
               w_f_checksum=np.array([O.flatten_params(P["fp"]).astype(np.float64).sum()]))
    for emu, tag in ((False, "f32"), (O.FUSED, "bf16")):      # "bf16" = the fused kernels' arithmetic (nerf_oracle.FUSED)
        c, f = O.predict_and_render_chunk(P["cp"], P["fp"], o, d, t, u, cfg, True, "zero", emulate_bf16=emu)
        out[f"{tag}_c_image"] = c["image"]; out[f"{tag}_c_depth"] = c["depth"]; out[f"{tag}_c_weights"] = c["weights"]
        out[f"{tag}_f_image"] = f["image"]; out[f"{tag}_t_fine"] = f["t"]
        rc, lc, gc = O.chunk_loss_and_grads(P["cp"], o, d, t, img, cfg, True, emulate_bf16=emu)
        g = O.flatten_params(gc)
        idx = np.linspace(0, g.size - 1, 4096).astype(np.int64)
        out[f"{tag}_coarse_loss"] = np.array([lc]); out[f"{tag}_grad_c_idx"] = idx; out[f"{tag}_grad_c_sample"] = g[idx]
        out[f"{tag}_grad_c_l2_per_tensor"] = np.array([np.linalg.norm(x.astype(np.float64)) for x in gc])
    np.savez_compressed(os.path.join(OUT, "fullsize_r64.npz"), **out)


def analytic():
    """hand-derivable answers kept as data: constant-sigma slab, PE of known angles, focal"""
    S, s, dl = 16, 0.7, 0.25
    w = (1 - np.exp(-s * dl)) * np.exp(-s * dl * np.arange(S)); w[-1] = (1 - np.exp(-s * 1e-10)) * np.exp(-s * dl * (S - 1))
    x = np.array([0.0, np.pi / 2, 1.0])
    pe = np.concatenate([x, np.sin(x), np.cos(x), np.sin(2 * x), np.cos(2 * x)])
    np.savez_compressed(os.path.join(OUT, "analytic.npz"), slab_sigma=np.array([s]), slab_delta=np.array([dl]), slab_weights=w,
                        pe_x=x, pe_L2=pe, focal_fov=np.array([0.6911112070083618]), focal_width=np.array([100.0]),
                        focal=np.array([138.88887889922103]))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    small(); fullsize(); analytic()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
