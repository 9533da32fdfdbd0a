"""Golden vectors FROM THE REFERENCE ITSELF -- the one route to a pinned oracle.  TEST INFRASTRUCTURE; NOT RUNNABLE HERE.

This image has no TensorFlow (ordinary ModuleNotFoundError), so this script has never been executed and
tests/test_golden_tf.py skips while its output is absent.  In ANY container that has tensorflow>=2.9, the reference checkout and
this repo (never the GPU box -- the reference does not travel):

    PYTHONPATH=/root/reference:/root/repo python -m oracle.make_tf_golden

It imports the reference (keras_nerf.model.nerf.nerf.NeRF), injects this repo's seeded inputs -- rays, targets, both nets' weights
(tests/problem.py) and the sampler's `u` (by patching tf.random.uniform for the one shape utils.py:72-73 asks for) -- runs it
eagerly and writes tests/golden/tf_default_r64.npz in the schema of oracle/make_golden.py (+ tf_coarse.h5 from save_weights):
coarse / fine images, depths, weights, merged t, losses, the 48 gradients of one train_step (captured at apply_gradients) and the
weights after two Adam steps.  `u` comes in two flavours: "inrange" (scaled below cdf[:, 62] of the reference's own coarse pass, so
that no mid-point gather leaves 0..62: valid on CPU and GPU TensorFlow alike) and "plain" uniform draws, ~3 % of which gather out
of range (tf.gather yields 0 on GPU and raises on CPU; what happened is recorded in `plain_status`)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")


def main():
    import tensorflow as tf
    from keras_nerf.model.nerf.nerf import NeRF          # the reference (PYTHONPATH)
    from oracle import nerf_oracle as O
    from tests.problem import make_problem
    P = make_problem(n_images=1, wh=8, seed=42, weight_scale=1.5, bias_std=0.05)
    N, cfg = P["N"], P["cfg"]
    rays = tuple(tf.constant(P[k]) for k in ("o", "d", "t"))
    u_var = tf.Variable(tf.zeros([N, cfg.n_fine]), trainable=False)
    real_uniform = tf.random.uniform

    def uniform(shape, *a, **kw):                          # utils.py:72-73: u = tf.random.uniform(shape=[ray_chunks, n_samples])
        return u_var.read_value() if [int(s) for s in shape] == [N, cfg.n_fine] else real_uniform(shape, *a, **kw)
    tf.random.uniform = uniform

    def fresh():
        nerf = NeRF()                                      # defaults = the kernels' shape: 64 + 128 samples, L = 10 / 4, 8 x 256, skip 4
        nerf.compile(optimizer="adam", loss=tf.keras.losses.MeanSquaredError(), batch_size=1, image_height=8, image_width=8,
                     ray_chunks=N, white_background=True, run_eagerly=True)
        nerf.coarse.set_weights([np.asarray(p) for p in P["cp"]]); nerf.fine.set_weights([np.asarray(p) for p in P["fp"]])
        return nerf
    nerf = fresh()
    flat = lambda x: tuple(tf.reshape(r, (N, -1)) for r in x)
    coarse = nerf._predict_and_render_chunk(flat(rays))
    w = coarse["weights"].numpy().astype(np.float64) + 1e-5
    cdf62 = (np.cumsum(w, -1) / w.sum(-1, keepdims=True))[:, 61:62]          # cdf[:, 62] = the sum of the first 62 pdf entries
    u_plain = P["u"].reshape(N, -1)
    out = dict(o=P["o"], d=P["d"], t=P["t"], img=P["img"], u_plain=u_plain, u_inrange=(u_plain * cdf62 * 0.999).astype(np.float32),
               tf_version=np.array(tf.__version__), device=np.array(coarse["image"].device),
               c_image=coarse["image"].numpy(), c_depth=coarse["depth"].numpy(), c_weights=coarse["weights"].numpy())
    for tag in ("inrange", "plain"):
        u_var.assign(out["u_" + tag])
        try:
            nerf = fresh()
            grads = []
            for opt in (nerf.coarse_optimizer, nerf.fine_optimizer):            # nerf.py:455-458: the accumulated gradients arrive here
                def spy(gv, _orig=opt.apply_gradients, **kw):
                    gv = list(gv); grads.append([g.numpy().copy() for g, _ in gv]); return _orig(gv, **kw)
                opt.apply_gradients = spy
            fine = nerf._predict_and_render_chunk(flat(rays), coarse["weights"])
            mid = 0.5 * (rays[2][..., 1:] + rays[2][..., :-1])
            t_fine = tf.sort(tf.concat([tf.reshape(rays[2], (N, -1)), nerf.nerf_utils.fine_hierarchical_sampling_chunk(
                tf.reshape(mid, (N, -1)), coarse["weights"], cfg.n_fine)], -1), -1)
            out.update({f"{tag}_f_image": fine["image"].numpy(), f"{tag}_f_depth": fine["depth"].numpy(),
                        f"{tag}_f_weights": fine["weights"].numpy(), f"{tag}_t_fine": t_fine.numpy()})
            for step in range(2):
                logs = nerf.train_step((tf.constant(P["img"]), rays))
                out[f"{tag}_step{step}_losses"] = np.array([float(logs["coarse_loss"]), float(logs["fine_loss"])])   # running means
            out[f"{tag}_grad_c"] = O.flatten_params(grads[0]); out[f"{tag}_grad_f"] = O.flatten_params(grads[1])    # first step
            out[f"{tag}_w_c_after"] = O.flatten_params(nerf.coarse.get_weights()); out[f"{tag}_w_f_after"] = O.flatten_params(nerf.fine.get_weights())
            out[f"{tag}_status"] = np.array("ok")
        except tf.errors.InvalidArgumentError as e:         # CPU TensorFlow: the out-of-range gather raises
            out[f"{tag}_status"] = np.array("raised: " + str(e)[:200])
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "tf_default_r64.npz"), **out)
    nerf = fresh(); nerf.coarse.save_weights(os.path.join(OUT, "tf_coarse.h5"))   # the Keras-written checkpoint io/hdf5_min.py must read
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})


if __name__ == "__main__":
    sys.exit(main())
