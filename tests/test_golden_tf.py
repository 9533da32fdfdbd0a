"""Oracle AND HIP path against vectors produced by the reference itself under TensorFlow (oracle/make_tf_golden.py).

The vectors do not exist in this repository yet: no TensorFlow in the build image, so the generator has
never run and every test here SKIPS with that reason.  The day someone runs `python -m oracle.make_tf_golden` in a container with
TensorFlow and commits tests/golden/tf_default_r64.npz (+ tf_coarse.h5), these tests are what turns "parity unpinned" into a pin:
the fp32 oracle must reproduce TensorFlow to fp32 round-off, the HIP path to its stated bf16 tolerances, and the dependency-free
HDF5 reader must read a checkpoint that Keras wrote."""
import os

import numpy as np
import pytest

from oracle import nerf_oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NPZ, H5 = os.path.join(G, "tf_default_r64.npz"), os.path.join(G, "tf_coarse.h5")
needs_vectors = pytest.mark.skipif(not os.path.exists(NPZ), reason="tests/golden/tf_default_r64.npz absent: TensorFlow is not available in the "
                                   "build image, run oracle/make_tf_golden.py where it is (DESIGN.md section 3)")


def _problem():
    from tests.problem import make_problem
    z = np.load(NPZ)
    P = make_problem(n_images=1, wh=8, seed=42, weight_scale=1.5, bias_std=0.05)
    np.testing.assert_array_equal(P["o"], z["o"]); np.testing.assert_array_equal(P["t"], z["t"])        # the generator used these inputs
    N = P["N"]
    return z, P, N, P["o"].reshape(N, 3), P["d"].reshape(N, 3), P["t"].reshape(N, -1), P["img"].reshape(N, 3)


def _tags(z):
    return [t for t in ("inrange", "plain") if str(z[f"{t}_status"]) == "ok"]


@needs_vectors
def test_fp32_oracle_reproduces_tensorflow():
    z, P, N, o, d, t, img = _problem()
    cfg = P["cfg"]
    c = O.predict_and_render_chunk_single(P["cp"], o, d, t, cfg, True)
    np.testing.assert_allclose(c["image"], z["c_image"], atol=2e-6); np.testing.assert_allclose(c["weights"], z["c_weights"], atol=2e-6)
    np.testing.assert_allclose(c["depth"], z["c_depth"], atol=1e-5)
    assert _tags(z), "neither sampling case ran under TensorFlow"
    for tag in _tags(z):
        u = z["u_" + tag]
        # TF GPU gathers 0 out of range ("zero"); the in-range case is the same under both modes
        _, f = O.predict_and_render_chunk(P["cp"], P["fp"], o, d, t, u, cfg, True, "zero")
        np.testing.assert_allclose(f["t"], z[f"{tag}_t_fine"], atol=1e-5)
        np.testing.assert_allclose(f["image"], z[f"{tag}_f_image"], atol=1e-4)
        cp, fp = [p.copy() for p in P["cp"]], [p.copy() for p in P["fp"]]
        oc, of_ = O.KerasAdam(cp), O.KerasAdam(fp)
        losses = []
        for step in range(2):
            m, _, _, (gc, gf) = O.train_step(cp, fp, oc, of_, P["img"], P["o"], P["d"], P["t"], u[None].reshape(1, 8, 8, -1), cfg, N, True, "zero")
            losses.append((m["coarse_loss"], m["fine_loss"]))
            if step == 0:
                for g, ref in ((gc, z[f"{tag}_grad_c"]), (gf, z[f"{tag}_grad_f"])):
                    g = O.flatten_params(g)
                    assert np.abs(g - ref).max() < 1e-4 * np.abs(ref).max()
        np.testing.assert_allclose(losses[0], z[f"{tag}_step0_losses"], rtol=1e-5)
        np.testing.assert_allclose(np.mean(losses, 0), z[f"{tag}_step1_losses"], rtol=1e-5)      # Keras' running Mean over the two steps
        for w, ref in ((cp, z[f"{tag}_w_c_after"]), (fp, z[f"{tag}_w_f_after"])):
            np.testing.assert_allclose(O.flatten_params(w), ref, atol=5e-6)                        # two Adam steps of 1e-3


@needs_vectors
@pytest.mark.gpu
def test_hip_path_meets_tensorflow_at_the_stated_tolerance():
    import torch
    from keras_nerf_amd.runtime import KnerfContext
    from tests.test_gpu_train import per_tensor_err
    z, P, N, o, d, t, img = _problem()
    for tag in _tags(z):
        ctx = KnerfContext(white_background=True, oob="zero")
        ctx.set_weights(0, O.flatten_params(P["cp"])); ctx.set_weights(1, O.flatten_params(P["fp"]))
        out = {k: v.cpu().numpy() for k, v in ctx.render_chunk(o, d, t, z["u_" + tag]).items()}
        assert np.abs(out["c_image"] - z["c_image"]).max() < 2e-2 and O.psnr(out["c_image"].reshape(1, 8, 8, 3), z["c_image"].reshape(1, 8, 8, 3))[0] > 45.0
        assert O.psnr(out["f_image"].reshape(1, 8, 8, 3), z[f"{tag}_f_image"].reshape(1, 8, 8, 3))[0] > 25.0       # through the sampler
        loss = torch.zeros(2, device="cuda")
        ctx.train_chunk(o, d, t, img, z["u_" + tag], loss=loss)
        g = ctx.grads_view().cpu().numpy(); n = ctx.param_count
        assert abs(float(loss[0]) - float(z[f"{tag}_step0_losses"][0])) < 3e-3
        assert per_tensor_err(g[:n], z[f"{tag}_grad_c"], P["cfg"])[0] < 8e-2          # GRAD_TOL_FP32 of tests/test_gpu_configs.py
        ctx.close()


@pytest.mark.skipif(not os.path.exists(H5), reason="tests/golden/tf_coarse.h5 absent (written by oracle/make_tf_golden.py under TensorFlow)")
def test_hdf5_reader_reads_a_keras_written_checkpoint():
    from keras_nerf_amd.io import hdf5_min
    from tests.problem import make_problem
    P = make_problem(n_images=1, wh=8, seed=42, weight_scale=1.5, bias_std=0.05)
    got = hdf5_min.read_keras_weights(H5, [n for n, _, _ in O.layer_shapes(P["cfg"])])
    assert len(got) == len(P["cp"])
    for a, b in zip(got, P["cp"]):
        np.testing.assert_array_equal(np.asarray(a), b)
