"""Host-side mirror of the reference interface: everything that is decidable without a GPU."""
import json
import os

import numpy as np
import pytest
import torch

from keras_nerf_amd.data import utils as DU
from keras_nerf_amd.model.nerf import nerf as N
from keras_nerf_amd.model.nerf.metrics import Mean, psnr, ssim
from keras_nerf_amd.model.nerf.mlp import NeRFMLP, layer_shapes
from oracle import nerf_oracle as O


def test_focal_known_answer_and_pose_match_oracle():
    assert DU.get_focal_from_fov(0.6911112070083618, 100) == pytest.approx(138.88887889922103, rel=1e-6)  # reference tests/data/test_utils.py:5-10
    for th, ph, r in ((0, -30, 4), (123.0, -10.0, 3.5)):
        np.testing.assert_allclose(DU.pose_spherical(th, ph, r), O.pose_spherical(th, ph, r), atol=1e-6)


def test_mlp_shapes_params_and_keras_order():
    m = NeRFMLP(8, 256, 4)
    with pytest.raises(ValueError, match="not built"):       # Keras: count_params of an unbuilt model raises
        m.count_params()
    assert m.get_weights() == [] and not m.built             # ... and it has no variables yet (mlp.py:11-27 names no input size)
    m.build()                                                # no shape given: the reference NeRF's own encodings, 63 / 27
    assert m.count_params() == 595844
    ws = m.get_weights()
    assert len(ws) == 24 and ws[0].shape == (63, 256) and ws[10].shape == (319, 256) and ws[16].shape == (256, 1)
    assert ws[20].shape == (283, 128) and ws[22].shape == (128, 3)
    assert all(np.all(b == 0) for b in ws[1::2])
    lim = np.sqrt(6.0 / (63 + 256))
    assert np.abs(ws[0]).max() <= lim and np.abs(ws[0]).max() > 0.9 * lim          # glorot_uniform
    assert [s[0] for s in layer_shapes(8, 256, 4, 63, 27)] == [s[0] for s in O.layer_shapes(O.NerfConfig())]
    assert m.get_config()["n_layers"] == 8 and m.get_config()["dense_units"] == 256 and m.get_config()["skip_layer"] == 4


def test_mlp_widths_are_fixed_by_build_weights_or_file_like_keras_dense(tmp_path):
    """the host half of Keras Dense's lazy input size (the call itself needs the GPU: tests/test_gpu_reference_shapes.py)"""
    m = NeRFMLP(8, 256, 4, seed=1)
    m.build(((None, 32, 99), (None, 32, 99)))                # what Keras passes to build(): the reference test's 99-wide inputs
    assert (m.xyz_dim, m.dir_dim) == (99, 99) and m.get_weights()[0].shape == (99, 256) and m.get_weights()[20].shape == (355, 128)
    assert m.count_params() == sum(w.size for w in m.get_weights())
    with pytest.raises(ValueError, match=r"99.*63"):
        m.build(((None, 63), (None, 27)))                    # a built model refuses other widths, naming both
    with pytest.raises(ValueError):
        NeRFMLP(8, 256, 4, xyz_dim=63)                        # the two extension arguments come together
    # an unbuilt model adopts the widths of the weights it is given ...
    m2 = NeRFMLP(8, 256, 4)
    m2.set_weights(m.get_weights())
    assert (m2.xyz_dim, m2.dir_dim) == (99, 99)
    np.testing.assert_array_equal(m2.get_flat_weights(), m.get_flat_weights())
    with pytest.raises(ValueError):
        NeRFMLP(8, 256, 4).set_weights(m.get_weights()[:-2])
    # ... or of the file it loads (Keras-layout HDF5 and .npz)
    for name in ("w.h5", "w.npz"):
        m.save_weights(str(tmp_path / name))
        m3 = NeRFMLP(8, 256, 4); m3.load_weights(str(tmp_path / name))
        assert (m3.xyz_dim, m3.dir_dim) == (99, 99)
        np.testing.assert_array_equal(m3.get_flat_weights(), m.get_flat_weights())
    with pytest.raises(ValueError, match="not built"):
        NeRFMLP(8, 256, 4).save_weights(str(tmp_path / "none.h5"))
    with pytest.raises(ValueError, match="not built"):
        NeRFMLP(8, 256, 4).summary()
    # a flat vector carries no shapes: only the reference NeRF's widths can be meant
    m4 = NeRFMLP(8, 256, 4); m4.set_flat_weights(np.zeros(595844, np.float32))
    assert (m4.xyz_dim, m4.dir_dim) == (63, 27)
    with pytest.raises(ValueError):
        NeRFMLP(8, 256, 4).set_flat_weights(np.zeros(m.count_params(), np.float32))


def test_weight_file_roundtrip(tmp_path):
    m = NeRFMLP(8, 256, 4, seed=3); m.build()
    p = str(tmp_path / "coarse.h5")
    m.save_weights(p)
    m2 = NeRFMLP(8, 256, 4, seed=4); m2.load_weights(p)
    np.testing.assert_array_equal(m.get_flat_weights(), m2.get_flat_weights())


def test_save_model_writes_reference_file_names(tmp_path):
    n = N.NeRF(seed=1)
    n.coarse.build(); n.fine.build()
    n.save_model(str(tmp_path / "m"))
    assert sorted(os.listdir(tmp_path / "m")) == ["coarse.h5", "fine.h5", "model_config.json"]
    cfg = json.load(open(tmp_path / "m" / "model_config.json"))
    assert cfg == dict(n_coarse=64, n_fine=128, pos_emb_xyz=10, pos_emb_dir=4, n_layers=8, dense_units=256, skip_layer=4)
    n2 = N.NeRF(n_coarse=1, model_path=str(tmp_path / "m"))
    assert n2.n_coarse == 64 and n2.n_fine == 128


def test_compile_asserts_divisibility_like_the_reference():
    n = N.NeRF()
    with pytest.raises(AssertionError, match="must be a divisor"):
        n.compile("adam", "mse", batch_size=1, image_height=400, image_width=400, ray_chunks=16384)   # 160000 % 16384 != 0


def test_loss_and_optimizer_validation():
    assert N._is_mse("mse") and N._is_mse(None) and N._is_mse(lambda a, b: torch.mean((a - b) ** 2))
    assert not N._is_mse(lambda a, b: torch.mean(torch.abs(a - b)))
    assert N._adam_hyper("adam") == dict(lr=1e-3, beta1=0.9, beta2=0.999, epsilon=1e-7)
    assert N._adam_hyper({"learning_rate": 5e-4})["lr"] == 5e-4
    with pytest.raises(ValueError):
        N._adam_hyper("sgd")
    with pytest.raises(ValueError):
        N.NeRF().compile("adam", lambda a, b: torch.mean(torch.abs(a - b)), 1, 8, 8, 8)


def test_metrics_against_definitions():
    """Mean is host-side; psnr / ssim run on the GPU (tests/test_gpu_api.py checks them against these oracle definitions)"""
    rng = np.random.default_rng(0)
    a = rng.random((2, 16, 16, 3), dtype=np.float32); b = rng.random((2, 16, 16, 3), dtype=np.float32)
    assert O.ssim(a, a) == pytest.approx([1.0, 1.0], abs=1e-12)
    s = O.ssim(a, b)
    assert s.shape == (2,) and float(s.max()) < 0.2
    # one window, constant images: luminance term only -> (2 x y + c1) / (x^2 + y^2 + c1)
    x, y = np.full((1, 11, 11, 1), 0.2), np.full((1, 11, 11, 1), 0.6)
    assert O.ssim(x, y)[0] == pytest.approx((2 * 0.12 + 1e-4) / (0.04 + 0.36 + 1e-4), rel=1e-9)
    from keras_nerf_amd.runtime import KnerfError
    with pytest.raises(KnerfError):
        ssim(torch.tensor(a), torch.tensor(b))          # CPU tensors: no CPU path
    m = Mean("x"); m.update_state(torch.tensor([1.0, 3.0])); m.update_state(5.0)
    assert m.result() == 3.0
    m.reset_state(); assert m.result() == 0.0


def test_image_loader_composites_and_resizes(tmp_path):
    from PIL import Image
    from keras_nerf_amd.data.image import ImageLoader
    rgba = np.zeros((8, 8, 4), np.uint8); rgba[2:6, 2:6] = (255, 0, 0, 255); rgba[0, 0] = (0, 255, 0, 128)
    p = str(tmp_path / "a.png"); Image.fromarray(rgba, "RGBA").save(p)
    w = ImageLoader(8, 8, white_background=True)(p); b = ImageLoader(8, 8, white_background=False)(p)
    assert w.shape == (8, 8, 4) and w.dtype == np.float32 and w.min() >= 0 and w.max() <= 1      # reference tests/data/test_image.py:12-20
    np.testing.assert_allclose(w[7, 7], [1, 1, 1, 0]); np.testing.assert_allclose(b[7, 7], [0, 0, 0, 0])    # transparent -> background
    np.testing.assert_allclose(w[3, 3], [1, 0, 0, 1]); np.testing.assert_allclose(b[3, 3], [1, 0, 0, 1])
    np.testing.assert_allclose(w[0, 0, :3], 128 / 255 * np.array([0, 1, 0]) + (1 - 128 / 255), atol=1e-6)   # alpha blend
    assert ImageLoader(4, 4)(p).shape == (4, 4, 4)


def test_antialiased_resize_follows_the_scale_and_translate_algorithm(tmp_path):
    """image.py:22-23 `tf.image.resize(image, size, antialias=True)`: triangle kernel widened by the reduction factor, taps on pixel
    centres, weights normalised per output pixel, float32 throughout (data/image.py).  Known answers of that algorithm."""
    from PIL import Image
    from keras_nerf_amd.data.image import ImageLoader, resize_antialiased, triangle_resize_weights
    rng = np.random.default_rng(0)
    np.testing.assert_array_equal(triangle_resize_weights(7, 7), np.eye(7, dtype=np.float32))             # equal size: identity
    W = triangle_resize_weights(16, 8)                                                                      # 2:1 -> taps (1, 3, 3, 1) / 8
    np.testing.assert_allclose(W[3, 5:9], [0.125, 0.375, 0.375, 0.125], atol=1e-7)
    assert np.count_nonzero(W[3]) == 4
    np.testing.assert_allclose(W[0, :3], np.array([3, 3, 1]) / 7.0, atol=1e-7)                             # border: the taps inside, renormalised
    W = triangle_resize_weights(800, 128)                                                                   # the lego images at --img_wh 128
    np.testing.assert_allclose(W.sum(axis=1), 1.0, atol=2e-6)
    assert (W >= 0).all() and np.count_nonzero(W[64]) in (12, 13)                                            # radius 6.25 input pixels
    np.testing.assert_allclose(W[10], W[117][::-1], atol=1e-7)                                              # mirror symmetry
    W = triangle_resize_weights(4, 8)                                                                       # enlarging: plain bilinear, half-pixel centres
    np.testing.assert_allclose(W[3, 1:3], [0.75, 0.25], atol=1e-7)
    np.testing.assert_allclose(W[0, :2], [1.0, 0.0], atol=1e-7)
    const = resize_antialiased(np.full((50, 37, 4), 0.3, np.float32), 11, 13)
    np.testing.assert_allclose(const, 0.3, atol=1e-6)                                                       # partition of unity
    img8 = rng.integers(0, 256, (100, 100, 4), dtype=np.uint8)
    img8[..., 3] = 255                                   # opaque: PIL resizes RGBA with PREMULTIPLIED alpha, the reference channel by channel (below)
    ours = resize_antialiased(img8.astype(np.float32) / 255, 16, 16)
    pil = np.asarray(Image.fromarray(img8, "RGBA").resize((16, 16), resample=Image.BILINEAR), np.float32) / 255
    assert np.abs(ours - pil).max() <= 1.01 / 255        # the same kernel; PIL rounds every output to 8 bits (rounds 1-4 used it)
    assert np.abs(ours - pil).mean() < 0.3 / 255
    # the reference resizes the FOUR channels independently and blends afterwards (image.py:22-31): at a silhouette the colour is mixed
    # with the transparent pixels' black AND weighted by the resized alpha.  2 x 2 -> 1 x 1, left column opaque red, right column empty:
    # rgb = (0.5, 0, 0), alpha = 0.5 -> over white (0.75, 0.5, 0.5), over black (0.25, 0, 0).  (A premultiplied-alpha resize, as PIL's,
    # would give (1, 0.5, 0.5) / (0.5, 0, 0): rounds 1-4 differed from the reference along every silhouette.)
    edge = np.zeros((2, 2, 4), np.uint8); edge[:, 0] = (255, 0, 0, 255)
    p = str(tmp_path / "edge.png"); Image.fromarray(edge, "RGBA").save(p)
    np.testing.assert_allclose(ImageLoader(1, 1, white_background=True)(p)[0, 0], [0.75, 0.5, 0.5, 0.5], atol=1e-6)
    np.testing.assert_allclose(ImageLoader(1, 1, white_background=False)(p)[0, 0], [0.25, 0.0, 0.0, 0.5], atol=1e-6)
    img8 = rng.integers(0, 256, (100, 100, 4), dtype=np.uint8)
    ours = resize_antialiased(img8.astype(np.float32) / 255, 16, 16)
    p = str(tmp_path / "r.png"); Image.fromarray(img8, "RGBA").save(p)
    out = ImageLoader(16, 16, white_background=True)(p)
    want = ours[..., 3:4] * ours[..., :3] + (1 - ours[..., 3:4])
    np.testing.assert_allclose(out[..., :3], np.clip(want, 0, 1), atol=1e-6)
    assert len(np.unique(np.round(out[..., 3] * 255, 3) % 1)) > 4                                           # not quantised to 1/255 any more


def test_dataset_loader_json_and_shuffle_semantics(tmp_path):
    from keras_nerf_amd.data.loader import DatasetLoader
    from tests.synthetic_scene import write
    root = write(str(tmp_path / "scene"), n=(5, 2, 3))
    tr, va, te = DatasetLoader(root, white_background=True).load_dataset(2, 24, 24, 2.0, 6.0, 64)
    assert len(tr.image_paths) == 5 and len(va.image_paths) == 2 and len(te.image_paths) == 3
    assert len(tr) == 2 and len(te) == 1                          # batch 2, drop_remainder
    assert tr.image_paths[0].endswith("train/r_0.png") and tr.camera_params[0].shape == (4, 4)
    from keras_nerf_amd.data.loader import shuffled_order
    order = shuffled_order(len(tr.image_paths), tr.batch_size, tr._rng)
    assert sorted(order) == list(range(5))
    assert all(order[k] <= k + 2 for k in range(5))                # a buffer of batch_size can pull an element at most that far forward


def test_bench_refuses_more_ranks_than_gpus_before_touching_hip():
    """`python bench.py --gpus N` starts its own N ranks; with fewer than N devices visible it must stop with a clear message
    (no silent one-GPU run).  Here no GPU is visible at all."""
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two GPUs visible: the refusal path needs fewer devices than ranks")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "KNERF_DIST_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert "only" in r.stderr and "GPU(s) visible" in r.stderr and "2 ranks" in r.stderr       # keras_nerf_amd/parallel.py launch
    assert "{" not in r.stdout              # no JSON line


def test_optimizer_objects_other_than_plain_adam_are_refused_not_ignored():
    """tf.keras.optimizers.get (nerf.py:163-165) honours any optimizer; the fused kernel is plain Adam with a constant
    learning rate, so everything else must raise instead of silently training with Adam defaults."""
    import types

    class Adam:                                   # duck-typed tf.keras.optimizers.Adam
        def __init__(self, **kw):
            self.cfg = dict(name="Adam", learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7, amsgrad=False); self.cfg.update(kw)

        def get_config(self):
            return dict(self.cfg)

    class SGD(Adam):
        def __init__(self):
            super().__init__(name="SGD", momentum=0.0)

    assert N._adam_hyper(Adam(learning_rate=2e-4, beta_1=0.8)) == dict(lr=2e-4, beta1=0.8, beta2=0.999, epsilon=1e-7)
    assert N._adam_hyper({"class_name": "Adam", "config": {"learning_rate": 3e-4}})["lr"] == 3e-4
    assert N._adam_hyper(types.SimpleNamespace(learning_rate=1e-2, epsilon=1e-8)) == dict(lr=1e-2, beta1=0.9, beta2=0.999, epsilon=1e-8)
    for bad in (SGD(), Adam(amsgrad=True), Adam(weight_decay=0.01), Adam(clipnorm=1.0), Adam(global_clipnorm=2.0), Adam(use_ema=True),
                Adam(learning_rate={"class_name": "ExponentialDecay", "config": {}}), Adam(learning_rate=lambda step: 1e-3),
                {"class_name": "RMSprop", "config": {}}):
        with pytest.raises(ValueError):
            N._adam_hyper(bad)


def test_data_parallel_batches_are_disjoint_slices_of_the_global_batches():
    """train.py:75-93: one dataset of GLOBAL batches (batch_size x replicas), replica r gets rows [r*b, (r+1)*b).  Every
    rank derives the same shuffled order from the shared seed and keeps its slice (keras_nerf_amd/data/loader.py)."""
    from keras_nerf_amd.data.loader import RayImageDataset, replica_batches, shuffled_order
    n, b, world = 23, 2, 3
    orders = [shuffled_order(n, b * world, np.random.default_rng(7)) for _ in range(world)]
    assert orders[0] == orders[1] == orders[2] and sorted(orders[0]) == list(range(n))
    per_rank = [replica_batches(orders[r], b, r, world) for r in range(world)]
    single = replica_batches(orders[0], b * world)                      # what one process with the global batch would see
    assert len(single) == n // (b * world) == len(per_rank[0])
    for k, g in enumerate(single):
        assert sum((per_rank[r][k] for r in range(world)), []) == g      # concatenated replica slices = the global batch
    seen = [i for r in range(world) for bt in per_rank[r] for i in bt]
    assert len(seen) == len(set(seen))                                   # no image is used twice in an epoch
    # tf.data shuffle(buffer) semantics: element k of the output comes from the first k + buffer inputs
    assert all(v < k + b * world for k, v in enumerate(orders[0]))
    # the dataset object applies exactly that plan (no GPU needed until a batch is materialised).  Its batch_size is the GLOBAL batch,
    # as the reference's train.py passes it (train.py:84-93: load_dataset(batch_size=args.batch_size * num_replicas_in_sync))
    ds = [RayImageDataset([f"img{i}" for i in range(n)], [np.eye(4)] * n, None, None, b * world, seed=5, rank=r, world=world) for r in range(world)]
    assert [len(d) for d in ds] == [3, 3, 3] and [d._local_batch(world) for d in ds] == [b, b, b]
    assert len(RayImageDataset(["x"] * n, [np.eye(4)] * n, None, None, b, seed=5)) == 11     # outside a process group: world 1
    with pytest.raises(ValueError, match="not divisible"):
        RayImageDataset(["x"] * n, [np.eye(4)] * n, None, None, 4, seed=5, rank=0, world=3)._local_batch(3)


def test_reference_import_names_resolve_to_this_implementation():
    """`keras_nerf.*` (the reference's package name, train_single.py:8-12 / inference.py:8-11) is an alias package of
    re-exports, so scripts written against the reference import this implementation without an import swap."""
    import importlib
    pairs = {"keras_nerf.model.nerf.nerf": ["NeRF"], "keras_nerf.model.nerf.utils": ["NeRFUtils"], "keras_nerf.model.nerf.mlp": ["NeRFMLP"],
             "keras_nerf.model.nerf.callback": ["NeRFTrainMonitor"], "keras_nerf.data.rays": ["RaysGenerator"],
             "keras_nerf.data.utils": ["pose_spherical", "get_focal_from_fov"], "keras_nerf.data.loader": ["DatasetLoader"],
             "keras_nerf.data.image": ["ImageLoader"]}
    for mod, names in pairs.items():
        a = importlib.import_module(mod)
        b = importlib.import_module(mod.replace("keras_nerf.", "keras_nerf_amd.", 1))
        for n in names:
            assert getattr(a, n) is getattr(b, n), (mod, n)


def test_nerfmlp_initializer_names_follow_keras():
    """mlp.py:5, 13-27: `initializer` goes to every Dense kernel_initializer.  The VarianceScaling family by name: limits of the
    uniform variants, standard deviations (and the +-2 sigma truncation) of the normal ones, zero biases; unknown names raise."""
    from keras_nerf_amd.model.nerf.mlp import NeRFMLP
    fi, fo = 256, 256
    want = {"glorot_uniform": ("u", np.sqrt(6 / (fi + fo))), "he_uniform": ("u", np.sqrt(6 / fi)), "lecun_uniform": ("u", np.sqrt(3 / fi)),
            "glorot_normal": ("n", np.sqrt(2 / (fi + fo))), "he_normal": ("n", np.sqrt(2 / fi)), "lecun_normal": ("n", np.sqrt(1 / fi))}
    for name, (kind, v) in want.items():
        m = NeRFMLP(initializer=name, seed=3); m.build()
        ws = m.get_weights()
        k = ws[2]                                    # layer_1: 256 x 256
        assert k.shape == (fi, fo) and not ws[3].any()
        if kind == "u":
            assert np.abs(k).max() <= v and np.abs(k).max() > 0.99 * v and abs(k.std() - v / np.sqrt(3)) < 0.02 * v
        else:
            assert abs(k.std() - v) < 0.02 * v and np.abs(k).max() <= 2.0 * v / 0.87962566103423978 + 1e-6
    a = NeRFMLP(seed=3); a.build(); b = NeRFMLP(initializer="glorot_uniform", seed=3); b.build()
    np.testing.assert_array_equal(a.get_flat_weights(), b.get_flat_weights())
    c = NeRFMLP(initializer=lambda shape: np.full(shape, 0.25, np.float32)); c.build()
    assert float(c.get_weights()[0][0, 0]) == 0.25
    with pytest.raises(ValueError):
        NeRFMLP(initializer="orthogonal")


def test_bench_quotes_pmc_traffic_of_the_instantiation_it_ran_only():
    """`roofline.traffic` must come from counters of the kernel instantiation the bench ran.  The weight-gradient
    kernel has a list-mode instantiation (`wgrad_kernel<Shape, NET, true>`, skip_dead_tiles on: what bench.py runs by default) and a
    contiguous one; bench.pmc_traffic returns the committed summary that holds the matching row, never the other one's."""
    import importlib.util
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("knerf_bench", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    for kernel in ("wgrad_fine", "wgrad_coarse", "mlp_bwd_fine", "mlp_fwd_fine"):
        for skip in (True, False):
            val, src = b.pmc_traffic(kernel, skip)
            assert val and src, (kernel, skip)
            rep = json.load(open(os.path.join(root, src)))
            assert rep["_layout"] == b.LAYOUT_TAG
            rows = [k for k, v in rep.items() if isinstance(v, dict) and v.get("hbm_bytes_per_launch") == val]
            assert rows, (kernel, skip, src)
            if kernel.startswith("wgrad"):
                assert all(k.split(" grid=")[0].endswith(", true>" if skip else ", false>") for k in rows), (rows, skip)
            if kernel.startswith("mlp_bwd"):
                assert bool(rep.get("_options", {}).get("skip_dead_tiles")) == skip
    # algorithmic bytes of the dominant kernel against the counters of its own instantiation: nothing re-read
    val, _ = b.pmc_traffic("wgrad_fine", True)
    assert 1.0 < val / (24576 * b.WGRAD_KIB_PER_TILE * 1024) < 1.05
    assert b.pmc_traffic("composite", True) == (None, None)


def test_bench_refuses_counters_of_other_kernels_and_names_its_binary(tmp_path):
    """(c) a PMC summary is quoted only when its recorded `_kernel_digest` (source of the three big kernels +
    flags, keras_nerf_amd/build.py) equals that of the library the process loaded -- a summary without a digest, one of an edited
    kernel, or a library whose record describes another file gives `traffic: null`; (b) the line's provenance names the binary."""
    import importlib.util
    import json
    import os
    import shutil
    from keras_nerf_amd import _lib, build as B
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("knerf_bench", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    row = {"wgrad_kernel<Shape<8, 4, 256>, 1, true> grid=65536": {"hbm_bytes_per_launch": 6.2e9}}
    good = B.kernel_digest()
    for name, digest in (("r01_pmc_traffic_old.json", None), ("r02_pmc_traffic_edited.json", "0123456789abcdef")):
        rep = {"_layout": b.LAYOUT_TAG, "_options": {"skip_dead_tiles": 1}, **row}
        if digest:
            rep["_kernel_digest"] = digest
        json.dump(rep, open(tmp_path / name, "w"))
    assert b.pmc_traffic("wgrad_fine", True, good, str(tmp_path)) == (None, None)          # nothing of THESE kernels: no number
    assert b.pmc_traffic("wgrad_fine", True, None, str(tmp_path)) == (None, None)          # a library without a valid record: no number
    json.dump({"_layout": b.LAYOUT_TAG, "_options": {"skip_dead_tiles": 1}, "_kernel_digest": good,
               "wgrad_kernel<Shape<8, 4, 256>, 1, true> grid=65536": {"hbm_bytes_per_launch": 6.1e9}}, open(tmp_path / "r00_pmc_traffic_mine.json", "w"))
    val, src = b.pmc_traffic("wgrad_fine", True, good, str(tmp_path))                      # an OLDER file name, but the right kernels
    assert val == 6.1e9 and src.endswith("r00_pmc_traffic_mine.json")
    # the digest follows the kernel sources and the flags, not the host code
    assert len(good) == 16 and B.kernel_digest(B.FLAGS + ["-DKNERF_MASK_LAYOUT=1"]) != good
    assert "knerf_api.hip" not in B.KERNEL_FILES and {"wgrad_body.h", "chain.h", "layout.h"} <= set(B.KERNEL_FILES)
    # build_info: the record beside a library counts only while it describes THAT file
    lib = tmp_path / "libfake.so"
    lib.write_bytes(b"\x7fELF one build")
    B._write_info(str(lib), B.FLAGS)
    info = _lib.build_info(str(lib))
    assert info["kernel_digest"] == good and info["lib_sha16"] == B.file_sha16(str(lib))
    lib.write_bytes(b"\x7fELF another build")                                              # re-linked, record not renewed
    stale = _lib.build_info(str(lib))
    assert stale["kernel_digest"] is None and stale["lib_sha16"] != info["lib_sha16"]
    if os.path.exists(_lib.LIB_PATH):                                                      # the product library of this tree carries its record
        prov = b.provenance()
        assert prov["lib_sha16"] == B.file_sha16(_lib.LIB_PATH) and prov["bench_py_sha16"] == B.file_sha16(os.path.join(root, "bench.py"))
        assert prov["kernel_digest"] == good, "libknerf_hip.so was not built from the kernel sources of this tree: run keras_nerf_amd/build.py"


def test_knerf_lib_override_is_announced(monkeypatch, caplog):
    """KNERF_LIB swaps the product binary for an A/B build (tools/kbench.py): never silently"""
    import importlib
    import logging
    from keras_nerf_amd import _lib
    monkeypatch.setenv("KNERF_LIB", "/nonexistent/libknerf_hip_variant.so")
    fresh = importlib.reload(_lib)
    try:
        with caplog.at_level(logging.WARNING):
            with pytest.raises(fresh.KnerfError):
                fresh.load()
        assert any("KNERF_LIB is set" in r.getMessage() and "/nonexistent/libknerf_hip_variant.so" in r.getMessage() for r in caplog.records)
    finally:
        monkeypatch.delenv("KNERF_LIB")
        importlib.reload(_lib)


def test_zero_gradient_messages_are_the_references(caplog):
    """nerf.py:430-451: the three log lines, their levels, and 'once per published step' -- the host side of the device-side count,
    driven by a stand-in context (the GPU test drives the real one)."""
    import logging
    from keras_nerf_amd.model.nerf.nerf import NeRF

    class Ctx:
        def __init__(self):
            self.seq, self.counts, self.stats = 0, (1, 1), ((5, 10), (7, 10))

        def grad_diagnostics(self, wait=True):
            return self.counts[0], self.counts[1], self.seq

        def get_option(self, name):
            return 1.0

        def tile_stats_net(self, reset=True):
            # the library's counters are RUNNING totals; the diagnostics must read them without resetting (other consumers
            # accumulate across steps) and difference against their previous read
            assert reset is False
            return self.running
    nerf = NeRF.__new__(NeRF)
    nerf._ctx, nerf._diag_seen = Ctx(), 0
    assert nerf._zero_gradient_diagnostics(wait=True) is None                      # nothing published yet
    for counts, stats, want in (((0, 0), ((0, 8), (0, 24)), [(logging.ERROR, "Both Coarse and Fine Gradient are zero")]),
                                ((0, 9), ((0, 8), (3, 24)), [(logging.WARNING, "Coarse Gradient is zero")]),
                                ((4, 0), ((2, 8), (0, 24)), [(logging.WARNING, "Fine Gradient is zero")]),
                                ((4, 9), ((2, 8), (3, 24)), [])):
        nerf._ctx.seq += 1; nerf._ctx.counts, nerf._ctx.stats = counts, stats
        prev = getattr(nerf._ctx, "running", ((0, 0), (0, 0)))
        nerf._ctx.running = tuple((p[0] + s_[0], p[1] + s_[1]) for p, s_ in zip(prev, stats))      # this step's tiles on top of the totals
        caplog.clear()
        with caplog.at_level(logging.WARNING):
            assert nerf._zero_gradient_diagnostics(wait=True) == counts
            assert nerf._zero_gradient_diagnostics(wait=True) is None              # the same step is not reported twice
        got = [(r.levelno, r.getMessage()) for r in caplog.records]
        for w in want:
            assert w in got, (counts, got)
        assert len([g for g in got if "Gradient" in g[1]]) == len(want)
        dead = [g[1] for g in got if "Every sample tile" in g[1]]
        assert len(dead) == sum(1 for live, total in stats if total > 0 and live == 0), (stats, dead)


def test_auto_build_does_not_compile_from_inside_a_multi_rank_job(monkeypatch, caplog, tmp_path):
    """KNERF_AUTO_BUILD is two minutes of hipcc behind a file lock -- eight ranks taking turns inside the process
    group's time-outs.  With torch.distributed initialised and world > 1 a shape that is not built yet is NOT compiled (warning with the
    command to build it beforehand; the context stays on the general-shape kernels); a library built earlier is adopted as it is."""
    import logging
    import torch.distributed as dist
    from keras_nerf_amd import build as B
    from keras_nerf_amd.runtime import KnerfContext
    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_world_size", lambda *a, **k: 8)
    monkeypatch.setattr(dist, "get_rank", lambda *a, **k: 3)
    monkeypatch.setattr(B, "build", lambda **kw: (_ for _ in ()).throw(AssertionError("hipcc must not be started from a rank")))
    ctx = object.__new__(KnerfContext)              # no GPU needed: the decision is taken before the library is touched
    adopted = []
    ctx._adopt_library = lambda path, spec: adopted.append((os.path.basename(path), spec))
    with caplog.at_level(logging.WARNING):
        assert ctx._rebuild_for("7,5,128") is None
    msg = " ".join(r.getMessage() for r in caplog.records)
    assert not adopted and "not compiling shape 7,5,128 from rank 3 of a 8-rank job" in msg and "--variant=auto_7_5_128 --add-shape=7,5,128" in msg
    # built beforehand (by one process): loaded without a compiler run
    here = os.path.dirname(os.path.abspath(B.__file__))
    built = os.path.join(here, "libknerf_hip_auto_7_5_128.so")
    open(built, "wb").close()
    try:
        ctx._rebuild_for("7,5,128")
        assert adopted == [("libknerf_hip_auto_7_5_128.so", "7,5,128")]
    finally:
        os.remove(built)
