"""The oracle's constants, keyword choices and comparison directions against the reference's TEXT (round 6).

tests/golden/ref_facts.json is derived mechanically -- `ast` of /root/reference, numbers and identifiers only, nothing imported or
executed (oracle/make_ref_facts.py; the build container regenerates it, this test only reads the committed file) -- and every fact
the oracle's restatement depends on is tied here to an EXECUTABLE check on the oracle: a mistyped epsilon, a `side` other than
"right", a dropped `exclusive`, an activation on the wrong layer, `>=` for `>` fails the CPU suite.  This removes transcription risk
for what is WRITTEN in the reference; what TensorFlow's ops compute for those arguments stays declared (parity unpinned, DESIGN.md
section 3)."""
import ast
import inspect
import json
import os
import textwrap

import numpy as np
import pytest

from oracle import nerf_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FACTS = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_facts.json")))
U = FACTS["keras_nerf/model/nerf/utils.py"]
M = FACTS["keras_nerf/model/nerf/mlp.py"]
N = FACTS["keras_nerf/model/nerf/nerf.py"]
R = FACTS["keras_nerf/data/rays.py"]
DU = FACTS["keras_nerf/data/utils.py"]

TRIVIAL = {"0", "1", "2", "3", "0.0", "1.0", "-1", "-2"}


def _nontrivial(f):
    return sorted(x for x in f["literals"] if x not in TRIVIAL)


def _calls(f, name):
    return f["calls"].get(name, [])


def _oracle_literals(*fns):
    out = []
    for fn in fns:
        for node in ast.walk(ast.parse(textwrap.dedent(inspect.getsource(fn)))):
            if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)) and not isinstance(node.value, bool):
                out.append(repr(node.value))
    return out


def test_the_committed_facts_are_current_where_the_reference_is_present():
    """build container only: the committed file is what the generator makes of the reference tree today"""
    if not os.path.isdir("/root/reference/keras_nerf"):
        pytest.skip("no reference tree on this machine (GPU box): the committed facts are used as they are")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_ref_facts", os.path.join(ROOT, "oracle", "make_ref_facts.py"))
    g = importlib.util.module_from_spec(spec); spec.loader.exec_module(g)
    for rel, names in g.FUNCTIONS.items():
        tree = ast.parse(open(os.path.join("/root/reference", rel)).read())
        seen = set()
        for node in ast.walk(tree):
            if isinstance(node, ast.FunctionDef) and node.name in names and node.name not in seen:
                seen.add(node.name)
                assert json.loads(json.dumps(g.facts_of(node))) == FACTS[rel][node.name], (rel, node.name)


def test_compositing_constants_and_operator_arguments():
    f = U["render_image_depth_chunk"]
    # ONE non-trivial number in the whole function: the default epsilon, used as the last delta AND inside the cumprod (utils.py:17,36,46)
    assert _nontrivial(f) == ["1e-10"] and f["arg_defaults"] == {"epsilon": 1e-10}
    assert inspect.signature(O.render_image_depth_chunk).parameters["epsilon"].default == 1e-10
    assert "1e-10" in _oracle_literals(O.render_image_depth_chunk)
    assert _calls(f, "tf.math.cumprod")[0]["const_kwargs"] == {"axis": -1, "exclusive": True}
    assert [c["const_kwargs"] for c in _calls(f, "tf.reduce_sum")] == [{"axis": -2}, {"axis": -1}, {"axis": -1}]
    assert f["binops"] == ["1.0 -"] * 3 and "0.0" in f["literals"] and _calls(f, "tf.clip_by_value")[0]["n_positional"] == 3
    # ... and the oracle does exactly that: known answer on one ray, two samples, fp64
    sig, t = np.array([[1e9, 3.0]]), np.array([[2.0, 2.5]])
    rgb = np.array([[[0.2, 0.4, 0.6], [1.0, 1.0, 1.0]]])
    img, depth, w = O.render_image_depth_chunk(rgb, sig, t, False)
    a0 = 1.0 - np.exp(-1e9 * 0.5)                        # delta_0 = t_1 - t_0
    a1 = 1.0 - np.exp(-3.0 * 1e-10)                      # the LAST delta is epsilon = 1e-10 (not 1e10: alpha_1 would be 1)
    T1 = (1.0 - a0) + 1e-10                              # exclusive product of (1 - alpha) + epsilon: T_0 = 1, T_1 = that
    assert w[0, 0] == a0 * 1.0 and w[0, 1] == pytest.approx(a1 * T1, rel=1e-12) and 0 < w[0, 1] < 1e-18
    np.testing.assert_allclose(img[0], a0 * rgb[0, 0] + a1 * T1 * rgb[0, 1], rtol=1e-12)            # sum over the SAMPLE axis (-2)
    assert depth[0] == pytest.approx(a0 * 2.0 + a1 * T1 * 2.5, rel=1e-12)
    # white background adds 1 - sum(w); the clip is to [0, 1]
    img_w, _, _ = O.render_image_depth_chunk(np.full((1, 2, 3), 0.9), np.array([[0.1, 0.1]]), t, True)
    wsum = float(O.render_image_depth_chunk(np.full((1, 2, 3), 0.9), np.array([[0.1, 0.1]]), t, True)[2].sum())
    assert img_w[0, 0] == pytest.approx(0.9 * wsum + 1.0 - wsum, rel=1e-12)
    assert O.render_image_depth_chunk(np.full((1, 2, 3), 5.0), np.array([[50.0, 50.0]]), t, False)[0].max() == 1.0


def test_sampling_constants_and_operator_arguments():
    for name in ("fine_hierarchical_sampling_chunk", "fine_hierarchical_sampling"):
        f = U[name]
        assert _nontrivial(f) == ["1e-05", "1e-05"] and f["compares"] == ["< 1e-05"]
        assert _calls(f, "tf.searchsorted")[0]["const_kwargs"] == {"side": "right"}
        assert _calls(f, "tf.cumsum")[0]["const_kwargs"] == {"axis": -1}
        assert _calls(f, "tf.reduce_sum")[0]["const_kwargs"] == {"axis": -1, "keepdims": True}
        assert len(_calls(f, "tf.maximum")) == 1 and len(_calls(f, "tf.minimum")) == 1
        assert f["binops"].count("- 1") == 2            # indices - 1 (below), cdf.shape[-1] - 1 (above)
    assert _oracle_literals(O.cdf_from_weights).count("1e-05") == 1 and _oracle_literals(O.fine_hierarchical_sampling_chunk).count("1e-05") == 1
    # w + 1e-5 BEFORE normalising: all-zero weights give the uniform cdf k / S, with the leading 0 column
    cdf = O.cdf_from_weights(np.zeros((1, 8)))
    np.testing.assert_allclose(cdf[0], np.arange(9) / 8.0, atol=1e-15)
    # ... and 1e-5, not another power: weights (1, 0): pdf = (1 + 1e-5, 1e-5) / (1 + 2e-5)
    np.testing.assert_allclose(O.cdf_from_weights(np.array([[1.0, 0.0]]))[0], [0.0, (1 + 1e-5) / (1 + 2e-5), 1.0], rtol=1e-14)
    # side="right" = the number of cdf entries <= u; below = max(0, idx - 1), above = min(len(cdf) - 1, idx); `denom < 1e-5` -> 1
    mids = np.array([[10.0, 20.0, 30.0]]); w = np.array([[1.0, 1.0, 1.0, 1.0]])        # four weights, THREE mid-points (nerf.py:182-187)
    u = np.array([[0.0, 0.25, 0.3, 0.999999]])
    s = O.fine_hierarchical_sampling_chunk(mids, w, u, oob="clamp")
    # u = 0: idx = 1 (cdf[0] = 0 <= 0), below 0, above 1, t = 0 -> mids[0].  u = 0.25 = cdf[1] exactly: idx = 2 -> below 1: mids[1] + 0
    assert s[0, 0] == 10.0 and s[0, 1] == 20.0
    assert s[0, 2] == pytest.approx(20.0 + (0.3 - 0.25) / 0.25 * 10.0, rel=1e-9)
    # a bin narrower than 1e-5 is divided by 1, not by its width: weights (1, 0, 1, 1): cdf = (0, 1/3, 1/3 + 3e-6, 2/3, 1)
    w4 = np.array([[1.0, 0.0, 1.0, 1.0]])
    c = O.cdf_from_weights(w4)[0]
    um = 0.5 * (c[1] + c[2])                                                   # the middle of the narrow bin
    s2 = O.fine_hierarchical_sampling_chunk(mids, w4, np.array([[um]]), oob="clamp")
    assert 0 < c[2] - c[1] < 1e-5 and s2[0, 0] == pytest.approx(20.0 + (um - c[1]) / 1.0 * 10.0, rel=1e-12) and s2[0, 0] < 20.0001      # (by its width: 25)


def test_positional_encoding_and_ray_points():
    f = U["positional_encoding"]
    assert _nontrivial(f) == ["2.0"] and f["binops"] == ["2.0 **"]                    # 2 ** i, NO pi factor anywhere in the function
    assert _calls(f, "tf.concat")[0]["const_kwargs"] == {"axis": -1}
    x = np.array([[0.3, -1.2, 2.0]])
    pe = O.positional_encoding(x, 3)
    want = np.concatenate([x] + [fn(2.0 ** i * x) for i in range(3) for fn in (np.sin, np.cos)], axis=-1)     # sin first, then cos, per frequency
    np.testing.assert_allclose(pe, want, rtol=1e-15)
    assert "3.141592653589793" not in _oracle_literals(O.positional_encoding)
    e = U["encode_position_and_directions"]
    assert _nontrivial(e) == [] and len(_calls(e, "self.positional_encoding")) == 2
    o, d, t = np.array([[1.0, 2.0, 3.0]]), np.array([[0.0, 0.6, 0.8]]), np.array([[2.0, 4.0]])
    xyz, dire = O.encode_position_and_directions(o, d, t, 2, 1)
    np.testing.assert_allclose(xyz[0, 1, :3], [1.0, 2.0 + 0.6 * 4.0, 3.0 + 0.8 * 4.0], rtol=1e-15)         # o + d * t
    np.testing.assert_allclose(dire[0, :, :3], [[0.0, 0.6, 0.8]] * 2, rtol=1e-15)                           # d broadcast over the samples, not normalised again


def test_mlp_layers_activations_and_the_concat_rule():
    dense = _calls(M["__init__"], "tf.keras.layers.Dense")
    assert [c["const_kwargs"] for c in dense] == [{"activation": "relu"}, {"activation": "relu", "name": "sigma", "units": 1},
                                                  {"name": "features"}, {"name": "rgb_features"},                     # NO activation: linear
                                                  {"activation": "sigmoid", "name": "rgb", "units": 3}]
    assert M["__init__"]["arg_defaults"] == {"dense_units": 256, "initializer": "glorot_uniform", "n_layers": 8, "skip_layer": 4}
    assert M["__init__"]["binops"] == ["// 2"]                                                                       # rgb_features: dense_units // 2
    assert sorted(M["call"]["compares"]) == ["== 0", "> 0"] and M["call"]["binops"] == []                           # i % skip_layer == 0 and i > 0
    assert [c["const_kwargs"] for c in _calls(M["call"], "tf.keras.layers.concatenate")] == [{"axis": -1}, {"axis": -1}]
    # the rule has NO "not the last layer" condition: 5 layers / skip 2 concatenates behind layers 2 AND 4, so sigma / features take 319
    cfg = O.NerfConfig(n_layers=5, skip_layer=2)
    assert [(n, i, o) for n, i, o in O.layer_shapes(cfg)] == [("layer_0", 63, 256), ("layer_1", 256, 256), ("layer_2", 256, 256), ("layer_3", 319, 256),
                                                              ("layer_4", 256, 256), ("sigma", 319, 1), ("features", 319, 256),
                                                              ("rgb_features", 283, 128), ("rgb", 128, 3)]
    # activations as executed: relu trunk and sigma, sigmoid rgb, LINEAR features / rgb_features (negating both of a linear layer's
    # kernel+bias and the rows that read it leaves the output unchanged; with a relu in between it would not)
    cfg = O.NerfConfig(n_layers=3, dense_units=8, skip_layer=2, pos_emb_xyz=1, pos_emb_dir=1)
    rng = np.random.default_rng(3)
    p = [rng.normal(0, 0.4, q.shape) for q in O.init_params(cfg, 0, dtype=np.float64)]
    xyz, dire = rng.normal(0, 1, (7, cfg.xyz_dim)), rng.normal(0, 1, (7, cfg.dir_dim))
    rgb, sigma = O.mlp_forward(p, xyz, dire, cfg)
    assert (sigma >= 0).all() and (sigma == 0).any() and ((rgb > 0) & (rgb < 1)).all()
    n = cfg.n_layers
    q = [a.copy() for a in p]
    q[2 * n + 2] *= -1; q[2 * n + 3] *= -1; q[2 * n + 4][:cfg.dense_units] *= -1        # features kernel, bias; rgb_features rows that read features
    np.testing.assert_allclose(O.mlp_forward(q, xyz, dire, cfg)[0], rgb, rtol=1e-12)
    q = [a.copy() for a in p]
    q[2 * n + 4] *= -1; q[2 * n + 5] *= -1; q[2 * n + 6] *= -1                           # rgb_features kernel, bias; rgb kernel
    np.testing.assert_allclose(O.mlp_forward(q, xyz, dire, cfg)[0], rgb, rtol=1e-12)
    q = [a.copy() for a in p]
    q[0] *= -1; q[1] *= -1; q[2] *= -1                                                  # a TRUNK layer is not linear: the same trick changes the output
    assert np.abs(O.mlp_forward(q, xyz, dire, cfg)[0] - rgb).max() > 1e-3


def test_nerf_defaults_chunk_forward_and_tape_arguments():
    assert N["__init__"]["arg_defaults"] == {"dense_units": 256, "model_path": None, "n_coarse": 64, "n_fine": 128, "n_layers": 8,
                                             "pos_emb_dir": 4, "pos_emb_xyz": 10, "skip_layer": 4}
    c = O.NerfConfig()
    assert (c.n_coarse, c.n_fine, c.pos_emb_xyz, c.pos_emb_dir, c.n_layers, c.dense_units, c.skip_layer) == (64, 128, 10, 4, 8, 256, 4)
    from keras_nerf_amd.model.nerf.nerf import NeRF
    sig = inspect.signature(NeRF.__init__).parameters
    assert {k: sig[k].default for k in N["__init__"]["arg_defaults"]} == N["__init__"]["arg_defaults"]
    assert N["compile"]["arg_defaults"] == {"is_training": True, "white_background": False} and N["compile"]["compares"] == ["== 0"]    # the divisibility assert
    csig = inspect.signature(NeRF.compile).parameters
    assert csig["is_training"].default is True and csig["white_background"].default is False
    f = N["_predict_and_render_chunk"]
    assert _nontrivial(f) == ["0.5"] and f["binops"] == ["0.5 *"]                       # mid-points
    assert _calls(f, "tf.sort")[0]["const_kwargs"] == {"axis": -1} and _calls(f, "tf.concat")[0]["const_kwargs"] == {"axis": -1}
    t = np.array([[2.0, 3.0, 5.0]]); w = np.array([[0.2, 0.5, 0.3]]); u = np.array([[0.1, 0.9]])
    pts = O.fine_points(t, w, u, oob="clamp")
    assert pts.shape == (1, 5) and (np.diff(pts) >= 0).all() and set(t[0]) <= set(pts[0])            # concat(coarse, fine), ascending
    mids = 0.5 * (t[..., 1:] + t[..., :-1])
    np.testing.assert_allclose(np.sort(np.concatenate([t, O.fine_hierarchical_sampling_chunk(mids, w, u, "clamp")], -1), -1), pts, rtol=0)
    # only the MLP's variables are watched by the two tapes: no gradient is taken with respect to anything else
    assert [c["const_kwargs"] for c in _calls(N["train_step"], "tf.GradientTape")] == [{"watch_accessed_variables": False}] * 2


def test_rays_and_pose_constants():
    f = R["__call__"]
    assert _nontrivial(f) == ["0.5", "0.5"] and sorted(f["binops"]) == ["* 0.5", "* 0.5", "/ 2"]
    assert _calls(f, "tf.meshgrid")[0]["const_kwargs"] == {"indexing": "xy"} and _calls(f, "tf.norm")[0]["const_kwargs"] == {"axis": -1, "keepdims": True}
    c2w = O.pose_spherical(30.0, -30.0, 4.0)
    o, d, t = O.generate_rays(c2w, 50.0, 4, 4, 2.0, 6.0, 8, np.full((4, 4, 8), 0.5, np.float32))
    np.testing.assert_allclose(np.linalg.norm(d, axis=-1), 1.0, rtol=1e-6)              # normalised
    np.testing.assert_allclose(t[0, 0], np.linspace(2.0, 6.0, 8), rtol=1e-6)           # noise 0.5: + interval / 2 - interval / 2
    cam = np.array([(0 - 4 * 0.5) / 50.0, -(0 - 4 * 0.5) / 50.0, -1.0])                # pixel (0, 0): x - W * 0.5, -(y - H * 0.5), -1
    want = (c2w[:3, :3] @ cam); want /= np.linalg.norm(want)
    np.testing.assert_allclose(d[0, 0], want, rtol=1e-5, atol=1e-6)
    assert _nontrivial(DU["get_focal_from_fov"]) == ["0.5", "0.5"] and _nontrivial(DU["pose_spherical"]) == ["180.0", "180.0"]
    assert DU["pose_spherical"]["binops"] == ["/ 180.0", "/ 180.0"]
    assert O.get_focal_from_fov(0.6911112070083618, 100) == pytest.approx(138.88887889922103, rel=1e-6)       # the reference's own known answer
    np.testing.assert_allclose(O.pose_spherical(0.0, 0.0, 4.0), [[-1, 0, 0, 0], [0, 0, 1, 4], [0, 1, 0, 0], [0, 0, 0, 1]], atol=1e-7)


def test_the_script_that_will_drive_the_reference_under_tensorflow_touches_only_what_exists():
    """oracle/make_tf_golden.py has never run (no TensorFlow in this pipeline).  Short of running it, everything it touches on the
    reference's objects is checked against the reference's text: the methods it calls and the attributes it reads on a `NeRF`, the
    keywords it passes, the positional counts, the result-dict keys and log names it reads (shake the bugs out
    before the day TensorFlow exists -- statically, since a stand-in library is not allowed)."""
    src = ast.parse(open(os.path.join(ROOT, "oracle", "make_tf_golden.py")).read())
    nerf_c, utils_c = FACTS["_classes"]["NeRF"], FACTS["_classes"]["NeRFUtils"]
    used = {n.attr for n in ast.walk(src) if isinstance(n, ast.Attribute) and isinstance(n.value, ast.Name) and n.value.id == "nerf"}
    assert used == {"compile", "coarse", "fine", "_predict_and_render_chunk", "coarse_optimizer", "fine_optimizer", "nerf_utils", "train_step"}
    assert used <= set(nerf_c["methods"]) | set(nerf_c["self_attrs"])
    calls = {}
    for n in ast.walk(src):
        if isinstance(n, ast.Call) and isinstance(n.func, ast.Attribute):
            calls.setdefault(n.func.attr, []).append(n)
    # compile: every keyword is a parameter of the reference's compile (run_eagerly travels in **kwargs to Keras, nerf.py:78-105)
    (comp,) = calls["compile"]
    kws = {k.arg for k in comp.keywords}
    params = [p for p in nerf_c["methods"]["compile"] if not p.startswith("**")]
    assert not comp.args and kws - {"run_eagerly"} <= set(params) and "**kwargs" in nerf_c["methods"]["compile"]
    assert {"optimizer", "loss", "batch_size", "image_height", "image_width", "ray_chunks"} <= kws           # the ones without a default
    # _predict_and_render_chunk(ray_chunks[, coarse_weights_chunk]); train_step(data); the sampler's three positional arguments
    assert nerf_c["methods"]["_predict_and_render_chunk"] == ["ray_chunks", "coarse_weights_chunk"]
    assert sorted(len(c.args) for c in calls["_predict_and_render_chunk"]) == [1, 2] and all(not c.keywords for c in calls["_predict_and_render_chunk"])
    assert nerf_c["methods"]["train_step"] == ["inputs"] and all(len(c.args) == 1 for c in calls["train_step"])
    assert utils_c["methods"]["fine_hierarchical_sampling_chunk"] == ["mid_points", "weights", "n_samples"]
    assert all(len(c.args) == 3 for c in calls["fine_hierarchical_sampling_chunk"])
    # NeRF() with no arguments relies on the reference's defaults being the kernels' shape (checked above against NerfConfig)
    assert all(not (c.args or c.keywords) for n_ in ast.walk(src) if isinstance(n_, ast.Call) and isinstance(n_.func, ast.Name) and n_.func.id == "NeRF" for c in [n_])
    # result keys read from the chunk forward, and the log names read from train_step's return value
    subs = {(n.value.id, n.slice.value) for n in ast.walk(src) if isinstance(n, ast.Subscript) and isinstance(n.value, ast.Name)
            and isinstance(n.slice, ast.Constant) and isinstance(n.slice.value, str)}
    chunk_keys = set(N["_predict_and_render_chunk"]["dict_keys"])
    assert chunk_keys == {"image", "depth", "weights"}
    assert {k for v, k in subs if v in ("coarse", "fine")} <= chunk_keys
    assert {k for v, k in subs if v == "logs"} == {"coarse_loss", "fine_loss"} <= set(nerf_c["dict_keys"])
    # ... and this implementation's NeRF offers the same surface the script uses (the import-swapped script would run here too)
    from keras_nerf_amd.model.nerf.nerf import NeRF
    for name in ("compile", "_predict_and_render_chunk", "train_step"):
        assert callable(getattr(NeRF, name))
    assert list(inspect.signature(NeRF._predict_and_render_chunk).parameters)[1:3] == ["ray_chunks", "coarse_weights_chunk"]
