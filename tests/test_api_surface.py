"""The call surface of the reference's scripts, callback and tests -- every constructor keyword, positional count, method and
attribute they use -- checked against this implementation's classes with inspect.signature / hasattr (CPU).

tests/golden/api_surface.json is DERIVED from the reference by oracle/make_api_surface.py (ast only: train_single.py, train.py,
inference.py, keras_nerf/model/nerf/callback.py, tests/**; identifiers only) in the build container; here it is data.  A keyword must
be a NAMED parameter of the shim (a bare **kwargs would accept anything and prove nothing) unless listed in VIA_KWARGS with the
place where the shim reads it."""
import inspect
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SURFACE = json.load(open(os.path.join(HERE, "golden", "api_surface.json")))
# keywords the shim takes through **kwargs exactly as the reference does (nerf.py:78 `compile(..., **kwargs)` hands run_eagerly to Keras)
VIA_KWARGS = {("NeRF", "compile"): {"run_eagerly"}}


def _classes():
    import keras_nerf.data.image as di
    import keras_nerf.data.loader as dl
    import keras_nerf.data.rays as dr
    import keras_nerf.data.utils as du
    import keras_nerf.model.nerf.callback as cb
    import keras_nerf.model.nerf.mlp as mlp
    import keras_nerf.model.nerf.nerf as nerf
    import keras_nerf.model.nerf.utils as ut
    # through the ALIAS package: the module paths the reference's imports name (train_single.py:8-12, inference.py:9-13)
    return ({"NeRF": nerf.NeRF, "NeRFMLP": mlp.NeRFMLP, "NeRFUtils": ut.NeRFUtils, "RaysGenerator": dr.RaysGenerator, "DatasetLoader": dl.DatasetLoader,
             "ImageLoader": di.ImageLoader, "NeRFTrainMonitor": cb.NeRFTrainMonitor},
            {"pose_spherical": du.pose_spherical, "get_focal_from_fov": du.get_focal_from_fov})


def _accepts(fn, slot, drop_self, via_kwargs=()):
    """None, or what the reference passes that `fn` would not take"""
    sig = inspect.signature(fn)
    params = list(sig.parameters.values())
    if drop_self and params and params[0].name in ("self", "cls"):
        params = params[1:]
    named = {p.name for p in params if p.kind in (p.POSITIONAL_OR_KEYWORD, p.KEYWORD_ONLY)}
    var_kw = any(p.kind == p.VAR_KEYWORD for p in params)
    n_pos = sum(p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD) for p in params)
    var_pos = any(p.kind == p.VAR_POSITIONAL for p in params)
    for k in slot["keywords"]:
        if k in named:
            continue
        if k in via_kwargs and var_kw:
            continue
        return f"keyword {k!r} is not a named parameter of {getattr(fn, '__qualname__', fn)}{sig}"
    if slot["max_positional"] > n_pos and not var_pos:
        return f"{slot['max_positional']} positional arguments, {getattr(fn, '__qualname__', fn)}{sig} takes {n_pos}"
    return None


def test_the_surface_file_covers_the_reference_scripts_and_tests():
    assert {"train_single.py", "train.py", "inference.py", "keras_nerf/model/nerf/callback.py", "tests/model/nerf/test_nerf_mlp.py",
            "tests/model/nerf/test_nerf_utils.py", "tests/data/test_rays.py", "tests/data/test_loader.py"} <= set(SURFACE["_files"])
    c = SURFACE["classes"]
    # spot checks that the extraction sees what a reader of the reference sees
    assert "model_path" in c["NeRF"]["init"]["keywords"] and "run_eagerly" in c["NeRF"]["methods"]["compile"]["keywords"]
    assert {"coarse", "fine", "n_coarse"} <= set(c["NeRF"]["attributes"]) and "weights_only" in c["NeRF"]["methods"]["save_model"]["keywords"]
    assert c["NeRFMLP"]["methods"]["__call__"]["max_positional"] == 1 and "get_config" in c["NeRFMLP"]["methods"]
    assert len(c["NeRFUtils"]["methods"]) == 5 and "last_epoch" in c["NeRFTrainMonitor"]["attributes"]
    assert all(v["init"]["sites"] >= 1 for v in c.values())


@pytest.mark.parametrize("name", sorted(SURFACE["classes"]))
def test_class_accepts_every_call_the_reference_makes(name, tmp_path):
    classes, _ = _classes()
    cls, want = classes[name], SURFACE["classes"][name]
    bad = _accepts(cls.__init__, want["init"], True, VIA_KWARGS.get((name, "__init__"), ()))
    assert bad is None, bad
    for m, slot in want["methods"].items():
        assert callable(getattr(cls, m, None)), f"{name}.{m} is missing"
        bad = _accepts(getattr(cls, m), slot, True, VIA_KWARGS.get((name, m), ()))
        assert bad is None, bad
    # attributes: on the class, or on an instance where one can be made without a GPU
    inst = None
    if name == "NeRF":
        inst = cls()
    elif name == "NeRFMLP":
        inst = cls(n_layers=8, dense_units=256, skip_layer=4)
    elif name == "DatasetLoader":
        inst = cls(str(tmp_path), False)
    elif name == "ImageLoader":
        inst = cls(8, 8)
    elif name == "NeRFTrainMonitor":
        from keras_nerf_amd.data.loader import _Iterator

        class OneBatch(list):                      # callback.py:41-48 reads `dataset.take(1)` and `iter(dataset).get_next()` in its constructor
            def take(self, n):
                return self[:n]

            def __iter__(self):
                return _Iterator(list.__iter__(self))
        import numpy as np
        z = np.zeros((1, 4, 4, 3), np.float32)
        inst = cls(dataset=OneBatch([(z, (z, z, z))]), log_dir=str(tmp_path / "log"), batch_size=1, update_freq=1)
    for a in want["attributes"]:
        assert hasattr(cls, a) or (inst is not None and hasattr(inst, a)), f"{name}.{a} is read by the reference and missing here"


@pytest.mark.parametrize("name", sorted(SURFACE["functions"]))
def test_function_accepts_every_call_the_reference_makes(name):
    _, funcs = _classes()
    bad = _accepts(funcs[name], SURFACE["functions"][name], False)
    assert bad is None, bad
