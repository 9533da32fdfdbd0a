#!/usr/bin/env python3
"""The call surface the reference's own scripts, callback and tests use -- derived MECHANICALLY.

BUILD CONTAINER ONLY (reads /root/reference; the GPU box has no reference tree).  The reference is PARSED with `ast`, never imported
or executed (TensorFlow is absent anyway), and nothing of its text is kept: the output, tests/golden/api_surface.json, holds
IDENTIFIERS ONLY -- for each of the public names of the hot path

    NeRF, NeRFMLP, NeRFUtils, RaysGenerator, DatasetLoader, ImageLoader, NeRFTrainMonitor (classes)
    pose_spherical, get_focal_from_fov (functions)

the keyword names passed and the largest number of positional arguments at any construction / call site, the methods called on
objects known to be instances (with their keywords / positional counts) and the attributes read from them.  "Known to be an
instance": a name assigned from a constructor call of the class (`nerf = NeRF(...)`), a pytest fixture that returns one (the
fixture's name as a test-function argument), `self.<name>` assigned from one, and `self.model` inside a `tf.keras.callbacks.Callback`
subclass (Keras sets it to the model `fit` was called on: a NeRF).  Files: train_single.py, train.py, inference.py,
keras_nerf/model/nerf/callback.py, tests/**.py.

tests/test_api_surface.py (CPU) checks every entry against the shim with inspect.signature / hasattr, so a call the reference makes
that this implementation would not accept fails the CPU suite instead of waiting for a user.

    python oracle/make_api_surface.py [--reference /root/reference] [--out tests/golden/api_surface.json]"""
import argparse
import ast
import glob
import json
import os

CLASSES = ("NeRF", "NeRFMLP", "NeRFUtils", "RaysGenerator", "DatasetLoader", "ImageLoader", "NeRFTrainMonitor")
FUNCTIONS = ("pose_spherical", "get_focal_from_fov")
FILES = ("train_single.py", "train.py", "inference.py", "keras_nerf/model/nerf/callback.py")
# call results that are instances as well: `train, val, test = loader.load_dataset(...)` is out of scope (tf.data objects)


def _callee(node):
    """NeRF(...) / module.NeRF(...) -> 'NeRF'"""
    f = node.func
    if isinstance(f, ast.Name):
        return f.id
    if isinstance(f, ast.Attribute):
        return f.attr
    return None


def _target_key(t):
    """assignment target -> a key for the instance table: 'name' or 'self.name'"""
    if isinstance(t, ast.Name):
        return t.id
    if isinstance(t, ast.Attribute) and isinstance(t.value, ast.Name) and t.value.id == "self":
        return "self." + t.attr
    return None


def _expr_key(e):
    if isinstance(e, ast.Name):
        return e.id
    if isinstance(e, ast.Attribute) and isinstance(e.value, ast.Name) and e.value.id == "self":
        return "self." + e.attr
    return None


class Surface:
    def __init__(self):
        self.s = {c: {"init": {"keywords": set(), "max_positional": 0, "sites": 0}, "methods": {}, "attributes": set()} for c in CLASSES}
        self.f = {f: {"keywords": set(), "max_positional": 0, "sites": 0} for f in FUNCTIONS}

    @staticmethod
    def _note(slot, call):
        slot["keywords"].update(k.arg for k in call.keywords if k.arg)
        slot["max_positional"] = max(slot["max_positional"], len(call.args))
        slot["sites"] += 1

    def scan(self, path, tree):
        # 1. instances visible anywhere in the file: fixtures (functions that return a constructor call), module / function level
        #    assignments, self.<x> assignments, and self.model in Keras callbacks
        inst = {}
        for fn in ast.walk(tree):
            if isinstance(fn, ast.FunctionDef):
                for st in ast.walk(fn):
                    if isinstance(st, ast.Return) and isinstance(st.value, ast.Call) and _callee(st.value) in CLASSES:
                        inst[fn.name] = _callee(st.value)          # a pytest fixture: its NAME is the instance in the tests' arguments
            if isinstance(fn, ast.ClassDef) and any("Callback" in ast.unparse(b) for b in fn.bases):
                inst["self.model"] = "NeRF"
        for st in ast.walk(tree):
            if isinstance(st, ast.Assign) and isinstance(st.value, ast.Call) and _callee(st.value) in CLASSES:
                for t in st.targets:
                    k = _target_key(t)
                    if k:
                        inst[k] = _callee(st.value)
            if isinstance(st, ast.Assign) and _expr_key(st.value) in inst:       # self.dataset = dataset and the like: aliases
                for t in st.targets:
                    k = _target_key(t)
                    if k:
                        inst.setdefault(k, inst[_expr_key(st.value)])
        # 2. constructions, function calls, method calls and attribute reads
        method_nodes = set()
        for node in ast.walk(tree):
            if isinstance(node, ast.Call):
                name = _callee(node)
                if isinstance(node.func, (ast.Name, ast.Attribute)) and name in CLASSES and not (isinstance(node.func, ast.Attribute) and _expr_key(node.func.value) in inst):
                    self._note(self.s[name]["init"], node)
                elif name in FUNCTIONS:
                    self._note(self.f[name], node)
                if _expr_key(node.func) in inst:                 # rays_generator(c2w), nerf_mlp((xyz, dirs)): the instance is called
                    m = self.s[inst[_expr_key(node.func)]]["methods"].setdefault("__call__", {"keywords": set(), "max_positional": 0, "sites": 0})
                    self._note(m, node)
                    method_nodes.add(id(node.func))
                if isinstance(node.func, ast.Attribute):
                    k = _expr_key(node.func.value)
                    if k in inst:
                        m = self.s[inst[k]]["methods"].setdefault(node.func.attr, {"keywords": set(), "max_positional": 0, "sites": 0})
                        self._note(m, node)
                        method_nodes.add(id(node.func))
        for node in ast.walk(tree):
            if isinstance(node, ast.Attribute) and id(node) not in method_nodes and isinstance(node.ctx, ast.Load):
                k = _expr_key(node.value)
                if k in inst:
                    self.s[inst[k]]["attributes"].add(node.attr)

    def to_json(self):
        def slot(x):
            return {"keywords": sorted(x["keywords"]), "max_positional": x["max_positional"], "sites": x["sites"]}
        return {"classes": {c: {"init": slot(v["init"]), "methods": {m: slot(s) for m, s in sorted(v["methods"].items())},
                                "attributes": sorted(v["attributes"])} for c, v in self.s.items()},
                "functions": {f: slot(v) for f, v in self.f.items()}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "api_surface.json"))
    args = ap.parse_args()
    files = [os.path.join(args.reference, f) for f in FILES] + sorted(glob.glob(os.path.join(args.reference, "tests", "**", "*.py"), recursive=True))
    S = Surface()
    for path in files:
        with open(path) as f:
            S.scan(path, ast.parse(f.read(), filename=path))
    out = {"_generator": "oracle/make_api_surface.py (ast only; identifiers only)",
           "_files": [os.path.relpath(p, args.reference) for p in files], **S.to_json()}
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print(json.dumps(out["classes"]["NeRF"], indent=1))


if __name__ == "__main__":
    main()
