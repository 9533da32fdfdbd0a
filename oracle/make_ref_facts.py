#!/usr/bin/env python3
"""The reference's hot-path CONSTANTS and OPERATOR ARGUMENTS, derived MECHANICALLY from its text (round 6).

TEST INFRASTRUCTURE, BUILD CONTAINER ONLY (reads /root/reference; the GPU box has no reference tree).  The reference is PARSED with
`ast` -- never imported, never executed, no stand-in for TensorFlow is written (the task's rules allow neither a shim library nor a
partial import; reading the source as text is study) -- and nothing of its text is kept.  The output, tests/golden/ref_facts.json,
holds NUMBERS AND IDENTIFIERS ONLY, per function of the path:

    literals   every numeric literal of the function (defaults of its arguments included), as a sorted list of reprs
    calls      {dotted name of a TensorFlow op or of a method of the object itself: [per occurrence: the keyword arguments whose
               values are constants (axis=-1, exclusive=True, side="right", activation="relu", units=3 ...) and the number of
               positional arguments]} -- grouped by name; the interleaving of different calls (the function's structure) is NOT kept
    compares   comparison operators with a constant on one side ("< 1e-05", "> 0", "== 0")
    binops     binary operators with a constant operand ("1.0 - ", "2.0 ** ", "% ", "// 2")
    dict_keys  the string keys of the dict literals it builds ("image", "depth", "weights"; the log names)
and, under "_classes", for NeRF / NeRFUtils / NeRFMLP: the method names with their parameter names and the attributes assigned on
`self` anywhere in the class (what a script that drives the reference -- oracle/make_tf_golden.py -- may touch).

What this buys: the oracle (oracle/nerf_oracle.py) is a RESTATEMENT; its constants, keyword choices and comparison directions were
typed by hand from the reference.  tests/test_ref_facts.py holds, per fact, the executable check on the oracle that corresponds to
it, so a mistyped epsilon, a `side` that is not "right", a missing `exclusive`, an activation on the wrong layer or a `>=` for a `>`
fails the CPU suite.  What it does NOT buy: TensorFlow's own semantics (what `searchsorted(side="right")` or `cumprod(exclusive=True)`
compute, out-of-range gathers, the clip gradient) stay DECLARED, not measured -- the oracle remains "parity unpinned".

    python oracle/make_ref_facts.py [--reference /root/reference] [--out tests/golden/ref_facts.json]"""
import argparse
import ast
import json
import os

# file -> functions of the path (class methods by bare name; SURVEY.md section 8a rows a-1 .. a-16)
FUNCTIONS = {
    "keras_nerf/model/nerf/utils.py": ["render_image_depth_chunk", "fine_hierarchical_sampling_chunk", "render_image_depth",
                                        "fine_hierarchical_sampling", "positional_encoding", "encode_position_and_directions"],
    "keras_nerf/model/nerf/mlp.py": ["__init__", "call"],
    "keras_nerf/model/nerf/nerf.py": ["__init__", "compile", "_predict_and_render_chunk", "train_step", "test_step"],
    "keras_nerf/data/rays.py": ["__call__"],
    "keras_nerf/data/utils.py": ["get_focal_from_fov", "get_translation_t", "get_rotation_phi", "get_rotation_theta", "pose_spherical"],
}


def _dotted(node):
    parts = []
    while isinstance(node, ast.Attribute):
        parts.append(node.attr)
        node = node.value
    if isinstance(node, ast.Name):
        parts.append(node.id)
        return ".".join(reversed(parts))
    return None


def _const(node):
    """a constant's JSON value (numbers, strings, booleans, None; a negated number), or the marker for 'not a constant'"""
    if isinstance(node, ast.Constant) and isinstance(node.value, (int, float, str, bool, type(None))):
        return node.value
    if isinstance(node, ast.UnaryOp) and isinstance(node.op, ast.USub) and isinstance(node.operand, ast.Constant) and isinstance(node.operand.value, (int, float)):
        return -node.operand.value
    return _const


OPS = {ast.Add: "+", ast.Sub: "-", ast.Mult: "*", ast.Div: "/", ast.FloorDiv: "//", ast.Mod: "%", ast.Pow: "**",
       ast.Lt: "<", ast.LtE: "<=", ast.Gt: ">", ast.GtE: ">=", ast.Eq: "==", ast.NotEq: "!="}


def facts_of(fn: ast.FunctionDef) -> dict:
    lits, calls, compares, binops = [], [], [], []
    doc = ast.get_docstring(fn, clean=False)
    for node in ast.walk(fn):
        if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)) and not isinstance(node.value, bool):
            lits.append(repr(node.value))
        elif isinstance(node, ast.Call):
            name = _dotted(node.func)
            if name is None:
                continue
            kw = {k.arg: _const(k.value) for k in node.keywords if k.arg and _const(k.value) is not _const}
            if name.startswith(("tf.", "self.")):          # TensorFlow ops and the object's own methods: what the oracle restates
                calls.append({"name": name, "n_positional": len(node.args), "const_kwargs": kw, "line_order": (node.lineno, node.col_offset)})
        elif isinstance(node, ast.Compare) and len(node.ops) == 1:
            l, r = _const(node.left), _const(node.comparators[0])
            op = OPS.get(type(node.ops[0]))
            if op and (l is not _const or r is not _const) and not (isinstance(l, str) or isinstance(r, str)):
                compares.append(f"{'' if l is _const else repr(l)} {op} {'' if r is _const else repr(r)}".strip())
        elif isinstance(node, ast.BinOp):
            l, r = _const(node.left), _const(node.right)
            op = OPS.get(type(node.op))
            if op and ((l is not _const) != (r is not _const)) and not (isinstance(l, str) or isinstance(r, str)):
                binops.append(f"{'' if l is _const else repr(l)} {op} {'' if r is _const else repr(r)}".strip())
    calls.sort(key=lambda c: c.pop("line_order"))
    grouped = {}
    for c in calls:
        grouped.setdefault(c.pop("name"), []).append(c)
    calls = dict(sorted(grouped.items()))
    del doc
    keys = sorted({k.value for node in ast.walk(fn) if isinstance(node, ast.Dict) for k in node.keys
                   if isinstance(k, ast.Constant) and isinstance(k.value, str)})
    return {"literals": sorted(lits), "calls": calls, "compares": sorted(compares), "binops": sorted(binops), "dict_keys": keys,
            "arg_defaults": {a.arg: _const(d) for a, d in zip(fn.args.args[len(fn.args.args) - len(fn.args.defaults):], fn.args.defaults) if _const(d) is not _const}}


def class_surface(tree: ast.Module, cls: str) -> dict:
    """identifiers only: {method: [parameter names]}, the attributes assigned on self, and the string keys of every dict literal"""
    for node in ast.walk(tree):
        if isinstance(node, ast.ClassDef) and node.name == cls:
            methods = {f.name: [a.arg for a in f.args.args[1:]] + ([f"**{f.args.kwarg.arg}"] if f.args.kwarg else [])
                       for f in node.body if isinstance(f, ast.FunctionDef)}
            attrs = sorted({t.attr for n in ast.walk(node) if isinstance(n, (ast.Assign, ast.AugAssign, ast.AnnAssign))
                            for t in (n.targets if isinstance(n, ast.Assign) else [n.target])
                            if isinstance(t, ast.Attribute) and isinstance(t.value, ast.Name) and t.value.id == "self"})
            keys = sorted({k.value for n in ast.walk(node) if isinstance(n, ast.Dict) for k in n.keys
                           if isinstance(k, ast.Constant) and isinstance(k.value, str)})
            return {"methods": methods, "self_attrs": attrs, "dict_keys": keys}
    raise SystemExit(f"class {cls} not found")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_facts.json"))
    args = ap.parse_args()
    out = {}
    for rel, names in FUNCTIONS.items():
        tree = ast.parse(open(os.path.join(args.reference, rel)).read())
        found = {}
        for node in ast.walk(tree):
            if isinstance(node, ast.FunctionDef) and node.name in names and node.name not in found:
                found[node.name] = facts_of(node)
        missing = [n for n in names if n not in found]
        if missing:
            raise SystemExit(f"{rel}: functions not found: {missing}")
        out[rel] = {n: found[n] for n in names}
    out["_classes"] = {}
    for rel, cls in (("keras_nerf/model/nerf/nerf.py", "NeRF"), ("keras_nerf/model/nerf/utils.py", "NeRFUtils"), ("keras_nerf/model/nerf/mlp.py", "NeRFMLP")):
        out["_classes"][cls] = class_surface(ast.parse(open(os.path.join(args.reference, rel)).read()), cls)
    with open(args.out, "w") as f:
        json.dump({"_made_by": "oracle/make_ref_facts.py (ast of the reference's text: numbers and identifiers only)", **out}, f, indent=1, sort_keys=True)
    print(args.out, {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
