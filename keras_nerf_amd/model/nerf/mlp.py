"""NeRFMLP -- counterpart of reference keras_nerf/model/nerf/mlp.py:4-59.

Holds the 24 trainable tensors of one MLP (Keras order: layer_0..layer_{n-1}, sigma, features, rgb_features, rgb; kernel
[in,out] then bias) as a flat fp32 vector; the arithmetic lives in the fused HIP kernels (csrc/mlp_fwd.hip).  When the
owning NeRF is compiled, the master copy is on the GPU and this object is a view onto it."""
from __future__ import annotations

import ctypes as C
from typing import List, Tuple

import numpy as np
import torch

from ... import _lib


def layer_shapes(n_layers: int, dense_units: int, skip_layer: int, xyz_dim: int, dir_dim: int) -> List[Tuple[str, int, int]]:
    """(name, fan_in, fan_out) in layer-creation order (mlp.py:11-27); the skip concat [h, xyz_enc] follows layer i when
    i % skip_layer == 0 and i > 0 (mlp.py:36-38)."""
    shapes, fan_in = [], xyz_dim
    for i in range(n_layers):
        shapes.append((f"layer_{i}", fan_in, dense_units))
        fan_in = dense_units
        if i % skip_layer == 0 and i > 0:
            fan_in = dense_units + xyz_dim
    shapes += [("sigma", fan_in, 1), ("features", fan_in, dense_units),
               ("rgb_features", dense_units + dir_dim, dense_units // 2), ("rgb", dense_units // 2, 3)]
    return shapes


def _truncated_normal(rng, n, stddev):
    """Keras' truncated normal: N(0, stddev) with draws beyond two standard deviations redrawn"""
    x = rng.standard_normal(n)
    bad = np.abs(x) > 2.0
    while bad.any():
        x[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(x) > 2.0
    return x * stddev


# tf.keras.initializers by name: the VarianceScaling family (scale, mode, distribution) as Keras defines it -- the `initializer`
# argument of the reference's NeRFMLP goes to every Dense layer's kernel_initializer (mlp.py:5, 13-27); biases stay zero.
# "normal" variants are TRUNCATED normals whose stddev is divided by 0.87962566103423978 (the std of a unit normal cut at +-2).
_VARIANCE_SCALING = {"glorot_uniform": (1.0, "avg", "uniform"), "glorot_normal": (1.0, "avg", "normal"),
                     "he_uniform": (2.0, "in", "uniform"), "he_normal": (2.0, "in", "normal"),
                     "lecun_uniform": (1.0, "in", "uniform"), "lecun_normal": (1.0, "in", "normal")}


def init_kernel(initializer, rng, fan_in: int, fan_out: int) -> np.ndarray:
    """one Dense kernel [fan_in, fan_out] flattened, drawn as tf.keras.initializers.get(initializer) would; a callable
    (shape) -> array is accepted as well"""
    n = fan_in * fan_out
    if callable(initializer):
        return np.asarray(initializer((fan_in, fan_out)), np.float32).reshape(-1)
    name = str(initializer).lower()
    if name in _VARIANCE_SCALING:
        scale, mode, dist = _VARIANCE_SCALING[name]
        k, den = (1.0, fan_in) if mode == "in" else (2.0, fan_in + fan_out)          # scale / fan, fan = fan_in or (fan_in + fan_out) / 2
        if dist == "uniform":
            lim = np.sqrt(3.0 * scale * k / den)                                     # glorot_uniform: sqrt(6 / (fan_in + fan_out))
            return rng.uniform(-lim, lim, size=n).astype(np.float32)
        return _truncated_normal(rng, n, np.sqrt(scale * k / den) / 0.87962566103423978).astype(np.float32)
    if name in ("zeros", "ones"):
        return np.full(n, float(name == "ones"), np.float32)
    if name in ("random_normal", "randomnormal"):
        return (rng.standard_normal(n) * 0.05).astype(np.float32)
    if name in ("truncated_normal", "truncatednormal"):
        return _truncated_normal(rng, n, 0.05).astype(np.float32)
    if name in ("random_uniform", "randomuniform"):
        return rng.uniform(-0.05, 0.05, size=n).astype(np.float32)
    raise ValueError(f"initializer {initializer!r}: known names are {sorted(_VARIANCE_SCALING)} + zeros, ones, random_normal, "
                     f"truncated_normal, random_uniform; or pass a callable (shape) -> array")


class NeRFMLP:
    """mlp.py:4-59.  The reference's Dense layers are created WITHOUT an input size (mlp.py:11-27): Keras builds each kernel from
    the last dimension of the first call, so the two input widths belong to the first call, not to the constructor -- the
    reference's own test feeds 99-wide tensors to both arguments (tests/model/nerf/test_nerf_mlp.py:13-26).  Same here: an MLP
    constructed without xyz_dim / dir_dim (extension arguments NeRF uses to fix 3 + 6 L up front) is UNBUILT; it takes its widths
    from the first call (or build(input_shape), set_weights, load_weights) and from then on a call with other widths raises
    ValueError naming both, as Keras does.  build() without a shape gives the reference NeRF's own encodings, 63 / 27
    (nerf.py:116-130 builds its MLPs with dummy inputs of exactly those widths)."""

    def __init__(self, n_layers: int = 8, dense_units: int = 256, skip_layer=4, initializer="glorot_uniform", name=None,
                 xyz_dim: int = None, dir_dim: int = None, seed=None, **kwargs):
        if not callable(initializer):
            init_kernel(initializer, np.random.default_rng(0), 1, 1)          # an unknown name fails here, as in Keras
        self.initializer = initializer
        self.n_layers, self.dense_units, self.skip_layer = n_layers, dense_units, skip_layer
        self.name = name or "nerf_mlp"
        self.xyz_dim = self.dir_dim = None
        self._shapes = None
        if (xyz_dim is None) != (dir_dim is None):
            raise ValueError("xyz_dim and dir_dim are given together or not at all")
        if xyz_dim is not None:
            self._set_widths(xyz_dim, dir_dim)
        self._seed = seed
        self._host = None          # flat fp32 weights until bound to a device context
        self._ctx = None
        self._net = None
        self._own_ctx = None

    # ---- input widths (Keras: fixed by the first call)
    def _set_widths(self, xyz_dim: int, dir_dim: int):
        xyz_dim, dir_dim = int(xyz_dim), int(dir_dim)
        if xyz_dim < 1 or dir_dim < 1:
            raise ValueError(f"{self.name}: input widths must be positive, got {xyz_dim} / {dir_dim}")
        self.xyz_dim, self.dir_dim = xyz_dim, dir_dim
        self._shapes = layer_shapes(self.n_layers, self.dense_units, self.skip_layer, xyz_dim, dir_dim)

    def _check_widths(self, xyz_dim: int, dir_dim: int):
        """Keras' input-compatibility error for a built Dense: expected vs. found last dimension"""
        for what, layer, want, got in (("ray_coordinate_inputs", "layer_0", self.xyz_dim, xyz_dim),
                                       ("direction_inputs", "rgb_features", self.dir_dim, dir_dim)):
            if int(got) != want:
                raise ValueError(f'Input 0 of layer "{layer}" of {self.name} is incompatible with the layer: expected the last axis of '
                                 f'{what} to have {want} features, found {got} (the widths were fixed when the model was built)')

    # ---- weights
    @property
    def built(self) -> bool:
        return self._host is not None or self._ctx is not None

    def _require_widths(self, what: str):
        if self._shapes is None:
            raise ValueError(f"{self.name}.{what}: the model is not built yet -- its input widths come from the first call "
                             f"(or build(input_shape) / set_weights / load_weights)")

    def count_params(self) -> int:
        self._require_widths("count_params")          # Keras: "You tried to call count_params ... but the layer isn't built"
        return sum(i * o + o for _, i, o in self._shapes)

    def build(self, input_shape=None):
        """kernels from `initializer` (default glorot_uniform: U(+-sqrt(6/(fan_in+fan_out)))), zero biases (Keras Dense defaults).
        input_shape: ((..., xyz_dim), (..., dir_dim)) as Keras passes it; None = the widths known so far, else 63 / 27."""
        if input_shape is not None:
            xs, ds = input_shape
            if self._shapes is None:
                self._set_widths(xs[-1], ds[-1])
            else:
                self._check_widths(xs[-1], ds[-1])
        if self.built:
            return
        if self._shapes is None:
            self._set_widths(63, 27)
        rng = np.random.default_rng(self._seed)
        parts = []
        for _, fi, fo in self._shapes:
            parts.append(init_kernel(self.initializer, rng, fi, fo))
            parts.append(np.zeros(fo, np.float32))
        self._host = np.concatenate(parts)

    def _bind(self, ctx, net: int):
        self.build()
        if self._host is not None:
            ctx.set_weights(net, self._host)
        self._ctx, self._net, self._host = ctx, net, None

    def get_flat_weights(self) -> np.ndarray:
        if self._ctx is not None:
            return self._ctx.get_weights(self._net)
        self.build()
        return self._host.copy()

    def set_flat_weights(self, flat):
        flat = np.ascontiguousarray(np.asarray(flat, np.float32).reshape(-1))
        if self._shapes is None:          # unbuilt: a flat vector carries no shapes; only the reference NeRF's widths can be meant
            self._set_widths(63, 27)
        if flat.size != self.count_params():
            raise ValueError(f"{self.name}: expected {self.count_params()} weights, got {flat.size}")
        if self._ctx is not None:
            self._ctx.set_weights(self._net, flat)
        else:
            self._host = flat.copy()

    def get_weights(self) -> List[np.ndarray]:
        """list of 24 arrays, Keras get_weights() order ([] while the model is unbuilt, as in Keras)"""
        if not self.built and self._shapes is None:
            return []
        flat, out, off = self.get_flat_weights(), [], 0
        for _, fi, fo in self._shapes:
            out.append(flat[off:off + fi * fo].reshape(fi, fo)); off += fi * fo
            out.append(flat[off:off + fo]); off += fo
        return out

    def set_weights(self, weights):
        weights = [np.asarray(w, np.float32) for w in weights]
        self._widths_from_kernels(weights[0::2], "set_weights")
        self.set_flat_weights(np.concatenate([w.reshape(-1) for w in weights]))

    def _widths_from_kernels(self, kernels, what: str):
        """an unbuilt model adopts the widths its weights were made for: layer_0/kernel is [xyz_dim, units], rgb_features/kernel
        [units + dir_dim, units / 2] (mlp.py:13-24)"""
        if self._shapes is not None:
            return
        if len(kernels) != self.n_layers + 4 or any(k.ndim != 2 for k in kernels):
            raise ValueError(f"{self.name}.{what}: expected {2 * (self.n_layers + 4)} arrays (kernel, bias per Dense layer), got {2 * len(kernels)}")
        self._set_widths(kernels[0].shape[0], kernels[self.n_layers + 2].shape[0] - self.dense_units)

    @property
    def trainable_variables(self):
        return self.get_weights()

    # ---- checkpoints.  The reference calls tf.keras.Model.save_weights / load_weights on 'coarse.h5' / 'fine.h5'
    # (nerf.py:63-64, 132-136): Keras' HDF5 weight format.  Files named *.h5 / *.hdf5 / *.keras are written and read as
    # REAL HDF5 in that layout (keras_nerf_amd/io/hdf5_min.py: no h5py needed), so a checkpoint trained with the reference
    # loads here and a Keras model can load ours.  Any other name is a NumPy .npz archive keyed by Keras weight names; on
    # load the container is told from the file's magic bytes, so round-1 checkpoints (an .npz named coarse.h5) still load.
    def _layer_names(self):
        return [f"layer_{i}" for i in range(self.n_layers)] + ["sigma", "features", "rgb_features", "rgb"]

    def save_weights(self, path: str, save_format: str = None):
        self._require_widths("save_weights")         # Keras refuses to save a model that has not created its variables
        fmt = save_format or ("h5" if str(path).lower().endswith((".h5", ".hdf5", ".keras")) else "npz")
        if fmt in ("h5", "hdf5"):
            from ...io.hdf5_min import write_keras_weights
            write_keras_weights(path, self.name, self._layer_names(), self.get_weights())
            return
        if fmt != "npz":
            raise ValueError(f"save_format {save_format!r}: expected 'h5' or 'npz'")
        arrs = {}
        for (name, _, _), k, b in zip(self._shapes, self.get_weights()[0::2], self.get_weights()[1::2]):
            arrs[f"{name}/kernel:0"] = k
            arrs[f"{name}/bias:0"] = b
        with open(path, "wb") as f:
            np.savez(f, **arrs)

    def load_weights(self, path: str):
        from ...io.hdf5_min import is_hdf5, read_keras_weights
        if is_hdf5(path):
            raw = read_keras_weights(path, self._layer_names())
            pairs = list(zip(raw[0::2], raw[1::2]))
        else:
            with open(path, "rb") as f:
                magic = f.read(4)
            if magic[:2] != b"PK":
                raise ValueError(f"{path}: neither an HDF5 file (Keras save_weights) nor a NumPy .npz archive")
            z = np.load(path)
            pairs = [(z[f"{name}/kernel:0"], z[f"{name}/bias:0"]) for name in self._layer_names()]
        self._widths_from_kernels([np.asarray(k) for k, _ in pairs], "load_weights")
        ws = []
        for (name, fi, fo), (k, b) in zip(self._shapes, pairs):
            if k.shape != (fi, fo) or b.shape != (fo,):
                raise ValueError(f"{path}: {name} has shape {k.shape} / {b.shape}, expected {(fi, fo)} / {(fo,)}")
            ws += [k, b]
        self.set_weights(ws)

    # ---- forward (rgb, sigma) = mlp((xyz_enc, dir_enc)): mlp.py:29-50
    def __call__(self, inputs):
        return self.call(inputs)

    def call(self, inputs):
        """Evaluates the MLP on already encoded inputs [..., xyz_dim] / [..., dir_dim] (mlp.py:29-50).  The reference's
        NeRF never calls its MLPs this way outside _build_model (weight creation) and a shape test; the fused kernels
        start from ray origins/directions instead, so this entry point runs on the general-shape HIP kernels
        (knerf_mlp_call: bf16 matmul operands, fp32 accumulate/bias/activation) for every shape.  The first call of an unbuilt
        model fixes its two input widths (any positive integers: Keras Dense builds from the last dimension it sees); later
        calls with other widths raise ValueError."""
        xyz, dire = inputs
        if not torch.cuda.is_available():
            from ...runtime import KnerfError
            raise KnerfError("keras_nerf_amd needs an MI355X (gfx950) GPU; there is no CPU path")
        xs, ds = tuple(np.shape(xyz) if not isinstance(xyz, torch.Tensor) else xyz.shape), \
            tuple(np.shape(dire) if not isinstance(dire, torch.Tensor) else dire.shape)
        if len(xs) < 1 or len(ds) < 1 or xs[:-1] != ds[:-1]:
            raise ValueError(f"{self.name}: inputs must share their leading dimensions, got {xs} and {ds}")
        if self._shapes is None:
            self._set_widths(xs[-1], ds[-1])
        else:
            self._check_widths(xs[-1], ds[-1])
        ctx, net = self._ctx, self._net
        if ctx is None:                      # stand-alone MLP (reference test_nerf_mlp.py): a private context of this shape
            if self._own_ctx is None:
                from ...runtime import KnerfContext
                self._own_ctx = KnerfContext(n_layers=self.n_layers, dense_units=self.dense_units, skip_layer=self.skip_layer,
                                             encoded_widths=(self.xyz_dim, self.dir_dim))
            ctx, net = self._own_ctx, 0
            ctx.set_weights(0, self.get_flat_weights())
        xyz = ctx.f32(xyz); dire = ctx.f32(dire)
        lead = tuple(xyz.shape[:-1])
        x2, d2 = xyz.reshape(-1, self.xyz_dim), dire.reshape(-1, self.dir_dim)
        n = x2.shape[0]
        if n <= self.CALL_ROWS:
            raw = ctx.mlp_call(net, x2, d2)
        else:           # bound the library's per-call workspace (every layer's activations of the rows in flight: ~5 KB per row at width 256)
            raw = torch.empty((n, 4), device=ctx.device, dtype=torch.float32)
            for r0 in range(0, n, self.CALL_ROWS):
                raw[r0:r0 + self.CALL_ROWS] = ctx.mlp_call(net, x2[r0:r0 + self.CALL_ROWS], d2[r0:r0 + self.CALL_ROWS])
        return raw[:, :3].reshape(lead + (3,)), raw[:, 3:4].reshape(lead + (1,))

    CALL_ROWS = 1 << 20      # rows per knerf_mlp_call: the reference test's 640,000 go in one call, a [2, 400, 400, 192, .] tensor in 59

    def get_config(self):
        return {"name": self.name, "n_layers": self.n_layers, "dense_units": self.dense_units, "skip_layer": self.skip_layer}

    def summary(self, print_fn=print):
        print_fn(f'Model: "{self.name}"')
        self._require_widths("summary")             # Keras: "This model has not yet been built"
        for name, fi, fo in self._shapes:
            print_fn(f"  {name:<14} Dense  in={fi:<4} out={fo:<4} params={fi * fo + fo}")
        print_fn(f"Total params: {self.count_params():,}")
