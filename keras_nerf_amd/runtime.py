"""Thin object wrapper over the C ABI (include/knerf.h): torch is used for device memory and streams only."""
from __future__ import annotations

import ctypes as C
import logging
import os
from typing import Optional

import numpy as np
import torch

from . import _lib
from ._lib import COARSE, FINE, KnerfConfig, KnerfError  # noqa: F401


class NonFiniteGradientError(ArithmeticError):
    """Counterpart of the InvalidArgumentError raised by tf.debugging.assert_all_finite (reference nerf.py:381-382)."""


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _f32(x, device) -> torch.Tensor:
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(np.asarray(x, dtype=np.float32))
    return x.to(device=device, dtype=torch.float32).contiguous()


_GENERIC_SHAPES_WARNED = set()      # shapes already reported as "coverable by build.py --add-shape" (one warning each per process)

FUSED_WIDTHS = (64, 128, 256)       # csrc/layout.h: the trunk widths the fused kernels are written for


def padded_width(dense_units: int):
    """the width a dense_units is run at, or None (as it is).  Below 256: the next FUSED width -- 50 -> 64, 96 -> 128, 191 -> 256 (odd
    widths too: the reference's rgb_features layer has dense_units // 2 outputs, mlp.py:25, and so has the map).  Above 256 (general-shape
    kernels): the next multiple of 128 -- their GEMMs tile the output in blocks of 8, 4, 2 or 1 thirty-two-column tiles, whichever
    divides it, and every block re-reads the layer's input: 352 (11 tiles: eleven blocks) costs 24.9 ms per train chunk, 384 (three
    blocks of four) 13.1; 416 29.8 against 512's 18.0."""
    if dense_units < 2 or dense_units in FUSED_WIDTHS:
        return None
    if dense_units < FUSED_WIDTHS[-1]:
        return next(w for w in FUSED_WIDTHS if w >= dense_units)
    wide = -(-dense_units // 128) * 128
    return None if wide == dense_units else wide


def width_pad_index(n_layers: int, units: int, padded: int, skip_layer: int, xyz_dim: int, dir_dim: int) -> np.ndarray:
    """int64 [n_params(units)]: where parameter i of the REAL network (flat Keras order, mlp.py:11-27) sits in the flat parameters
    of the same network at width `padded`.  A kernel's h rows / h columns keep their index; the rows behind them ([h ; xyz_enc] of a
    concat layer, [features ; dir_enc] of rgb_features) move behind the padded h block."""
    from .model.nerf.mlp import layer_shapes
    real = layer_shapes(n_layers, units, skip_layer, xyz_dim, dir_dim)
    pad = layer_shapes(n_layers, padded, skip_layer, xyz_dim, dir_dim)
    idx, off = [], 0
    for (name, fi, fo), (_, fip, fop) in zip(real, pad):
        h_rows = {"layer_0": 0, "rgb": units // 2}.get(name, units)                 # leading rows that are h features of width `units`
        h_rows_p = {"layer_0": 0, "rgb": padded // 2}.get(name, padded)
        rows = np.arange(fi)
        rows_p = np.where(rows < h_rows, rows, rows - h_rows + h_rows_p)
        idx.append((off + rows_p[:, None] * fop + np.arange(fo)[None, :]).reshape(-1))
        off += fip * fop
        idx.append(off + np.arange(fo))
        off += fop
    return np.concatenate(idx).astype(np.int64)



class _CudaView:
    def __init__(self, ptr: int, n: int, typestr: str):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


class KnerfContext:
    """Owns one knerf_ctx on the current CUDA(HIP) device."""

    def __init__(self, n_coarse=64, n_fine=128, pos_emb_xyz=10, pos_emb_dir=4, n_layers=8, dense_units=256, skip_layer=4,
                 white_background=False, oob="zero", lr=1e-3, beta1=0.9, beta2=0.999, epsilon=1e-7, device=None,
                 force_generic=None, options=None, encoded_widths=None, auto_build=None, pad_width=None):
        """force_generic: run the default MLP shape through the general-shape kernels as well (tests).  options: {name: value}
        for knerf_set_option.  encoded_widths = (xyz_dim, dir_dim): a stand-alone NeRFMLP of those two input widths
        (KNERF_FLAG_ENCODED_WIDTHS: weights and mlp_call only; pos_emb_* are ignored).  auto_build (default: $KNERF_AUTO_BUILD): a
        shape the fused kernels COULD cover but the loaded library does not hold is compiled for them on first use (hipcc, a few
        minutes once per shape; the build is kept in keras_nerf_amd/build_auto_*/) instead of running on the general-shape kernels.
        pad_width (default on; $KNERF_NO_WIDTH_PAD=1 turns it off): a dense_units below 256 that is not 64 / 128 / 256 runs on the
        FUSED kernels of the next of those widths with zero-padded weights -- an exact identity (padded neurons have zero kernel and
        bias: their activations, every gradient that touches them and hence their Adam updates are exactly zero; `_set_up_padding`) --
        whenever the library holds (or auto_build builds) that shape; set_weights / get_weights / grads() speak the REAL layout,
        weights_view / grads_view expose the padded buffers (what a data-parallel broadcast / all-reduce needs).  The LIBRARY reads no environment variables; for tools and sweeps this wrapper translates
        KNERF_FORCE_GENERIC, KNERF_WGRAD_GROUP_MAX, KNERF_WGRAD_GROUP_GB, KNERF_WGRAD_COSTS ("c0,...,c<n_layers>": one per weight-gradient job), KNERF_DETERMINISTIC and
        KNERF_SKIP_DEAD_TILES, KNERF_MERGE_CHUNK_RAYS, KNERF_MERGE_RENDER_RAYS into the config flag / options below (explicit arguments win)."""
        self._ctx = C.c_void_p()
        if not torch.cuda.is_available():
            raise KnerfError("keras_nerf_amd needs an MI355X (gfx950) GPU; there is no CPU path")
        self.lib = _lib.load()
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        torch.cuda.set_device(self.device)
        if oob not in ("zero", "clamp"):
            raise ValueError("oob must be 'zero' or 'clamp'")
        if force_generic is None:
            force_generic = bool(os.environ.get("KNERF_FORCE_GENERIC"))
        flags = _lib.FLAG_FORCE_GENERIC if force_generic else 0
        if encoded_widths is not None:
            pos_emb_xyz, pos_emb_dir = (int(v) for v in encoded_widths)
            flags |= _lib.FLAG_ENCODED_WIDTHS
        self.cfg = KnerfConfig(n_coarse, n_fine, pos_emb_xyz, pos_emb_dir, n_layers, dense_units, skip_layer,
                               int(bool(white_background)), int(oob == "clamp"), lr, beta1, beta2, epsilon, flags)
        self.n_coarse, self.n_fine = n_coarse, n_fine
        self._pad_index = None          # torch int64 [real params] -> position in the padded flat parameters (width padding), or None
        if pad_width is None:
            pad_width = os.environ.get("KNERF_NO_WIDTH_PAD", "") in ("", "0")
        wide = padded_width(dense_units) if (pad_width and not force_generic and encoded_widths is None) else None
        # ... and where the library lacks this (n_layers, skip_layer) pair at that fused width but holds it one width up, that one: the
        # fused chain at 256 (3.4 ms per 8-layer chunk) still beats the general-shape kernels at 128 (3.8); two widths up it does not
        for cand in ([wide] + [w for w in FUSED_WIDTHS if wide < w <= 2 * wide][:1] if wide is not None else []):
            if self._set_up_padding(cand, auto_build if cand == wide else False):
                dense_units = cand                                      # from here on the context IS the padded shape
                break
        rc = 0 if self._ctx.value else self.lib.knerf_create(C.byref(self.cfg), C.byref(self._ctx))
        if rc != 0:
            msg = self.lib.knerf_last_error(None).decode()
            self._ctx = C.c_void_p()
            raise (ValueError if rc == _lib.KNERF_ERR_INVALID else KnerfError)(msg)
        if self._pad_index is None:
            self.param_count = int(self.lib.knerf_param_count_for(C.byref(self.cfg)))
        if not force_generic and encoded_widths is None and self.get_option("general_shape_path"):
            # a shape outside the library's list: where the fused kernels could cover it, build them (opt-in) or say how; else say what it costs
            eff = (padded_width(dense_units) if pad_width else None) or dense_units       # the width the fused kernels would run it at
            coverable = (eff in FUSED_WIDTHS and 3 <= n_layers <= 16 and skip_layer >= 1
                         and 1 <= pos_emb_xyz <= 16 and 1 <= pos_emb_dir <= 8 and not (eff == 256 and pos_emb_xyz == 16 and pos_emb_dir >= 5))
            spec = f"{n_layers},{skip_layer},{eff}" + ("" if (pos_emb_xyz, pos_emb_dir) == (10, 4) else f",{pos_emb_xyz},{pos_emb_dir}")
            if auto_build is None:
                auto_build = os.environ.get("KNERF_AUTO_BUILD", "") not in ("", "0")
            if coverable and auto_build and eff == dense_units:          # (a padded width was tried, and built if allowed, in _set_up_padding)
                self._rebuild_for(spec)
        if not force_generic and encoded_widths is None and self.get_option("general_shape_path"):
            # visible at the default log level when there is something the user can do about it, once per shape
            say = logging.info
            if coverable and spec not in _GENERIC_SHAPES_WARNED:
                _GENERIC_SHAPES_WARNED.add(spec)
                say = logging.warning
            say("NeRFMLP(n_layers=%d, dense_units=%d, skip_layer=%d), pos_emb %d/%d runs on the general-shape kernels (about 2x slower "
                "than the fused chain)%s", n_layers, dense_units, skip_layer, pos_emb_xyz, pos_emb_dir,
                f"; `python keras_nerf_amd/build.py --add-shape={spec}` builds the fused kernels for it (KNERF_AUTO_BUILD=1 does that on first use)" if coverable else "")
        opts = {}
        env = os.environ
        for key, name in (("KNERF_WGRAD_GROUP_MAX", "wgrad_group_max"), ("KNERF_WGRAD_GROUP_GB", "wgrad_group_gb"),
                          ("KNERF_DETERMINISTIC", "deterministic"), ("KNERF_SKIP_DEAD_TILES", "skip_dead_tiles"),
                          ("KNERF_MERGE_CHUNK_RAYS", "merge_chunk_rays"), ("KNERF_MERGE_RENDER_RAYS", "merge_render_rays")):
            if env.get(key) not in (None, ""):
                opts[name] = float(env[key])
        env_costs = {}
        if env.get("KNERF_WGRAD_COSTS"):          # one entry per weight-gradient job (n_layers + 1 of them); surplus entries are ignored
            for j, v in enumerate(env["KNERF_WGRAD_COSTS"].split(",")[:n_layers + 1]):
                env_costs[f"wgrad_cost{j}"] = float(v)
        opts.update(options or {})
        for k, v in opts.items():
            self.set_option(k, v)
        for k, v in env_costs.items():
            if k not in opts:
                try:
                    self.set_option(k, v)
                except ValueError:           # the context has fewer jobs than n_layers + 1 (general-shape path)
                    break

    def _set_up_padding(self, wide: int, auto_build) -> bool:
        """Try the network at width `wide` (zero-padded) on the fused kernels.  True: self._ctx is a fused context of the padded shape,
        self.cfg describes it, self.param_count / self._pad_index the real one.  False: nothing changed (the real shape then runs on
        the general-shape kernels as before).

        Why the padding is exact: a padded neuron j has a zero kernel column and zero bias, so z_j = 0 and h_j = relu(0) = 0 (a
        linear head layer: 0); the next layer's kernel rows for j are zero, so nothing reads it.  Backward: dz_j = [h_j > 0] (...) = 0
        at a relu, and W_next[j, :] . dz_next = 0 at a linear layer, so dW[:, j] = x^T dz_j = 0, db_j = 0 and dW_next[j, :] = h_j^T
        dz_next = 0: Adam's m and v stay 0 and its update 0 / (0 + eps) = 0 -- the zeros stay zeros, bit for bit, and the sums over the
        real neurons only ever gain exact zeros."""
        c = self.cfg
        cfg_p = KnerfConfig(c.n_coarse, c.n_fine, c.pos_emb_xyz, c.pos_emb_dir, c.n_layers, wide, c.skip_layer, c.white_background,
                            c.oob_clamp, c.lr, c.beta1, c.beta2, c.epsilon, c.flags)
        ctx = C.c_void_p()
        if self.lib.knerf_create(C.byref(cfg_p), C.byref(ctx)) != 0:
            return False
        real_cfg, real_ctx = self.cfg, self._ctx
        self.cfg, self._ctx = cfg_p, ctx
        generic = bool(self.get_option("general_shape_path")) and wide <= FUSED_WIDTHS[-1]      # (above 256 the general-shape kernels ARE the target)
        if generic:                                        # the padded shape is not in the library either: build it when allowed
            if auto_build is None:
                auto_build = os.environ.get("KNERF_AUTO_BUILD", "") not in ("", "0")
            coverable = (3 <= c.n_layers <= 16 and c.skip_layer >= 1 and 1 <= c.pos_emb_xyz <= 16
                         and 1 <= c.pos_emb_dir <= 8 and not (wide == 256 and c.pos_emb_xyz == 16 and c.pos_emb_dir >= 5))
            if coverable and auto_build:
                self._rebuild_for(f"{c.n_layers},{c.skip_layer},{wide}" + ("" if (c.pos_emb_xyz, c.pos_emb_dir) == (10, 4) else f",{c.pos_emb_xyz},{c.pos_emb_dir}"))
                generic = bool(self.get_option("general_shape_path"))
        if generic:
            self.lib.knerf_destroy(self._ctx)
            self.cfg, self._ctx = real_cfg, real_ctx
            return False
        self.param_count = int(self.lib.knerf_param_count_for(C.byref(real_cfg)))
        self.padded_param_count = int(self.lib.knerf_param_count_for(C.byref(cfg_p)))
        self.real_dense_units = int(real_cfg.dense_units)
        idx = width_pad_index(c.n_layers, int(real_cfg.dense_units), wide, c.skip_layer, 3 + 6 * c.pos_emb_xyz, 3 + 6 * c.pos_emb_dir)
        assert idx.size == self.param_count and int(idx.max()) < self.padded_param_count
        self._pad_index_host = idx
        self._pad_index = torch.as_tensor(idx, device=self.device)
        logging.info("dense_units=%d runs at width %d with zero-padded weights (exact) on the %s kernels", real_cfg.dense_units, wide,
                     "fused" if wide <= FUSED_WIDTHS[-1] else "general-shape")
        return True

    def _rebuild_for(self, spec: str):
        """KNERF_AUTO_BUILD: compile the fused kernels for this shape into their own library (build.py --variant=auto_<shape>
        --add-shape=<shape>; kept on disk, so the next process finds it built), load it beside the product library and re-create
        the context on it.  One process builds at a time (several ranks start together): a file lock around the build."""
        import fcntl
        from . import build as B
        variant = "auto_" + spec.replace(",", "_")
        lock_path = os.path.join(os.path.dirname(os.path.abspath(B.__file__)), f".build_{variant}.lock")
        import torch.distributed as dist
        built = os.path.join(os.path.dirname(os.path.abspath(B.__file__)), f"libknerf_hip_{variant}.so")
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            # N ranks behind one file lock, two minutes of hipcc each turn, inside the process group's time-outs: not from a rank of a
            # running job.  A library built earlier (one process, before the job) is loaded as it is; otherwise the general-shape path.
            if not os.path.exists(built):
                logging.warning("KNERF_AUTO_BUILD: not compiling shape %s from rank %d of a %d-rank job (it runs on the general-shape kernels); "
                                "build it once beforehand: `python keras_nerf_amd/build.py --variant=%s --add-shape=%s`",
                                spec, dist.get_rank(), dist.get_world_size(), variant, spec)
                return
            path = built
            return self._adopt_library(path, spec)
        logging.warning("KNERF_AUTO_BUILD: building the fused kernels for shape %s (hipcc, a few minutes the first time)", spec)
        try:
            with open(lock_path, "w") as lock:
                fcntl.flock(lock, fcntl.LOCK_EX)
                try:
                    path = B.build(verbose=False, variant=variant, add_shapes=[spec])
                finally:
                    fcntl.flock(lock, fcntl.LOCK_UN)
        except Exception as e:                     # noqa: BLE001 -- no hipcc on this machine, a read-only tree, a shape that spills ...
            logging.warning("KNERF_AUTO_BUILD: building shape %s failed (%s: %s); it runs on the general-shape kernels", spec, type(e).__name__, e)
            return
        self._adopt_library(path, spec)

    def _adopt_library(self, path: str, spec: str):
        """re-create this context on the library at `path` (a build that holds shape `spec` for the fused kernels)"""
        self.lib.knerf_destroy(self._ctx)
        self._ctx = C.c_void_p()
        self.lib = _lib.load_path(path)
        rc = self.lib.knerf_create(C.byref(self.cfg), C.byref(self._ctx))
        if rc != 0:
            msg = self.lib.knerf_last_error(None).decode()
            self._ctx = C.c_void_p()
            raise KnerfError(f"KNERF_AUTO_BUILD: {path} refused the context: {msg}")
        if self.get_option("general_shape_path"):
            raise KnerfError(f"KNERF_AUTO_BUILD: {path} does not hold shape {spec}")

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self.lib.knerf_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- run-time options (include/knerf.h knerf_set_option)
    def set_option(self, name: str, value) -> None:
        self._check(self.lib.knerf_set_option(self._ctx, name.encode(), float(value)))

    def get_option(self, name: str) -> float:
        v = C.c_double()
        self._check(self.lib.knerf_get_option(self._ctx, name.encode(), C.byref(v)))
        return v.value

    def tile_stats(self, reset: bool = True):
        """(live, total) 32-sample tiles seen by the dgrad launches since the last reset (skip_dead_tiles on)"""
        a, b = C.c_int64(), C.c_int64()
        self._check(self.lib.knerf_tile_stats(self._ctx, self._stream(), C.byref(a), C.byref(b), int(reset)))
        return a.value, b.value

    def tile_stats_net(self, reset: bool = True):
        """((live, total) of the coarse passes, (live, total) of the fine passes) since the last reset"""
        a, b = (C.c_int64 * 2)(), (C.c_int64 * 2)()
        self._check(self.lib.knerf_tile_stats_net(self._ctx, self._stream(), a, b, int(reset)))
        return (a[0], b[0]), (a[1], b[1])

    def grad_diagnostics(self, wait: bool = True):
        """(non-zero entries of the last chunk's coarse gradient, of its fine gradient, steps published so far): the reference's
        tf.math.count_nonzero check (nerf.py:430-451); needs the option grad_diagnostics"""
        out = (C.c_int64 * 3)()
        self._check(self.lib.knerf_grad_diagnostics(self._ctx, self._stream(), int(wait), out))
        return int(out[0]), int(out[1]), int(out[2])

    # ---- helpers
    def _check(self, rc: int):
        if rc == 0:
            return
        msg = self.lib.knerf_last_error(self._ctx).decode()
        if rc == _lib.KNERF_ERR_NONFINITE:
            raise NonFiniteGradientError(msg)
        if rc == _lib.KNERF_ERR_INVALID:
            raise ValueError(msg)
        raise KnerfError(f"knerf error {rc}: {msg}")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def f32(self, x) -> torch.Tensor:
        return _f32(x, self.device)

    # ---- weights
    def set_weights(self, net: int, flat):
        flat = np.ascontiguousarray(np.asarray(flat, dtype=np.float32).reshape(-1))
        if self._pad_index is not None:                # width padding: the real parameters into a zero vector of the padded layout
            if flat.size != self.param_count:
                raise ValueError(f"set_weights: {flat.size} values for a network of {self.param_count} parameters")
            wide = np.zeros(self.padded_param_count, np.float32)
            wide[self._pad_index_host] = flat
            flat = wide
        self._check(self.lib.knerf_set_weights(self._ctx, net, flat.ctypes.data_as(C.POINTER(C.c_float)), flat.size))

    def get_weights(self, net: int) -> np.ndarray:
        n = self.param_count if self._pad_index is None else self.padded_param_count
        out = np.empty(n, np.float32)
        self._check(self.lib.knerf_get_weights(self._ctx, net, out.ctypes.data_as(C.POINTER(C.c_float)), out.size))
        return out if self._pad_index is None else out[self._pad_index_host]

    def grads(self, net: int) -> torch.Tensor:
        """the accumulated gradient of one net in the REAL flat layout (a copy): grads_view()'s half, un-padded where the width is padded"""
        g = self.grads_view()
        n = g.numel() // 2
        half = g[net * n:(net + 1) * n]
        return half.clone() if self._pad_index is None else half[self._pad_index]

    def weights_view(self, net: int) -> torch.Tensor:
        """torch view (no copy) of the library-owned fp32 master weights of one net"""
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self.lib.knerf_weights_device(self._ctx, net, C.byref(p), C.byref(n)))
        return torch.as_tensor(_CudaView(p.value, n.value, "<f4"), device=self.device)

    def grads_view(self) -> torch.Tensor:
        """torch view (no copy) of the gradient accumulators [coarse | fine]: the buffer a DP step all-reduces"""
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self.lib.knerf_grads_device(self._ctx, C.byref(p), C.byref(n)))
        return torch.as_tensor(_CudaView(p.value, n.value, "<f4"), device=self.device)

    def refresh_weights(self):
        self._check(self.lib.knerf_refresh_weights(self._ctx, self._stream()))

    # ---- forward
    def forward_chunk(self, net: int, o, d, t):
        o, d, t = self.f32(o), self.f32(d), self.f32(t)
        R, S = t.shape
        image = torch.empty((R, 3), device=self.device); depth = torch.empty((R,), device=self.device)
        weights = torch.empty((R, S), device=self.device)
        self._check(self.lib.knerf_forward_chunk(self._ctx, self._stream(), net, _ptr(o), _ptr(d), _ptr(t), R, S,
                                                 _ptr(image), _ptr(depth), _ptr(weights)))
        return image, depth, weights

    def sample_fine(self, t_coarse, w_coarse, u=None, seed=0, stream_id=0, ray_offset=0):
        t_coarse, w_coarse = self.f32(t_coarse), self.f32(w_coarse)
        u = None if u is None else self.f32(u)
        R = t_coarse.shape[0]
        out = torch.empty((R, self.n_coarse + self.n_fine), device=self.device)
        self._check(self.lib.knerf_sample_fine(self._ctx, self._stream(), _ptr(t_coarse), _ptr(w_coarse), _ptr(u), seed,
                                               stream_id, ray_offset, R, _ptr(out)))
        return out

    def render_chunk(self, o, d, t, u=None, seed=0, ray_offset=0, out=None):
        """out: optional dict of preallocated [R,...] tensors (c_image, c_depth, c_weights, f_image, f_depth, f_weights,
        t_fine); only the two images are mandatory"""
        o, d, t = self.f32(o), self.f32(d), self.f32(t)
        u = None if u is None else self.f32(u)
        R = t.shape[0]
        Na = self.n_coarse + self.n_fine
        if out is None:
            def e(*s):
                return torch.empty(s, device=self.device)
            out = dict(c_image=e(R, 3), c_depth=e(R), c_weights=e(R, self.n_coarse), f_image=e(R, 3), f_depth=e(R),
                       f_weights=e(R, Na), t_fine=e(R, Na))
        self._check(self.lib.knerf_render_chunk(self._ctx, self._stream(), _ptr(o), _ptr(d), _ptr(t), _ptr(u), seed,
                                                ray_offset, R, _ptr(out["c_image"]), _ptr(out.get("c_depth")),
                                                _ptr(out.get("c_weights")), _ptr(out["f_image"]), _ptr(out.get("f_depth")),
                                                _ptr(out.get("f_weights")), _ptr(out.get("t_fine"))))
        return out

    def render_batch(self, o, d, t, u=None, seed=0, ray_chunks=None, out=None):
        """the whole chunk loop of predict_and_render_images in one host call; `out` as in render_chunk but [N,...]
        (an optional out["t_fine"] [N, n_coarse+n_fine] receives the merged t-values)"""
        o, d, t = self.f32(o), self.f32(d), self.f32(t)
        u = None if u is None else self.f32(u)
        N = t.shape[0]
        Na = self.n_coarse + self.n_fine
        if out is None:
            def e(*s):
                return torch.empty(s, device=self.device)
            out = dict(c_image=e(N, 3), c_depth=e(N), c_weights=e(N, self.n_coarse), f_image=e(N, 3), f_depth=e(N), f_weights=e(N, Na))
        self._check(self.lib.knerf_render_batch(self._ctx, self._stream(), _ptr(o), _ptr(d), _ptr(t), _ptr(u), seed, N,
                                                int(ray_chunks or N), _ptr(out["c_image"]), _ptr(out.get("c_depth")),
                                                _ptr(out.get("c_weights")), _ptr(out["f_image"]), _ptr(out.get("f_depth")),
                                                _ptr(out.get("f_weights")), _ptr(out.get("t_fine"))))
        return out

    # ---- training
    def train_chunk(self, o, d, t, target, u=None, seed=0, ray_offset=0, inv_chunks=1.0, loss=None, c_image=None,
                    f_image=None):
        o, d, t, target = self.f32(o), self.f32(d), self.f32(t), self.f32(target)
        u = None if u is None else self.f32(u)
        self._check(self.lib.knerf_train_chunk(self._ctx, self._stream(), _ptr(o), _ptr(d), _ptr(t), _ptr(target), _ptr(u),
                                               seed, ray_offset, t.shape[0], float(inv_chunks), _ptr(loss), _ptr(c_image),
                                               _ptr(f_image)))

    def train_batch(self, o, d, t, target, u=None, seed=0, ray_chunks=None, loss=None, c_image=None, f_image=None):
        """the whole chunk loop of one train step (all arrays [N,...]; N % ray_chunks == 0)"""
        o, d, t, target = self.f32(o), self.f32(d), self.f32(t), self.f32(target)
        u = None if u is None else self.f32(u)
        n = t.shape[0]
        self._check(self.lib.knerf_train_batch(self._ctx, self._stream(), _ptr(o), _ptr(d), _ptr(t), _ptr(target), _ptr(u), seed, n,
                                               int(ray_chunks or n), _ptr(loss), _ptr(c_image), _ptr(f_image)))

    def apply_adam(self, check: bool = True):
        """finite check + 2x Keras-form Adam + accumulator reset, enqueued without waiting for the GPU.  check=True then
        waits and raises NonFiniteGradientError if the step was skipped (the reference's assert_all_finite, nerf.py:381-382);
        check=False leaves that to a later poll_nonfinite() at a point where the caller synchronises anyway."""
        self._check(self.lib.knerf_apply_adam(self._ctx, self._stream()))
        if check:
            self.poll_nonfinite(wait=True)

    def poll_nonfinite(self, wait: bool = False):
        self._check(self.lib.knerf_poll_nonfinite(self._ctx, self._stream(), int(wait)))

    def mlp_call(self, net: int, xyz_enc, dir_enc) -> torch.Tensor:
        """NeRFMLP.__call__ on already encoded inputs [n, xyz_dim] / [n, dir_dim]: raw [n,4] = (rgb, sigma)"""
        x, dd = self.f32(xyz_enc).contiguous(), self.f32(dir_enc).contiguous()
        n = x.shape[0]
        raw = torch.empty((n, 4), device=self.device, dtype=torch.float32)
        self._check(self.lib.knerf_mlp_call(self._ctx, self._stream(), int(net), _ptr(x), _ptr(dd), n, _ptr(raw)))
        return raw

    def zero_grads(self):
        self._check(self.lib.knerf_zero_grads(self._ctx, self._stream()))

    @property
    def step(self) -> int:
        """optimizer steps applied so far (steps skipped for a non-finite gradient leave it once they have been polled)"""
        return int(self.lib.knerf_step_count(self._ctx))

    @step.setter
    def step(self, v: int):
        self._check(self.lib.knerf_set_step_count(self._ctx, int(v)))

    def generate_rays(self, c2w, focal, height, width, near, far, n_samples, noise=None, seed=0, stream_id=0):
        c2w = self.f32(c2w).reshape(-1, 4, 4)
        B = c2w.shape[0]
        noise = None if noise is None else self.f32(noise)
        o = torch.empty((B, height, width, 3), device=self.device); d = torch.empty_like(o)
        t = torch.empty((B, height, width, n_samples), device=self.device)
        self._check(self.lib.knerf_generate_rays(self._ctx, self._stream(), _ptr(c2w), _ptr(noise), seed, stream_id, B, height,
                                                 width, n_samples, float(focal), float(near), float(far), _ptr(o), _ptr(d),
                                                 _ptr(t)))
        return o, d, t

    PROFILE_CLASSES = ("mlp_fwd_coarse", "mlp_fwd_fine", "composite", "sample_fine", "mlp_bwd_coarse", "mlp_bwd_fine",
                       "wgrad_coarse", "wgrad_fine", "adam_repack")

    def profile_enable(self, on: bool = True):
        self._check(self.lib.knerf_profile_enable(self._ctx, int(on)))

    def profile_read(self):
        """{class: (total_ms, launches)} since the last read"""
        n = len(self.PROFILE_CLASSES)
        ms = (C.c_double * n)(); cnt = (C.c_int64 * n)()
        self._check(self.lib.knerf_profile_read(self._ctx, ms, cnt, n))
        return {k: (ms[i], int(cnt[i])) for i, k in enumerate(self.PROFILE_CLASSES)}
