"""NeRF -- counterpart of reference keras_nerf/model/nerf/nerf.py:10-508 on the MI355X hot path.

Same constructor / compile / fit / train_step / test_step / predict_and_render_* / save_model / load_model surface and
the same chunk semantics (equal ray chunks, mean of chunk losses, one optimizer step per batch, divisibility assert);
the per-chunk body runs in hand-written HIP behind the C ABI (include/knerf.h).  Data parallelism is one process per GPU:
when torch.distributed is initialised, the accumulated gradients of both MLPs are all-reduced (SUM, as Keras optimizers
aggregate under MirroredStrategy -- reference train.py:75, nerf.py:455-458) over RCCL before the two Adam updates.
"""
from __future__ import annotations

import json
import logging
import os
from typing import Dict, Optional

import numpy as np
import torch

from ... import parallel
from ...runtime import COARSE, FINE, KnerfContext, NonFiniteGradientError  # noqa: F401
from .metrics import Mean, MetricLogs, MetricState
from .mlp import NeRFMLP
from .utils import NeRFUtils


def _is_mse(loss) -> bool:
    """The fused kernels implement the reference's loss, mean squared error over [R,3] (train_single.py:127,
    train.py:130-136).  Strings / None are accepted by name; a callable is probed numerically."""
    if loss is None or (isinstance(loss, str) and loss.lower() in ("mse", "mean_squared_error", "meansquarederror")):
        return True
    if callable(loss):
        a = torch.linspace(0, 1, 24).reshape(8, 3); b = torch.flip(a, [0]) * 0.5
        try:
            v = float(torch.as_tensor(loss(a, b)))
        except Exception:
            return False
        return abs(v - float(torch.mean((a - b) ** 2))) < 1e-6
    return False


def _adam_hyper(optimizer) -> Dict[str, float]:
    """tf.keras.optimizers.get('adam') defaults (nerf.py:163-165); a dict or an object with the Keras attribute names
    overrides them.  Only PLAIN Adam with a constant learning rate is implemented (what the reference trains with): any
    other optimizer, amsgrad, weight decay, gradient clipping, EMA or a learning-rate schedule is refused, not ignored."""
    h = dict(lr=1e-3, beta1=0.9, beta2=0.999, epsilon=1e-7)
    if optimizer is None or (isinstance(optimizer, str) and optimizer.lower() == "adam"):
        return h
    if isinstance(optimizer, str):
        raise ValueError(f"optimizer '{optimizer}': only Adam is implemented (the reference trains with 'adam')")
    cfg = {}
    if isinstance(optimizer, dict):
        cfg = dict(optimizer.get("config", optimizer))          # Keras serialised form {"class_name", "config"} or a flat dict
        name = str(optimizer.get("class_name", cfg.get("name", "adam")))
    else:
        name = type(optimizer).__name__
        if callable(getattr(optimizer, "get_config", None)):
            try:
                cfg = dict(optimizer.get_config())
            except Exception:
                cfg = {}
        for k in ("learning_rate", "lr", "beta_1", "beta_2", "epsilon", "amsgrad", "weight_decay", "clipnorm", "clipvalue",
                  "global_clipnorm", "use_ema"):
            if k not in cfg and getattr(optimizer, k, None) is not None:
                cfg[k] = getattr(optimizer, k)
        name = str(cfg.get("name", name))
    if name.lower() not in ("adam", "dict", "simplenamespace", "namespace"):
        raise ValueError(f"optimizer {name}: only plain Adam is implemented (the reference trains with 'adam')")
    for k in ("amsgrad", "weight_decay", "clipnorm", "clipvalue", "global_clipnorm", "use_ema"):
        if cfg.get(k) not in (None, False, 0, 0.0):
            raise ValueError(f"Adam option {k}={cfg[k]!r} is not implemented by the fused optimizer kernel")
    for src, dst in (("learning_rate", "lr"), ("lr", "lr"), ("beta_1", "beta1"), ("beta_2", "beta2"), ("epsilon", "epsilon")):
        v = cfg.get(src)
        if v is None:
            continue
        if isinstance(v, dict) or callable(v) or (hasattr(v, "get_config") and not hasattr(v, "__float__")):
            raise ValueError(f"Adam {src}: learning-rate schedules are not implemented (constant learning rate only)")
        try:
            h[dst] = float(v)
        except (TypeError, ValueError):
            raise ValueError(f"Adam {src}={v!r} is not a number (learning-rate schedules are not implemented)") from None
    return h


class History(dict):
    """what tf.keras.Model.fit returns: `.history` = {log key: [one value per epoch]}, `.epoch` = the epochs run, `.params`.  Also a
    dict itself (rounds 1-4 returned the plain dictionary), so both `h["fine_loss"]` and `h.history["fine_loss"]` read the curves."""

    def __init__(self, history, epoch, params):
        super().__init__(history)
        self.history, self.epoch, self.params = self, list(epoch), dict(params)


class NeRF:
    def __init__(self, n_coarse: int = 64, n_fine: int = 128, pos_emb_xyz: int = 10, pos_emb_dir: int = 4, n_layers: int = 8,
                 dense_units: int = 256, skip_layer=4, model_path: str = None, oob: str = "zero", seed: int = 42, **kwargs):
        logging.info("Initializing NeRF model")
        self.model_path = model_path
        if self.model_path is None:
            self.n_coarse, self.n_fine = n_coarse, n_fine
            self.pos_emb_xyz, self.pos_emb_dir = pos_emb_xyz, pos_emb_dir
            self.n_layers, self.dense_units, self.skip_layer = n_layers, dense_units, skip_layer
        else:
            self.load_model(model_path)                       # nerf.py:33-35: config comes from model_config.json
        # NB the reference builds the MLPs from the constructor arguments even when a model_path is given (nerf.py:37-40);
        # here the loaded config wins, which is what the saved weights need.
        xyz_dim, dir_dim = 3 + 6 * self.pos_emb_xyz, 3 + 6 * self.pos_emb_dir
        self.coarse = NeRFMLP(self.n_layers, self.dense_units, self.skip_layer, name="coarse_nerf", xyz_dim=xyz_dim,
                              dir_dim=dir_dim, seed=seed)
        self.fine = NeRFMLP(self.n_layers, self.dense_units, self.skip_layer, name="fine_nerf", xyz_dim=xyz_dim,
                            dir_dim=dir_dim, seed=seed + 1)
        self.oob = oob
        self.epsilon = 1e-10
        self.seed = seed
        self._ctx: Optional[KnerfContext] = None
        self._compiled = False
        self._global_step = 0
        self.stop_training = False

    # ------------------------------------------------------------------ save / load (nerf.py:45-76)
    def save_model(self, path, weights_only=False):
        logging.info("Saving NeRF model")
        if self._ctx is not None:
            self._ctx.poll_nonfinite(wait=True)      # never checkpoint past an unreported skipped step
        os.makedirs(path, exist_ok=True)
        if not weights_only:
            cfg = dict(n_coarse=self.n_coarse, n_fine=self.n_fine, pos_emb_xyz=self.pos_emb_xyz, pos_emb_dir=self.pos_emb_dir,
                       n_layers=self.n_layers, dense_units=self.dense_units, skip_layer=self.skip_layer)
            with open(os.path.join(path, "model_config.json"), "w") as f:
                json.dump(cfg, f)
        # same file names AND container as the reference: Keras-layout HDF5 (keras_nerf_amd/io/hdf5_min.py)
        self.coarse.save_weights(os.path.join(path, "coarse.h5"))
        self.fine.save_weights(os.path.join(path, "fine.h5"))

    def load_model(self, path):
        with open(os.path.join(path, "model_config.json"), "r") as f:
            c = json.load(f)
        self.n_coarse, self.n_fine = c["n_coarse"], c["n_fine"]
        self.pos_emb_xyz, self.pos_emb_dir = c["pos_emb_xyz"], c["pos_emb_dir"]
        self.n_layers, self.dense_units, self.skip_layer = c["n_layers"], c["dense_units"], c["skip_layer"]

    # ------------------------------------------------------------------ compile (nerf.py:78-173)
    def compile(self, optimizer="adam", loss="mse", batch_size=1, image_height=128, image_width=128, ray_chunks=2048,
                white_background=False, is_training=True, all_reduce="sum", deterministic=False, skip_dead_tiles=None, **kwargs):
        """nerf.py:78-173.  Extensions (keyword-only in spirit; the reference's callers never pass them):
        all_reduce 'sum' | 'mean' (data parallel), deterministic (bit-reproducible gradient sums, slower),
        skip_dead_tiles (None = the library default, on, unless KNERF_SKIP_DEAD_TILES says otherwise: the backward skips 32-sample
        tiles whose dL/d(rgb, sigma) is exactly zero -- same gradients, less work once the scene has empty space)."""
        logging.info("Compiling NeRF model")
        if not _is_mse(loss):
            raise ValueError("the fused HIP path implements the reference's mean-squared-error loss only")
        self.optimizer, self.loss = optimizer, loss
        self.batch_size, self.image_height, self.image_width = batch_size, image_height, image_width
        self.white_background = white_background
        self.run_eagerly = bool(kwargs.get("run_eagerly", False))
        self.ray_chunks = ray_chunks
        self.num_rays = batch_size * image_height * image_width
        if self.ray_chunks >= self.num_rays:                                      # nerf.py:95-98
            self.ray_chunks = self.num_rays
            logging.info(f"ray_chunks is greater than num_rays, setting ray_chunks to num_rays: {self.num_rays}")
        assert self.num_rays % self.ray_chunks == 0, \
            f"ray_chunks {self.ray_chunks} must be a divisor of the number of rays {self.num_rays}"   # nerf.py:100
        self.sequential_chunks = self.num_rays // self.ray_chunks
        if all_reduce not in ("sum", "mean"):
            raise ValueError("all_reduce must be 'sum' (Keras/MirroredStrategy semantics) or 'mean'")
        self.all_reduce = all_reduce
        h = _adam_hyper(optimizer)
        if self._ctx is not None:
            self.coarse._host, self.fine._host = self.coarse.get_flat_weights(), self.fine.get_flat_weights()
            self.coarse._ctx = self.fine._ctx = None
            self._ctx.close()
        opts = {}
        if deterministic:
            opts["deterministic"] = 1
        if self.run_eagerly:                             # nerf.py:430-451: the zero-gradient check exists in eager mode only
            opts["grad_diagnostics"] = 1
        if skip_dead_tiles is not None:                  # only an explicit argument overrides KNERF_SKIP_DEAD_TILES (runtime.py)
            opts["skip_dead_tiles"] = int(bool(skip_dead_tiles))
        self._ctx = KnerfContext(self.n_coarse, self.n_fine, self.pos_emb_xyz, self.pos_emb_dir, self.n_layers, self.dense_units,
                                 self.skip_layer, white_background, self.oob, h["lr"], h["beta1"], h["beta2"], h["epsilon"],
                                 options=opts)
        self.device = self._ctx.device
        self.nerf_utils = NeRFUtils(batch_size, image_height, image_width, self.ray_chunks, self.pos_emb_xyz, self.pos_emb_dir,
                                    white_background, self.oob)
        self._build_model()
        self.is_training = is_training
        self._dist = parallel.is_distributed()
        if self._dist:                                   # mirrored variables start identical (train.py:110-148)
            parallel.broadcast_weights([self._ctx.weights_view(COARSE), self._ctx.weights_view(FINE)])
            self._ctx.refresh_weights()
        self._loss_acc = torch.zeros(2, device=self.device)
        self._initialize_metrics()
        self._diag_seen = 0
        self._compiled = True

    def _zero_gradient_diagnostics(self, wait: bool):
        """nerf.py:430-451 (eager mode only): the non-zero counts of the LAST chunk's coarse and fine gradients, counted on the
        device (knerf_grad_diagnostics), with the reference's three messages.  wait=False (the asynchronous `fit` path) reports the
        newest step that has completed, once.  One further message the reference cannot give: every 32-sample tile of a net's
        passes was exactly dead in the backward (knerf_tile_stats_net) -- the sigma == 0 collapse in which the ReLU on sigma is
        closed everywhere and no gradient can ever re-open it."""
        c, f, seq = self._ctx.grad_diagnostics(wait=wait)
        if seq == self._diag_seen:
            return None
        self._diag_seen = seq
        if c == 0 and f == 0:
            logging.error('Both Coarse and Fine Gradient are zero')
        elif c == 0:
            logging.warning('Coarse Gradient is zero')
        elif f == 0:
            logging.warning('Fine Gradient is zero')
        if wait and self._ctx.get_option("skip_dead_tiles_active"):
            # this step's share of the RUNNING totals: read without resetting them -- tile_stats() consumers (bench.py's dead_tile_frac,
            # fit's per-epoch report, tools/) accumulate across steps -- and difference against the previous read
            now = self._ctx.tile_stats_net(reset=False)
            prev = getattr(self, "_tile_stats_seen", ((0, 0), (0, 0)))
            self._tile_stats_seen = now
            for name, (l1, t1), (l0, t0) in zip(("coarse", "fine"), now, prev):
                if t1 < t0:                       # somebody reset the counters in between: the totals ARE the delta
                    l0 = t0 = 0
                live, total = l1 - l0, t1 - t0
                if total > 0 and live == 0:
                    logging.warning(f'Every sample tile of the {name} passes is dead (no sample passes a gradient): sigma has collapsed to zero '
                                    f'or the pixel error is exactly zero everywhere')
        return c, f

    def _build_model(self):                              # nerf.py:116-136
        self.coarse._bind(self._ctx, COARSE)
        self.fine._bind(self._ctx, FINE)
        if self.model_path is not None:
            logging.info("Loading NeRF model weights")
            self.coarse.load_weights(os.path.join(self.model_path, "coarse.h5"))
            self.fine.load_weights(os.path.join(self.model_path, "fine.h5"))

    def _initialize_metrics(self):                       # nerf.py:167-173
        # the six running means share one device-side state (metrics.py): a step enqueues its contribution, nothing is read back
        st = self._metric_state = MetricState(self.device)
        self.coarse_loss_tracker = Mean("coarse_loss", st, 0); self.coarse_psnr_metric = Mean("coarse_psnr", st, 1)
        self.corase_ssim_metric = Mean("coarse_ssim", st, 2)
        self.fine_loss_tracker = Mean("fine_loss", st, 3); self.fine_psnr_metric = Mean("fine_psnr", st, 4)
        self.fine_ssim_metric = Mean("fine_ssim", st, 5)

    @property
    def metrics(self):                                   # nerf.py:499-508
        return [self.coarse_loss_tracker, self.coarse_psnr_metric, self.corase_ssim_metric,
                self.fine_loss_tracker, self.fine_psnr_metric, self.fine_ssim_metric]

    @property
    def metrics_names(self):
        return [m.name for m in self.metrics]

    def reset_metrics(self):
        self._metric_state.state.zero_()

    # ------------------------------------------------------------------ forward
    def _flat_rays(self, rays):
        o, d, t = rays
        f = self._ctx.f32
        return f(o).reshape(self.num_rays, 3), f(d).reshape(self.num_rays, 3), f(t).reshape(self.num_rays, self.n_coarse)

    def _next_seed(self):
        self._global_step += 1
        rank = torch.distributed.get_rank() if getattr(self, "_dist", False) else 0
        return (self.seed << 20) ^ (rank << 40) ^ self._global_step        # replicas draw different u

    def _predict_and_render_chunk(self, ray_chunks, coarse_weights_chunk=None, u=None, seed=0):
        """nerf.py:175-216.  ray_chunks = (o [R,3], d [R,3], t_coarse [R,n_coarse])."""
        o, d, t = [self._ctx.f32(x) for x in ray_chunks]
        net = COARSE
        if coarse_weights_chunk is not None:
            t = self._ctx.sample_fine(t, coarse_weights_chunk, u, seed=seed)
            net = FINE
        image, depth, weights = self._ctx.forward_chunk(net, o, d, t)
        return {"image": image, "depth": depth, "weights": weights}

    def predict_and_render_chunk(self, ray_chunks, u=None, seed=0, ray_offset=0):
        """nerf.py:218-227"""
        out = self._ctx.render_chunk(*ray_chunks, u=u, seed=seed, ray_offset=ray_offset)
        return ({"image": out["c_image"], "depth": out["c_depth"], "weights": out["c_weights"]},
                {"image": out["f_image"], "depth": out["f_depth"], "weights": out["f_weights"]})

    def predict_and_render_images(self, rays, u=None, outputs=None):
        """nerf.py:229-304: returns (coarse_results, fine_results), each {image [B,H,W,3], depth [B,H,W], weights [B,H,W,S]}.
        outputs (extension): the keys wanted, e.g. ("image", "depth") -- what inference.py:108-114 reads -- or ("image",) (test_step);
        the others are neither allocated nor written (knerf_render_batch takes NULL for them: at 256 x 256 the two `weights` arrays
        are 67 MB per frame).  Default: the reference's full dictionaries."""
        keys = ("image", "depth", "weights") if outputs is None else tuple(outputs)
        if "image" not in keys or any(k not in ("image", "depth", "weights") for k in keys):
            raise ValueError("outputs must contain 'image' and may contain 'depth' and 'weights'")
        o, d, t = self._flat_rays(rays)
        N, R, Nc, Na = self.num_rays, self.ray_chunks, self.n_coarse, self.n_coarse + self.n_fine
        e = lambda *s: torch.empty(s, device=self.device)
        buf = dict(c_image=e(N, 3), f_image=e(N, 3))
        if "depth" in keys:
            buf.update(c_depth=e(N), f_depth=e(N))
        if "weights" in keys:
            buf.update(c_weights=e(N, Nc), f_weights=e(N, Na))
        uf = None if u is None else self._ctx.f32(u).reshape(N, self.n_fine)
        seed = self._next_seed()
        # the chunk loop of nerf.py:236-288 runs inside the library: one host call per batch of images
        self._ctx.render_batch(o, d, t, uf, seed, R, out=buf)
        B, H, W = self.batch_size, self.image_height, self.image_width
        shape = {"image": (B, H, W, 3), "depth": (B, H, W)}
        coarse = {k: buf["c_" + k].reshape(shape.get(k, (B, H, W, Nc))) for k in keys}
        fine = {k: buf["f_" + k].reshape(shape.get(k, (B, H, W, Na))) for k in keys}
        return coarse, fine

    call = predict_and_render_images          # the reference defines no call(); Keras users expect one
    __call__ = predict_and_render_images

    # ------------------------------------------------------------------ metrics (nerf.py:306-330)
    def update_and_return_metrics(self, images, coarse_images, fine_images, coarse_loss, fine_loss):
        """two metric launches (each yields both the SSIM and the squared-error sums of one image pair) + one update of the
        device-side means; returns the six running means as a mapping that reads them back lazily (metrics.MetricLogs)"""
        losses = torch.stack([torch.as_tensor(coarse_loss, device=self.device).reshape(()).float(),
                              torch.as_tensor(fine_loss, device=self.device).reshape(()).float()])
        self._metric_state.update(images, coarse_images, fine_images, losses)
        return self._metric_state.snapshot()

    # ------------------------------------------------------------------ train / test step
    def train_step(self, inputs, u=None, with_metrics=True, sync=True):
        """nerf.py:332-473.  inputs = (images [B,H,W,3|4], (o, d, t)).  Returns the six running means (read back lazily).
        sync=True (a direct call): waits for the step, so a non-finite gradient raises from THIS call as in the reference
        (nerf.py:381-382).  sync=False (what `fit` uses): nothing waits for the GPU; a skipped step raises from the first later
        poll (fit polls at every epoch end, save_model and get_weights poll before they read weights)."""
        images, rays = inputs
        images = self._ctx.f32(images)[..., :3].contiguous()                      # nerf.py:335
        o, d, t = self._flat_rays(rays)
        N, R, C = self.num_rays, self.ray_chunks, self.sequential_chunks
        tgt = images.reshape(N, 3)
        uf = None if u is None else self._ctx.f32(u).reshape(N, self.n_fine)
        ci = torch.empty((N, 3), device=self.device); fi = torch.empty((N, 3), device=self.device)
        self._loss_acc.zero_()
        seed = self._next_seed()
        # the chunk loop of nerf.py:351-421 (C chunks of R rays, gradients and losses accumulated with weight 1/C) runs
        # inside the library: one host call per step
        self._ctx.train_batch(o, d, t, tgt, uf, seed, R, self._loss_acc, ci, fi)
        if self._dist:                                                            # nerf.py:455-458 under MirroredStrategy
            timing = getattr(self, "_allreduce_events", None)                     # bench.py: events around the collective
            if timing is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            parallel.all_reduce_gradients(self._ctx.grads_view(), self.all_reduce)
            if timing is not None:
                e1.record(); timing.append((e0, e1))
        # finite check (nerf.py:381-382), 2x Adam, accumulators zeroed (nerf.py:464-471): enqueued, nothing waits for the GPU
        self._ctx.apply_adam(check=False)
        if self.run_eagerly:
            self._zero_gradient_diagnostics(wait=bool(sync))
        if not with_metrics:
            # fast path (no host synchronisation per step): a step skipped for a non-finite gradient is reported by the
            # first later call that finds it completed
            self._ctx.poll_nonfinite(wait=False)
            return {"coarse_loss": self._loss_acc[0], "fine_loss": self._loss_acc[1]}
        B, H, W = self.batch_size, self.image_height, self.image_width
        self._metric_state.update(images, ci.reshape(B, H, W, 3), fi.reshape(B, H, W, 3), self._loss_acc)
        logs = self._metric_state.snapshot()
        self._ctx.poll_nonfinite(wait=bool(sync))      # sync: as the reference, the failing batch raises from its own train_step
        return logs

    def test_step(self, inputs, u=None):
        """nerf.py:475-497"""
        images, rays = inputs
        images = self._ctx.f32(images)[..., :3].contiguous()
        coarse, fine = self.predict_and_render_images(rays, u, outputs=("image",))     # nerf.py:489-497 reads the two images only
        # whole-image MSE (nerf.py:484-487) from the metrics kernel's squared-difference sums (losses=None)
        self._metric_state.update(images, coarse["image"], fine["image"], None)
        return self._metric_state.snapshot()

    def evaluate(self, dataset, return_dict=False, callbacks=None, verbose=0):
        """tf.keras.Model.evaluate on the reference's test_step (nerf.py:475-497): the metrics averaged over the dataset's batches, as
        a list in `metrics_names` order or as a dict (data-parallel: the replicas' mean, as fit's validation logs)."""
        callbacks = list(callbacks or [])
        for cb in callbacks:
            getattr(cb, "on_test_begin", lambda logs=None: None)({})
        self.reset_metrics()
        logs = {}
        for b, batch in enumerate(dataset):
            logs = self.test_step(batch)
            for cb in callbacks:
                getattr(cb, "on_test_batch_end", lambda i, logs=None: None)(b, logs)
        logs = parallel.reduce_logs({k: float(v) for k, v in dict(logs).items()}, self.device)
        for cb in callbacks:
            getattr(cb, "on_test_end", lambda logs=None: None)(dict(logs))
        if verbose:
            logging.info("evaluate - %s", " - ".join(f"{k}: {v:.4f}" for k, v in logs.items()))
        return logs if return_dict else [logs[k] for k in self.metrics_names if k in logs]

    # ------------------------------------------------------------------ fit: the part of tf.keras.Model.fit the reference uses
    def fit(self, dataset, epochs=1, validation_data=None, callbacks=None, initial_epoch=0, verbose=1):
        """train_single.py:137-143.  dataset yields (images, (o, d, t)) batches and is re-iterable; callbacks get the Keras
        hooks NeRFTrainMonitor uses (set_model, on_train_batch_end, on_epoch_end).  Returns a Keras-style History ({key: [per-epoch
        values]} as `.history`, and as the object itself)."""
        callbacks = list(callbacks or [])
        for cb in callbacks:
            if hasattr(cb, "set_model"):
                cb.set_model(self)
            else:
                cb.model = self
        history: Dict[str, list] = {}
        epochs_run = []
        self.stop_training = False
        for cb in callbacks:
            getattr(cb, "on_train_begin", lambda logs=None: None)({})
        for epoch in range(initial_epoch, epochs):
            self.reset_metrics()
            for cb in callbacks:
                getattr(cb, "on_epoch_begin", lambda e, logs=None: None)(epoch, {})
            logs = {}
            for b, batch in enumerate(dataset):
                # asynchronous: the step is enqueued, its logs stay on the device until a callback reads them (MetricLogs)
                logs = self.train_step(batch, sync=False)
                for cb in callbacks:
                    getattr(cb, "on_train_batch_end", lambda i, logs=None: None)(b, logs)
            logs = {k: float(v) for k, v in dict(logs).items()}            # one read-back per epoch
            self._ctx.poll_nonfinite(wait=True)                            # a skipped step of this epoch raises here at the latest
            if validation_data is not None:
                for cb in callbacks:
                    getattr(cb, "on_test_begin", lambda logs=None: None)({})
                self.reset_metrics()
                vlogs = {}
                for batch in validation_data:
                    vlogs = self.test_step(batch)
                logs.update({"val_" + k: float(v) for k, v in dict(vlogs).items()})
                for cb in callbacks:
                    getattr(cb, "on_test_end", lambda logs=None: None)(dict(logs))
            logs = parallel.reduce_logs(logs, self.device)    # replica means (one tiny all-reduce)
            for k, v in logs.items():
                history.setdefault(k, []).append(v)
            epochs_run.append(epoch)
            if verbose:
                logging.info("Epoch %d/%d - %s", epoch + 1, epochs, " - ".join(f"{k}: {v:.4f}" for k, v in logs.items()))
            for cb in callbacks:
                getattr(cb, "on_epoch_end", lambda e, logs=None: None)(epoch, logs)
            if self.stop_training:
                break
        for cb in callbacks:
            getattr(cb, "on_train_end", lambda logs=None: None)({})
        try:
            steps = len(dataset)
        except TypeError:
            steps = None
        self.history = History(history, epochs_run, {"epochs": epochs, "steps": steps, "verbose": verbose})
        return self.history
