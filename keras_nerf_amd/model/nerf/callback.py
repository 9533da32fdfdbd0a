"""NeRFTrainMonitor -- counterpart of reference keras_nerf/model/nerf/callback.py:8-226 (host-side observability).

Same artefacts: `<log_dir>/log.csv` (columns `epoch` + the 12 log keys, header written at epoch 0 only, one row per
`update_freq` epochs), `<log_dir>/model/` checkpoints (config json only at epoch 0), PNG panels `test_{i}_{epoch}.png`,
`test_sample_{i}_{epoch}.png`, `debug_{i}_{batch}.png` (verbose), and the resume rule `last_epoch = last CSV epoch + 1`
(the reference's reader skips the first data row when rebuilding its loss history, callback.py:38-46; kept).

Data parallel (one process per GPU): every rank keeps the loss history, only rank 0 renders panels and writes files (the
replicas hold identical weights and `fit` has already averaged the logs); the other ranks wait at a barrier so that nobody
runs ahead of a checkpoint that is still being written.  The barrier is reached even when rank 0's rendering, plotting or
writing raises (try/finally: the error then surfaces on rank 0 instead of as a collective time-out everywhere else), and the
"other test images" iterator walks a PRIVATE view of the dataset (own shuffle generator), so that its rank-0-only iteration
never advances the generator from which every rank derives the shared batch order."""
from __future__ import annotations

import logging
import os
from csv import DictReader, DictWriter

import numpy as np

from ... import parallel


def _np(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


class NeRFTrainMonitor:
    def __init__(self, dataset, log_dir: str, batch_size: int, update_freq: int = 1, verbose: bool = False, plots: bool = True, **kwargs):
        self.dataset, self.log_dir, self.batch_size, self.update_freq, self.verbose = dataset, log_dir, batch_size, update_freq, verbose
        self.plots = plots
        self.model = None
        self.log_model_dir = os.path.join(log_dir, "model")
        os.makedirs(self.log_model_dir, exist_ok=True)
        self.coarse_log_list, self.val_coarse_log_list, self.fine_log_list, self.val_fine_log_list = [], [], [], []
        if self.verbose:
            self.coarse_log_list_batch, self.fine_log_list_batch = [], []
        self.last_epoch = 0
        self.log_csv = os.path.join(log_dir, "log.csv")
        if os.path.exists(self.log_csv):
            with open(self.log_csv, "r") as f:
                for i, row in enumerate(DictReader(f)):
                    if i > 0:
                        self.coarse_log_list.append(float(row["coarse_loss"])); self.val_coarse_log_list.append(float(row["val_coarse_loss"]))
                        self.fine_log_list.append(float(row["fine_loss"])); self.val_fine_log_list.append(float(row["val_fine_loss"]))
                        self.last_epoch = int(row["epoch"])
            self.last_epoch += 1
        for inputs in self.dataset.take(1):
            self.images, self.rays = inputs
            o, d, t = self.rays
            self.ray_origin, self.ray_direction, self.coarse_points = o[:batch_size], d[:batch_size], t[:batch_size]
        # rank-0-only iteration below must not touch the dataset's own generator (loader.py: the ranks' shared order)
        self._sample_view = self.dataset.private_view(seed=20240229) if hasattr(self.dataset, "private_view") else self.dataset
        self.dataset_iterator = iter(self._sample_view)
        self.dataset_iterator.get_next()

    def set_model(self, model):
        self.model = model

    # ---- plotting (matplotlib, Agg)
    def _panel(self, path, coarse, fine, gt, i, curves=None, title=None):
        if not self.plots:
            return
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        rows = 2 if curves else 1
        fig = plt.figure(figsize=(20, 5 * rows))
        gs = fig.add_gridspec(rows, 5)
        for k, (img, name, cmap) in enumerate([(coarse["image"], "Coarse Image", None), (coarse["depth"], "Coarse Depth", "inferno"),
                                               (fine["image"], "Fine Image", None), (fine["depth"], "Fine Depth", "inferno"),
                                               (gt, "Ground Truth", None)]):
            ax = fig.add_subplot(gs[0, k]); ax.imshow(np.clip(_np(img)[i][..., :3] if cmap is None else _np(img)[i], 0, None), cmap=cmap); ax.set_title(name)
        if curves:
            ax = fig.add_subplot(gs[1, :])
            for ys, color, style, label in curves:
                ax.plot(ys, color=color, linestyle=style, label=label)
            ax.legend(); ax.set_yscale("log"); ax.set_title(title)
        plt.savefig(path); plt.close(fig)

    def on_train_batch_end(self, batch, logs=None):
        if not self.verbose:
            return
        logging.debug(f"Batch {batch}: {logs}")
        self.coarse_log_list_batch.append(logs["coarse_loss"]); self.fine_log_list_batch.append(logs["fine_loss"])
        if not parallel.is_main():
            return
        coarse, fine = self.model.predict_and_render_images((self.ray_origin, self.ray_direction, self.coarse_points), outputs=("image", "depth"))
        curves = [(self.coarse_log_list_batch, "blue", "solid", "Coarse Train Loss"), (self.fine_log_list_batch, "orange", "solid", "Fine Train Loss")]
        for i in range(self.batch_size):
            self._panel(os.path.join(self.log_dir, f"debug_{i}_{batch}.png"), coarse, fine, self.images, i, curves, f"Loss Batch Plot: {batch}")

    def on_epoch_end(self, epoch, logs):
        self.coarse_log_list.append(logs["coarse_loss"]); self.val_coarse_log_list.append(logs["val_coarse_loss"])
        self.fine_log_list.append(logs["fine_loss"]); self.val_fine_log_list.append(logs["val_fine_loss"])
        if epoch % self.update_freq == 0 and not parallel.is_main():
            parallel.barrier()                                     # rank 0 is writing panels, log.csv and the checkpoint
        elif epoch % self.update_freq == 0:
            try:
                self._write_epoch(epoch, logs)
            finally:
                parallel.barrier()                                 # also when the writing failed: nobody is left waiting
        if self.verbose:
            self.coarse_log_list_batch, self.fine_log_list_batch = [], []

    def _write_epoch(self, epoch, logs):
        coarse, fine = self.model.predict_and_render_images((self.ray_origin, self.ray_direction, self.coarse_points), outputs=("image", "depth"))
        curves = [(self.coarse_log_list, "blue", "solid", "Coarse Train Loss"), (self.val_coarse_log_list, "blue", "dashed", "Coarse Val Loss"),
                  (self.fine_log_list, "orange", "solid", "Fine Train Loss"), (self.val_fine_log_list, "orange", "dashed", "Fine Val Loss")]
        for i in range(self.batch_size):
            self._panel(os.path.join(self.log_dir, f"test_{i}_{epoch}.png"), coarse, fine, self.images, i, curves, f"Loss Plot: {epoch}")
        try:                                                   # "Predict other test images" (callback.py:168-209)
            images, rays = self.dataset_iterator.get_next()
        except StopIteration:
            self.dataset_iterator = iter(self._sample_view)
            images, rays = self.dataset_iterator.get_next()
        o, d, t = [r[:self.batch_size] for r in rays]
        coarse, fine = self.model.predict_and_render_images((o, d, t), outputs=("image", "depth"))     # the panels show images and depths only
        for i in range(self.batch_size):
            self._panel(os.path.join(self.log_dir, f"test_sample_{i}_{epoch}.png"), coarse, fine, images, i)
        with open(self.log_csv, "a") as f:                     # callback.py:211-218
            new_logs = {"epoch": epoch}
            new_logs.update(logs)
            w = DictWriter(f, new_logs.keys())
            if epoch == 0:
                w.writeheader()
            w.writerow(new_logs)
        self.model.save_model(self.log_model_dir, weights_only=(epoch != 0))   # callback.py:220-222
