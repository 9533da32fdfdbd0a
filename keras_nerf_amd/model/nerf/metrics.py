"""Keras-style running Mean and the two image metrics the reference logs (nerf.py:306-330): tf.image.psnr and tf.image.ssim with
their defaults, computed by one HIP kernel per image pair (knerf_image_metrics).

The running means live ON THE DEVICE (`MetricState`: six {total, count} pairs of doubles): a train step enqueues two metric
launches and one update (knerf_metrics_update) and nothing returns to the host until a result is read -- `fit` reads them once
per epoch (or when a callback looks at a batch's logs), not six times per step."""
from __future__ import annotations

import collections.abc
import math

import torch

NAMES = ("coarse_loss", "coarse_psnr", "coarse_ssim", "fine_loss", "fine_psnr", "fine_ssim")


def _check_cuda(a):
    if not (isinstance(a, torch.Tensor) and a.is_cuda):
        from ...runtime import KnerfError
        raise KnerfError("image metrics run on the MI355X (knerf_image_metrics); there is no CPU path")


def _image_sums(a, b):
    """[B,H,W,C] CUDA tensors -> per-image (SSIM term sum, squared-difference sum) from the HIP kernel (csrc/utils_ops.hip)"""
    from ... import _lib
    _check_cuda(a)
    a = a.to(torch.float32).contiguous(); b = b.to(device=a.device, dtype=torch.float32).contiguous()
    if a.shape != b.shape or a.dim() != 4:
        raise ValueError("expected two [B,H,W,C] tensors of the same shape")
    B, H, W, C = a.shape
    if H < 11 or W < 11:
        raise ValueError("ssim needs images of at least 11x11")
    sums = torch.empty((B, 2), device=a.device, dtype=torch.float32)
    rc = _lib.load().knerf_image_metrics(torch.cuda.current_stream(a.device).cuda_stream, a.data_ptr(), b.data_ptr(), B, H, W, C,
                                         sums.data_ptr())
    if rc != 0:
        from ...runtime import KnerfError
        raise KnerfError(f"knerf_image_metrics failed ({rc})")
    return sums, (H - 10) * (W - 10) * C, H * W * C


def psnr(a, b, max_val=1.0):
    """tf.image.psnr: per image over the last three axes"""
    sums, _, n = _image_sums(a, b)
    return 20.0 * math.log10(max_val) - 10.0 * torch.log10(sums[:, 1] / n)


def ssim(a, b, max_val=1.0):
    """tf.image.ssim defaults: 11x11 Gaussian (sigma 1.5), VALID windows, k1 0.01, k2 0.03, mean over windows and channels"""
    if max_val != 1.0:
        raise ValueError("the kernel implements max_val = 1 (the reference's call, nerf.py:310-321)")
    sums, nwin, _ = _image_sums(a, b)
    return sums[:, 0] / nwin


class MetricState:
    """the six running means of NeRF.metrics as one device tensor [6, 2] = {total, count} (float64)"""

    def __init__(self, device):
        self.device = device
        self.state = torch.zeros((len(NAMES), 2), device=device, dtype=torch.float64)
        self.events = None           # a list: update() appends a pair of HIP events recorded on the stream around its three launches (bench.py)

    def update(self, images, coarse_images, fine_images, losses):
        """one step's contribution (nerf.py:306-330): `losses` = device tensor [2] (coarse, fine), or None for the whole-image
        mean squared errors of test_step (nerf.py:484-487).  Three launches, no synchronisation."""
        from ... import _lib
        if self.events is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(self.device))
        sc, _, _ = _image_sums(images, coarse_images)
        sf, _, _ = _image_sums(images, fine_images)
        B, H, W, C = images.shape
        if losses is not None:
            losses = losses.to(device=self.device, dtype=torch.float32).contiguous()
        rc = _lib.load().knerf_metrics_update(torch.cuda.current_stream(self.device).cuda_stream, sc.data_ptr(), sf.data_ptr(),
                                              None if losses is None else losses.data_ptr(), B, H, W, C, self.state.data_ptr())
        if rc != 0:
            from ...runtime import KnerfError
            raise KnerfError(f"knerf_metrics_update failed ({rc})")
        if self.events is not None:
            e1.record(torch.cuda.current_stream(self.device))
            self.events.append((e0, e1))

    def snapshot(self) -> "MetricLogs":
        return MetricLogs(self.state.clone())


class MetricLogs(collections.abc.Mapping):
    """What train_step / test_step return: the six running means AS OF that step.  The values sit in a device-side snapshot and
    come to the host (one copy, one synchronisation) the first time one of them is read."""

    def __init__(self, snapshot: torch.Tensor):
        self._snap, self._host = snapshot, None

    def _values(self):
        if self._host is None:
            s = self._snap.cpu()
            self._host = {n: (float(s[i, 0] / s[i, 1]) if float(s[i, 1]) else 0.0) for i, n in enumerate(NAMES)}
            self._snap = None
        return self._host

    def __getitem__(self, k):
        return self._values()[k]

    def __iter__(self):
        return iter(NAMES)

    def __len__(self):
        return len(NAMES)

    def __repr__(self):
        return f"MetricLogs({self._values()!r})"


class Mean:
    """tf.keras.metrics.Mean: running mean of every value fed to update_state (a tensor contributes all its elements).  Backed by
    row `index` of a MetricState (device) when given one, so that NeRF.metrics[i].result() and the fused update agree; a
    stand-alone Mean keeps its own one-row state."""

    def __init__(self, name, state: MetricState = None, index: int = 0):
        self.name = name
        self._owner, self._index = state, index
        self._own = None if state is not None else [0.0, 0]

    def _row(self):
        return self._owner.state[self._index]

    def reset_state(self):
        if self._owner is not None:
            self._row().zero_()
        else:
            self._own = [0.0, 0]

    reset_states = reset_state

    def update_state(self, v):
        if self._owner is not None:
            v = torch.as_tensor(v).to(device=self._owner.device, dtype=torch.float64).reshape(-1)
            row = self._row()
            row[0] += v.sum(); row[1] += v.numel()
        else:
            v = torch.as_tensor(v, dtype=torch.float32).reshape(-1)
            self._own[0] += float(v.sum()); self._own[1] += v.numel()

    def result(self):
        if self._owner is not None:
            t, c = (float(x) for x in self._row().cpu())
        else:
            t, c = self._own
        return t / c if c else 0.0
