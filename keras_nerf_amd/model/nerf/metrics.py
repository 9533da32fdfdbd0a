"""Keras-style running Mean and the two image metrics the reference logs (nerf.py:306-330): tf.image.psnr and
tf.image.ssim with their defaults, computed by one HIP kernel (knerf_image_metrics); the running means are host-side."""
from __future__ import annotations

import math

import torch


class Mean:
    """tf.keras.metrics.Mean: running mean of every value fed to update_state (a tensor contributes all its elements)"""

    def __init__(self, name):
        self.name = name
        self.reset_state()

    def reset_state(self):
        self.total, self.count = 0.0, 0

    reset_states = reset_state

    def update_state(self, v):
        v = torch.as_tensor(v, dtype=torch.float32).reshape(-1)
        self.total += float(v.sum()); self.count += v.numel()

    def result(self):
        return self.total / self.count if self.count else 0.0


def _image_sums(a, b):
    """[B,H,W,C] CUDA tensors -> per-image (SSIM term sum, squared-difference sum) from the HIP kernel (csrc/utils_ops.hip)"""
    from ... import _lib
    if not (isinstance(a, torch.Tensor) and a.is_cuda):
        from ...runtime import KnerfError
        raise KnerfError("image metrics run on the MI355X (knerf_image_metrics); there is no CPU path")
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
