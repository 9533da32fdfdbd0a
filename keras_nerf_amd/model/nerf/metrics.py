"""Keras-style running Mean and the two image metrics the reference logs (nerf.py:306-330): tf.image.psnr and
tf.image.ssim with their defaults.  Host-side bookkeeping on CUDA tensors; not part of the fused hot path."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


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


def psnr(a, b, max_val=1.0):
    """tf.image.psnr: per image over the last three axes"""
    mse = torch.mean((a - b) ** 2, dim=(-3, -2, -1))
    return 20.0 * math.log10(max_val) - 10.0 * torch.log10(mse)


def _gauss(size=11, sigma=1.5, device="cpu"):
    x = torch.arange(size, dtype=torch.float32, device=device) - (size - 1) / 2.0
    g = torch.exp(-(x ** 2) / (2 * sigma ** 2))
    g = g / g.sum()
    return torch.outer(g, g)


def ssim(a, b, max_val=1.0, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03):
    """tf.image.ssim defaults: 11x11 Gaussian (sigma 1.5), VALID windows, mean over windows and channels; [B,H,W,C]."""
    x = a.permute(0, 3, 1, 2); y = b.permute(0, 3, 1, 2)
    C = x.shape[1]
    if x.shape[-1] < filter_size or x.shape[-2] < filter_size:
        raise ValueError(f"ssim needs images of at least {filter_size}x{filter_size}")
    w = _gauss(filter_size, filter_sigma, x.device)[None, None].repeat(C, 1, 1, 1)
    conv = lambda z: F.conv2d(z, w, groups=C)
    c1, c2 = (k1 * max_val) ** 2, (k2 * max_val) ** 2
    mx, my = conv(x), conv(y)
    sxx, syy, sxy = conv(x * x) - mx * mx, conv(y * y) - my * my, conv(x * y) - mx * my
    lum = (2 * mx * my + c1) / (mx * mx + my * my + c1)
    cs = (2 * sxy + c2) / (sxx + syy + c2)
    return torch.mean(lum * cs, dim=(1, 2, 3))
