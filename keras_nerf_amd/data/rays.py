"""RaysGenerator -- counterpart of reference keras_nerf/data/rays.py:4-130, generated on the GPU (csrc/raygen.hip)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib
from ..runtime import KnerfError


class RaysGenerator:
    def __init__(self, focal_length: float, image_width: int, image_height: int, near: float, far: float, n_sample: int,
                 seed: int = 0, **kwargs):
        self.focal_length, self.image_width, self.image_height = float(focal_length), int(image_width), int(image_height)
        self.near, self.far, self.n_sample = float(near), float(far), int(n_sample)
        self.seed, self._calls = seed, 0
        if not torch.cuda.is_available():
            raise KnerfError("keras_nerf_amd needs an MI355X (gfx950) GPU; there is no CPU path")
        self._lib = _lib.load()

    def __call__(self, camera_params, noise=None):
        """camera_params: 4x4 camera-to-world (or [B,4,4]).  Returns (ray_origin, ray_direction [...,H,W,3],
        sample_points [...,H,W,n_sample]); the jitter is redrawn on every call (rays.py:122-123) unless `noise` is given."""
        c2w = torch.as_tensor(np.asarray(camera_params, np.float32) if not isinstance(camera_params, torch.Tensor)
                              else camera_params).to("cuda", torch.float32).contiguous()
        single = c2w.dim() == 2
        c2w = c2w.reshape(-1, 4, 4)
        B, H, W, N = c2w.shape[0], self.image_height, self.image_width, self.n_sample
        nz = None if noise is None else torch.as_tensor(np.asarray(noise, np.float32) if not isinstance(noise, torch.Tensor)
                                                        else noise).to("cuda", torch.float32).reshape(B, H, W, N).contiguous()
        o = torch.empty((B, H, W, 3), device="cuda"); d = torch.empty_like(o); t = torch.empty((B, H, W, N), device="cuda")
        self._calls += 1
        p = lambda x: None if x is None else C.c_void_p(x.data_ptr())
        rc = self._lib.knerf_generate_rays(None, C.c_void_p(torch.cuda.current_stream().cuda_stream), p(c2w), p(nz), self.seed,
                                           self._calls, B, H, W, N, self.focal_length, self.near, self.far, p(o), p(d), p(t))
        if rc != 0:
            raise KnerfError(f"knerf_generate_rays failed ({rc})")
        return (o[0], d[0], t[0]) if single else (o, d, t)
