"""NeRFUtils -- counterpart of reference keras_nerf/model/nerf/utils.py:4-210, served by stand-alone HIP ops
(csrc/utils_ops.hip, csrc/composite.hip).  Inputs may be numpy arrays or torch tensors; outputs are CUDA tensors."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ... import _lib
from ...runtime import KnerfError


def _dev(x):
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(np.asarray(x, np.float32))
    return x.to("cuda", torch.float32).contiguous()


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class NeRFUtils:
    def __init__(self, batch_size, image_height, image_width, ray_chunks, pos_emb_xyz, pos_emb_dir, white_background=False,
                 oob="zero"):
        self.batch_size, self.image_height, self.image_width = batch_size, image_height, image_width
        self.ray_chunks, self.pos_emb_xyz, self.pos_emb_dir = ray_chunks, pos_emb_xyz, pos_emb_dir
        self.num_rays = batch_size * image_height * image_width
        self.sequential_chunks = self.num_rays // ray_chunks
        self.white_background = white_background
        self.oob = oob
        if not torch.cuda.is_available():
            raise KnerfError("keras_nerf_amd needs an MI355X (gfx950) GPU; there is no CPU path")
        self._lib = _lib.load()

    def _chk(self, rc):
        if rc != 0:
            raise (ValueError if rc == _lib.KNERF_ERR_INVALID else KnerfError)(f"knerf utils op failed ({rc})")

    # utils.py:176-186
    def positional_encoding(self, inputs, pos_embedding_dim):
        x = _dev(inputs)
        lead = x.shape[:-1]
        out = torch.empty(lead + (3 + 6 * pos_embedding_dim,), device="cuda")
        self._chk(self._lib.knerf_positional_encoding(_stream(), _p(x), x.numel() // 3, int(pos_embedding_dim), _p(out)))
        return out

    # utils.py:188-210
    def encode_position_and_directions(self, ray_origin, ray_direction, coarse_points):
        o, d, t = _dev(ray_origin), _dev(ray_direction), _dev(coarse_points)
        S = t.shape[-1]
        R = t.numel() // S
        pos = torch.empty(tuple(t.shape) + (3,), device="cuda")
        self._chk(self._lib.knerf_ray_points(_stream(), _p(o.reshape(R, 3).contiguous()), _p(d.reshape(R, 3).contiguous()),
                                             _p(t.reshape(R, S).contiguous()), R, S, _p(pos)))   # ray(t) = o + t d (utils.py:193-194)
        enc = self.positional_encoding(pos, self.pos_emb_xyz)
        dirs = d[..., None, :].expand_as(pos).contiguous()
        return enc, self.positional_encoding(dirs, self.pos_emb_dir)

    def _composite(self, rgb, sigma, sample_points, white):
        rgb, sigma, t = _dev(rgb), _dev(sigma), _dev(sample_points)
        lead, S = t.shape[:-1], t.shape[-1]
        R = t.numel() // S
        raw = torch.cat([rgb.reshape(R, S, 3), sigma.reshape(R, S, 1)], dim=-1).contiguous()
        image = torch.empty((R, 3), device="cuda"); depth = torch.empty((R,), device="cuda"); w = torch.empty((R, S), device="cuda")
        self._chk(self._lib.knerf_composite(_stream(), _p(raw), _p(t.reshape(R, S)), R, S, int(white), _p(image), _p(depth), _p(w)))
        return image.reshape(lead + (3,)), depth.reshape(lead), w.reshape(lead + (S,))

    # utils.py:16-58 (epsilon is the reference's fixed 1e-10)
    def render_image_depth_chunk(self, rgb, sigma, sample_points, epsilon=1e-10):
        if epsilon != 1e-10:
            raise ValueError("the kernel implements the reference's epsilon = 1e-10")
        return self._composite(rgb, sigma, sample_points, 1 if self.white_background else 0)

    # utils.py:99-134: the non-chunk twin omits the white background AND the clip
    def render_image_depth(self, rgb, sigma, sample_points, epsilon=1e-10):
        return self._composite(rgb, sigma, sample_points, 2)          # mode bit 1: black background, no clip

    def _inverse_cdf(self, mid_points, weights, n_samples, u=None):
        m, w = _dev(mid_points), _dev(weights)
        lead = w.shape[:-1]
        R = w.numel() // w.shape[-1]
        u = torch.rand((R, n_samples), device="cuda") if u is None else _dev(u).reshape(R, n_samples)
        out = torch.empty((R, n_samples), device="cuda")
        self._chk(self._lib.knerf_inverse_cdf(_stream(), _p(m.reshape(R, -1)), _p(w.reshape(R, -1)), _p(u), R, m.shape[-1], w.shape[-1],
                                              int(n_samples), int(self.oob == "clamp"), _p(out)))
        return out.reshape(lead + (n_samples,))

    # utils.py:60-97 / 136-174 (u is drawn like tf.random.uniform unless given)
    def fine_hierarchical_sampling_chunk(self, mid_points, weights, n_samples, u=None):
        return self._inverse_cdf(mid_points, weights, n_samples, u)

    def fine_hierarchical_sampling(self, mid_points, weights, n_samples, u=None):
        return self._inverse_cdf(mid_points, weights, n_samples, u)
