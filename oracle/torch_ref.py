"""Op-for-op, unfused torch-CPU restatement of the reference train step  --  TEST INFRASTRUCTURE.

Two uses only (see oracle/nerf_oracle.py header for the import rule):
  * tests: autograd cross-check of the NumPy oracle's hand-written backward;
  * bench.py ``cpu_baseline`` leg (kind "port"): the stand-in for the reference's TF-CPU path, which cannot
    run here (TensorFlow absent).  Same op order as SURVEY.md section 2.1, fp32, autograd backward, Keras-form
    Adam, multi-threaded torch CPU matmuls (thread count is reported by the caller).

Cites are relative to the reference root.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch

from .nerf_oracle import NerfConfig


def positional_encoding(x: torch.Tensor, L: int) -> torch.Tensor:  # utils.py:176-186
    parts = [x]
    for i in range(L):
        parts.append(torch.sin((2.0 ** i) * x))
        parts.append(torch.cos((2.0 ** i) * x))
    return torch.cat(parts, dim=-1)


def encode(o, d, t, Lx, Ld):  # utils.py:188-210
    p = o[..., None, :] + d[..., None, :] * t[..., None]
    xyz = positional_encoding(p, Lx)
    dire = positional_encoding(d[..., None, :].expand_as(p), Ld)
    return xyz, dire


def mlp(params: Sequence[torch.Tensor], xyz, dire, cfg: NerfConfig):  # mlp.py:29-50
    h = xyz
    p = 0
    for i in range(cfg.n_layers):
        h = torch.relu(h @ params[p] + params[p + 1]); p += 2
        if i % cfg.skip_layer == 0 and i > 0:
            h = torch.cat([h, xyz], dim=-1)
    sigma = torch.relu(h @ params[p] + params[p + 1]); p += 2
    feat = h @ params[p] + params[p + 1]; p += 2
    f2 = torch.cat([feat, dire], dim=-1) @ params[p] + params[p + 1]; p += 2
    rgb = torch.sigmoid(f2 @ params[p] + params[p + 1])
    return rgb, sigma


def render(rgb, sigma, t, white_background, eps=1e-10):  # utils.py:16-58
    sigma = sigma[..., 0]
    delta = torch.cat([t[..., 1:] - t[..., :-1], torch.full_like(t[..., :1], eps)], dim=-1)
    alpha = 1.0 - torch.exp(-sigma * delta)
    e = 1.0 - alpha
    x = e + eps
    T = torch.cat([torch.ones_like(x[..., :1]), torch.cumprod(x[..., :-1], dim=-1)], dim=-1)
    w = alpha * T
    image = torch.sum(w[..., None] * rgb, dim=-2)
    depth = torch.sum(w * t, dim=-1)
    if white_background:
        image = image + (1.0 - torch.sum(w, dim=-1)[..., None])
    # TF clip_by_value passes gradient on the closed interval; torch.clamp does too
    image = torch.clamp(image, 0.0, 1.0)
    return image, depth, w


def fine_sampling(mids, weights, u, oob="zero"):  # utils.py:60-97
    w = weights + 1e-5
    pdf = w / torch.sum(w, dim=-1, keepdim=True)
    cdf = torch.cumsum(pdf, dim=-1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], dim=-1)
    idx = torch.searchsorted(cdf.contiguous(), u.contiguous(), right=True)
    below = torch.clamp(idx - 1, min=0)
    above = torch.clamp(idx, max=cdf.shape[-1] - 1)
    cb, ca = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    nm = mids.shape[-1]

    def g(ix):
        v = torch.gather(mids, -1, torch.clamp(ix, max=nm - 1))
        if oob == "zero":
            v = torch.where(ix < nm, v, torch.zeros_like(v))
        return v
    mb, ma = g(below), g(above)
    denom = ca - cb
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    tt = (u - cb) / denom
    return mb + tt * (ma - mb)


def chunk_forward(params, o, d, t, cfg, white_background):  # nerf.py:175-216
    xyz, dire = encode(o, d, t, cfg.pos_emb_xyz, cfg.pos_emb_dir)
    rgb, sigma = mlp(params, xyz, dire, cfg)
    return render(rgb, sigma, t, white_background)


class TorchKerasAdam:  # nerf.py:163-165, Keras form (eps outside the bias-corrected root)
    def __init__(self, params, lr=1e-3, b1=0.9, b2=0.999, eps=1e-7):
        self.lr, self.b1, self.b2, self.eps, self.t = lr, b1, b2, eps, 0
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]

    @torch.no_grad()
    def apply(self, params, grads):
        self.t += 1
        lr_t = self.lr * np.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        for p, g, m, v in zip(params, grads, self.m, self.v):
            m.mul_(self.b1).add_(g, alpha=1 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            p.sub_(lr_t * m / (v.sqrt() + self.eps))


def train_step(cp: List[torch.Tensor], fp: List[torch.Tensor], oc, of_, images, o, d, t, u, cfg: NerfConfig,
               ray_chunks: int, white_background: bool, oob="zero", coarse_only=False, chunk_order=None):
    """nerf.py:332-473 with autograd.  ``coarse_only`` is BASELINE config 1 (coarse net, no fine pass).
    ``chunk_order`` (a permutation of range(C)) visits the chunks in another order: the same sums in a different
    floating-point order, used to measure the fp32 arithmetic's own run-to-run spread (tools/convergence128.py)."""
    N = o.numel() // 3
    R = min(ray_chunks, N)
    assert N % R == 0
    C = N // R
    im, o_, d_, t_, u_ = images[..., :3].reshape(N, 3), o.reshape(N, 3), d.reshape(N, 3), t.reshape(N, -1), \
        u.reshape(N, -1)
    acc_c = [torch.zeros_like(p) for p in cp]
    acc_f = [torch.zeros_like(p) for p in fp]
    lc_tot = 0.0; lf_tot = 0.0
    for i in (range(C) if chunk_order is None else chunk_order):
        sl = slice(i * R, (i + 1) * R)
        img, _, w = chunk_forward(cp, o_[sl], d_[sl], t_[sl], cfg, white_background)
        lc = torch.mean((im[sl] - img) ** 2)
        gc = torch.autograd.grad(lc, cp)
        for a, g in zip(acc_c, gc):
            a.add_(g / C)
        lc_tot += float(lc.detach()) / C
        if coarse_only:
            continue
        with torch.no_grad():
            mids = 0.5 * (t_[sl][..., 1:] + t_[sl][..., :-1])
            tf_ = fine_sampling(mids, w.detach(), u_[sl], oob)
            tall = torch.sort(torch.cat([t_[sl], tf_], dim=-1), dim=-1).values
        img, _, _ = chunk_forward(fp, o_[sl], d_[sl], tall, cfg, white_background)
        lf = torch.mean((im[sl] - img) ** 2)
        gf = torch.autograd.grad(lf, fp)
        for a, g in zip(acc_f, gf):
            a.add_(g / C)
        lf_tot += float(lf.detach()) / C
    if oc is not None:
        oc.apply(cp, acc_c)
        if not coarse_only:
            of_.apply(fp, acc_f)
    return lc_tot, lf_tot, acc_c, acc_f
