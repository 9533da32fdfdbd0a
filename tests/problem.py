"""Seeded synthetic problems shared by the tests (poses from pose_spherical, jittered coarse t,
glorot weights, injected u)."""
import numpy as np

from oracle import nerf_oracle as O


def make_problem(n_images=1, wh=16, seed=42, weight_scale=1.0, bias_std=0.0, cfg=None):
    cfg = cfg or O.NerfConfig()
    rng = np.random.default_rng(seed)
    focal = O.get_focal_from_fov(0.6911112070083618, wh)
    os_, ds_, ts_ = [], [], []
    for i in range(n_images):
        c2w = O.pose_spherical(360.0 * i / max(n_images, 1) + 20.0, -30.0, 4.0)
        o, d, t = O.generate_rays(c2w, focal, wh, wh, 2.0, 6.0, cfg.n_coarse, rng.random((wh, wh, cfg.n_coarse)))
        os_.append(o); ds_.append(d); ts_.append(t)
    o, d, t = np.stack(os_), np.stack(ds_), np.stack(ts_)
    N = n_images * wh * wh
    u = np.random.default_rng(7).random((n_images, wh, wh, cfg.n_fine), dtype=np.float32)
    img = rng.random((n_images, wh, wh, 3), dtype=np.float32)
    cp = [p * np.float32(weight_scale) for p in O.init_params(cfg, 0)]
    fp = [p * np.float32(weight_scale) for p in O.init_params(cfg, 1)]
    if bias_std:
        for p in cp[1::2] + fp[1::2]:
            p += rng.normal(0, bias_std, p.shape).astype(np.float32)
    return dict(cfg=cfg, o=o, d=d, t=t, u=u, img=img, cp=cp, fp=fp, N=N, focal=focal)
