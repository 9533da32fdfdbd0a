#!/usr/bin/env python3
"""Where a cfg5 frame's time goes under the two read-back forms of bench.py bench_render (one GPU box, one process):
round 3's loop verbatim (full dictionaries, `.cpu().numpy()` of image and depth), the same with only the two outputs computed, and
round 4's pipelined read-back.  Prints ms per frame of each, twice (alternating)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from keras_nerf_amd.data.rays import RaysGenerator  # noqa: E402
from keras_nerf_amd.data.utils import get_focal_from_fov, pose_spherical  # noqa: E402
from keras_nerf_amd.model.nerf.nerf import NeRF  # noqa: E402

wh, n = 256, 60
nerf = NeRF(seed=0)
nerf.compile("adam", "mse", batch_size=1, image_height=wh, image_width=wh, ray_chunks=4096, white_background=True, is_training=False)
rg = RaysGenerator(get_focal_from_fov(0.6911112070083618, wh), wh, wh, 2.0, 6.0, nerf.n_coarse, seed=0)
poses = [pose_spherical(360.0 * i / n, -30.0, 4.0) for i in range(n)]


def r03(i):
    o, d, t = rg(poses[i])
    _, fine = nerf.predict_and_render_images((o[None], d[None], t[None]))
    return fine["image"].cpu().numpy(), fine["depth"].cpu().numpy()


def two_outputs_blocking(i):
    o, d, t = rg(poses[i])
    _, fine = nerf.predict_and_render_images((o[None], d[None], t[None]), outputs=("image", "depth"))
    return fine["image"].cpu().numpy(), fine["depth"].cpu().numpy()


side = torch.cuda.Stream()
pinned = [(torch.empty((1, wh, wh, 3), pin_memory=True), torch.empty((1, wh, wh), pin_memory=True)) for _ in range(2)]
done = [torch.cuda.Event(), torch.cuda.Event()]


def pipelined_loop():
    out = []
    for i in range(n):
        k = i & 1
        o, d, t = rg(poses[i])
        _, fine = nerf.predict_and_render_images((o[None], d[None], t[None]), outputs=("image", "depth"))
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            pinned[k][0].copy_(fine["image"], non_blocking=True); pinned[k][1].copy_(fine["depth"], non_blocking=True)
            fine["image"].record_stream(side); fine["depth"].record_stream(side)
            done[k].record(side)
        if i:
            done[(i - 1) & 1].synchronize(); out.append((pinned[(i - 1) & 1][0].numpy().copy(), pinned[(i - 1) & 1][1].numpy().copy()))
    done[(n - 1) & 1].synchronize(); out.append((pinned[(n - 1) & 1][0].numpy().copy(), pinned[(n - 1) & 1][1].numpy().copy()))
    return out


def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r


res = {}
for rep in range(2):
    for name, fn in (("r03_full_dict_blocking", lambda: [r03(i) for i in range(n)]), ("two_outputs_blocking", lambda: [two_outputs_blocking(i) for i in range(n)]),
                     ("two_outputs_pipelined", pipelined_loop)):
        for i in range(3):
            r03(i) if name.startswith("r03") else two_outputs_blocking(i)
        ms, _ = timed(fn)
        res.setdefault(name, []).append(round(ms, 3))
print(json.dumps({"ms_per_frame": res, "frames_per_s": {k: [round(1e3 / x, 2) for x in v] for k, v in res.items()}}))
