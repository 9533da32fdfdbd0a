"""A tiny nerf_synthetic-layout dataset written to a temp directory (transforms_{train,val,test}.json + RGBA PNGs) for
the loader / monitor tests.  The images are procedural (a shaded disc with alpha), not renders of a 3-D scene."""
import json
import os

import numpy as np
from PIL import Image

from oracle import nerf_oracle as O


def write(root, n=(4, 2, 3), wh=24):
    os.makedirs(root, exist_ok=True)
    rng = np.random.default_rng(0)
    for subset, cnt in zip(("train", "val", "test"), n):
        os.makedirs(os.path.join(root, subset), exist_ok=True)
        frames = []
        for i in range(cnt):
            yy, xx = np.mgrid[0:wh, 0:wh]
            r = np.hypot(xx - wh / 2 + 3 * np.sin(i), yy - wh / 2) / (wh / 3)
            alpha = (r < 1).astype(np.float32)
            rgb = np.stack([1 - r, 0.5 + 0.5 * np.cos(i + r), r], -1).clip(0, 1) * alpha[..., None]
            img = (np.concatenate([rgb, alpha[..., None]], -1) * 255).astype(np.uint8)
            Image.fromarray(img, "RGBA").save(os.path.join(root, subset, f"r_{i}.png"))
            frames.append({"file_path": f"./{subset}/r_{i}", "transform_matrix": O.pose_spherical(40.0 * i, -30.0, 4.0).tolist()})
        json.dump({"camera_angle_x": 0.6911112070083618, "frames": frames}, open(os.path.join(root, f"transforms_{subset}.json"), "w"))
    return root
