"""DatasetLoader -- counterpart of reference keras_nerf/data/loader.py:12-113 for nerf_synthetic-layout directories.

`load_dataset(...)` returns [train, val, test]; each is a re-iterable `RayImageDataset` that yields
(images [B,H,W,4] float32, (ray_origin [B,H,W,3], ray_direction [B,H,W,3], sample_points [B,H,W,N])) exactly like the
reference's zipped/shuffled/batched tf.data pipeline: shuffle with a buffer of `batch_size` elements, batches of
`batch_size`, remainder dropped, ray jitter redrawn on every pass (the tf.data `map` re-executes per epoch).
Images are decoded on the host (ImageLoader); rays are generated on the GPU (RaysGenerator)."""
from __future__ import annotations

import json
import logging
import os
from typing import List

import numpy as np

from .image import ImageLoader
from .utils import get_focal_from_fov


class _Iterator:
    def __init__(self, it):
        self._it = it

    def get_next(self):
        return next(self._it)

    __next__ = get_next

    def __iter__(self):
        return self


class RayImageDataset:
    def __init__(self, image_paths, camera_params, image_loader, rays_generator_factory, batch_size, seed=0, limit=None):
        self.image_paths, self.camera_params = list(image_paths), [np.asarray(c, np.float32) for c in camera_params]
        self.image_loader, self._rg_factory, self._rg = image_loader, rays_generator_factory, None
        self.batch_size, self._rng, self._limit = batch_size, np.random.default_rng(seed), limit
        self._cache = {}

    def __len__(self):
        n = len(self.image_paths) // self.batch_size
        return n if self._limit is None else min(n, self._limit)

    def take(self, n):
        return RayImageDataset(self.image_paths, self.camera_params, self.image_loader, self._rg_factory, self.batch_size,
                               seed=int(self._rng.integers(1 << 31)), limit=n)

    def _image(self, i):
        if i not in self._cache:
            self._cache[i] = self.image_loader(self.image_paths[i])
        return self._cache[i]

    def _order(self):
        """tf.data shuffle(buffer_size): keep a buffer of `batch_size` elements, emit a random one, refill in order"""
        n, buf, out, nxt = len(self.image_paths), [], [], 0
        while nxt < n or buf:
            while nxt < n and len(buf) < self.batch_size:
                buf.append(nxt); nxt += 1
            out.append(buf.pop(int(self._rng.integers(len(buf)))))
        return out

    def __iter__(self):
        import torch
        if self._rg is None:
            self._rg = self._rg_factory()
        order, b = self._order(), self.batch_size
        nb = len(order) // b                                           # drop_remainder=True
        if self._limit is not None:
            nb = min(nb, self._limit)

        def gen():
            for k in range(nb):
                idx = order[k * b:(k + 1) * b]
                images = np.stack([self._image(i) for i in idx])
                o, d, t = self._rg(np.stack([self.camera_params[i] for i in idx]))
                yield torch.as_tensor(images).to(o.device), (o, d, t)
        return _Iterator(gen())


class DatasetLoader:
    def __init__(self, data_dir: str, white_background: bool = False, **kwargs):
        self.data_dir, self.white_background = data_dir, white_background

    def _load_json(self, filename: str) -> dict:
        with open(filename, "r") as f:
            return json.load(f)

    def _load_image_path_and_camera_param(self, json_config: dict) -> tuple:
        paths = [os.path.join(self.data_dir, f"{fr['file_path']}.png") for fr in json_config["frames"]]
        return paths, [fr["transform_matrix"] for fr in json_config["frames"]]

    def load_dataset(self, batch_size: int, image_width: int, image_height: int, near: float, far: float, n_sample: int) -> List[RayImageDataset]:
        image_loader = ImageLoader(image_width, image_height, self.white_background)
        out = []
        for k, subset in enumerate(["train", "val", "test"]):
            cfg = self._load_json(os.path.join(self.data_dir, f"transforms_{subset}.json"))
            focal = get_focal_from_fov(cfg["camera_angle_x"], image_width)

            def factory(focal=focal, k=k):
                from .rays import RaysGenerator
                return RaysGenerator(focal_length=focal, image_width=image_width, image_height=image_height, near=near, far=far,
                                     n_sample=n_sample, seed=1000 + k)
            paths, cams = self._load_image_path_and_camera_param(cfg)
            out.append(RayImageDataset(paths, cams, image_loader, factory, batch_size, seed=k))
            logging.info(f"Loaded {subset} dataset. {len(paths)} images.")
        return out
