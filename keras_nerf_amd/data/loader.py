"""DatasetLoader -- counterpart of reference keras_nerf/data/loader.py:12-113 for nerf_synthetic-layout directories.

`load_dataset(...)` returns [train, val, test]; each is a re-iterable `RayImageDataset` that yields
(images [B,H,W,4] float32, (ray_origin [B,H,W,3], ray_direction [B,H,W,3], sample_points [B,H,W,N])) exactly like the
reference's zipped/shuffled/batched tf.data pipeline: shuffle with a buffer of `batch_size` elements, batches of
`batch_size`, remainder dropped, ray jitter redrawn on every pass (the tf.data `map` re-executes per epoch).
Images are decoded on the host (ImageLoader); rays are generated on the GPU (RaysGenerator).

Data parallel (one process per GPU, reference train.py:75-93): the reference builds ONE dataset with the GLOBAL batch
(`load_dataset(batch_size=args.batch_size * strategy.num_replicas_in_sync)`, train.py:84-93) and Keras hands replica r images
[r*b, (r+1)*b) of every global batch (b = args.batch_size, what NeRF.compile and NeRFTrainMonitor are given).  Same here, so that
an import-swapped train.py needs no change: `batch_size` IS THE GLOBAL BATCH; inside a torch.distributed job of W ranks every
rank builds the same dataset, all ranks draw the same shuffled order (one shared seed, shuffle buffer = the global batch as in
loader.py:97-113), cut it into global batches (remainder dropped) and yield their own slice of batch_size / W images (W must divide
it), so the replicas see disjoint images and the SUM all-reduce of NeRF.train_step adds gradients of different data.  The ray
jitter stream is keyed by the rank as well.  (Rounds 1-4 took the per-replica batch here; train.py passes the global one.)

Feeding the GPU (the reference prefetches with tf.data, loader.py:104-106): the decoded images and the camera matrices of a
dataset are kept RESIDENT ON THE DEVICE (100 views of 800 x 800 x 4 floats are 1 GB of 288), filled as the first pass touches
them, so from the second epoch on a batch is one gather on the device plus the ray-generation kernel -- no host-to-device copy
and, above all, no pageable copy that would make the host wait for the previous train step.  The few indices a batch needs
travel through pinned memory with a non-blocking copy.  A dataset larger than `device_cache_gb` (default 64) is staged through
two pinned buffers on a side stream instead, one batch ahead of the step that consumes it."""
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


def _dist_rank_world():
    """(rank, world size) of the torch.distributed job this process belongs to, (0, 1) outside one"""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except Exception:
        pass
    return 0, 1


def shuffled_order(n, buffer_size, rng):
    """tf.data shuffle(buffer_size): keep a buffer of `buffer_size` elements, emit a random one, refill in order"""
    buf, out, nxt = [], [], 0
    while nxt < n or buf:
        while nxt < n and len(buf) < buffer_size:
            buf.append(nxt); nxt += 1
        out.append(buf.pop(int(rng.integers(len(buf)))))
    return out


def replica_batches(order, batch_size, rank=0, world=1, limit=None):
    """index lists of this replica's batches: `order` cut into global batches of batch_size x world (drop_remainder=True),
    replica r keeps elements [r*b, (r+1)*b) of each (train.py:84-93: Keras splits the global batch along dim 0)"""
    g = batch_size * world
    nb = len(order) // g
    if limit is not None:
        nb = min(nb, limit)
    return [order[k * g + rank * batch_size:k * g + (rank + 1) * batch_size] for k in range(nb)]


class RayImageDataset:
    def __init__(self, image_paths, camera_params, image_loader, rays_generator_factory, batch_size, seed=0, limit=None,
                 rank=None, world=None, device_cache_gb=64.0, _shared=None):
        """rank / world: data-parallel placement; None = taken from torch.distributed when an iteration starts.
        device_cache_gb: keep the whole dataset on the device when it fits this budget (see the module docstring)."""
        self.image_paths, self.camera_params = list(image_paths), [np.asarray(c, np.float32) for c in camera_params]
        self.image_loader, self._rg_factory, self._rg = image_loader, rays_generator_factory, None
        self.batch_size, self._rng, self._limit = batch_size, np.random.default_rng(seed), limit
        self._rank, self._world = rank, world
        self.device_cache_gb = device_cache_gb
        # decoded host images and the device-resident copies are shared between a dataset and its take() views
        self._shared = _shared if _shared is not None else {"host": {}, "dev": None, "have": None, "cams": None}
        self._cache = self._shared["host"]

    def _placement(self):
        if self._rank is not None and self._world is not None:
            return self._rank, self._world
        return _dist_rank_world()

    def _local_batch(self, world):
        """images per step of ONE replica: the global batch split along dim 0 (train.py:84-93)"""
        if self.batch_size % world:
            raise ValueError(f"the global batch of {self.batch_size} images is not divisible by {world} replicas "
                             f"(train.py:84: global_batch_size = batch_size x num_replicas_in_sync)")
        return self.batch_size // world

    def __len__(self):
        n = len(self.image_paths) // self.batch_size          # global batches; every replica takes part in each
        return n if self._limit is None else min(n, self._limit)

    def take(self, n):
        return RayImageDataset(self.image_paths, self.camera_params, self.image_loader, self._rg_factory, self.batch_size,
                               seed=int(self._rng.integers(1 << 31)), limit=n, rank=self._rank, world=self._world,
                               device_cache_gb=self.device_cache_gb, _shared=self._shared)

    def private_view(self, seed=0, limit=None):
        """the same data behind an INDEPENDENT shuffle stream: iterating the view never advances this dataset's generator.
        For consumers that iterate on one rank only (NeRFTrainMonitor on rank 0), so that the ranks' shared order of THIS
        dataset stays in lock-step."""
        return RayImageDataset(self.image_paths, self.camera_params, self.image_loader, self._rg_factory, self.batch_size,
                               seed=seed, limit=self._limit if limit is None else limit, rank=self._rank, world=self._world,
                               device_cache_gb=self.device_cache_gb, _shared=self._shared)

    def _image(self, i):
        if i not in self._cache:
            self._cache[i] = self.image_loader(self.image_paths[i])
        return self._cache[i]

    # ---- device-resident copy
    def _resident(self, shape):
        """(images [N,H,W,4], have [N] host bools, cams [N,4,4]) on the device, or None when the dataset exceeds the budget"""
        import torch
        sh = self._shared
        if sh["dev"] is None and sh.get("dev_refused") is None:
            n = len(self.image_paths)
            if n * int(np.prod(shape)) * 4 > self.device_cache_gb * 1e9:
                sh["dev_refused"] = True
            else:
                sh["dev"] = torch.empty((n, *shape), device="cuda", dtype=torch.float32)
                sh["have"] = np.zeros(n, bool)
                sh["cams"] = torch.as_tensor(np.stack(self.camera_params)).pin_memory().to("cuda", non_blocking=True)
        return None if sh["dev"] is None else (sh["dev"], sh["have"], sh["cams"])

    @staticmethod
    def _to_device_async(array):
        """small host array -> device without a pageable copy (which would make the host wait for the stream's earlier work)"""
        import torch
        return torch.as_tensor(array).pin_memory().to("cuda", non_blocking=True)

    def __iter__(self):
        import torch
        rank, world = self._placement()
        if self._rg is None:
            self._rg = self._rg_factory(rank)
        # the same order on every rank (the rngs are seeded alike and advance alike), each rank keeps its slice
        order = shuffled_order(len(self.image_paths), self.batch_size, self._rng)
        batches = replica_batches(order, self._local_batch(world), rank, world, self._limit)

        def gen_resident(res):
            dev, have, cams = res
            for idx in batches:
                missing = [i for i in idx if not have[i]]
                if missing:                                   # first touch: decode on the host, one pinned non-blocking upload
                    up = self._to_device_async(np.stack([self._image(i) for i in missing]))
                    dev[self._to_device_async(np.asarray(missing, np.int64))] = up
                    have[missing] = True
                    for i in missing:
                        self._cache.pop(i, None)              # the device copy is the cache now
                sel = self._to_device_async(np.asarray(idx, np.int64))
                o, d, t = self._rg(cams.index_select(0, sel))
                yield dev.index_select(0, sel), (o, d, t)

        def gen_staged():
            # two pinned staging buffers, copies on a side stream one batch ahead of the consumer
            side = torch.cuda.Stream()
            main = torch.cuda.current_stream()
            pinned, copied, pending = [None, None], [None, None], None

            def stage(k, idx):
                images = np.stack([self._image(i) for i in idx])
                if copied[k] is not None:
                    copied[k].synchronize()                   # the buffer's previous upload (two batches ago) has left it
                if pinned[k] is None or pinned[k].shape != images.shape:
                    pinned[k] = torch.empty(images.shape, dtype=torch.float32).pin_memory()
                pinned[k].copy_(torch.as_tensor(images))
                with torch.cuda.stream(side):
                    dev = pinned[k].to("cuda", non_blocking=True)
                    ev = torch.cuda.Event(); ev.record(side)
                copied[k] = ev
                return dev, ev
            for n, idx in enumerate(batches):
                cur = pending if pending is not None else stage(n & 1, idx)
                pending = stage((n + 1) & 1, batches[n + 1]) if n + 1 < len(batches) else None
                dev, ev = cur
                main.wait_event(ev)
                dev.record_stream(main)
                o, d, t = self._rg(self._to_device_async(np.stack([self.camera_params[i] for i in idx])))
                yield dev, (o, d, t)

        if not batches:
            return _Iterator(iter(()))
        res = self._resident(self._image(batches[0][0]).shape)
        return _Iterator(gen_resident(res) if res is not None else gen_staged())


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
        """loader.py:55-113.  batch_size: the GLOBAL batch (train.py:84-93 passes batch_size x replicas; train_single.py:60-67 the
        only batch there is); a rank of a data-parallel job yields its 1/world slice of every batch"""
        image_loader = ImageLoader(image_width, image_height, self.white_background)
        out = []
        for k, subset in enumerate(["train", "val", "test"]):
            cfg = self._load_json(os.path.join(self.data_dir, f"transforms_{subset}.json"))
            focal = get_focal_from_fov(cfg["camera_angle_x"], image_width)

            def factory(rank=0, focal=focal, k=k):
                from .rays import RaysGenerator
                return RaysGenerator(focal_length=focal, image_width=image_width, image_height=image_height, near=near, far=far,
                                     n_sample=n_sample, seed=1000 + k + 4096 * rank)       # replicas jitter differently
            paths, cams = self._load_image_path_and_camera_param(cfg)
            out.append(RayImageDataset(paths, cams, image_loader, factory, batch_size, seed=k))
            logging.info(f"Loaded {subset} dataset. {len(paths)} images.")
        return out
