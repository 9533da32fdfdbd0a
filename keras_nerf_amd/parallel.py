"""Data-parallel glue: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).

The reference's only parallelism is tf.distribute.MirroredStrategy (train.py:75-157): every replica holds both MLPs,
processes `batch_size` whole images of the global batch, and the optimizer SUMs the accumulated gradients across
replicas inside apply_gradients (nerf.py:455-458).  Here that is ONE all-reduce of the flat [coarse | fine] gradient
buffer (2 x 595,844 fp32 = 4.77 MB) per step, issued after the last chunk and before the two Adam updates; rays never
cross GPUs, so there is no other data-path collective.  These helpers are backend-agnostic (they run under gloo in the
CPU tests)."""
from __future__ import annotations

from typing import Sequence

import torch
import torch.distributed as dist


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def rank() -> int:
    return dist.get_rank() if is_distributed() else 0


def is_main() -> bool:
    """the one process that writes logs, images and checkpoints (replicas hold identical weights and logs)"""
    return rank() == 0


def barrier() -> None:
    if is_distributed():
        dist.barrier()


def all_reduce_gradients(flat_grads: torch.Tensor, mode: str = "sum") -> torch.Tensor:
    """In-place all-reduce of the flat gradient accumulator.  mode 'sum' = Keras/MirroredStrategy semantics (the applied
    gradient is world_size x the replica mean; train.py:130-136 leaves the 1/global_batch factor commented out);
    'mean' divides by the world size."""
    if mode not in ("sum", "mean"):
        raise ValueError("mode must be 'sum' or 'mean'")
    if is_distributed():
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM)
        if mode == "mean":
            flat_grads.div_(dist.get_world_size())
    return flat_grads


def broadcast_weights(flat_weights: Sequence[torch.Tensor], src: int = 0) -> None:
    """mirrored variables start identical on every replica (strategy.scope(), train.py:110-148)"""
    if is_distributed():
        for w in flat_weights:
            dist.broadcast(w, src=src)


def shard_batch(global_batch, rank: int, world_size: int):
    """Keras splits each global batch along dim 0: replica r gets images [r*b, (r+1)*b) (train.py:84-93)."""
    n = global_batch.shape[0]
    if n % world_size:
        raise ValueError(f"global batch {n} is not divisible by {world_size} replicas")
    b = n // world_size
    return global_batch[rank * b:(rank + 1) * b]


def reduce_logs(logs: dict, device=None) -> dict:
    """logged scalars are replica means (Keras Mean metrics under MirroredStrategy): one small all-reduce"""
    if not is_distributed() or not logs:
        return logs
    keys = sorted(logs)
    vec = torch.tensor([float(logs[k]) for k in keys], dtype=torch.float64, device=device)
    dist.all_reduce(vec)
    return {k: float(v) / dist.get_world_size() for k, v in zip(keys, vec)}
