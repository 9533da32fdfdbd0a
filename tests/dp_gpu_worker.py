"""Worker of tests/test_gpu_api.py::test_two_rank_data_parallel_step_on_the_gpu: one rank of a 2-process data-parallel
NeRF.train_step (torch.distributed, gloo -- one GPU cannot host two RCCL ranks; the collective is exchanged through the
host, everything else is the product path: library-owned gradient buffer, broadcast + refresh at compile, SUM all-reduce,
identical Adam on both ranks).  Launched by torch.distributed.run; writes <out>/rank<r>.npz."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


# (image side, n_coarse, n_fine, ray_chunks): the small problem the oracle can follow, and cfg4's per-GPU workload as BASELINE.json
# states it (chair-shaped 128 x 128, one image per GPU, coarse 64 + fine 128, ray_chunks 4096) -- KNERF_DP_SHAPE=cfg4
SHAPES = {"small": (16, 32, 32, 128), "cfg4": (128, 64, 128, 4096)}


def problem(shape="small"):
    from keras_nerf_amd.data.utils import get_focal_from_fov, pose_spherical
    wh, nc, nf, _ = SHAPES[shape]
    rng = np.random.default_rng(5)
    poses = np.stack([pose_spherical(30.0 + 70.0 * i, -30.0, 4.0) for i in range(2)])
    img = rng.random((2, wh, wh, 3), dtype=np.float32)
    u = rng.random((2, wh, wh, nf), dtype=np.float32)
    return poses, get_focal_from_fov(0.6911112070083618, wh), img, u


def main():
    out = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    from keras_nerf_amd.model.nerf.nerf import NeRF
    shape = os.environ.get("KNERF_DP_SHAPE", "small")
    wh, nc, nf, chunks = SHAPES[shape]
    nerf = NeRF(n_coarse=nc, n_fine=nf, seed=100 + rank)             # different initial weights per rank: compile() must mirror rank 0
    nerf.compile("adam", "mse", batch_size=1, image_height=wh, image_width=wh, ray_chunks=chunks, white_background=True)
    poses, focal, img, u = problem(shape)
    o, d, t = nerf._ctx.generate_rays(poses[rank:rank + 1], focal, wh, wh, 2.0, 6.0, nc, None, seed=9, stream_id=rank)
    w_start = np.concatenate([nerf._ctx.get_weights(0), nerf._ctx.get_weights(1)])
    logs = nerf.train_step((img[rank:rank + 1], (o, d, t)), u=u[rank:rank + 1])
    w_end = np.concatenate([nerf._ctx.get_weights(0), nerf._ctx.get_weights(1)])
    np.savez(os.path.join(out, f"rank{rank}.npz"), w_start=w_start, w_end=w_end, o=o.cpu().numpy(), d=d.cpu().numpy(), t=t.cpu().numpy(),
             coarse_loss=logs["coarse_loss"], fine_loss=logs["fine_loss"])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
