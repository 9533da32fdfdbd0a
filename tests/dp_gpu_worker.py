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


def problem():
    from keras_nerf_amd.data.utils import get_focal_from_fov, pose_spherical
    rng = np.random.default_rng(5)
    poses = np.stack([pose_spherical(30.0 + 70.0 * i, -30.0, 4.0) for i in range(2)])
    img = rng.random((2, 16, 16, 3), dtype=np.float32)
    u = rng.random((2, 16, 16, 32), dtype=np.float32)
    return poses, get_focal_from_fov(0.6911112070083618, 16), img, u


def main():
    out = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    from keras_nerf_amd.model.nerf.nerf import NeRF
    nerf = NeRF(n_coarse=32, n_fine=32, seed=100 + rank)             # different initial weights per rank: compile() must mirror rank 0
    nerf.compile("adam", "mse", batch_size=1, image_height=16, image_width=16, ray_chunks=128, white_background=True)
    poses, focal, img, u = problem()
    o, d, t = nerf._ctx.generate_rays(poses[rank:rank + 1], focal, 16, 16, 2.0, 6.0, 32, None, seed=9, stream_id=rank)
    w_start = np.concatenate([nerf._ctx.get_weights(0), nerf._ctx.get_weights(1)])
    logs = nerf.train_step((img[rank:rank + 1], (o, d, t)), u=u[rank:rank + 1])
    w_end = np.concatenate([nerf._ctx.get_weights(0), nerf._ctx.get_weights(1)])
    np.savez(os.path.join(out, f"rank{rank}.npz"), w_start=w_start, w_end=w_end, o=o.cpu().numpy(), d=d.cpu().numpy(), t=t.cpu().numpy(),
             coarse_loss=logs["coarse_loss"], fine_loss=logs["fine_loss"])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
