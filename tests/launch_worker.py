"""Rank bodies for tests/test_dp_gloo.py::test_launch_*: importable by the spawned rank processes (keras_nerf_amd/parallel.py launch
uses multiprocessing "spawn", so the function must live in a module), and runnable as a SCRIPT for the MirroredStrategy() form --
`python tests/launch_worker.py OUT N` re-runs itself as N ranks the way an import-swapped train.py would (reference train.py:75)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def body(out_dir, fail_rank=-1):
    """what a training script does with its process group: rank 0's "weights" everywhere, one SUM all-reduce, replica-mean logs"""
    import torch
    import torch.distributed as dist
    from keras_nerf_amd import parallel
    rank, world = dist.get_rank(), dist.get_world_size()
    assert parallel.is_distributed() and parallel.rank() == rank and int(os.environ["LOCAL_RANK"]) == rank
    if rank == fail_rank:
        raise RuntimeError(f"rank {rank} fails on purpose")
    w = [torch.full((1000,), float(rank + 1))]
    parallel.broadcast_weights(w)
    g = torch.arange(4096, dtype=torch.float32) * (rank + 1)
    parallel.all_reduce_gradients(g, "sum")
    logs = parallel.reduce_logs({"loss": float(rank)})
    parallel.barrier()
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump({"rank": rank, "world": world, "w": float(w[0][0]), "g1": float(g[1]), "loss": logs["loss"], "pid": os.getpid(),
                   "master": os.environ["MASTER_ADDR"], "ipc": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                   "nccl_debug": os.environ.get("NCCL_DEBUG"), "is_main": parallel.is_main(), **parallel.launch_fields()}, f)
    if rank == 0:                # a benchmark's one line: a job that needed the launcher's second attempt says so
        print(json.dumps({"n_ranks": world, **parallel.launch_fields()}), flush=True)


def hang_unless_zero(out_dir):
    """rank 1 raises at once; the others would sit in a collective for ever if the launcher did not stop them"""
    import torch
    import torch.distributed as dist
    if dist.get_rank() == 1:
        raise RuntimeError("rank 1 fails on purpose")
    open(os.path.join(out_dir, f"alive{dist.get_rank()}"), "w").close()
    dist.all_reduce(torch.ones(1))           # never completes: rank 1 is gone


def record_pid_and_wait(out_dir):
    """a healthy rank that is simply busy: the launcher is what gets the signal"""
    import time
    import torch.distributed as dist
    with open(os.path.join(out_dir, f"pid{dist.get_rank()}"), "w") as f:
        f.write(str(os.getpid()))
    time.sleep(600)


def ignore_sigterm_and_hang(out_dir):
    """a rank that cannot be talked out of its collective: SIGTERM ignored (as inside a driver call) -- the launcher must escalate"""
    import signal
    import torch
    import torch.distributed as dist
    signal.signal(signal.SIGTERM, signal.SIG_IGN)
    if dist.get_rank() == 1:
        raise RuntimeError("rank 1 fails on purpose")
    with open(os.path.join(out_dir, f"pid{dist.get_rank()}"), "w") as f:
        f.write(str(os.getpid()))
    dist.all_reduce(torch.ones(1))


if __name__ == "__main__":
    from keras_nerf_amd import parallel
    out, n = sys.argv[1], int(sys.argv[2])
    strategy = parallel.MirroredStrategy(devices=n, backend="gloo")      # the parent never gets past this line: it becomes the launcher
    with strategy.scope():
        assert strategy.num_replicas_in_sync == n
        body(out)
