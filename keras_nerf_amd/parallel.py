"""Data-parallel glue: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).

The reference's only parallelism is tf.distribute.MirroredStrategy (train.py:75-157): every replica holds both MLPs,
processes `batch_size` whole images of the global batch, and the optimizer SUMs the accumulated gradients across
replicas inside apply_gradients (nerf.py:455-458).  Here that is ONE all-reduce of the flat [coarse | fine] gradient
buffer (2 x 595,844 fp32 = 4.77 MB) per step, issued after the last chunk and before the two Adam updates; rays never
cross GPUs, so there is no other data-path collective.  These helpers are backend-agnostic (they run under gloo in the
CPU tests)."""
from __future__ import annotations

import os
import sys
from typing import Sequence

import torch
import torch.distributed as dist


def single_rank_rehearsal() -> bool:
    """KNERF_DIST_SINGLE=1: a ONE-rank process group counts as distributed -- every collective of the N > 1 path (weight broadcast,
    gradient all-reduce, log means, barriers) really runs, over the real backend, on a box with one GPU.  RCCL refuses two ranks on
    one device, so this is the only way its code path executes before an 8-GPU node does it for the first time (bench.py, tests)."""
    return os.environ.get("KNERF_DIST_SINGLE", "") not in ("", "0")


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or single_rank_rehearsal())


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


# ---------------------------------------------------------------------------------------------------------------------
# Launching: what `tf.distribute.MirroredStrategy()` does for the reference (train.py:75-93, 110-148) -- plain `python train.py`
# uses every GPU -- as one PROCESS per GPU.  Rules of this pool (and good manners anywhere): rank processes are CHILDREN started
# before the parent has touched the GPU; a process that has initialised HIP never execs; a failing rank names itself and the
# launcher stops the others by their own PIDs (no pattern kills, no collective time-outs).
# ---------------------------------------------------------------------------------------------------------------------
RANK_FAILED = 3          # exit code of a rank that raised (and of the launcher that saw it)
RANK_FAILED_SETUP = 4    # ... of a rank whose process group never worked (init_process_group / the first all_reduce): the one
                         # failure the launcher answers with a second set of ranks under the other IPC setting (launch)
IPC_VAR = "HSA_ENABLE_IPC_MODE_LEGACY"


def dist_env(world: int = 1) -> None:
    """The environment every rank needs, whichever way it was started (self-spawned or under an external torch.distributed.run):
    rendezvous on 127.0.0.1 (a container's hostname may not resolve) and dmabuf IPC (this host driver supports nothing else:
    without it RCCL fails with `hipIpcGetMemHandle: invalid argument`).  Called before anything touches HIP; existing values win.
    world > 1: RCCL's warnings go to one file per rank (NCCL_DEBUG=WARN, NCCL_DEBUG_FILE), whose tail `rank_fail` prints."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if os.environ.get("KNERF_LAUNCH_IPC") != "unset":          # the launcher's second attempt runs WITHOUT the variable (launch)
        os.environ.setdefault(IPC_VAR, "0")
    if world > 1 or single_rank_rehearsal():
        os.environ.setdefault("NCCL_DEBUG", "WARN")
        if "NCCL_DEBUG_FILE" not in os.environ:
            # one file per rank: in the launcher's job directory when there is one (removed with it after a clean job), else under
            # TMPDIR and removed by this process when it leaves normally (rank_fail leaves through os._exit: the file stays)
            os.environ["NCCL_DEBUG_FILE"] = os.path.join(os.environ.get("KNERF_LAUNCH_DIR") or os.environ.get("TMPDIR", "/tmp"), "knerf_rccl_%h_%p.log")
            if not os.environ.get("KNERF_LAUNCH_DIR"):
                import atexit
                atexit.register(_remove_own_rccl_log, os.getpid())


def _own_rccl_log() -> str:
    pat = os.environ.get("NCCL_DEBUG_FILE", "")
    if not pat:
        return ""
    import socket
    return pat.replace("%h", socket.gethostname()).replace("%p", str(os.getpid()))


def _remove_own_rccl_log(pid: int) -> None:
    if pid == os.getpid():                 # not in a forked child that inherited the handler
        try:
            os.remove(_own_rccl_log())
        except OSError:
            pass


def rccl_log_tail(max_bytes: int = 4000) -> str:
    """the tail of THIS process's RCCL debug file (dist_env), or '' -- NCCL_DEBUG_FILE's %h / %p are host name and pid"""
    path = _own_rccl_log()
    if not path:
        return ""
    try:
        with open(path, "rb") as f:
            f.seek(0, 2); n = f.tell(); f.seek(max(0, n - max_bytes))
            return f.read().decode(errors="replace")
    except OSError:
        return ""


def rank_fail(what: str, exc: BaseException, tag: str = "knerf", code: int = RANK_FAILED) -> None:
    """A rank that cannot join or use the process group says who and where it is and leaves with exit code 3 (4 = the group never
    worked, RANK_FAILED_SETUP) -- the launcher then stops the other ranks -- instead of letting them sit in a collective until its
    time-out.  Never re-execs (the process may have initialised the GPU).  Prints the tail of this rank's RCCL warnings when there
    are any; under `launch` the same text is left in the job directory, for the line of a second attempt to quote."""
    import traceback
    r, w = os.environ.get("RANK", "0"), os.environ.get("WORLD_SIZE", "1")
    n_dev = torch.cuda.device_count()          # no GPU initialisation: the device this rank was (or would have been) given
    dev = f"cuda:{int(os.environ.get('LOCAL_RANK', '0')) % n_dev}" if n_dev else "cpu"
    backend = dist.get_backend() if dist.is_available() and dist.is_initialized() else os.environ.get("KNERF_DIST_BACKEND", "nccl")
    tb = "".join(traceback.format_exception(type(exc), exc, exc.__traceback__))
    tail = rccl_log_tail()
    text = (f"[{tag} rank {r}/{w}] FAILED in {what} on {dev} (backend {backend}, {IPC_VAR}={os.environ.get(IPC_VAR, '<unset>')}): "
            f"{type(exc).__name__}: {exc}\n{tb}" + (f"[{tag} rank {r}/{w}] RCCL log tail ({os.environ.get('NCCL_DEBUG', '')}):\n{tail}\n" if tail else ""))
    if code == RANK_FAILED_SETUP and not os.environ.get("KNERF_LAUNCH_DIR"):
        # under an external launcher (torch.distributed.run) nothing is retried: say what the self-spawning launcher would have tried
        alt = "unset" if os.environ.get(IPC_VAR) == "0" else "0"
        text += (f"[{tag} rank {r}/{w}] hint: the process group never came up under {IPC_VAR}={os.environ.get(IPC_VAR, '<unset>')}; if the log above names "
                 f"hipIpc* / IPC handles, rerun with {IPC_VAR} {alt} (`python bench.py --gpus N` without a launcher tries that by itself, once)\n")
    print(text, file=sys.stderr, flush=True)
    job = os.environ.get("KNERF_LAUNCH_DIR")
    if job:
        try:
            with open(os.path.join(job, f"fail_attempt{os.environ.get('KNERF_LAUNCH_ATTEMPT', '1')}_rank{r}.txt"), "w") as f:
                f.write(text)
        except OSError:
            pass
    sys.stderr.flush(); sys.stdout.flush()
    os._exit(code)


def inject(stage: str) -> None:
    """fault injection for the fail-fast tests: KNERF_INJECT_FAILURE (or KNERF_BENCH_INJECT_FAILURE) = "<rank>:<stage>" makes that
    rank raise at that stage (init | first_all_reduce | compile | warmup | body); "<rank>:<stage>@<k>" only in the launcher's k-th
    attempt, "*" = every rank"""
    spec = os.environ.get("KNERF_INJECT_FAILURE") or os.environ.get("KNERF_BENCH_INJECT_FAILURE", "")
    spec, _, attempt = spec.partition("@")
    if attempt and attempt != os.environ.get("KNERF_LAUNCH_ATTEMPT", "1"):
        return
    if spec and spec in (f"{os.environ.get('RANK', '0')}:{stage}", f"*:{stage}"):
        raise RuntimeError(f"injected failure at stage '{stage}'")


def init_rank(backend: str = None, timeout_s: float = 300.0, tag: str = "knerf") -> tuple:
    """Join the process group as the rank the environment describes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*: what
    torch.distributed.run and `launch` set) and prove it works with a one-word all-reduce (RCCL builds its rings over xGMI in
    there).  Returns (rank, world, device_index or None).  World size 1: nothing to join.  Every failure ends in rank_fail."""
    world = int(os.environ.get("WORLD_SIZE", "1")); rank_ = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = backend or os.environ.get("KNERF_DIST_BACKEND", "nccl")
    dist_env(world)
    n_dev = torch.cuda.device_count()              # counts devices without initialising the GPU
    dev = None
    if n_dev:
        if world > 1 and backend == "nccl" and local >= n_dev:
            raise SystemExit(f"rank {rank_}: LOCAL_RANK {local} but only {n_dev} GPUs visible (RCCL needs one GPU per rank; "
                             f"KNERF_DIST_BACKEND=gloo rehearses the control flow with ranks sharing devices)")
        dev = local % n_dev
        torch.cuda.set_device(dev)
    elif backend == "nccl" and world > 1:
        raise SystemExit("no GPU visible: the nccl (RCCL) backend needs one per rank")
    if (world == 1 and not single_rank_rehearsal()) or (dist.is_available() and dist.is_initialized()):
        return rank_, world, dev
    if world == 1 and "MASTER_PORT" not in os.environ:      # the one-rank rehearsal has no launcher that picked a port
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
    if dev is not None:      # which device every rank really sits on (a job that silently shares devices would still train)
        pr = torch.cuda.get_device_properties(dev)
        pci = ":".join(f"{getattr(pr, k):02x}" for k in ("pci_domain_id", "pci_bus_id", "pci_device_id") if hasattr(pr, k)) or "n/a"
        print(f"[{tag} rank {rank_}/{world}] local_rank {local} -> cuda:{dev} {pr.name} pci {pci} uuid {getattr(pr, 'uuid', 'n/a')} "
              f"backend {backend} visible_devices {n_dev}", file=sys.stderr, flush=True)
    import datetime
    try:
        inject("init")
        kw = {"device_id": torch.device("cuda", dev)} if backend == "nccl" else {}
        dist.init_process_group(backend, timeout=datetime.timedelta(seconds=timeout_s), **kw)
    except Exception as e:                         # noqa: BLE001 -- whatever the backend raises: say which rank and leave
        rank_fail("init_process_group", e, tag, RANK_FAILED_SETUP)
    try:
        inject("first_all_reduce")
        probe = torch.ones(1, device="cuda" if dev is not None else "cpu")
        dist.all_reduce(probe)
        if dev is not None:
            torch.cuda.synchronize()
        if int(probe[0]) != world:
            raise RuntimeError(f"all_reduce(1) over {world} ranks returned {float(probe[0])}")
    except Exception as e:                         # noqa: BLE001
        rank_fail("the first all_reduce", e, tag, RANK_FAILED_SETUP)
    return rank_, world, dev


def _free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _wait_ranks(procs, poll, kill, what: str, hard_kill=None, grace_s: float = 10.0) -> int:
    """wait for rank processes; the first one that fails ends the job: the others (still blocked in a collective with the dead
    rank) are terminated BY HANDLE (SIGTERM, then `hard_kill` = SIGKILL for those that have not left after `grace_s` seconds -- a
    rank waiting inside a driver call does not see SIGTERM).  The same when the launcher itself is interrupted (Ctrl-C, SIGTERM
    turned into an exception): no orphan ranks.  Returns the job's exit code."""
    import time
    code = 0
    live = list(procs)
    deadline = None
    try:
        while live:
            for p in list(live):
                rc = poll(p)
                if rc is None:
                    continue
                live.remove(p)
                if rc != 0 and code == 0:
                    code = rc if rc > 0 else RANK_FAILED
                    print(f"[knerf launch] a rank of {what} exited with code {rc}: stopping the other {len(live)}", file=sys.stderr, flush=True)
                    for q in live:
                        kill(q)
                    deadline = time.time() + grace_s
            if live and deadline is not None and time.time() > deadline and hard_kill is not None:
                for q in live:
                    hard_kill(q)
                deadline = time.time() + grace_s
            if live:
                time.sleep(0.05)
    except BaseException:                          # the launcher is going down (KeyboardInterrupt, SystemExit from a signal handler): take the ranks along
        for q in live:
            if poll(q) is None:
                kill(q)
        t_end = time.time() + grace_s
        while time.time() < t_end and any(poll(q) is None for q in live):
            time.sleep(0.05)
        if hard_kill is not None:
            for q in live:
                if poll(q) is None:
                    hard_kill(q)
        raise
    return code


def _rank_entry(fn, args, rank_: int, world: int, port: int, backend: str):
    os.environ.update(RANK=str(rank_), LOCAL_RANK=str(rank_), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_PORT=str(port))
    if backend:
        os.environ["KNERF_DIST_BACKEND"] = backend
    init_rank(backend)
    try:
        inject("body")
        fn(*args)
    except Exception as e:                         # noqa: BLE001 -- its peers would meet it as "connection closed by peer"
        rank_fail(f"{getattr(fn, '__name__', 'fn')}()", e)
    if dist.is_initialized():
        dist.destroy_process_group()


def launch(fn=None, nprocs: int = None, args=(), backend: str = None) -> int:
    """Run a training job on every GPU of this node, one process per GPU -- the MirroredStrategy() of this implementation.

    launch(fn, nprocs, args): `fn(*args)` runs in `nprocs` fresh processes (multiprocessing "spawn": `fn` must be importable, i.e.
        defined at module level, and the call sits under `if __name__ == "__main__":`); each has joined the process group
        (init_rank: RCCL, or gloo when backend / KNERF_DIST_BACKEND says so) and has its GPU selected, so NeRF.compile finds
        torch.distributed initialised: it broadcasts rank 0's weights and train_step all-reduces the gradients (train.py:110-148).
    launch() without fn: re-run THIS script (sys.argv) as the ranks and exit with their code -- for an import-swapped train.py
        that should behave like the reference's: put `parallel.launch()` (or `strategy = parallel.MirroredStrategy()`) where
        train.py:75 builds its strategy.  Inside a rank it returns at once, after joining the process group.

    nprocs None = torch.cuda.device_count() (which does not initialise the GPU).  With one device (or nprocs 1) nothing is
    spawned: fn runs here / the script continues.  Inside a rank of ANY launcher (WORLD_SIZE set, e.g. torch.distributed.run) no
    second level is spawned either.  Returns the job's exit code (0) -- a failed rank raises SystemExit(code) in the parent."""
    if "WORLD_SIZE" in os.environ:                 # already a rank: join the group, run, leave
        init_rank(backend)
        if fn is not None:
            try:
                fn(*args)
            except Exception as e:                 # noqa: BLE001
                if int(os.environ["WORLD_SIZE"]) > 1:
                    rank_fail(f"{getattr(fn, '__name__', 'fn')}()", e)
                raise
        return 0
    backend = backend or os.environ.get("KNERF_DIST_BACKEND", "nccl")
    dist_env(1)                                    # rendezvous address and IPC mode BEFORE the first call into torch.cuda, whatever it initialises
    n_dev = torch.cuda.device_count()
    n = int(nprocs) if nprocs is not None else max(n_dev, 1)
    if n < 1:
        raise ValueError("nprocs must be >= 1")
    if n == 1:
        dist_env(1)
        if single_rank_rehearsal():                # KNERF_DIST_SINGLE: this process IS the one rank of a real process group
            init_rank(backend)
        if fn is not None:
            fn(*args)
        return 0
    if backend == "nccl" and n > n_dev:
        raise SystemExit(f"launch: {n} ranks but only {n_dev} GPU(s) visible; RCCL needs one GPU per rank "
                         f"(KNERF_DIST_BACKEND=gloo rehearses the control flow with ranks sharing devices)")
    if torch.cuda.is_available() and torch.cuda.is_initialized():
        raise RuntimeError("launch must be called before this process touches the GPU (rank processes are started first)")
    what = os.path.basename(sys.argv[0]) if fn is None else getattr(fn, "__name__", "fn")
    code = _launch_attempts(fn, tuple(args), n, backend, what)
    if fn is None:
        raise SystemExit(code)                     # the parent was only the launcher: the script's body ran in the ranks
    if code:
        raise SystemExit(code)
    return 0


def _start_ranks(fn, args, n: int, backend: str, port: int):
    """one fresh process per rank, environment taken from os.environ as it is NOW; returns (procs, poll, kill, hard_kill)"""
    if fn is None:
        import subprocess
        procs = []
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_PORT=str(port),
                       KNERF_DIST_BACKEND=backend)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(sys.argv[0]), *sys.argv[1:]], env=env))
        return procs, (lambda p: p.poll()), (lambda p: p.terminate()), (lambda p: p.kill())
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_rank_entry, args=(fn, args, r, n, port, backend), daemon=False) for r in range(n)]
    for p in procs:
        p.start()
    return procs, (lambda p: p.exitcode if not p.is_alive() else None), (lambda p: p.terminate()), (lambda p: p.kill())


def first_attempt_failure(max_chars: int = 1500):
    """inside a rank of the launcher's SECOND attempt: what the first attempt's failing rank printed (its exception and the tail
    of its RCCL log), shortened -- bench.py puts it on its line.  None in a first attempt or outside `launch`."""
    path = os.environ.get("KNERF_LAUNCH_FIRST_FAILURE")
    if not path:
        return None
    try:
        with open(path) as f:
            text = f.read()
    except OSError:
        return "(the first attempt left no report)"
    return text if len(text) <= max_chars else text[:max_chars // 3] + "\n[...]\n" + text[-2 * max_chars // 3:]


def launch_fields() -> dict:
    """for a benchmark line: which attempt of the launcher this job is and the IPC setting it runs under"""
    return {"launch_attempts": int(os.environ.get("KNERF_LAUNCH_ATTEMPT", "1")), "ipc_mode_legacy_env": os.environ.get(IPC_VAR),
            "first_attempt_failure": first_attempt_failure()}


def _launch_attempts(fn, args, n: int, backend: str, what: str) -> int:
    """Start the ranks; when the job ends because its process group never worked (the first rank to fail left with
    RANK_FAILED_SETUP: init_process_group or the first all_reduce), start ONE more, entirely fresh, set of ranks under the other
    IPC setting.  dist_env defaults HSA_ENABLE_IPC_MODE_LEGACY=0 because this pool's host driver supports dmabuf IPC only; a node
    with another driver may need the variable absent, and the first N > 1 run there has nobody to try that by hand.  This
    process never touches the GPU, nothing execs, the first set has been stopped by handle before the second starts.  The ranks
    learn the attempt number and the first failure's text from the environment (launch_fields) -- a job that needed two attempts
    says so on its line.  Any other failure, and a second set-up failure, end the job as before."""
    import shutil
    import signal
    import tempfile
    job = tempfile.mkdtemp(prefix="knerf_job_")
    saved = {k: os.environ.get(k) for k in (IPC_VAR, "KNERF_LAUNCH_DIR", "KNERF_LAUNCH_ATTEMPT", "KNERF_LAUNCH_IPC", "KNERF_LAUNCH_FIRST_FAILURE",
                                            "NCCL_DEBUG", "NCCL_DEBUG_FILE")}

    def on_term(signum, frame):                    # a plain SIGTERM must not orphan the ranks: _wait_ranks stops them on its way out
        raise SystemExit(128 + signum)
    try:
        old_term = signal.signal(signal.SIGTERM, on_term)
    except ValueError:                             # not the main thread: the caller keeps its own handling
        old_term = None
    code = 0
    try:
        os.environ["KNERF_LAUNCH_DIR"] = job
        if os.environ.get("NCCL_DEBUG_FILE", "").endswith("knerf_rccl_%h_%p.log"):     # dist_env's own default, set before the job had a directory
            del os.environ["NCCL_DEBUG_FILE"]
        for attempt in (1, 2):
            os.environ["KNERF_LAUNCH_ATTEMPT"] = str(attempt)
            if attempt == 2:                       # the other setting: absent where it was "0", "0" where it was something else
                if os.environ.get(IPC_VAR) == "0":
                    del os.environ[IPC_VAR]; os.environ["KNERF_LAUNCH_IPC"] = "unset"
                else:
                    os.environ[IPC_VAR] = "0"
                reports = sorted(f for f in os.listdir(job) if f.startswith("fail_attempt1_"))
                os.environ["KNERF_LAUNCH_FIRST_FAILURE"] = os.path.join(job, reports[0]) if reports else os.path.join(job, "none")
                print(f"[knerf launch] the process group of {what} never came up (exit code {code}): ONE more attempt with fresh ranks "
                      f"and {IPC_VAR}={os.environ.get(IPC_VAR, '<unset>')}", file=sys.stderr, flush=True)
            dist_env(n)
            port = (int(os.environ.get("MASTER_PORT", 0)) if attempt == 1 else 0) or _free_port()
            procs, poll, kill, hard_kill = _start_ranks(fn, args, n, backend, port)
            code = _wait_ranks(procs, poll, kill, what, hard_kill=hard_kill)
            if code != RANK_FAILED_SETUP:
                break
    finally:
        if old_term is not None:
            signal.signal(signal.SIGTERM, old_term)
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        if code == 0:
            shutil.rmtree(job, ignore_errors=True)
        else:
            print(f"[knerf launch] reports and RCCL logs of the failed job: {job}", file=sys.stderr, flush=True)
    return code


class MirroredStrategy:
    """`strategy = tf.distribute.MirroredStrategy()` (train.py:75) for an import-swapped script: constructing it in a plain
    `python train.py` starts one rank process per GPU that re-runs the script (launch()), and in each rank joins the process group;
    `num_replicas_in_sync` (train.py:76, 84) is the world size, `scope()` (train.py:110) a no-op context -- NeRF.compile mirrors
    the variables itself (weight broadcast from rank 0).  devices: a count or a list, like the reference's optional argument."""

    def __init__(self, devices=None, backend: str = None):
        n = len(devices) if isinstance(devices, (list, tuple)) else devices
        launch(None, n, backend=backend)
        self.num_replicas_in_sync = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    def scope(self):
        import contextlib
        return contextlib.nullcontext(self)
