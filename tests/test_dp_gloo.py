"""world_size-2 data-parallel test on CPU (gloo): the DP glue (keras_nerf_amd/parallel.py) + the oracle's train step
reproduce the single-process "two mirrored replicas" result of the reference's train.py semantics: each replica runs the
chunk loop on its own images, the accumulated gradients are SUMmed, both replicas apply identical Adam updates."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import nerf_oracle as O


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _problem():
    cfg = O.NerfConfig(n_coarse=8, n_fine=16, pos_emb_xyz=4, pos_emb_dir=2, n_layers=8, dense_units=32, skip_layer=4)
    rng = np.random.default_rng(0)
    os_, ds_, ts_ = [], [], []
    for i in range(2):
        o, d, t = O.generate_rays(O.pose_spherical(40.0 + 90 * i, -30.0, 4.0), 8.0, 4, 4, 2.0, 6.0, 8, rng.random((4, 4, 8)))
        os_.append(o); ds_.append(d); ts_.append(t)
    o, d, t = np.stack(os_), np.stack(ds_), np.stack(ts_)
    u = rng.random((2, 4, 4, 16)).astype(np.float32)
    img = rng.random((2, 4, 4, 3)).astype(np.float32)
    cp = [p * 3 for p in O.init_params(cfg, 1)]
    fp = [p * 3 for p in O.init_params(cfg, 2)]
    return cfg, o, d, t, u, img, cp, fp


def _worker(rank, world, port, mode, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from keras_nerf_amd import parallel
    cfg, o, d, t, u, img, cp, fp = _problem()
    if rank != 0:                                     # replicas must end up with rank 0's weights
        cp = [p * 0 for p in cp]; fp = [p * 0 for p in fp]
    flat_c, flat_f = torch.tensor(O.flatten_params(cp)), torch.tensor(O.flatten_params(fp))
    parallel.broadcast_weights([flat_c, flat_f])
    cp, fp = O.unflatten_params(flat_c.numpy().copy(), cfg), O.unflatten_params(flat_f.numpy().copy(), cfg)
    cp, fp = [p.copy() for p in cp], [p.copy() for p in fp]
    oc, of_ = O.KerasAdam(cp), O.KerasAdam(fp)
    sh = lambda x: parallel.shard_batch(x, rank, world)

    def hook(gc, gf):
        flat = torch.tensor(np.concatenate([O.flatten_params(gc), O.flatten_params(gf)]))
        parallel.all_reduce_gradients(flat, mode)
        n = flat.numel() // 2
        return O.unflatten_params(flat[:n].numpy(), cfg), O.unflatten_params(flat[n:].numpy(), cfg)
    logs = {}
    for step in range(2):
        m, _, _, _ = O.train_step(cp, fp, oc, of_, sh(img), sh(o), sh(d), sh(t), sh(u), cfg, 8, True, grad_scale_hook=hook)
        logs = parallel.reduce_logs({k: float(v) for k, v in m.items()})
    np.savez(out.format(rank=rank), c=O.flatten_params(cp), f=O.flatten_params(fp), **logs)
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["sum", "mean"])
def test_two_replicas_match_single_process_reference(tmp_path, mode):
    port = _free_port()
    out = str(tmp_path / "rank{rank}.npz")
    mp.spawn(_worker, args=(2, port, mode, out), nprocs=2, join=True)
    r0, r1 = np.load(out.format(rank=0)), np.load(out.format(rank=1))
    np.testing.assert_array_equal(r0["c"], r1["c"])           # mirrored variables stay identical
    np.testing.assert_array_equal(r0["f"], r1["f"])
    assert r0["coarse_loss"] == r1["coarse_loss"]
    # single-process reference: both replicas' gradients computed here, summed (or averaged), one Adam per net
    cfg, o, d, t, u, img, cp, fp = _problem()
    oc, of_ = O.KerasAdam(cp), O.KerasAdam(fp)
    losses = []
    for step in range(2):
        accs = []
        for r in range(2):
            sl = slice(r, r + 1)
            m, _, _, acc = O.train_step(cp, fp, None, None, img[sl], o[sl], d[sl], t[sl], u[sl], cfg, 8, True)
            accs.append(acc); losses.append(float(m["coarse_loss"]))
        k = 1.0 if mode == "sum" else 0.5
        gc = [np.float32(k) * (a + b) for a, b in zip(accs[0][0], accs[1][0])]
        gf = [np.float32(k) * (a + b) for a, b in zip(accs[0][1], accs[1][1])]
        oc.apply(cp, gc); of_.apply(fp, gf)
    np.testing.assert_allclose(r0["c"], O.flatten_params(cp), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(r0["f"], O.flatten_params(fp), rtol=1e-5, atol=1e-7)
    assert float(r0["coarse_loss"]) == pytest.approx(np.mean(losses[-2:]), rel=1e-5)


def test_shard_batch_divisibility():
    from keras_nerf_amd import parallel
    with pytest.raises(ValueError):
        parallel.shard_batch(torch.zeros(3, 2), 0, 2)
    assert parallel.shard_batch(torch.arange(8).reshape(4, 2), 1, 2).tolist() == [[4, 5], [6, 7]]


def _worker8(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from keras_nerf_amd import parallel
    from keras_nerf_amd.data.loader import replica_batches, shuffled_order
    w = [torch.full((1000,), float(rank + 1)), torch.full((7,), -float(rank))]          # every rank starts with its own "weights"
    parallel.broadcast_weights(w)
    g = torch.arange(2 * 595844, dtype=torch.float32) % 17 * (rank + 1)                  # the real operand size: 4,766,752 bytes
    parallel.all_reduce_gradients(g, "sum")
    logs = parallel.reduce_logs({"fine_loss": float(rank), "coarse_loss": 2.0})
    # the shared-seed batch plan (data/loader.py): all ranks draw the same order, each keeps its slice of every global batch
    order = shuffled_order(100, 1 * world, np.random.default_rng(7))
    mine = replica_batches(order, 1, rank, world)
    # the bench's replica-drift check on CPU tensors: exact integer checksum, MAX - MIN over the ranks
    chk = (w[0].view(torch.int32).to(torch.int64)).sum().reshape(1)
    hi, lo = chk.clone(), chk.clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX); dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    np.savez(out.format(rank=rank), w0=w[0].numpy(), w1=w[1].numpy(), g=g[:64].numpy(), gsum=float(g.double().sum()), drift=int(hi[0]) - int(lo[0]),
             mine=np.asarray(mine).reshape(-1), **logs)
    dist.destroy_process_group()


def test_eight_ranks_cfg4_shape_of_the_glue():
    """cfg4 is 8 GPUs; the largest job a one-GPU box can host on its card is 4-5 ranks (tests/test_gpu_api.py), so the EIGHT-rank
    case of the glue runs here on the CPU: mirrored start from rank 0, one SUM all-reduce of the real 4.77 MB operand, replica
    means of the logs, zero drift of the weight checksum, and the loader's batch plan (train.py:84-93: global batch = batch_size x
    replicas, replica r takes elements [r b, (r+1) b) of each): 12 global batches of 8 out of 100 views, slices disjoint."""
    port = _free_port()
    import tempfile
    d = tempfile.mkdtemp()
    out = os.path.join(d, "rank{rank}.npz")
    mp.spawn(_worker8, args=(8, port, out), nprocs=8, join=True)
    rs = [np.load(out.format(rank=r)) for r in range(8)]
    seen = []
    for r, z in enumerate(rs):
        assert (z["w0"] == 1.0).all() and (z["w1"] == 0.0).all() and int(z["drift"]) == 0
        np.testing.assert_array_equal(z["g"], (np.arange(64) % 17 * 36).astype(np.float32))                  # 1 + 2 + ... + 8 = 36
        assert float(z["fine_loss"]) == pytest.approx(3.5) and float(z["coarse_loss"]) == pytest.approx(2.0)
        assert z["mine"].size == 12
        seen += z["mine"].tolist()
    assert len(set(seen)) == 96 and rs[0]["gsum"] == rs[7]["gsum"]


def _launch_env():
    return {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "KNERF_DIST_BACKEND",
                                                           "HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG", "NCCL_DEBUG_FILE")}


def _check_ranks(out_dir, n):
    import json
    rs = [json.load(open(os.path.join(out_dir, f"rank{r}.json"))) for r in range(n)]
    for r, z in enumerate(rs):
        assert z["rank"] == r and z["world"] == n and z["w"] == 1.0 and z["g1"] == n * (n + 1) / 2 and z["loss"] == pytest.approx((n - 1) / 2)
        assert z["master"] == "127.0.0.1" and z["ipc"] == "0" and z["nccl_debug"] == "WARN" and z["is_main"] == (r == 0)
    assert len({z["pid"] for z in rs}) == n          # one process per rank
    return rs


def test_launch_runs_a_function_on_eight_ranks(tmp_path):
    """keras_nerf_amd.parallel.launch(fn, nprocs): the MirroredStrategy() of this implementation (train.py:75-93, 110-148) -- eight
    fresh rank processes (started before the parent touches any GPU, never an exec), each inside the process group with the launch
    environment set for it (rendezvous on 127.0.0.1, dmabuf IPC, RCCL warnings to a per-rank file), run `fn`; here over gloo on the
    CPU, on the driver's node over RCCL."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); from keras_nerf_amd import parallel; from tests.launch_worker import body; "
            "rc = parallel.launch(body, 8, args=(%r,), backend='gloo'); print('launch returned', rc)") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=_launch_env())
    assert r.returncode == 0 and "launch returned 0" in r.stdout, r.stderr[-3000:]
    _check_ranks(str(tmp_path), 8)


def test_launch_rerun_script_form_like_mirrored_strategy(tmp_path):
    """`strategy = parallel.MirroredStrategy()` at the top of a script (where train.py:75 builds its strategy): plain `python script.py`
    becomes the launcher of N ranks that re-run the script; `num_replicas_in_sync` is the world size inside them"""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "launch_worker.py"), str(tmp_path), "4"],
                       capture_output=True, text=True, timeout=300, env=_launch_env())
    assert r.returncode == 0, r.stderr[-3000:]
    _check_ranks(str(tmp_path), 4)


@pytest.mark.parametrize("form", ["function", "script"])
def test_launch_retries_once_with_the_other_ipc_setting_when_the_group_never_comes_up(tmp_path, form):
    """The first unattended N > 1 run (train.py:75-93 on the driver's 8-GPU node) most likely dies where RCCL builds its rings:
    dist_env defaults HSA_ENABLE_IPC_MODE_LEGACY=0 because THIS pool's driver needs it, and nobody knows what that node needs.
    A job whose first failing rank leaves from init_process_group / the first all_reduce (exit code 4) gets ONE more set of FRESH
    ranks from the launcher (which never touched a GPU) with the variable absent; the line of the job says launch_attempts 2, the
    setting that worked and what the first attempt's rank reported.  Here: every rank of attempt 1 fails by injection, over gloo."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(_launch_env(), KNERF_INJECT_FAILURE="*:first_all_reduce@1")
    if form == "function":
        code = ("import sys; sys.path.insert(0, %r); from keras_nerf_amd import parallel; from tests.launch_worker import body; "
                "rc = parallel.launch(body, 2, args=(%r,), backend='gloo'); print('launch returned', rc)") % (root, str(tmp_path))
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    else:
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "launch_worker.py"), str(tmp_path), "2"],
                           capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, r.stdout                                     # ONE line: the first attempt printed none
    line = lines[0]
    assert line["launch_attempts"] == 2 and line["ipc_mode_legacy_env"] is None and line["n_ranks"] == 2
    assert "injected failure at stage 'first_all_reduce'" in line["first_attempt_failure"] and "HSA_ENABLE_IPC_MODE_LEGACY=0" in line["first_attempt_failure"]
    assert "ONE more attempt with fresh ranks and HSA_ENABLE_IPC_MODE_LEGACY=<unset>" in r.stderr
    rs = [json.load(open(os.path.join(str(tmp_path), f"rank{k}.json"))) for k in range(2)]
    assert all(z["ipc"] is None and z["launch_attempts"] == 2 and z["world"] == 2 for z in rs)
    # the launcher's own environment is as it found it, and a clean job leaves no job directory behind
    assert "reports and RCCL logs of the failed job" not in r.stderr


def test_launch_does_not_retry_other_failures_and_gives_up_after_the_second_set_up_failure(tmp_path):
    """a rank that fails in the BODY ends the job at once (exit 3, one attempt); a group that does not come up under either setting
    ends it with exit 4 after exactly two attempts; a caller who chose a value other than "0" gets "0" as the second attempt"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); from keras_nerf_amd import parallel; from tests.launch_worker import body; "
            "parallel.launch(body, 2, args=(%r,), backend='gloo')") % (root, str(tmp_path))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(_launch_env(), KNERF_INJECT_FAILURE="1:body"))
    assert r.returncode == 3 and "ONE more attempt" not in r.stderr, (r.returncode, r.stderr[-2000:])
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(_launch_env(), KNERF_INJECT_FAILURE="0:init", HSA_ENABLE_IPC_MODE_LEGACY="1"))
    assert r.returncode == 4 and r.stderr.count("ONE more attempt") == 1, (r.returncode, r.stderr[-2000:])
    assert "fresh ranks and HSA_ENABLE_IPC_MODE_LEGACY=0" in r.stderr and "reports and RCCL logs of the failed job" in r.stderr
    assert r.stderr.count("[knerf rank 0/2] FAILED in init_process_group") == 2


def test_launch_stops_everything_when_one_rank_fails(tmp_path):
    """a rank that raises names itself, leaves with code 3, and the launcher terminates its peers (which are blocked in a collective
    with it) by their handles within seconds -- no collective time-out, no orphan processes"""
    import subprocess
    import sys
    import time
    code = ("import sys; sys.path.insert(0, %r); from keras_nerf_amd import parallel; from tests.launch_worker import hang_unless_zero; "
            "parallel.launch(hang_unless_zero, 4, args=(%r,), backend='gloo')") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path))
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=_launch_env())
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert time.time() - t0 < 120
    assert "[knerf rank 1/4] FAILED in hang_unless_zero()" in r.stderr and "rank 1 fails on purpose" in r.stderr
    assert "stopping the other" in r.stderr


def test_launch_escalates_to_sigkill_for_ranks_that_ignore_sigterm(tmp_path, monkeypatch):
    """a rank blocked where SIGTERM does not reach it (inside a driver call) must not keep the job alive: after a grace period the
    launcher kills it by handle; afterwards none of the ranks' pids exists any more"""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); from keras_nerf_amd import parallel; from tests.launch_worker import ignore_sigterm_and_hang; "
            "import functools; parallel._wait_ranks = functools.partial(parallel._wait_ranks, grace_s=2.0); "
            "parallel.launch(ignore_sigterm_and_hang, 3, args=(%r,), backend='gloo')") % (root, str(tmp_path))
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=_launch_env())
    assert r.returncode == 3 and time.time() - t0 < 120, (r.returncode, r.stderr[-2000:])
    for k in (0, 2):
        pid = int(open(tmp_path / f"pid{k}").read())
        assert not os.path.exists(f"/proc/{pid}"), f"rank {k} (pid {pid}) survived its launcher"


def test_a_plain_sigterm_to_the_launcher_takes_the_ranks_along(tmp_path):
    """a launcher killed by a plain SIGTERM (a scheduler's time limit, `timeout` without -k) had no handler -- no exception,
    no clean-up, orphan ranks holding their GPUs.  launch() turns the signal into SystemExit, and _wait_ranks stops the ranks by handle
    on its way out."""
    import signal
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); from keras_nerf_amd import parallel; from tests.launch_worker import record_pid_and_wait; "
            "parallel.launch(record_pid_and_wait, 2, args=(%r,), backend='gloo')") % (root, str(tmp_path))
    p = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=_launch_env())
    t_end = time.time() + 120
    while time.time() < t_end and not all(os.path.exists(tmp_path / f"pid{k}") and open(tmp_path / f"pid{k}").read() for k in range(2)):
        time.sleep(0.2)
    pids = [int(open(tmp_path / f"pid{k}").read()) for k in range(2)]
    assert all(os.path.exists(f"/proc/{pid}") for pid in pids)
    p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=60)
    assert p.returncode == 128 + signal.SIGTERM, (p.returncode, err[-2000:])
    t_end = time.time() + 20
    while time.time() < t_end and any(os.path.exists(f"/proc/{pid}") for pid in pids):
        time.sleep(0.2)
    for pid in pids:
        assert not os.path.exists(f"/proc/{pid}"), f"rank pid {pid} survived its launcher"


def test_init_rank_refusals_name_the_problem(monkeypatch):
    """the checks in front of init_process_group: a rank without a GPU under the RCCL backend, and a world of one (nothing to join)"""
    import torch
    from keras_nerf_amd import parallel
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert parallel.init_rank("gloo")[:2] == (0, 1)                   # world size 1: no process group is created
    assert not dist.is_initialized()
    if torch.cuda.device_count() == 0:
        monkeypatch.setenv("WORLD_SIZE", "2"); monkeypatch.setenv("RANK", "1"); monkeypatch.setenv("LOCAL_RANK", "1")
        with pytest.raises(SystemExit, match="no GPU visible"):
            parallel.init_rank("nccl")
    # the launch environment is set without overriding what the caller chose
    monkeypatch.setenv("MASTER_ADDR", "10.0.0.7"); monkeypatch.delenv("NCCL_DEBUG", raising=False); monkeypatch.delenv("NCCL_DEBUG_FILE", raising=False)
    parallel.dist_env(8)
    assert os.environ["MASTER_ADDR"] == "10.0.0.7" and os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert os.environ["NCCL_DEBUG"] == "WARN" and "%p" in os.environ["NCCL_DEBUG_FILE"]
    assert parallel.rccl_log_tail() == ""                             # no file yet: nothing to show
