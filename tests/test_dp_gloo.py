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
