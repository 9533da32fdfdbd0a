"""INTEGRATION.md section B, executed: the ctypes stub a maintainer of the reference would write -- its OWN struct and argtypes from
include/knerf.h, no keras_nerf_amd import on the binding side -- drives one train step (chunk loop of nerf.py:351-421, optimizer
step of nerf.py:455-471) through `libknerf_hip.so` and gets, in deterministic mode, the bits the Python shim gets."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from tests.problem import make_problem

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Cfg(C.Structure):            # struct knerf_config, include/knerf.h -- as printed in INTEGRATION.md
    _fields_ = [(n, C.c_int32) for n in ("n_coarse", "n_fine", "pos_emb_xyz", "pos_emb_dir", "n_layers", "dense_units",
                                         "skip_layer", "white_background", "oob_clamp")] + \
               [(n, C.c_float) for n in ("lr", "beta1", "beta2", "epsilon")] + [("flags", C.c_int32)]


def test_the_documented_ctypes_stub_trains_one_step_and_matches_the_shim():
    P = make_problem(n_images=1, wh=16, weight_scale=1.5, bias_std=0.05)
    N, R = P["N"], 64
    n_chunks = N // R
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device="cuda")
    o, d, t, target, u = (dev(P[k].reshape(N, -1)) for k in ("o", "d", "t", "img", "u"))
    flat_c, flat_f = O.flatten_params(P["cp"]).astype(np.float32), O.flatten_params(P["fp"]).astype(np.float32)
    torch.cuda.synchronize()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    # ---- the stub --------------------------------------------------------------------------------
    lib = C.CDLL(os.path.join(ROOT, "keras_nerf_amd", "libknerf_hip.so"))
    ctx = C.c_void_p()
    assert lib.knerf_create(C.byref(Cfg(64, 128, 10, 4, 8, 256, 4, 1, 0, 1e-3, .9, .999, 1e-7, 0)), C.byref(ctx)) == 0
    lib.knerf_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
    assert lib.knerf_set_option(ctx, b"deterministic", 1.0) == 0
    lib.knerf_set_weights.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_size_t]
    for net, flat in ((0, flat_c), (1, flat_f)):
        assert lib.knerf_set_weights(ctx, net, flat.ctypes.data_as(C.POINTER(C.c_float)), flat.size) == 0
    lib.knerf_last_error.restype, lib.knerf_last_error.argtypes = C.c_char_p, [C.c_void_p]
    lib.knerf_train_chunk.argtypes = [C.c_void_p] * 7 + [C.c_uint64, C.c_uint64, C.c_int, C.c_float] + [C.c_void_p] * 3
    loss = torch.zeros(2, device="cuda")
    for i in range(n_chunks):
        rc = lib.knerf_train_chunk(ctx, stream, o.data_ptr() + i * R * 12, d.data_ptr() + i * R * 12, t.data_ptr() + i * R * 256,
                                   target.data_ptr() + i * R * 12, u.data_ptr() + i * R * 512, 0, i * R, R, 1.0 / n_chunks,
                                   loss.data_ptr(), None, None)
        assert rc == 0, (rc, lib.knerf_last_error(ctx))
    gp, gn = C.c_void_p(), C.c_size_t()
    lib.knerf_grads_device.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    assert lib.knerf_grads_device(ctx, C.byref(gp), C.byref(gn)) == 0 and gn.value == 2 * flat_c.size
    torch.cuda.synchronize()
    g_raw = np.empty(gn.value, np.float32)
    # (a maintainer would all-reduce gp here: it is the library-owned [coarse | fine] buffer, nerf.py:455-458 under MirroredStrategy)
    from keras_nerf_amd.runtime import _CudaView              # only to READ the device buffer back in this test
    g_raw[:] = torch.as_tensor(_CudaView(gp.value, gn.value, "<f4"), device="cuda").cpu().numpy()
    lib.knerf_apply_adam.argtypes = [C.c_void_p, C.c_void_p]
    lib.knerf_poll_nonfinite.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    assert lib.knerf_apply_adam(ctx, stream) == 0 and lib.knerf_poll_nonfinite(ctx, stream, 1) == 0
    w_raw = np.empty(flat_c.size, np.float32)
    lib.knerf_get_weights.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_size_t]
    assert lib.knerf_get_weights(ctx, 0, w_raw.ctypes.data_as(C.POINTER(C.c_float)), w_raw.size) == 0
    lib.knerf_destroy.argtypes = [C.c_void_p]
    assert lib.knerf_destroy(ctx) == 0

    # ---- the same step through the Python shim ------------------------------------------------------------------------------
    from keras_nerf_amd.runtime import KnerfContext
    k = KnerfContext(white_background=True, options=dict(deterministic=1))
    k.set_weights(0, flat_c); k.set_weights(1, flat_f)
    loss2 = torch.zeros(2, device="cuda")
    for i in range(n_chunks):
        sl = slice(i * R, (i + 1) * R)
        k.train_chunk(o[sl], d[sl], t[sl], target[sl], u[sl], inv_chunks=1.0 / n_chunks, loss=loss2, ray_offset=i * R)
    torch.cuda.synchronize()
    g_shim = k.grads_view().cpu().numpy().copy()
    k.apply_adam(); k.poll_nonfinite(wait=True)
    w_shim = k.get_weights(0)
    k.close()
    assert np.abs(g_raw).max() > 1e-6 and np.array_equal(g_raw.view(np.int32), g_shim.view(np.int32))
    assert np.array_equal(w_raw.view(np.int32), w_shim.view(np.int32)) and np.abs(w_raw - flat_c).max() > 1e-5
    assert torch.equal(loss, loss2)
