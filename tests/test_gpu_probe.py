"""Pins, on the real GPU, the two hardware facts the fused kernels are built on (gfx950):
the operand/result lane maps of v_mfma_f32_32x32x16_bf16 and the semantics of ds_read_b64_tr_b16."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from tests import mfma_sim as M

pytestmark = pytest.mark.gpu
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def _p(t):
    return C.c_void_p(t.data_ptr())


def test_mfma_32x32x16_bf16_lane_maps():
    from keras_nerf_amd import debug
    lib = debug.load()
    rng = np.random.default_rng(0)
    a = rng.integers(-4, 5, (64, 8)).astype(np.float32)      # small integers: exact in bf16 and in the f32 sums
    b = rng.integers(-4, 5, (64, 8)).astype(np.float32)      # asymmetric on purpose
    ta = torch.tensor(a, device="cuda").to(torch.bfloat16).contiguous()
    tb = torch.tensor(b, device="cuda").to(torch.bfloat16).contiguous()
    out = torch.zeros((64, 16), device="cuda", dtype=torch.float32)
    assert lib.knerf_debug_probe(0, _p(ta), _p(tb), _p(out), None) == 0
    torch.cuda.synchronize()
    exp = M.mfma(a, b, np.zeros((64, 16), np.float32))
    np.testing.assert_array_equal(out.cpu().numpy(), exp)


def tr_expected(img_u16, addr):
    """ds_read_b64_tr_b16 as described in the CDNA4 guide (T10): per group of 16 lanes, lane 4q+p supplies the address
    of row q / column group p (4 elements); lane i receives column i of the 4 rows (row q in element q)."""
    out = np.zeros((64, 4), np.uint16)
    for l in range(64):
        g, i = l >> 4, l & 15
        for q in range(4):
            src_lane = 16 * g + 4 * q + (i >> 2)
            out[l, q] = img_u16[addr[src_lane] // 2 + (i & 3)]
    return out


def test_ds_read_b64_tr_b16_semantics():
    from keras_nerf_amd import debug
    lib = debug.load()
    rng = np.random.default_rng(1)
    img = np.arange(2048, dtype=np.uint16)
    addr = (rng.permutation(512)[:64] * 8).astype(np.int32)
    timg = torch.tensor(img.view(np.int16), device="cuda")
    taddr = torch.tensor(addr, device="cuda")
    out = torch.zeros((64, 4), device="cuda", dtype=torch.int16)
    assert lib.knerf_debug_probe(1, _p(timg), _p(taddr), _p(out), None) == 0
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint16)
    os.makedirs(OUT, exist_ok=True)
    np.savez(os.path.join(OUT, "tr_probe.npz"), addr=addr, got=got)
    np.testing.assert_array_equal(got, tr_expected(img, addr))
