#!/usr/bin/env python3
"""Which bf16 MFMA shape sustains more FLOP/s on this device with the chain kernels' operand traffic (random data)?
GPU box:  python tools/mfma_shape_probe.py"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keras_nerf_amd import debug as _lib

lib = _lib.load()
dev = torch.device("cuda")
a = torch.randn(96 * 1024, device=dev).to(torch.bfloat16)
b = torch.randn(64 * 64 * 8, device=dev).to(torch.bfloat16)
blocks, iters = 1024, 200
out = torch.empty(blocks * 512, device=dev)
s = torch.cuda.current_stream().cuda_stream
flop = blocks * 8 * iters * 96 * 32768
for rep in range(2):
    for shape in (32, 16, 64, 32, 16, 64):
        for _ in range(2):
            assert lib.knerf_debug_rate_probe(shape, a.data_ptr(), b.data_ptr(), out.data_ptr(), blocks, iters, s) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            lib.knerf_debug_rate_probe(shape, a.data_ptr(), b.data_ptr(), out.data_ptr(), blocks, iters, s)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(json.dumps({"shape": {32: "32x32x16", 16: "16x16x32", 64: "32x32x16, 64 samples per wave, 4 waves per CU"}[shape], "ms": round(ms, 3), "TFLOPs": round(flop / ms / 1e9, 1)}), flush=True)
