#!/usr/bin/env python3
"""HBM write rate of the saved-activation store pattern vs a linear fill (diagnostic).  GPU box: python tools/write_probe.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keras_nerf_amd import debug as _lib
lib = _lib.load()
wgs, blocks, stride = 3072, 118, 118 * 1024 + 256      # the forward's saved-activation run (csrc/layout.h kActBlocks)
buf = torch.empty(wgs * 8 * stride, dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
byts = wgs * 8 * blocks * 1024
def run(mode, spin):
    for _ in range(2): lib.knerf_debug_write_probe(buf.data_ptr(), wgs, blocks, stride, mode, spin, s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): lib.knerf_debug_write_probe(buf.data_ptr(), wgs, blocks, stride, mode, spin, s)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5
for pol, pname in ((0, "nt"), (1, "plain"), (2, "sc1"), (3, "sc0 sc1")):
    ms = run(0, pol << 16)
    print(json.dumps({"layout": "tile-major", "policy": pname, "ms": round(ms, 3), "TBs": round(byts / ms / 1e9, 2)}), flush=True)
ms = run(0, 1 << 30)
print(json.dumps({"layout": "tile-major", "policy": "nt", "resident": "one workgroup per CU", "ms": round(ms, 3), "TBs": round(byts / ms / 1e9, 2)}), flush=True)
for spin in (0, 16):
    for mode in (0, 8, 64, 512, 4096, 24576):
        ms = run(mode, spin)
        print(json.dumps({"layout": "tile-major" if mode == 0 else f"block-major in groups of {mode} tiles", "spin_nops": spin,
                          "ms": round(ms, 3), "TBs": round(byts / ms / 1e9, 2)}), flush=True)
x = torch.empty(byts // 4, dtype=torch.float32, device="cuda")
x.fill_(1.0); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): x.fill_(2.0)
e1.record(); torch.cuda.synchronize()
print(json.dumps({"mode": "linear fill", "TBs": round(byts / (e0.elapsed_time(e1) / 5) / 1e9, 2)}))

# read side: the ceiling of wgrad's streaming pattern (8 GB per launch, one workgroup per CU)
src = torch.empty(8 << 30, dtype=torch.uint8, device="cuda"); src.zero_()
o = torch.zeros(512, device="cuda")
for wg in (252, 504):
    per_wave = (8 << 30) // (wg * 8) // 8192 * 8192
    for mode in (0, 1, 2, 3):
        for _ in range(2): lib.knerf_debug_read_probe(src.data_ptr(), wg, per_wave, mode, o.data_ptr(), s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): lib.knerf_debug_read_probe(src.data_ptr(), wg, per_wave, mode, o.data_ptr(), s)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(json.dumps({"read": ["nt LDS-DMA", "register loads", "nt LDS-DMA, barrier per 4 KiB/wave (3 tiles in flight)", "nt LDS-DMA, barrier per 8 KiB/wave"][mode], "workgroups": wg, "ms": round(ms, 3),
                          "TBs": round(per_wave * wg * 8 / ms / 1e9, 2)}), flush=True)
