#!/usr/bin/env python3
"""Per-workgroup timeline of the fine inference forward launch (diagnostic build: build.py -DKNERF_FWD_STAMPS --variant=fst).
s_memtime at kernel entry / in front of the first MFMA / at exit, and the CU each workgroup ran on: ramp, body and the gap between
consecutive workgroups of one CU, in s_memtime ticks (1.66 GHz on this device: body 79.6 k ticks = 0.575 ms / 12 workgroups per CU).
GPU box:  python tools/fwd_stamps.py keras_nerf_amd/libknerf_hip_fst.so"""
import os, sys, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["KNERF_LIB"] = os.path.abspath(sys.argv[1])
import numpy as np, torch
from keras_nerf_amd.runtime import KnerfContext
from keras_nerf_amd.data.utils import get_focal_from_fov, pose_spherical
from keras_nerf_amd.model.nerf.mlp import NeRFMLP
ctx = KnerfContext(white_background=True)
for net in (0, 1):
    m = NeRFMLP(8, 256, 4, seed=net); m.build(); ctx.set_weights(net, m.get_flat_weights())
wh = 128
o, d, t = ctx.generate_rays(pose_spherical(20.0, -30.0, 4.0)[None], get_focal_from_fov(0.6911112070083618, wh), wh, wh, 2.0, 6.0, 64, None, seed=1)
R = 4096
o, d, t = o.reshape(-1, 3)[:R].contiguous(), d.reshape(-1, 3)[:R].contiguous(), t.reshape(-1, 64)[:R].contiguous()
for _ in range(3): ctx.render_chunk(o, d, t, None, seed=1)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (4096 * 4))()
ctx.lib.knerf_debug_fwd_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
assert ctx.lib.knerf_debug_fwd_stamps(buf, 4096 * 4) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 4)[:3072].astype(np.int64)
st0, st1, st2, hw = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
xcc = hw >> 32; hwid = hw & 0xffffffff
cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 0x1; se = (hwid >> 13) & 0x7     # gfx9 HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
key = xcc * 1024 + se * 32 + sh * 16 + cu
ramp = st1 - st0; body = st2 - st1
# every XCD has its own s_memtime counter: only differences on one CU mean anything
res = {"lib": os.path.basename(sys.argv[1]), "ramp_mean": float(ramp.mean()), "ramp_p90": float(np.percentile(ramp, 90)), "body_mean": float(body.mean()),
       "distinct_cus": int(len(np.unique(key))), "wgs_per_cu_mean": 3072 / len(np.unique(key))}
gaps, busy = [], []
for k in np.unique(key):
    idx = np.where(key == k)[0]
    order = idx[np.argsort(st0[idx])]
    gaps += list(st0[order][1:] - st2[order][:-1])
    busy.append(float(body[order].sum()) / float(st2[order].max() - st0[order].min()))
res.update({"gap_mean": float(np.mean(gaps)), "gap_p90": float(np.percentile(gaps, 90)), "body_share_of_cu_time_mean": float(np.mean(busy))})
print(json.dumps(res))
