"""ctypes binding of libknerf_probe.so (include/knerf_debug.h): diagnostics for tests/ and tools/ ONLY.

The product path (runtime.py, model/, data/) never imports this module; tests/test_abi.py checks that, and that
libknerf_hip.so exports no `knerf_debug_*` symbol."""
from __future__ import annotations

import ctypes as C
import os

from . import _lib

_HERE = os.path.dirname(os.path.abspath(__file__))
PROBE_PATH = os.environ.get("KNERF_PROBE_LIB") or os.path.join(_HERE, "libknerf_probe.so")

_P = C.c_void_p
SIGNATURES = {
    "knerf_debug_table": (C.c_int, [C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_size_t)]),
    "knerf_debug_generic_plan": (C.c_int, [_P, _P, _P]),
    "knerf_debug_buffer": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "knerf_debug_probe": (C.c_int, [C.c_int, _P, _P, _P, _P]),
    "knerf_debug_write_probe": (C.c_int, [_P, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_int, _P]),
    "knerf_debug_read_probe": (C.c_int, [_P, C.c_int, C.c_longlong, C.c_int, _P, _P]),
    "knerf_debug_rate_probe": (C.c_int, [C.c_int, _P, _P, _P, C.c_int, C.c_int, _P]),
}
_probe = None


def load() -> C.CDLL:
    global _probe
    if _probe is None:
        _lib.load()                                    # the product library (and torch's HIP runtime) first
        if not os.path.exists(PROBE_PATH):
            raise _lib.KnerfError(f"{PROBE_PATH} is missing: run `python keras_nerf_amd/build.py`")
        lib = C.CDLL(PROBE_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _probe = lib
    return _probe


def debug_table(kind: int, shape: int = 0):
    """host-side packing / destination table `kind` of built-in trunk shape `shape` (see include/knerf_debug.h)"""
    import numpy as np
    lib = load()
    kind = int(kind) + 16 * int(shape)
    n = C.c_size_t(0)
    if lib.knerf_debug_table(kind, None, C.byref(n)) != 0:
        raise _lib.KnerfError("knerf_debug_table failed")
    out = np.empty(n.value, np.int32)
    if lib.knerf_debug_table(kind, out.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(n)) != 0:
        raise _lib.KnerfError("knerf_debug_table failed")
    return out


def debug_buffer(ctx, which: int, net: int = 0):
    """uint8 torch view of a workspace of a runtime.KnerfContext (see knerf_debug_buffer)"""
    import torch
    from .runtime import _CudaView
    p, n = C.c_void_p(), C.c_size_t()
    if load().knerf_debug_buffer(ctx._ctx, int(net), int(which), C.byref(p), C.byref(n)) != 0:
        raise ValueError("debug_buffer: unknown or unallocated buffer (no pass has run yet, or it belongs to the other MLP path)")
    return torch.as_tensor(_CudaView(p.value, n.value, "|u1"), device=ctx.device)
