"""ctypes binding of libknerf_hip.so (include/knerf.h).  There is no CPU fallback: a missing library or a missing
gfx950 device is an error, never a silent detour."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KNERF_LIB") or os.path.join(_HERE, "libknerf_hip.so")   # KNERF_LIB: A/B builds (tools/kbench.py)

KNERF_OK, KNERF_ERR_INVALID, KNERF_ERR_HIP, KNERF_ERR_NONFINITE, KNERF_ERR_NODEVICE = 0, -1, -2, -3, -4
COARSE, FINE = 0, 1


class KnerfConfig(C.Structure):
    _fields_ = [("n_coarse", C.c_int32), ("n_fine", C.c_int32), ("pos_emb_xyz", C.c_int32), ("pos_emb_dir", C.c_int32),
                ("n_layers", C.c_int32), ("dense_units", C.c_int32), ("skip_layer", C.c_int32),
                ("white_background", C.c_int32), ("oob_clamp", C.c_int32),
                ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("epsilon", C.c_float), ("flags", C.c_int32)]


FLAG_FORCE_GENERIC = 1
FLAG_ENCODED_WIDTHS = 2      # pos_emb_xyz / pos_emb_dir hold the two encoded input widths of a stand-alone NeRFMLP (include/knerf.h)


_P = C.c_void_p
_F = C.c_void_p  # device float* passed as integers (tensor.data_ptr())
# name -> (restype, argtypes): exactly the declarations of include/knerf.h
SIGNATURES = {
    "knerf_param_count": (C.c_size_t, []),
    "knerf_param_count_for": (C.c_size_t, [_P]),
    "knerf_create": (C.c_int, [C.POINTER(KnerfConfig), C.POINTER(_P)]),
    "knerf_destroy": (C.c_int, [_P]),
    "knerf_last_error": (C.c_char_p, [_P]),
    "knerf_set_weights": (C.c_int, [_P, C.c_int, C.POINTER(C.c_float), C.c_size_t]),
    "knerf_get_weights": (C.c_int, [_P, C.c_int, C.POINTER(C.c_float), C.c_size_t]),
    "knerf_weights_device": (C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "knerf_grads_device": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "knerf_refresh_weights": (C.c_int, [_P, _P]),
    "knerf_forward_chunk": (C.c_int, [_P, _P, C.c_int, _F, _F, _F, C.c_int, C.c_int, _F, _F, _F]),
    "knerf_sample_fine": (C.c_int, [_P, _P, _F, _F, _F, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, _F]),
    "knerf_render_chunk": (C.c_int, [_P, _P, _F, _F, _F, _F, C.c_uint64, C.c_uint64, C.c_int, _F, _F, _F, _F, _F, _F, _F]),
    "knerf_train_chunk": (C.c_int, [_P, _P, _F, _F, _F, _F, _F, C.c_uint64, C.c_uint64, C.c_int, C.c_float, _F, _F, _F]),
    "knerf_train_batch": (C.c_int, [_P, _P, _F, _F, _F, _F, _F, C.c_uint64, C.c_int, C.c_int, _F, _F, _F]),
    "knerf_apply_adam": (C.c_int, [_P, _P]),
    "knerf_poll_nonfinite": (C.c_int, [_P, _P, C.c_int]),
    "knerf_zero_grads": (C.c_int, [_P, _P]),
    "knerf_set_option": (C.c_int, [_P, C.c_char_p, C.c_double]),
    "knerf_get_option": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_double)]),
    "knerf_tile_stats": (C.c_int, [_P, _P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int]),
    "knerf_tile_stats_net": (C.c_int, [_P, _P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int]),
    "knerf_grad_diagnostics": (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_int64)]),
    "knerf_render_batch": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_uint64, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P]),
    "knerf_ray_points": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, _P]),
    "knerf_image_metrics": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "knerf_metrics_update": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "knerf_mlp_call": (C.c_int, [_P, _P, C.c_int, _P, _P, C.c_uint64, _P]),
    "knerf_step_count": (C.c_int, [_P]),
    "knerf_set_step_count": (C.c_int, [_P, C.c_int]),
    "knerf_generate_rays": (C.c_int, [_P, _P, _F, _F, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_float, C.c_float, C.c_float, _F, _F, _F]),
    "knerf_positional_encoding": (C.c_int, [_P, _F, C.c_longlong, C.c_int, _F]),
    "knerf_composite": (C.c_int, [_P, _F, _F, C.c_int, C.c_int, C.c_int, _F, _F, _F]),
    "knerf_inverse_cdf": (C.c_int, [_P, _F, _F, _F, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _F]),
    "knerf_profile_enable": (C.c_int, [_P, C.c_int]),
    "knerf_profile_read": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]),
}

_lib = None


class KnerfError(RuntimeError):
    pass


_by_path = {}


def load_path(path: str) -> C.CDLL:
    """A library of this ABI by path (bound once per path): the product library, or a build with further fused shapes
    (build.py --add-shape / KNERF_AUTO_BUILD in runtime.py).  Raises (never falls back) when the file is missing."""
    path = os.path.abspath(path)
    lib = _by_path.get(path)
    if lib is None:
        if not os.path.exists(path):
            raise KnerfError(f"{path} is missing: run `python keras_nerf_amd/build.py` (hipcc, gfx950). "
                             "keras_nerf_amd has no CPU path.")
        # torch first: its bundled HIP runtime must be the one already mapped when libknerf_hip.so resolves
        # libamdhip64, otherwise two runtimes coexist and the second one sees no device / foreign pointers
        import torch  # noqa: F401
        lib = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _by_path[path] = lib
    return lib


def build_info(path: str = None) -> dict:
    """What is inside the library at `path` (default: the one load() binds): {"lib_path", "lib_sha16", "kernel_digest",
    "git_head_at_build"}.  kernel_digest (build.py: the source of the three big kernels + flags) comes from the record build.py
    writes beside the library and is None when that record is missing or describes ANOTHER file (its lib_sha16 differs from the
    file's): a digest is only as good as its tie to the binary."""
    import json
    from .build import file_sha16
    path = os.path.abspath(path or LIB_PATH)
    out = {"lib_path": os.path.relpath(path, os.path.dirname(_HERE)), "lib_sha16": file_sha16(path) if os.path.exists(path) else None,
           "kernel_digest": None, "git_head_at_build": None}
    try:
        info = json.load(open(path + ".info.json"))
    except (OSError, ValueError):
        return out
    if info.get("lib_sha16") == out["lib_sha16"]:
        out["kernel_digest"] = info.get("kernel_digest"); out["git_head_at_build"] = info.get("git_head_at_build")
    return out


def load() -> C.CDLL:
    """Load the HIP library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if os.environ.get("KNERF_LIB"):          # an A/B build takes the product library's place: never silently
            import logging
            logging.warning("KNERF_LIB is set: keras_nerf_amd runs on %s instead of the product library %s", LIB_PATH,
                            os.path.join(_HERE, "libknerf_hip.so"))
        _lib = load_path(LIB_PATH)
    return _lib
