"""ctypes binding of include/retinaface_amd.h (one declaration per exported symbol)."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path() -> str:
    return os.environ.get("RETINAFACE_AMD_LIB", os.path.join(_HERE, "lib", "libretinaface_amd.so"))


class RFError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"retinaface_amd error {status}: {message}")
        self.status = status


RF_OK, RF_ERR_INVALID_ARG, RF_ERR_IO, RF_ERR_MODEL, RF_ERR_HIP, RF_ERR_UNSUPPORTED, RF_ERR_TRUNCATED = 0, -1, -2, -3, -4, -5, -6


class rf_face(C.Structure):
    _fields_ = [("score", C.c_float), ("x1", C.c_float), ("y1", C.c_float), ("x2", C.c_float), ("y2", C.c_float),
                ("px", C.c_float * 5), ("py", C.c_float * 5)]


class rf_options(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("precision", C.c_int32), ("net_h", C.c_int32), ("net_w", C.c_int32),
                ("max_batch", C.c_int32), ("device", C.c_int32), ("max_candidates", C.c_int32),
                ("max_detections", C.c_int32), ("use_graph", C.c_int32), ("keep_outputs", C.c_int32),
                ("model_stem", C.c_char_p), ("lanes", C.c_int32), ("coalesce", C.c_int32),
                ("copy_threads", C.c_int32), ("n_devices", C.c_int32), ("devices", C.POINTER(C.c_int32)),
                ("plan_cache", C.c_int32), ("oversize_resize", C.c_int32)]


# every symbol include/retinaface_amd.h declares: name -> (restype, argtypes)
_PP = C.POINTER
SYMBOLS = {
    "rf_abi_version": (C.c_int, []),
    "rf_create": (C.c_int, [C.c_char_p, C.c_char_p, C.c_float, _PP(rf_options), _PP(C.c_void_p)]),
    "rf_preset_anchors": (C.c_int, [C.c_char_p, C.c_int, _PP(C.c_float), C.c_int]),
    "rf_destroy": (None, [C.c_void_p]),
    "rf_last_error": (C.c_char_p, [C.c_void_p]),
    "rf_get_net_size": (C.c_int, [C.c_void_p, _PP(C.c_int), _PP(C.c_int), _PP(C.c_int)]),
    "rf_detect_batch": (C.c_int, [C.c_void_p, _PP(C.c_void_p), _PP(C.c_int), _PP(C.c_int), _PP(C.c_int), C.c_int,
                                  C.c_float, _PP(rf_face), C.c_int, _PP(C.c_int)]),
    "rf_detect_batch_device": (C.c_int, [C.c_void_p, _PP(C.c_void_p), _PP(C.c_int), _PP(C.c_int), _PP(C.c_int),
                                         C.c_int, C.c_float, _PP(rf_face), C.c_int, _PP(C.c_int)]),
    "rf_detect_batch_pad32": (C.c_int, [C.c_void_p, _PP(C.c_void_p), _PP(C.c_int), _PP(C.c_int), _PP(C.c_int), C.c_int,
                                        C.c_int, C.c_float, _PP(rf_face), C.c_int, _PP(C.c_int)]),
    "rf_frame_scale": (C.c_float, [C.c_void_p, C.c_int, C.c_int]),
    "rf_num_slots": (C.c_int, [C.c_void_p]),
    "rf_enqueue_batch_device": (C.c_int, [C.c_void_p, _PP(C.c_void_p), _PP(C.c_int), _PP(C.c_int), _PP(C.c_int),
                                          C.c_int, C.c_float, _PP(C.c_int)]),
    "rf_wait": (C.c_int, [C.c_void_p, C.c_int, _PP(rf_face), C.c_int, _PP(C.c_int)]),
    "rf_enqueue_batch": (C.c_int, [C.c_void_p, _PP(C.c_void_p), _PP(C.c_int), _PP(C.c_int), _PP(C.c_int),
                                   C.c_int, C.c_float, _PP(C.c_int)]),
    "rf_host_register": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "rf_host_unregister": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rf_invalidate_residency": (C.c_int, [C.c_void_p]),
    "rf_num_devices": (C.c_int, [C.c_void_p]),
    "rf_scatter_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "rf_last_anchor_indices": (C.c_int, [C.c_void_p, C.c_int, _PP(C.c_int32), C.c_int]),
    "rf_last_candidate_counts": (C.c_int, [C.c_void_p, _PP(C.c_int), C.c_int]),
    "rf_last_timings": (C.c_int, [C.c_void_p, _PP(C.c_float), _PP(C.c_float), _PP(C.c_float), _PP(C.c_float)]),
    "rf_get_output": (C.c_long, [C.c_void_p, C.c_char_p, C.c_int, _PP(C.c_float), C.c_size_t]),
    "rf_debug_activation": (C.c_long, [C.c_void_p, C.c_char_p, C.c_int, _PP(C.c_float), C.c_size_t, _PP(C.c_int)]),
    "rf_profile": (C.c_int, [C.c_void_p, _PP(C.c_void_p), C.c_int, C.c_int, C.c_int, _PP(C.c_char_p), _PP(C.c_char_p),
                             _PP(C.c_float), _PP(C.c_double), _PP(C.c_double)]),
    "rf_profile_compulsory_bytes": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _PP(C.c_double)]),
    "rf_convert_model": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]),
    "rf_plan_cache_probe": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, _PP(C.c_size_t)]),
    "rf_plan_folded": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, _PP(C.c_float), C.c_size_t, _PP(C.c_float),
                                 C.c_size_t, _PP(C.c_int)]),
    "rf_plan_int8_gemm": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, _PP(C.c_float), C.c_size_t, _PP(C.c_float), C.c_size_t,
                                    _PP(C.c_float), _PP(C.c_float), C.c_size_t, _PP(C.c_int)]),
    "rf_attach_calibration": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]),
}

ABI_VERSION = 2      # include/retinaface_amd.h RF_ABI_VERSION
_lib = None


def load_library() -> C.CDLL:
    """Load the HIP library; there is no fallback -- a missing build is a hard error."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm wheels bundle their own libamdhip64.so / libhsa-runtime64.so (same SONAME as /opt/rocm's).  If this
    # library were loaded first it would pull in /opt/rocm's copy and a later `import torch` would bring a SECOND HIP
    # runtime into the process (the second one then finds no device).  Importing torch first makes both share one.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           f"or `make -C retinaface_amd/csrc` (there is no CPU fallback)")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.rf_abi_version() != ABI_VERSION:
        raise RuntimeError("libretinaface_amd.so ABI version mismatch")
    _lib = lib
    return lib


def abi_version() -> int:
    return load_library().rf_abi_version()


def check(status: int, handle=None) -> int:
    if status >= 0 or status == RF_ERR_TRUNCATED:
        return status
    msg = load_library().rf_last_error(handle)
    raise RFError(status, msg.decode("utf-8", "replace") if msg else "")
