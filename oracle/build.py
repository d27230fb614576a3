"""Build / load the plain-C restatement of decode + NMS (oracle/csrc/rf_post_ref.c) -- test infrastructure.

No -mfma / -ffast-math: every float op must round where the reference's x86-64 build rounds.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "csrc", "rf_post_ref.c")
_OUT = os.path.join(_HERE, "_build", "librf_post_ref.so")


def build(force: bool = False) -> str:
    newest = max(os.path.getmtime(_SRC), os.path.getmtime(os.path.join(_HERE, "csrc", "cv_resize_linear.h")))
    if force or not os.path.exists(_OUT) or os.path.getmtime(_OUT) < newest:
        os.makedirs(os.path.dirname(_OUT), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-ffp-contract=off", "-o", _OUT, _SRC, "-lm"])
    return _OUT


class _Face(C.Structure):
    _fields_ = [("v", C.c_float * 15)]


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.rfo_decode.restype = C.c_int
        _lib.rfo_decode.argtypes = [C.POINTER(C.POINTER(C.c_float)), C.c_int, C.c_int, C.c_float, C.POINTER(_Face),
                                    C.POINTER(C.c_int32), C.c_int]
        _lib.rfo_nms.restype = C.c_int
        _lib.rfo_nms.argtypes = [C.POINTER(_Face), C.POINTER(C.c_int32), C.c_int, C.c_float]
        _lib.rfo_anchors_plane.restype = None
        _lib.rfo_anchors_plane.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
    return _lib


def cv_resize_linear(img, fx: float, fy: float):
    """C restatement of cv::resize(img, Size(), fx, fy), INTER_LINEAR, CV_8UC3 (csrc/cv_resize_linear.h)."""
    import numpy as np
    l = lib()
    l.rfo_cv_resize_dsize.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    l.rfo_cv_resize_linear.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_double]
    img = np.ascontiguousarray(img, dtype=np.uint8)
    dr, dc = C.c_int(), C.c_int()
    l.rfo_cv_resize_dsize(img.shape[0], img.shape[1], fx, fy, C.byref(dr), C.byref(dc))
    out = np.empty((dr.value, dc.value, 3), np.uint8)
    l.rfo_cv_resize_linear(img.ctypes.data, img.shape[0], img.shape[1], out.ctypes.data, fx, fy)
    return out


def decode_nms(heads9, net_h: int, net_w: int, threshold: float, nms_threshold: float, cap: int = 1 << 16):
    """heads9: the 9 blobs of ONE image (C,H,W fp32) in binding order (stride 32,16,8 x prob,bbox,landmark).
    Returns (candidate rows [n,15], candidate anchor idx [n], kept rows [k,15], kept anchor idx [k])."""
    l = lib()
    arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in heads9]
    ptrs = (C.POINTER(C.c_float) * 9)(*[a.ctypes.data_as(C.POINTER(C.c_float)) for a in arrs])
    faces = (_Face * cap)()
    idx = (C.c_int32 * cap)()
    n = l.rfo_decode(ptrs, net_h, net_w, threshold, faces, idx, cap)
    if n > cap:
        return decode_nms(heads9, net_h, net_w, threshold, nms_threshold, cap=n)
    cand = np.ctypeslib.as_array(C.cast(faces, C.POINTER(C.c_float)), shape=(cap, 15))[:n].copy()
    cidx = np.array(idx[:n], dtype=np.int32)
    k = l.rfo_nms(faces, idx, n, nms_threshold)
    kept = np.ctypeslib.as_array(C.cast(faces, C.POINTER(C.c_float)), shape=(cap, 15))[:k].copy()
    kidx = np.array(idx[:k], dtype=np.int32)
    return cand, cidx, kept, kidx


def anchors_plane(stride_index: int, h: int, w: int) -> np.ndarray:
    out = np.empty((2 * h * w, 4), np.float32)
    lib().rfo_anchors_plane(stride_index, h, w, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out
