"""Build / load oracle/_ref/libretinaface_ref.so: the reference's OWN retinaface/RetinaFace.cpp, compiled unmodified from
where it lies under /root/reference -- TEST INFRASTRUCTURE ONLY.

Recipe (the reference's build system is not run; it needs cmake + OpenCV + Caffe + TensorRT + CUDA):

    g++ -std=c++11 -O2 -fomit-frame-pointer  (the flags of reference CMakeLists.txt:30)
        -DUSE_TENSORRT                        (CMakeLists.txt:69, the default build; USE_NPP off -> the cv:: preprocess branch)
        -DTRTRETINAFACENET_H -include oracle/ref_shim/trt_shim.h   (skip the real TensorRT header, use the callback engine)
        -I oracle/ref_shim  -I /root/reference/retinaface
        /root/reference/retinaface/RetinaFace.cpp  oracle/ref_harness.cpp  -shared -fPIC -o oracle/_ref/libretinaface_ref.so

oracle/ref_shim/README.md lists exactly which pieces are the reference's code and which are third-party stand-ins.
The output lives in oracle/_ref/ (git-ignored, NOT gpurun-ignored: the prebuilt .so travels to the GPU box, where
/root/reference does not exist and `available()` simply reports what was built here).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Callable, List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE_DIR = os.environ.get("RETINAFACE_REFERENCE_DIR", "/root/reference")
_SRC = os.path.join(REFERENCE_DIR, "retinaface", "RetinaFace.cpp")
_HARNESS = os.path.join(_HERE, "ref_harness.cpp")
_SHIM = os.path.join(_HERE, "ref_shim")
_OUT = os.path.join(_HERE, "_ref", "libretinaface_ref.so")

HEAD_BLOBS = [f"{stem}{s}" for s in (32, 16, 8)
              for stem in ("face_rpn_cls_prob_reshape_stride", "face_rpn_bbox_pred_stride", "face_rpn_landmark_pred_stride")]


def can_build() -> bool:
    return os.path.exists(_SRC)


def build(force: bool = False) -> Optional[str]:
    """Compile when the reference sources are present; otherwise return the prebuilt library (or None)."""
    if can_build():
        deps = [_SRC, _HARNESS] + [os.path.join(r, f) for r, _, fs in os.walk(_SHIM) for f in fs]
        stale = not os.path.exists(_OUT) or os.path.getmtime(_OUT) < max(os.path.getmtime(d) for d in deps)
        if force or stale:
            os.makedirs(os.path.dirname(_OUT), exist_ok=True)
            subprocess.check_call([
                "g++", "-std=c++11", "-O2", "-fomit-frame-pointer", "-w", "-fPIC", "-shared",
                "-DUSE_TENSORRT", "-DTRTRETINAFACENET_H", "-include", os.path.join(_SHIM, "trt_shim.h"),
                "-I", _SHIM, "-I", os.path.join(REFERENCE_DIR, "retinaface"),
                _SRC, _HARNESS, "-o", _OUT])
    return _OUT if os.path.exists(_OUT) else None


def available() -> bool:
    return build() is not None


_FWD = C.CFUNCTYPE(None, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.c_void_p)
_lib = None


def _load() -> C.CDLL:
    global _lib
    if _lib is None:
        path = build()
        if path is None:
            raise RuntimeError("oracle/_ref/libretinaface_ref.so is not built and /root/reference is absent")
        l = C.CDLL(path)
        f32p, u8p = C.POINTER(C.c_float), C.POINTER(C.c_uint8)
        l.rfref_create.restype = C.c_int
        l.rfref_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_float, _FWD, C.c_void_p]
        l.rfref_create_net.restype = C.c_int
        l.rfref_create_net.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _FWD, C.c_void_p]
        l.rfref_num_levels.restype = C.c_int
        l.rfref_base_anchors.restype = C.c_int
        l.rfref_base_anchors.argtypes = [C.c_int, C.POINTER(C.c_float), C.c_int]
        l.rfref_destroy.restype = None
        l.rfref_set_output.restype = None
        l.rfref_set_output.argtypes = [C.c_char_p, C.c_int, f32p, C.c_size_t]
        l.rfref_postprocess.restype = C.c_int
        l.rfref_postprocess.argtypes = [C.c_int, C.c_float, f32p, C.c_int]
        l.rfref_detect.restype = C.c_int
        l.rfref_detect.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_float, f32p, C.c_int]
        l.rfref_detect_batch.restype = None
        l.rfref_detect_batch.argtypes = [C.POINTER(u8p), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_float]
        l.rfref_last_input.restype = C.c_long
        l.rfref_last_input.argtypes = [f32p, C.c_size_t]
        l.rfref_nms.restype = C.c_int
        l.rfref_nms.argtypes = [f32p, C.c_int, C.c_float, f32p, C.c_int]
        l.rfref_anchors.restype = C.c_long
        l.rfref_anchors.argtypes = [C.c_int, f32p, C.c_size_t]
        l.rfref_anchors_plane.restype = C.c_long
        l.rfref_anchors_plane.argtypes = [C.c_int, C.c_int, C.c_int, f32p, C.c_size_t]
        l.rfref_bbox_pred.restype = None
        l.rfref_bbox_pred.argtypes = [f32p, f32p, f32p]
        l.rfref_landmark_pred.restype = None
        l.rfref_landmark_pred.argtypes = [f32p, f32p, f32p]
        _lib = l
    return _lib


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class ReferenceRetinaFace:
    """The reference's RetinaFace object (TensorRT build) with the engine replaced by `forward`.

    forward(chw: np.ndarray [n,3,H,W] f32) -> list over images of the 9 head blobs (C,H,W f32) in HEAD_BLOBS order.
    Only one instance may exist at a time (the harness keeps a single global, like the reference's main.cpp)."""

    def __init__(self, net_h: int, net_w: int, forward: Optional[Callable[[np.ndarray], Sequence[Sequence[np.ndarray]]]] = None,
                 nms: float = 0.4, max_batch: int = 8, model_dir: str = "unused", network: str = "net3", head_anchors: int = 2):
        self._l = _load()
        self.net_h, self.net_w, self.max_batch = net_h, net_w, max_batch
        self._forward = forward
        self.forward_calls = 0

        def _cb(inp, n, h, w, _user):
            self.forward_calls += 1
            if self._forward is None:
                return
            x = np.ctypeslib.as_array(inp, shape=(n, 3, h, w)).copy()
            outs = self._forward(x)
            for i in range(n):
                self.set_heads(i, outs[i])

        self._cb = _FWD(_cb)          # keep alive
        self._l.rfref_create_net(model_dir.encode(), network.encode(), head_anchors, net_h, net_w, max_batch, nms, self._cb, None)

    def close(self):
        self._l.rfref_destroy()

    def num_levels(self) -> int:
        """Strides the constructor gave an anchor configuration (0 for the fmc != 3 presets)."""
        return self._l.rfref_num_levels()

    def base_anchors(self, stride: int) -> np.ndarray:
        """_anchors_fpn["stride<S>"] as built by the constructor for its `network` preset: (A, 4); A = 0 without ratios."""
        out = np.zeros((16, 4), np.float32)
        n = self._l.rfref_base_anchors(stride, _fp(out), 16)
        return out[:max(n, 0)].copy()

    def set_heads(self, image: int, heads9: Sequence[np.ndarray]):
        for name, a in zip(HEAD_BLOBS, heads9):
            a = np.ascontiguousarray(a, dtype=np.float32)
            self._l.rfref_set_output(name.encode(), image, _fp(a), a.size)

    def _faces(self, fn, cap=4096):
        out = np.zeros((cap, 15), np.float32)
        n = fn(out, cap)
        if n > cap:
            return self._faces(fn, cap=n)
        return out[:n].copy()

    def postprocess(self, image: int, threshold: float) -> np.ndarray:
        """RetinaFace::postProcess on the stored blobs: decode + NMS(0.4 literal) -> rows [score, x1,y1,x2,y2, x[5], y[5]] (FaceDetectInfo order)."""
        return self._faces(lambda o, cap: self._l.rfref_postprocess(image, threshold, _fp(o), cap))

    def detect(self, img_bgr: np.ndarray, threshold: float) -> np.ndarray:
        img = np.ascontiguousarray(img_bgr, dtype=np.uint8)
        p = img.ctypes.data_as(C.POINTER(C.c_uint8))
        return self._faces(lambda o, cap: self._l.rfref_detect(p, img.shape[0], img.shape[1], img.strides[0], threshold, _fp(o), cap))

    def detect_batch(self, imgs: List[np.ndarray], threshold: float) -> List[np.ndarray]:
        imgs = [np.ascontiguousarray(i, dtype=np.uint8) for i in imgs]
        u8p = C.POINTER(C.c_uint8)
        ptrs = (u8p * len(imgs))(*[i.ctypes.data_as(u8p) for i in imgs])
        rows = (C.c_int * len(imgs))(*[i.shape[0] for i in imgs])
        cols = (C.c_int * len(imgs))(*[i.shape[1] for i in imgs])
        self._l.rfref_detect_batch(ptrs, rows, cols, len(imgs), threshold)
        return [self.postprocess(i, threshold) for i in range(len(imgs))]

    def last_input(self) -> np.ndarray:
        n = self._l.rfref_last_input(None, 0)
        a = np.zeros(n, np.float32)
        self._l.rfref_last_input(_fp(a), n)
        return a.reshape(-1, 3, self.net_h, self.net_w)

    def nms(self, faces15: np.ndarray, threshold: float) -> np.ndarray:
        f = np.ascontiguousarray(faces15, dtype=np.float32).reshape(-1, 15)
        return self._faces(lambda o, cap: self._l.rfref_nms(_fp(f), len(f), threshold, _fp(o), cap), cap=max(len(f), 1))

    def anchors(self, stride: int) -> np.ndarray:
        n = self._l.rfref_anchors(stride, None, 0)
        a = np.zeros((n, 4), np.float32)
        self._l.rfref_anchors(stride, _fp(a), n)
        return a

    def anchors_plane(self, height: int, width: int, level: int) -> np.ndarray:
        n = self._l.rfref_anchors_plane(height, width, level, None, 0)
        a = np.zeros((n, 4), np.float32)
        self._l.rfref_anchors_plane(height, width, level, _fp(a), n)
        return a

    def bbox_pred(self, anchor4, regress4) -> np.ndarray:
        a, r, o = (np.asarray(anchor4, np.float32).copy(), np.asarray(regress4, np.float32).copy(), np.zeros(4, np.float32))
        self._l.rfref_bbox_pred(_fp(a), _fp(r), _fp(o))
        return o

    def landmark_pred(self, anchor4, pts10) -> np.ndarray:
        a, p, o = (np.asarray(anchor4, np.float32).copy(), np.asarray(pts10, np.float32).copy(), np.zeros(10, np.float32))
        self._l.rfref_landmark_pred(_fp(a), _fp(p), _fp(o))
        return o
