"""Host-side mirror of the reference's ``RetinaFace`` class (retinaface/RetinaFace.h:63-78) over the C ABI.

Same constructor arguments (model directory, network preset, NMS threshold), same two entry points --
``detect(img, threshold)`` and ``detectBatchImages(imgs, threshold)`` -- taking OpenCV-style ``uint8``
H x W x 3 BGR arrays.  Unlike the reference, which returns ``void`` and drops its result
(RetinaFace.cpp:726-747), the detections are returned.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import rf_face, rf_options

PRECISION_FP32, PRECISION_FP16, PRECISION_INT8 = 0, 1, 2


@dataclass
class Detection:
    """FaceDetectInfo (RetinaFace.h:37-42) + the global anchor index that produced it."""
    score: float
    rect: tuple            # x1, y1, x2, y2 in network-input pixels
    xs: tuple              # 5 landmark x
    ys: tuple              # 5 landmark y
    anchor_index: int

    def as_row(self) -> np.ndarray:
        return np.array([self.score, *self.rect, *self.xs, *self.ys], dtype=np.float32)


def _faces_to_array(buf, n: int) -> np.ndarray:
    return np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_float)), shape=(n, 15)).copy()


class RetinaFace:
    def __init__(self, model: str, network: str = "net3", nms: float = 0.4, *, precision: int = PRECISION_FP16,
                 net_hw: Optional[tuple] = None, max_batch: int = 8, model_stem: Optional[str] = None,
                 max_candidates: int = 0, max_detections: int = 0, use_graph: bool = True,
                 keep_outputs: bool = False, device: Optional[int] = None, lanes: int = 0, coalesce: int = 0,
                 devices: Optional[Sequence[int]] = None, copy_threads: int = 0, plan_cache: bool = True,
                 oversize_resize: str = "area"):
        self._lib = _lib.load_library()
        o = rf_options()
        o.struct_size = C.sizeof(rf_options)
        o.precision = precision
        if net_hw:
            o.net_h, o.net_w = int(net_hw[0]), int(net_hw[1])
        o.max_batch = max_batch
        o.device = 0 if device is None else device + 1
        o.max_candidates = max_candidates
        o.max_detections = max_detections
        o.use_graph = 1 if use_graph else 2
        o.keep_outputs = 1 if keep_outputs else 0
        o.lanes = lanes
        o.coalesce = coalesce
        o.copy_threads = copy_threads
        o.plan_cache = 1 if plan_cache else 2
        o.oversize_resize = {"area": 1, "bilinear": 2}[oversize_resize]
        if devices:       # more than one entry: one engine per entry, detectBatchImages sharded by image over them
            self._devices = (C.c_int32 * len(devices))(*devices)
            o.devices, o.n_devices = self._devices, len(devices)
        self._stem = model_stem.encode() if model_stem else None
        o.model_stem = self._stem
        h = C.c_void_p()
        _lib.check(self._lib.rf_create(model.encode(), network.encode(), float(nms), C.byref(o), C.byref(h)))
        self._h = h
        nh, nw, mb = C.c_int(), C.c_int(), C.c_int()
        self._lib.rf_get_net_size(self._h, C.byref(nh), C.byref(nw), C.byref(mb))
        self.net_h, self.net_w, self.max_batch = nh.value, nw.value, mb.value
        self.max_detections = max_detections or 256
        self.truncated = False

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ reference entry points
    def detect(self, img: np.ndarray, threshold: float = 0.5, scales: float = 1.0) -> List[Detection]:
        """RetinaFace::detect(const Mat&, float threshold = 0.5, float scales = 1.0); `scales` is unused there too."""
        if img is None or img.size == 0:
            return []
        return self.detectBatchImages([img], threshold)[0]

    def detectBatchImages(self, imgs: Sequence[np.ndarray], threshold: float = 0.5) -> List[List[Detection]]:
        n = len(imgs)
        if n == 0:
            return []
        ptrs = (C.c_void_p * n)()
        rows, cols, steps = (C.c_int * n)(), (C.c_int * n)(), (C.c_int * n)()
        keep = []
        for i, im in enumerate(imgs):
            if im is None or im.size == 0:
                ptrs[i], rows[i], cols[i], steps[i] = None, 0, 0, 0
                continue
            if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
                raise ValueError("frames must be uint8 H x W x 3 (CV_8UC3, BGR)")
            if im.strides[2] != 1 or im.strides[1] != 3:
                im = np.ascontiguousarray(im)
            keep.append(im)
            ptrs[i], rows[i], cols[i], steps[i] = im.ctypes.data, im.shape[0], im.shape[1], im.strides[0]
        return self._run(self._lib.rf_detect_batch, ptrs, rows, cols, steps, n, threshold)

    def frame_scale(self, rows: int, cols: int) -> float:
        """Multiply returned coordinates by this to get source-frame pixels (1.0 unless the frame is larger than the net)."""
        return float(self._lib.rf_frame_scale(self._h, int(rows), int(cols)))

    def detect_pad32(self, imgs: Sequence[np.ndarray], threshold: float = 0.5) -> List[List[Detection]]:
        """The reference's Caffe-build detect (RetinaFace.cpp:943-1075): no resize, each frame zero-padded to the next
        multiple of 32 and run at that size, boxes clipped to the padded size, coordinates in source-frame pixels.
        anchor_index of the returned detections is -1 (anchor tables differ per size)."""
        n = len(imgs)
        if n == 0:
            return []
        ptrs = (C.c_void_p * n)()
        rows, cols, steps = (C.c_int * n)(), (C.c_int * n)(), (C.c_int * n)()
        keep = []
        for i, im in enumerate(imgs):
            if im is None or im.size == 0:
                ptrs[i], rows[i], cols[i], steps[i] = None, 0, 0, 0
                continue
            if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
                raise ValueError("frames must be uint8 H x W x 3 (CV_8UC3, BGR)")
            if im.strides[2] != 1 or im.strides[1] != 3:
                im = np.ascontiguousarray(im)
            keep.append(im)
            ptrs[i], rows[i], cols[i], steps[i] = im.ctypes.data, im.shape[0], im.shape[1], im.strides[0]
        cap = self.max_detections
        out = (rf_face * (n * cap))()
        counts = (C.c_int * n)()
        st = _lib.check(self._lib.rf_detect_batch_pad32(self._h, ptrs, rows, cols, steps, n, 0, float(threshold), out, cap, counts), self._h)
        self.truncated = st == _lib.RF_ERR_TRUNCATED
        return self._collect(out, counts, n, cap, anchors=False)

    # ------------------------------------------------------------------ device-resident frames
    def detect_device(self, ptrs: Sequence[int], rows: Sequence[int], cols: Sequence[int], threshold: float = 0.5,
                      steps: Optional[Sequence[int]] = None) -> List[List[Detection]]:
        n = len(ptrs)
        p = (C.c_void_p * n)(*ptrs)
        r, c = (C.c_int * n)(*rows), (C.c_int * n)(*cols)
        s = (C.c_int * n)(*(steps if steps is not None else [3 * x for x in cols]))
        return self._run(self._lib.rf_detect_batch_device, p, r, c, s, n, threshold)

    def enqueue_device(self, ptrs, rows, cols, threshold: float = 0.5) -> int:
        n = len(ptrs)
        p = (C.c_void_p * n)(*ptrs)
        r, c = (C.c_int * n)(*rows), (C.c_int * n)(*cols)
        s = (C.c_int * n)(*[3 * x for x in cols])
        t = C.c_int()
        _lib.check(self._lib.rf_enqueue_batch_device(self._h, p, r, c, s, n, float(threshold), C.byref(t)), self._h)
        return t.value

    def enqueue_host(self, imgs: Sequence[np.ndarray], threshold: float = 0.5) -> int:
        """rf_enqueue_batch: frames in host memory (numpy, OpenCV layout); they are staged before this returns."""
        return self.enqueue_prepared_host(self.prepare_host_batch(imgs), threshold)

    def prepare_host_batch(self, imgs: Sequence[np.ndarray]):
        """C argument arrays of rf_enqueue_batch for host frames that are submitted repeatedly (what a C caller keeps)."""
        n = len(imgs)
        keep = [np.ascontiguousarray(im) if (im.strides[2] != 1 or im.strides[1] != 3) else im for im in imgs]
        return ((C.c_void_p * n)(*[im.ctypes.data for im in keep]), (C.c_int * n)(*[im.shape[0] for im in keep]),
                (C.c_int * n)(*[im.shape[1] for im in keep]), (C.c_int * n)(*[im.strides[0] for im in keep]), n, C.c_int(), keep)

    def enqueue_prepared_host(self, batch, threshold: float = 0.5) -> int:
        p, r, c, s, n, t, _keep = batch
        _lib.check(self._lib.rf_enqueue_batch(self._h, p, r, c, s, n, threshold, C.byref(t)), self._h)
        return t.value

    def host_register(self, arr: np.ndarray) -> None:
        """Pin a caller-owned buffer (rf_host_register): frames inside it are DMA'd in place, without the staging copy."""
        _lib.check(self._lib.rf_host_register(self._h, arr.ctypes.data, arr.nbytes), self._h)

    def host_unregister(self, arr: np.ndarray) -> None:
        _lib.check(self._lib.rf_host_unregister(self._h, arr.ctypes.data), self._h)

    def invalidate_residency(self) -> None:
        """Forget where device frame pointers live (call after freeing / re-allocating frame buffers; rf_invalidate_residency)."""
        _lib.check(self._lib.rf_invalidate_residency(self._h), self._h)

    def num_devices(self) -> int:
        return self._lib.rf_num_devices(self._h)

    def scatter_stats(self) -> dict:
        """rf_scatter_stats: device frames pulled from other GPUs since the handle was built, and the peer copies that carried them."""
        f, c = C.c_longlong(), C.c_longlong()
        _lib.check(self._lib.rf_scatter_stats(self._h, C.byref(f), C.byref(c)), self._h)
        return {"frames": f.value, "peer_copies": c.value}

    def prepare_device_batch(self, ptrs, rows, cols, steps=None):
        """Build the C argument arrays of rf_enqueue_batch_device once for a batch of device frames that is submitted
        repeatedly (a ring of camera buffers): what a C/C++ caller keeps on its side anyway.  Returns an opaque batch."""
        n = len(ptrs)
        steps = [3 * x for x in cols] if steps is None else steps
        return ((C.c_void_p * n)(*ptrs), (C.c_int * n)(*rows), (C.c_int * n)(*cols), (C.c_int * n)(*steps), n, C.c_int())

    def enqueue_prepared(self, batch, threshold: float = 0.5) -> int:
        p, r, c, s, n, t = batch
        _lib.check(self._lib.rf_enqueue_batch_device(self._h, p, r, c, s, n, threshold, C.byref(t)), self._h)
        return t.value

    def wait(self, ticket: int, n: int) -> List[List[Detection]]:
        cap = self.max_detections
        out = (rf_face * (n * cap))()
        counts = (C.c_int * n)()
        st = _lib.check(self._lib.rf_wait(self._h, ticket, out, cap, counts), self._h)
        self.truncated = st == _lib.RF_ERR_TRUNCATED
        return self._collect(out, counts, n, cap)

    def wait_counts(self, ticket: int, n: int) -> List[int]:
        """rf_wait without materialising Python objects (benchmark loop)."""
        if not hasattr(self, "_wc_buf") or len(self._wc_buf[1]) < n:
            self._wc_buf = ((rf_face * (self.max_batch * self.max_detections))(), (C.c_int * self.max_batch)())
        out, counts = self._wc_buf
        st = _lib.check(self._lib.rf_wait(self._h, ticket, out, self.max_detections, counts), self._h)
        self.truncated = st == _lib.RF_ERR_TRUNCATED
        return list(counts[:n])

    def last_wait_faces(self) -> np.ndarray:
        """The result block the most recent wait_counts() filled, as a (max_batch, max_detections, 15) float view (no copy)."""
        return np.ctypeslib.as_array(C.cast(self._wc_buf[0], C.POINTER(C.c_float)),
                                     shape=(self.max_batch, self.max_detections, 15))

    def num_slots(self) -> int:
        return self._lib.rf_num_slots(self._h)

    # ------------------------------------------------------------------ inspection
    def last_timings(self):
        a, b, c, d = C.c_float(), C.c_float(), C.c_float(), C.c_float()
        self._lib.rf_last_timings(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        return {"pre_ms": a.value, "infer_ms": b.value, "post_ms": c.value, "total_ms": d.value}

    def last_candidate_counts(self, n: int) -> List[int]:
        buf = (C.c_int * n)()
        _lib.check(self._lib.rf_last_candidate_counts(self._h, buf, n), self._h)
        return list(buf)

    def get_output(self, blob: str, image: int = 0) -> np.ndarray:
        """blob_by_name(name)->result[image] (trtretinafacenet.cpp:104-114) as a (C, H, W) fp32 array."""
        need = self._lib.rf_get_output(self._h, blob.encode(), image, None, 0)
        if need < 0:
            _lib.check(int(need), self._h)
        arr = np.empty(need, dtype=np.float32)
        got = self._lib.rf_get_output(self._h, blob.encode(), image, arr.ctypes.data_as(C.POINTER(C.c_float)), need)
        if got < 0:
            _lib.check(int(got), self._h)
        stride = int(blob.rsplit("stride", 1)[1])
        return arr.reshape(-1, self.net_h // stride, self.net_w // stride)

    def debug_activation(self, blob: str, image: int = 0) -> np.ndarray:
        """Internal NHWC activation named after the reference blob it equals, as a (H, W, C) fp32 array."""
        dims = (C.c_int * 3)()
        need = self._lib.rf_debug_activation(self._h, blob.encode(), image, None, 0, dims)
        if need < 0:
            _lib.check(int(need), self._h)
        arr = np.empty(need, dtype=np.float32)
        got = self._lib.rf_debug_activation(self._h, blob.encode(), image, arr.ctypes.data_as(C.POINTER(C.c_float)),
                                            need, dims)
        if got < 0:
            _lib.check(int(got), self._h)
        return arr.reshape(dims[0], dims[1], dims[2])

    def profile(self, ptrs: Sequence[int], iters: int = 20):
        """Per-kernel HIP-event timing: list of dicts {name, kernel, ms, alg_bytes, macs, compulsory_bytes} in launch order
        (alg_bytes: layer-wise; compulsory_bytes: what the fused launch must move through HBM at least)."""
        n = len(ptrs)
        p = (C.c_void_p * n)(*ptrs)
        cap = 128
        names = (C.c_char_p * cap)()
        kernels = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        ab = (C.c_double * cap)()
        mc = (C.c_double * cap)()
        k = _lib.check(self._lib.rf_profile(self._h, p, n, iters, cap, names, kernels, ms, ab, mc), self._h)
        cb = (C.c_double * cap)()
        _lib.check(self._lib.rf_profile_compulsory_bytes(self._h, n, cap, cb), self._h)
        return [{"name": names[i].decode(), "kernel": kernels[i].decode(), "ms": ms[i], "alg_bytes": ab[i], "macs": mc[i],
                 "compulsory_bytes": cb[i]} for i in range(k)]

    # ------------------------------------------------------------------ internals
    def _run(self, fn, ptrs, rows, cols, steps, n, threshold):
        cap = self.max_detections
        out = (rf_face * (n * cap))()
        counts = (C.c_int * n)()
        st = _lib.check(fn(self._h, ptrs, rows, cols, steps, n, float(threshold), out, cap, counts), self._h)
        self.truncated = st == _lib.RF_ERR_TRUNCATED
        return self._collect(out, counts, n, cap)

    def _collect(self, out, counts, n, cap, anchors=True):
        rows = _faces_to_array(out, n * cap)
        res: List[List[Detection]] = []
        for i in range(n):
            k = min(counts[i], cap)
            idx = (C.c_int32 * max(k, 1))()
            got = self._lib.rf_last_anchor_indices(self._h, i, idx, k) if anchors else -1
            dets = []
            for j in range(k):
                r = rows[i * cap + j]
                dets.append(Detection(float(r[0]), tuple(float(v) for v in r[1:5]), tuple(float(v) for v in r[5:10]),
                                      tuple(float(v) for v in r[10:15]), int(idx[j]) if got >= 0 else -1))
            res.append(dets)
        return res
