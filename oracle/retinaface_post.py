"""Literal numpy restatement of the reference's host-side pre/post-processing
(test infrastructure, see oracle/__init__.py).  Every function cites what it follows.

Float discipline: the reference mixes ``float`` storage with ``double`` intermediates
(``0.5 * (w - 1)`` is double arithmetic in C++); the helpers below reproduce the same
rounding points with np.float32 / np.float64 scalars.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Sequence, Tuple

import numpy as np

f32 = np.float32
f64 = np.float64

# exp() at RetinaFace.cpp:389-390 resolves to std::exp(float) (<cmath> + `using namespace std`), i.e. glibc's expf.
# numpy's float32 exp is a different (SIMD) implementation that is 1 ulp off on some inputs -- found when this file was
# pinned against the reference build (oracle/build_ref.py) -- so call the same libm the reference links.
import ctypes as _C
import ctypes.util as _Cu

_libm = _C.CDLL(_Cu.find_library("m") or "libm.so.6")
_libm.expf.restype = _C.c_float
_libm.expf.argtypes = [_C.c_float]


def expf(x) -> np.float32:
    return f32(_libm.expf(float(f32(x))))

FEAT_STRIDES = (32, 16, 8)                       # RetinaFace.cpp:246 (_feat_stride_fpn)
ANCHOR_SCALES = {32: (32, 16), 16: (8, 4), 8: (2, 1)}   # RetinaFace.cpp:247-268
ANCHOR_BASE_SIZE = 16
ANCHOR_RATIOS = (1.0,)                           # network "net3": RetinaFace.cpp:215-217


# ------------------------------------------------------------------ anchors (RetinaFace.cpp:9-154)

def _whctrs(a):
    """RetinaFace.cpp:9-19"""
    x1, y1, x2, y2 = a
    w = f32(f32(x2 - x1) + f32(1))
    h = f32(f32(y2 - y1) + f32(1))
    x_ctr = f32(f64(x1) + 0.5 * (f64(w) - 1))
    y_ctr = f32(f64(y1) + 0.5 * (f64(h) - 1))
    return w, h, x_ctr, y_ctr


def _mkanchors(w, h, x_ctr, y_ctr):
    """RetinaFace.cpp:21-32"""
    return (f32(f64(x_ctr) - 0.5 * (f64(w) - 1)), f32(f64(y_ctr) - 0.5 * (f64(h) - 1)),
            f32(f64(x_ctr) + 0.5 * (f64(w) - 1)), f32(f64(y_ctr) + 0.5 * (f64(h) - 1)))


def _c_round(x: float) -> float:
    """std::round: half away from zero"""
    return float(np.floor(abs(x) + 0.5) * (1 if x >= 0 else -1))


def generate_anchors(base_size: int, ratios: Sequence[float], scales: Sequence[int]):
    """RetinaFace.cpp:34-103 (dense_anchor = false)"""
    base = (f32(0), f32(0), f32(base_size - 1), f32(base_size - 1))
    ratio_anchors = []
    for r in ratios:                                   # _ratio_enum, :34-51
        w, h, xc, yc = _whctrs(base)
        size = f32(w * h)
        scale = f32(size / f32(r))
        w2 = f32(_c_round(float(np.sqrt(f64(scale)))))     # std::round(sqrt(float)) -> sqrt is double overload? float
        h2 = f32(_c_round(float(f32(w2 * f32(r)))))
        ratio_anchors.append(_mkanchors(w2, h2, xc, yc))
    anchors = []
    for ra in ratio_anchors:                           # _scale_enum, :53-68
        for s in scales:
            w, h, xc, yc = _whctrs(ra)
            anchors.append(_mkanchors(f32(w * f32(s)), f32(h * f32(s)), xc, yc))
    return np.array(anchors, dtype=np.float32)         # [A, 4]


# `network` presets of the constructor (RetinaFace.cpp:209-271): only "net3" / "net3a" set ratios; every other name leaves the
# ratio list empty (-> zero anchors per stride) or, for fmc != 3, the whole cfg empty (-> no strides): either way no faces.
PRESET_RATIOS = {"net3": (1.0,), "net3a": (1.0, 1.5)}


def preset_ratios(network: str) -> Tuple[float, ...]:
    return PRESET_RATIOS.get(network, ())


def base_anchors(ratios: Sequence[float] = ANCHOR_RATIOS) -> Dict[int, np.ndarray]:
    """generate_anchors_fpn, RetinaFace.cpp:105-125, with the fmc == 3 cfg of :245-268 and the preset's ratios."""
    return {s: generate_anchors(ANCHOR_BASE_SIZE, ratios, ANCHOR_SCALES[s]).reshape(-1, 4)
            for s in FEAT_STRIDES}


def anchors_plane(height: int, width: int, stride: int, base: np.ndarray) -> np.ndarray:
    """RetinaFace.cpp:127-154.  Order: k-major, then row, then col -> [A*H*W, 4]."""
    out = np.empty((base.shape[0], height, width, 4), dtype=np.float32)
    sw = (np.arange(width, dtype=np.int64) * stride).astype(np.float32)
    sh = (np.arange(height, dtype=np.int64) * stride).astype(np.float32)
    for k in range(base.shape[0]):
        out[k, :, :, 0] = base[k, 0] + sw[None, :]
        out[k, :, :, 1] = base[k, 1] + sh[:, None]
        out[k, :, :, 2] = base[k, 2] + sw[None, :]
        out[k, :, :, 3] = base[k, 3] + sh[:, None]
    return out.reshape(-1, 4)


def anchor_offsets(net_h: int, net_w: int, anchors_per_cell: int = 2) -> Dict[int, int]:
    """Global anchor index = offset(stride) + a*h*w + iy*w + ix, strides visited 32,16,8
    (visiting order of RetinaFace.cpp:667,:1000)."""
    offs, acc = {}, 0
    for s in FEAT_STRIDES:
        offs[s] = acc
        acc += anchors_per_cell * (net_h // s) * (net_w // s)
    offs["total"] = acc  # type: ignore[index]
    return offs


# ------------------------------------------------------------------ preprocess

def preprocess_caffe(img_bgr: np.ndarray) -> Tuple[np.ndarray, int, int]:
    """Caffe variant, RetinaFace.cpp:950-981: zero-pad right/bottom to a multiple of 32,
    u8 -> f32, BGR -> RGB, HWC -> CHW, raw 0..255 (no mean, no scale)."""
    assert img_bgr.dtype == np.uint8 and img_bgr.ndim == 3 and img_bgr.shape[2] == 3
    rows, cols = img_bgr.shape[:2]
    ws = (cols + 31) // 32 * 32
    hs = (rows + 31) // 32 * 32
    padded = np.zeros((hs, ws, 3), dtype=np.uint8)
    padded[:rows, :cols] = img_bgr
    chw = padded[:, :, ::-1].astype(np.float32).transpose(2, 0, 1)
    return np.ascontiguousarray(chw[None]), hs, ws


def preprocess_trt_identity(img_bgr: np.ndarray, net_h: int, net_w: int) -> np.ndarray:
    """TRT+NPP variant for frames that already fit the net (scale factor clamps to 1):
    RetinaFace.cpp:594-608 + resizeconvertion.cu:298-303 -- 1:1 copy into the top-left of a
    zeroed net_h x net_w buffer, then BGR->RGB f32 CHW."""
    rows, cols = img_bgr.shape[:2]
    assert rows <= net_h and cols <= net_w
    padded = np.zeros((net_h, net_w, 3), dtype=np.uint8)
    padded[:rows, :cols] = img_bgr
    chw = padded[:, :, ::-1].astype(np.float32).transpose(2, 0, 1)
    return np.ascontiguousarray(chw[None])


def cv_resize_linear(img: np.ndarray, fx: float, fy: float) -> np.ndarray:
    """cv::resize(img, dst, Size(), fx, fy) with the default INTER_LINEAR for CV_8UC3: numpy twin of
    oracle/csrc/cv_resize_linear.h (OpenCV's published legacy fixed-point path; see that header for the algorithm and for why
    it is a restatement of a third-party dependency: parity unpinned)."""
    assert img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 3
    rows, cols = img.shape[:2]
    dcols, drows = int(np.rint(f64(cols) * f64(fx))), int(np.rint(f64(rows) * f64(fy)))          # cvRound: half to even
    scale_x, scale_y = f64(1.0) / f64(fx), f64(1.0) / f64(fy)

    def taps(n_dst, n_src, scale, clamp):
        f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        if clamp:                                                    # columns: fraction forced to 0 at the borders
            lo, hi = s < 0, s >= n_src - 1
            f = np.where(lo | hi, f32(0), f)
            s = np.where(lo, 0, np.where(hi, n_src - 1, s))
        a0 = np.clip(np.rint((f32(1) - f) * f32(2048)), -32768, 32767).astype(np.int64)
        a1 = np.clip(np.rint(f * f32(2048)), -32768, 32767).astype(np.int64)
        return s, a0, a1

    sx, a0, a1 = taps(dcols, cols, scale_x, True)
    sy, b0, b1 = taps(drows, rows, scale_y, False)
    sx1 = np.minimum(sx + 1, cols - 1)
    y0, y1 = np.clip(sy, 0, rows - 1), np.clip(sy + 1, 0, rows - 1)
    src = img.astype(np.int64)
    h = src[:, sx] * a0[None, :, None] + src[:, sx1] * a1[None, :, None]            # horizontal pass, all source rows
    out = (((b0[:, None, None] * (h[y0] >> 4)) >> 16) + ((b1[:, None, None] * (h[y1] >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def resize_area_reference(img_bgr: np.ndarray, net_h: int, net_w: int) -> np.ndarray:
    """Independent statement of what the product's area-average step computes for an oversize frame (the NPP build's
    NPPI_INTER_SUPER is closed source, so this is a DEFINITION, not a pin): factor f = min(netW / cols, netH / rows) < 1, the
    destination is floor(cols f) x floor(rows f) pixels top-left on a zero canvas, and destination pixel (y, x) is the mean of
    the source over the rectangle [x / f, (x + 1) / f) x [y / f, (y + 1) / f) with fractional coverage at the edges.  Written
    as two coverage matrices (rows and columns) -- a different formulation from the kernel's per-pixel loops."""
    rows, cols = img_bgr.shape[:2]
    f = min(f32(net_w) / f32(cols), f32(net_h) / f32(rows))
    f = f32(min(f, f32(1.0)))
    dw, dh = int(f32(cols) * f), int(f32(rows) * f)
    inv = f64(f32(1.0) / f)

    def coverage(n_dst, n_src):
        lo = np.arange(n_dst, dtype=np.float64) * inv
        hi = np.minimum((np.arange(n_dst, dtype=np.float64) + 1) * inv, n_src)
        edges = np.arange(n_src + 1, dtype=np.float64)
        return np.clip(np.minimum(hi[:, None], edges[None, 1:]) - np.maximum(lo[:, None], edges[None, :-1]), 0, None)

    wy, wx = coverage(dh, rows), coverage(dw, cols)
    src = img_bgr.astype(np.float64)
    tmp = (wy @ src.reshape(rows, cols * 3)).reshape(dh, cols, 3)             # rows first, then columns: two small matmuls
    acc = np.einsum("xc,ycs->yxs", wx, tmp, optimize=True)
    mean = acc / (wy.sum(1)[:, None, None] * wx.sum(1)[None, :, None])
    out = np.zeros((net_h, net_w, 3), np.uint8)
    out[:dh, :dw] = np.clip(np.rint(mean), 0, 255).astype(np.uint8)
    return out


def preprocess_trt_cvresize(img_bgr: np.ndarray, net_h: int, net_w: int) -> np.ndarray:
    """The TensorRT build WITHOUT NPP (CMake default), RetinaFace.cpp:585-647: scale = max(cols / netW, rows / netH, 1) in float;
    frames that fit are zero-padded bottom / right to the net size (:621-624, identical to preprocess_trt_identity); larger frames
    are shrunk by cv::resize(img, Size(), 1 / scale, 1 / scale) (bilinear) and padded on the ONE side that is short (:611-620 --
    the code assumes the other side came out at exactly the net size, which holds whenever cvRound(side / scale) does)."""
    rows, cols = img_bgr.shape[:2]
    sw = f32(f64(1.0) * f64(cols) / f64(net_w))        # `float sw = 1.0 * img.cols / inputW`: double arithmetic, stored to float
    sh = f32(f64(1.0) * f64(rows) / f64(net_h))
    scale = sw if sw > sh else sh
    scale = scale if scale > 1.0 else f32(1.0)
    if scale > 1:
        inv = f64(f32(1) / scale)                   # `1 / scale`: int / float -> float, then widened to cv::resize's double fx
        small = cv_resize_linear(img_bgr, inv, inv)
        if sw > sh:
            assert small.shape[1] == net_w, "the reference pads only the bottom here"
        else:
            assert small.shape[0] == net_h, "the reference pads only the right side here"
        assert small.shape[0] <= net_h and small.shape[1] <= net_w
        img_bgr = small
    return preprocess_trt_identity(img_bgr, net_h, net_w)


# ------------------------------------------------------------------ regression (RetinaFace.cpp:378-432)

def bbox_pred(anchor, regress):
    """RetinaFace.cpp:378-398"""
    x1, y1, x2, y2 = (f32(v) for v in anchor)
    dx, dy, dw, dh = (f32(v) for v in regress)
    width = f32(f32(x2 - x1) + f32(1))
    height = f32(f32(y2 - y1) + f32(1))
    ctr_x = f32(f64(x1) + 0.5 * (f64(width) - 1.0))
    ctr_y = f32(f64(y1) + 0.5 * (f64(height) - 1.0))
    pred_ctr_x = f32(f32(dx * width) + ctr_x)
    pred_ctr_y = f32(f32(dy * height) + ctr_y)
    pred_w = f32(expf(dw) * width)
    pred_h = f32(expf(dh) * height)
    return (f32(f64(pred_ctr_x) - 0.5 * (f64(pred_w) - 1.0)),
            f32(f64(pred_ctr_y) - 0.5 * (f64(pred_h) - 1.0)),
            f32(f64(pred_ctr_x) + 0.5 * (f64(pred_w) - 1.0)),
            f32(f64(pred_ctr_y) + 0.5 * (f64(pred_h) - 1.0)))


def clip_box(box, width: int, height: int):
    """RetinaFace.cpp:179-199 (one-sided clips)"""
    x1, y1, x2, y2 = box
    if x1 < 0:
        x1 = f32(0)
    if y1 < 0:
        y1 = f32(0)
    if x2 > width - 1:
        x2 = f32(width - 1)
    if y2 > height - 1:
        y2 = f32(height - 1)
    return x1, y1, x2, y2


def landmark_pred(anchor, pts_xy):
    """RetinaFace.cpp:418-432; pts_xy = 10 floats x0,y0,...,x4,y4 -> (xs[5], ys[5])"""
    x1, y1, x2, y2 = (f32(v) for v in anchor)
    width = f32(f32(x2 - x1) + f32(1))
    height = f32(f32(y2 - y1) + f32(1))
    ctr_x = f32(f64(x1) + 0.5 * (f64(width) - 1.0))
    ctr_y = f32(f64(y1) + 0.5 * (f64(height) - 1.0))
    xs = [f32(f32(f32(pts_xy[2 * k]) * width) + ctr_x) for k in range(5)]
    ys = [f32(f32(f32(pts_xy[2 * k + 1]) * height) + ctr_y) for k in range(5)]
    return xs, ys


@dataclass
class Detection:
    score: np.float32
    rect: Tuple[np.float32, np.float32, np.float32, np.float32]
    xs: List[np.float32]
    ys: List[np.float32]
    anchor_index: int   # global anchor index

    def as_row(self) -> np.ndarray:
        """15 floats in FaceDetectInfo order (RetinaFace.h:37-42): score, x1,y1,x2,y2, x[5], y[5]"""
        return np.array([self.score, *self.rect, *self.xs, *self.ys], dtype=np.float32)


def decode(heads: Dict[str, np.ndarray], net_h: int, net_w: int, threshold: float,
           image: int = 0, ratios: Sequence[float] = ANCHOR_RATIOS) -> List[Detection]:
    """The threshold scan + regression loop, RetinaFace.cpp:666-724 (== :999-1072).
    `heads` maps the 9 output blob names to NCHW arrays; anchors come from anchors_plane
    on the blob's own H x W as at :301 / :1035."""
    thr = f32(threshold)
    base = base_anchors(ratios)
    out: List[Detection] = []
    goff = 0
    for s in FEAT_STRIDES:
        prob = heads[f"face_rpn_cls_prob_reshape_stride{s}"][image]
        bbox = heads[f"face_rpn_bbox_pred_stride{s}"][image]
        lmk = heads[f"face_rpn_landmark_pred_stride{s}"][image]
        h, w = prob.shape[1:]
        count = h * w
        score = prob.reshape(-1)[prob.size // 2:]            # second half of the blob, :671-674
        bbox_f = bbox.reshape(-1)
        lmk_f = lmk.reshape(-1)
        num_anchor = base[s].shape[0]                        # _num_anchors[key], :297
        if num_anchor == 0:
            continue                                         # a preset without ratios: the a-loop of :684 runs zero times
        assert prob.shape[0] == 2 * num_anchor, "the reference would read past the score blob here"
        anchors = anchors_plane(h, w, s, base[s])
        for idx in np.nonzero(score > thr)[0]:               # same visiting order as the a/j loops
            a, j = divmod(int(idx), count)
            conf = score[j + count * a]
            regress = [bbox_f[j + count * (c + a * 4)] for c in range(4)]
            anchor = anchors[j + count * a]
            rect = clip_box(bbox_pred(anchor, regress), net_w, net_h)
            pts = []
            for k in range(5):
                pts.append(lmk_f[j + count * (a * 10 + k * 2)])
                pts.append(lmk_f[j + count * (a * 10 + k * 2 + 1)])
            xs, ys = landmark_pred(anchor, pts)
            out.append(Detection(f32(conf), rect, xs, ys, goff + a * count + j))
        goff += num_anchor * count
    return out


# ------------------------------------------------------------------ NMS (RetinaFace.cpp:434-492)

def nms(dets: List[Detection], threshold: float) -> List[Detection]:
    """Greedy NMS of RetinaFace.cpp:439-492.  std::sort there is unstable, so ties are
    undefined in the reference; this repo fixes the total order (score desc, global anchor
    index asc) -- which equals a *stable* sort of the decode order."""
    thr = f32(threshold)
    order = sorted(range(len(dets)), key=lambda i: (-float(dets[i].score), dets[i].anchor_index))
    boxes = [dets[i] for i in order]
    n = len(boxes)
    merged = [False] * n
    keep: List[Detection] = []
    for i in range(n):
        if merged[i]:
            continue
        keep.append(boxes[i])
        merged[i] = True
        x1, y1, x2, y2 = boxes[i].rect
        area1 = f32(f32(f32(x2 - x1) + f32(1)) * f32(f32(y2 - y1) + f32(1)))
        for k in range(i + 1, n):
            if merged[k]:
                continue
            bx1, by1, bx2, by2 = boxes[k].rect
            x = max(x1, bx1)
            y = max(y1, by1)
            w = f32(f32(min(x2, bx2) - x) + f32(1))
            h = f32(f32(min(y2, by2) - y) + f32(1))
            if w <= 0 or h <= 0:
                continue
            area2 = f32(f32(f32(bx2 - bx1) + f32(1)) * f32(f32(by2 - by1) + f32(1)))
            inter = f32(w * h)
            if f32(inter / f32(f32(area1 + area2) - inter)) > thr:
                merged[k] = True
    return keep


def iou_plus1(a, b) -> float:
    """IoU with the reference's +1 pixel convention (RetinaFace.cpp:462-485), in float64."""
    ax1, ay1, ax2, ay2 = (float(v) for v in a)
    bx1, by1, bx2, by2 = (float(v) for v in b)
    w = min(ax2, bx2) - max(ax1, bx1) + 1
    h = min(ay2, by2) - max(ay1, by1) + 1
    if w <= 0 or h <= 0:
        return 0.0
    inter = w * h
    return inter / ((ax2 - ax1 + 1) * (ay2 - ay1 + 1) + (bx2 - bx1 + 1) * (by2 - by1 + 1) - inter)
