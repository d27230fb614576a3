"""Synthetic face-bearing frames for the benchmark and the parity tests.

Uniform-noise frames give zero detections (max foreground probability ~0.02), so throughput in faces/s needs
frames that contain faces.  Recipe: mid-grey canvas + N(0, 8) noise, then k in 1..6 face patches cut from the
reference's only image fixture (data/img.jpg, shipped losslessly as assets/faces_1280x886.png) around the 6 boxes
the fp32 oracle finds there, each expanded 1.5x, resampled by an integer factor (x0.5 area-average, x1, x2
nearest) and pasted at random non-overlapping positions.  Frames are generated AT net size, so preprocess is the
identity-resize path (what BASELINE.json's configs use).  Everything is seeded: frame i of config c uses
numpy.random.default_rng(1000*c + i).
"""
from __future__ import annotations

import os
from typing import List, Tuple

import numpy as np

_ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")

# boxes (x1, y1, x2, y2) of the 6 faces the fp32 oracle (0517 weights, thr 0.5, nms 0.4) finds in the fixture,
# rounded outwards; regenerate with tools/make_golden.py --print-boxes
FACE_BOXES = [(462, 268, 573, 417), (745, 342, 844, 478), (903, 57, 1011, 205),
              (1131, 276, 1227, 399), (60, 262, 165, 397), (273, 146, 371, 269)]


def load_base_frame() -> np.ndarray:
    """The fixture photo as a BGR uint8 array (886 x 1280 x 3), i.e. what cv::imread would hand the reference."""
    from PIL import Image
    rgb = np.array(Image.open(os.path.join(_ASSETS, "faces_1280x886.png")).convert("RGB"))
    return np.ascontiguousarray(rgb[:, :, ::-1])


def padded_base_frame() -> np.ndarray:
    """The fixture zero-padded to 896 x 1280 (the Caffe path's pad-to-32, RetinaFace.cpp:950-953)."""
    base = load_base_frame()
    out = np.zeros((896, 1280, 3), np.uint8)
    out[:base.shape[0], :base.shape[1]] = base
    return out


def _patches(base: np.ndarray) -> List[np.ndarray]:
    out = []
    H, W = base.shape[:2]
    for x1, y1, x2, y2 in FACE_BOXES:
        cx, cy, w, h = (x1 + x2) / 2, (y1 + y2) / 2, (x2 - x1) * 1.5, (y2 - y1) * 1.5
        xa, xb = max(0, int(cx - w / 2)), min(W, int(cx + w / 2))
        ya, yb = max(0, int(cy - h / 2)), min(H, int(cy + h / 2))
        out.append(base[ya:yb, xa:xb].copy())
    return out


def _rescale(p: np.ndarray, num: int, den: int) -> np.ndarray:
    if den == 2:   # x0.5, 2x2 area average
        h, w = p.shape[0] // 2 * 2, p.shape[1] // 2 * 2
        q = p[:h, :w].astype(np.uint16)
        return ((q[0::2, 0::2] + q[1::2, 0::2] + q[0::2, 1::2] + q[1::2, 1::2] + 2) // 4).astype(np.uint8)
    if num == 2:   # x2 nearest
        return np.repeat(np.repeat(p, 2, axis=0), 2, axis=1)
    return p


def synth_frames(h: int, w: int, n: int, config: int = 0, base: np.ndarray = None, faces=None) -> List[np.ndarray]:
    """n seeded BGR uint8 frames of h x w with 1..6 pasted faces each.  `faces` restricts the patches to a subset of the six
    fixture faces (indices into FACE_BOXES): the int8 calibration set and the held-out int8 parity frames use disjoint subsets."""
    if base is None:
        base = load_base_frame()
    patches = _patches(base)
    if faces is not None:
        patches = [patches[i] for i in faces]
    frames = []
    for i in range(n):
        rng = np.random.default_rng(1000 * config + i)
        img = np.clip(128.0 + rng.normal(0.0, 8.0, size=(h, w, 3)), 0, 255).astype(np.uint8)
        k = int(rng.integers(1, 7))
        placed: List[Tuple[int, int, int, int]] = []
        for _ in range(k):
            p = patches[int(rng.integers(0, len(patches)))]
            num, den = [(1, 2), (1, 1), (2, 1)][int(rng.integers(0, 3))]
            q = _rescale(p, num, den)
            if q.shape[0] >= h or q.shape[1] >= w:
                q = p if (p.shape[0] < h and p.shape[1] < w) else _rescale(p, 1, 2)
            ph, pw = q.shape[:2]
            for _try in range(20):
                y = int(rng.integers(0, h - ph + 1))
                x = int(rng.integers(0, w - pw + 1))
                if all(x + pw <= a or a + c <= x or y + ph <= b or b + d <= y for a, b, c, d in placed):
                    img[y:y + ph, x:x + pw] = q
                    placed.append((x, y, pw, ph))
                    break
        frames.append(img)
    return frames
