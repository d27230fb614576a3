"""Calibration-set readers for tools/calibrate_int8.py.

The reference's INT8-Calibration-Tool turns a directory of images into `.batch` files and hands those to TensorRT's entropy
calibrator (INT8-Calibration-Tool/calibrationtable.cpp:399-461): per file an `int[4]` header {N, C, H, W} followed by
N*C*H*W float32 values, planar (CHW) **RGB**, raw 0..255 -- its preprocess is cvtColor(BGR2RGB) + convertTo(CV_32FC3) and
nothing else (CalibrationTableImpl.cpp:28-34; prepareData :463-476 splits the channels).  This module reads and writes that
format and loads plain image directories, both as the BGR uint8 frames the engine takes.
"""
from __future__ import annotations

import os
import struct
from typing import Iterable, List

import numpy as np


def read_batch_file(path: str) -> List[np.ndarray]:
    """One reference `.batch` file -> N frames, H x W x 3 uint8 BGR."""
    with open(path, "rb") as f:
        head = f.read(16)
        if len(head) != 16:
            raise ValueError(f"{path}: shorter than the int[4] header")
        n, c, h, w = struct.unpack("<4i", head)
        if n <= 0 or c != 3 or h <= 0 or w <= 0 or n * c * h * w > (1 << 30):
            raise ValueError(f"{path}: bad header {(n, c, h, w)} (expected N x 3 x H x W)")
        data = np.fromfile(f, dtype="<f4", count=n * c * h * w)
    if data.size != n * c * h * w:
        raise ValueError(f"{path}: truncated ({data.size} of {n * c * h * w} floats)")
    chw = data.reshape(n, c, h, w)
    rgb = np.clip(np.rint(chw), 0, 255).astype(np.uint8).transpose(0, 2, 3, 1)
    return [np.ascontiguousarray(x[:, :, ::-1]) for x in rgb]


def write_batch_file(path: str, frames_bgr: Iterable[np.ndarray]) -> None:
    """Frames (same size, H x W x 3 uint8 BGR) -> one `.batch` file in the reference's layout."""
    frames = list(frames_bgr)
    h, w = frames[0].shape[:2]
    if any(f.shape != (h, w, 3) or f.dtype != np.uint8 for f in frames):
        raise ValueError("all frames of a batch must be uint8 H x W x 3 of one size")
    chw = np.stack([f[:, :, ::-1].transpose(2, 0, 1) for f in frames]).astype("<f4")
    with open(path, "wb") as f:
        f.write(struct.pack("<4i", len(frames), 3, h, w))
        chw.tofile(f)


def read_batch_dir(directory: str) -> List[np.ndarray]:
    """Every `*.batch` of a directory in name order (the tool writes batch_calibration<i>.batch)."""
    out: List[np.ndarray] = []
    for name in sorted(os.listdir(directory)):
        if name.endswith(".batch"):
            out.extend(read_batch_file(os.path.join(directory, name)))
    return out


def read_image_dir(directory: str) -> List[np.ndarray]:
    """Every image PIL can open in a directory (what the reference's getFileList + cv::imread would feed), as BGR uint8."""
    from PIL import Image
    out = []
    for name in sorted(os.listdir(directory)):
        p = os.path.join(directory, name)
        if not os.path.isfile(p):
            continue
        try:
            rgb = np.array(Image.open(p).convert("RGB"))
        except Exception:  # noqa: BLE001  (not an image)
            continue
        out.append(np.ascontiguousarray(rgb[:, :, ::-1]))
    return out
