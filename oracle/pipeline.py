"""End-to-end CPU oracle of RetinaFace::detect() (test infrastructure, see oracle/__init__.py).

Caffe variant  = reference retinaface/RetinaFace.cpp:943-1075 (pad to x32, anchors per call,
                 clip to the padded size) -- the semantic oracle named by BASELINE.json.
TRT variant    = reference retinaface/RetinaFace.cpp:576-747 for frames that fit the net
                 (identity resize): frame copied top-left into a zeroed netH x netW canvas,
                 clip to the net size.  For net-sized frames both variants coincide.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np

from .caffe_forward import CaffeNet, HEAD_STRIDES, head_names
from .caffe_io import NetSpec
from .retinaface_post import (Detection, decode, nms, preprocess_caffe, preprocess_trt_cvresize, preprocess_trt_identity)


@dataclass
class OracleResult:
    net_h: int
    net_w: int
    heads: Dict[str, np.ndarray]          # the 9 output blobs, NCHW fp32 (batch 1)
    candidates: List[Detection]           # pre-NMS, reference visiting order
    detections: List[Detection]           # post-NMS, score-descending

    def rows(self) -> np.ndarray:
        return np.stack([d.as_row() for d in self.detections]) if self.detections else \
            np.zeros((0, 15), np.float32)

    def anchor_indices(self) -> np.ndarray:
        return np.array([d.anchor_index for d in self.detections], dtype=np.int32)


class OracleDetector:
    def __init__(self, net: NetSpec, backend: str = "torch"):
        self.net = net
        self.caffe = CaffeNet(net, backend=backend)

    def forward(self, chw: np.ndarray, keep_all: bool = False):
        return self.caffe.forward(chw, keep_all=keep_all)

    def detect(self, img_bgr: np.ndarray, threshold: float = 0.5, nms_threshold: float = 0.4,
               net_hw: Optional[tuple] = None) -> OracleResult:
        if net_hw is None:
            chw, hs, ws = preprocess_caffe(img_bgr)
        else:
            hs, ws = net_hw
            # frames that fit: 1:1 on a zero canvas (both TensorRT builds agree); larger frames: the cv::resize branch of the
            # build without NPP (the one oracle/_ref compiles) -- NPP's area filter is closed source
            fits = img_bgr.shape[0] <= hs and img_bgr.shape[1] <= ws
            chw = preprocess_trt_identity(img_bgr, hs, ws) if fits else preprocess_trt_cvresize(img_bgr, hs, ws)
        blobs = self.caffe.forward(chw)
        heads = {n: blobs[n] for s in HEAD_STRIDES for n in head_names(s)}
        cands = decode(heads, hs, ws, threshold)
        dets = nms(list(cands), nms_threshold)
        return OracleResult(hs, ws, heads, cands, dets)
