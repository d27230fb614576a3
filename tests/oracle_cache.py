"""Oracle results of the contract tests' LARGE frames, minted once (round 6; VERDICT r5 next #7: keep the driver's `-m gpu` run short).

The fp16 and int8 contracts (tests/test_gpu_parity.py) hold the engine against the fp32 oracle on 104 frames per model each; 40 of
them are 1280 x 896 and cost the CPU oracle ~0.9 s apiece -- 160 frames, 2.5 of the suite's 6 minutes, spent re-deriving numbers that
only change when the oracle or the frame generator changes.  tools/make_contract_golden.py runs oracle.pipeline.OracleDetector on
exactly those frames and stores, per frame, what the tests read of an OracleResult: the kept detections and the pre-NMS candidates (15
floats + global anchor index each), the number of anchors within the fp16 score noise of the threshold (the candidate-count band), and
the SHA-1 of the frame's pixels.  `detect()` here serves a frame from that file when its key AND its pixel hash match, and runs the live
oracle otherwise (every 448 x 448 frame still does).  tests/test_oracle.py::test_contract_golden_equals_a_live_oracle_run re-derives
entries live on the CPU, so the file cannot drift from the oracle unnoticed.
"""
import hashlib
import os
from typing import List

import numpy as np

from oracle.retinaface_post import Detection

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "contract_oracle_1280x896.npz")
SCORE_NOISE = 2e-3            # = tests/test_gpu_parity.py SCORE_NOISE (asserted there)
_file = None


def frame_key(stem, hw, cfg, faces, i) -> str:
    return f"{stem}/{hw[0]}x{hw[1]}/cfg{cfg}/{'all' if faces is None else ''.join(str(f) for f in faces)}/{i}"


def frame_hash(frame: np.ndarray) -> np.ndarray:
    return np.frombuffer(hashlib.sha1(np.ascontiguousarray(frame).tobytes()).digest(), np.uint8)


def band_of(heads, thr=0.5) -> int:
    """anchors whose oracle foreground probability lies within SCORE_NOISE of the threshold (ncand_band of the parity tests)"""
    from oracle.caffe_forward import HEAD_STRIDES, head_names
    n = 0
    for s in HEAD_STRIDES:
        p = heads[head_names(s)[0]]
        n += int((np.abs(p[:, p.shape[1] // 2:] - np.float32(thr)) <= SCORE_NOISE).sum())
    return n


def _dets(rows, idx) -> List[Detection]:
    return [Detection(np.float32(r[0]), tuple(np.float32(v) for v in r[1:5]), [np.float32(v) for v in r[5:10]], [np.float32(v) for v in r[10:15]], int(a))
            for r, a in zip(rows, idx)]


class CachedResult:
    """The part of oracle.pipeline.OracleResult the contract tests read."""
    heads = None

    def __init__(self, net_h, net_w, det_rows, det_idx, cand_rows, cand_idx, band):
        self.net_h, self.net_w = net_h, net_w
        self.detections, self.candidates = _dets(det_rows, det_idx), _dets(cand_rows, cand_idx)
        self.band = int(band)
        self._rows, self._idx = np.asarray(det_rows, np.float32).reshape(-1, 15), np.asarray(det_idx, np.int32)

    def rows(self):
        return self._rows

    def anchor_indices(self):
        return self._idx


def pack(ref, frame) -> dict:
    """OracleResult -> the arrays stored per frame"""
    crow = np.stack([d.as_row() for d in ref.candidates]) if ref.candidates else np.zeros((0, 15), np.float32)
    return {"det_rows": ref.rows(), "det_idx": ref.anchor_indices(), "cand_rows": crow,
            "cand_idx": np.array([d.anchor_index for d in ref.candidates], np.int32), "band": np.int32(band_of(ref.heads)), "sha1": frame_hash(frame)}


def lookup(key, frame, hw):
    global _file
    if _file is None:
        _file = np.load(GOLDEN) if os.path.exists(GOLDEN) else {}
    files = getattr(_file, "files", ())
    if key + "/sha1" not in files or not np.array_equal(_file[key + "/sha1"], frame_hash(frame)):
        return None
    g = lambda n: _file[f"{key}/{n}"]          # noqa: E731
    return CachedResult(hw[0], hw[1], g("det_rows"), g("det_idx"), g("cand_rows"), g("cand_idx"), g("band"))


def detect(oracle, stem, frame, hw, cfg, faces, i, thr=0.5, nms=0.4):
    """The oracle's result for frame i of synth_frames(hw, config=cfg, faces=faces): from the minted file when it holds this very frame
    (thr 0.5 / nms 0.4 only), live otherwise.  The live result gets the same `.band` attribute."""
    if thr == 0.5 and nms == 0.4:
        hit = lookup(frame_key(stem, hw, cfg, faces, i), frame, hw)
        if hit is not None:
            return hit
    ref = oracle.detect(frame, thr, nms, net_hw=hw)
    ref.band = band_of(ref.heads, thr)
    return ref
