"""Host numerics of the INT8 calibration tool (tools/calibrate_int8.py) that go beyond activation scales.

The reference's INT8-Calibration-Tool only feeds batches to TensorRT (INT8-Calibration-Tool/calibrationtable.cpp:399-583) and
TensorRT quantises the weights itself (closed source).  This engine quantises them per output channel on the grid
`row_scale[o] = max_k |w[o][k] * in_scale[k]| / 127` (retinaface_amd/csrc/weights.h put_gemm).  Round-to-nearest on that grid
turned out to be the LARGER half of the int8 engine's box-regression noise once activations are per channel (the per-input-channel
activation scale folds into the weights and spreads their dynamic range; tools/probes/int8_mix_sim.py: 13.8e-6 of 23.3e-6 box-delta
variance on mnet25).  The calibration run therefore also picks the rounding DIRECTION of every weight with the activations it has
anyway:

  gptq_round   error-compensated rounding (Frantar et al., "GPTQ", 2022 -- optimal-brain-quantisation column sweep): columns are
               rounded one at a time and the rounding error of each is pushed onto the columns not yet rounded, weighted by the
               inverse Gram matrix of the layer's calibration inputs, so that the layer's OUTPUT error (not the weight error) is
               minimised.  The grid (row scales) is untouched: the result is a plain int8 matrix.
  bias delta   minus the mean output error that is left, per output channel (Nagel et al., "Data-free quantization through weight
               equalization and bias correction", 2019).

Both are offline and cost nothing at run time; the integers travel next to the activation table (`<stem>.qweights.int8`, RFQ1, or
inside the .rfw) and the integer oracle (oracle/int8_forward.py) takes them from the same place, so the engine stays bit-exact
against it.
"""
from __future__ import annotations

import struct
from typing import Dict, Tuple

import numpy as np


def im2col_hwc(x: np.ndarray, k: int) -> np.ndarray:
    """x (H, W, C) -> (k*k*C, H*W) columns of the stride-1, pad k//2 convolution, rows ordered (ky, kx, c) = FoldedConv::w's K order."""
    h, w, c = x.shape
    if k == 1:
        return np.ascontiguousarray(x.reshape(h * w, c).T)
    p = k // 2
    xp = np.zeros((h + 2 * p, w + 2 * p, c), x.dtype)
    xp[p:p + h, p:p + w] = x
    rows = [xp[ky:ky + h, kx:kx + w].reshape(h * w, c).T for ky in range(k) for kx in range(k)]
    return np.ascontiguousarray(np.concatenate(rows, axis=0))


class Gram:
    """running sum of x x^T and of x over the calibration pixels of one fused conv's input (real units, float64)"""

    def __init__(self, ktot: int):
        self.g = np.zeros((ktot, ktot), np.float64)
        self.s = np.zeros(ktot, np.float64)
        self.n = 0

    def add(self, x_hwc: np.ndarray, k: int) -> None:
        cols = im2col_hwc(x_hwc.astype(np.float32), k).astype(np.float64)
        self.g += cols @ cols.T
        self.s += cols.sum(axis=1)
        self.n += cols.shape[1]


def gptq_round(quanta: np.ndarray, gram: np.ndarray, mean: np.ndarray, damp: float = 0.01) -> Tuple[np.ndarray, np.ndarray]:
    """quanta (cout, K): the unrounded weights in units of their row's grid step (w * in_scale / row_scale); gram (K, K) and mean (K):
    E[x x^T] and E[x] of the layer input IN INPUT QUANTA (x / in_scale).  Returns (q int8 (cout, K) in [-127, 127], the output-mean
    error (cout,) in row-grid units that the bias correction must remove: (q - quanta) @ mean)."""
    w = np.array(quanta, np.float64)
    w0 = w.copy()
    kk = w.shape[1]
    h = np.array(gram, np.float64)
    d = np.diag(h).copy()
    dead = d <= 0
    h[dead, :] = 0
    h[:, dead] = 0
    h[dead, dead] = 1.0                                   # an input that never fired: decoupled, plain rounding
    h += np.eye(kk) * (damp * float(np.diag(h).mean()))
    # upper Cholesky factor of the inverse: row i holds how column i's rounding error is spread over the columns after it
    hinv = np.linalg.inv(h)
    u = np.linalg.cholesky((hinv + hinv.T) / 2).T
    q = np.zeros_like(w)
    for i in range(kk):
        qi = np.clip(np.rint(w[:, i]), -127, 127)
        q[:, i] = qi
        err = (w[:, i] - qi) / u[i, i]
        if i + 1 < kk:
            w[:, i + 1:] -= err[:, None] * u[i, i + 1:][None, :]
    return q.astype(np.int8), (q - w0) @ np.asarray(mean, np.float64)


def output_error(quanta: np.ndarray, q: np.ndarray, gram: np.ndarray) -> np.ndarray:
    """per output channel: E[((q - quanta) . x)^2] on the calibration inputs (row-grid units squared)"""
    d = q.astype(np.float64) - quanta.astype(np.float64)
    return np.einsum("ok,kl,ol->o", d, gram, d)


def write_qweights(qw: Dict[str, Tuple[np.ndarray, np.ndarray]], path: str) -> None:
    """RFQ1 (retinaface_amd/csrc/model.h QWeights): "RFQ1", u32 n, per op: str name, u32 cout, u32 ktot, i8 q[cout*ktot], f32 bias_delta[cout]"""
    out = bytearray(b"RFQ1" + struct.pack("<I", len(qw)))
    for name, (q, db) in qw.items():
        q = np.ascontiguousarray(q, np.int8)
        db = np.ascontiguousarray(db, "<f4")
        if q.ndim != 2 or db.shape != (q.shape[0],) or q.min() < -127:
            raise ValueError(f"bad calibrated weights for '{name}'")
        b = name.encode("utf-8")
        out += struct.pack("<I", len(b)) + b + struct.pack("<II", q.shape[0], q.shape[1]) + q.tobytes() + db.tobytes()
    with open(path, "wb") as f:
        f.write(bytes(out))
