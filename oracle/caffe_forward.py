"""Layer-by-layer, *unfused* fp32 restatement of BVLC Caffe's ``Net::Forward()`` for the layer
types the RetinaFace prototxts use (test infrastructure, see oracle/__init__.py).

Reference call site: ``Net_->Forward()`` at retinaface/RetinaFace.cpp:988; the graph is
model/mnet-deconv-0517.prototxt / model/mnet25.prototxt.  Caffe itself is not vendored in
the reference (include path /home/ubuntu/caffe-office/caffe, CMakeLists.txt:91), so the layer
semantics below restate BVLC Caffe's published definitions:

  Convolution    y = conv(x, W; stride, pad, group) (+ b)      out = floor((H + 2p - k)/s) + 1
  Deconvolution  y = conv_transpose(x, W[Cin, Cout/g, k, k])   out = s(H-1) + k - 2p
  BatchNorm      use_global_stats: y = (x - mean/sf) / sqrt(var/sf + eps), sf = blobs[2] (0 -> 0)
  Scale          y = gamma * x (+ beta), broadcast over axis 1
  ReLU, Eltwise(SUM), Concat(axis), Crop(axis, offsets), Reshape(0 = copy, -1 = infer), Softmax(axis)

Two independent back-ends evaluate the conv arithmetic:
  ``backend="torch"``  torch.nn.functional.conv2d / conv_transpose2d on CPU (oneDNN/MKL)
  ``backend="numpy"``  explicit im2col + matmul, and a scatter-add transposed conv
tests/test_oracle.py requires them to agree to fp32 round-off.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional

import numpy as np

from .caffe_io import LayerSpec, NetSpec


# ----------------------------------------------------------------------------- numpy back-end


def _conv2d_numpy(x: np.ndarray, w: np.ndarray, b: Optional[np.ndarray], stride: int, pad: int,
                  group: int) -> np.ndarray:
    n, c, h, wd = x.shape
    o, cg, kh, kw = w.shape
    assert c == cg * group and o % group == 0
    ho = (h + 2 * pad - kh) // stride + 1
    wo = (wd + 2 * pad - kw) // stride + 1
    xp = np.zeros((n, c, h + 2 * pad, wd + 2 * pad), dtype=np.float32)
    xp[:, :, pad:pad + h, pad:pad + wd] = x
    # cols[n, c, ky, kx, ho, wo]
    cols = np.empty((n, c, kh, kw, ho, wo), dtype=np.float32)
    for ky in range(kh):
        for kx in range(kw):
            cols[:, :, ky, kx] = xp[:, :, ky:ky + stride * (ho - 1) + 1:stride,
                                    kx:kx + stride * (wo - 1) + 1:stride]
    og = o // group
    y = np.empty((n, o, ho, wo), dtype=np.float32)
    for g in range(group):
        a = w[g * og:(g + 1) * og].reshape(og, cg * kh * kw)
        for i in range(n):
            bmat = cols[i, g * cg:(g + 1) * cg].reshape(cg * kh * kw, ho * wo)
            y[i, g * og:(g + 1) * og] = (a @ bmat).reshape(og, ho, wo)
    if b is not None:
        y += b.reshape(1, o, 1, 1)
    return y


def _deconv2d_numpy(x: np.ndarray, w: np.ndarray, b: Optional[np.ndarray], stride: int, pad: int,
                    group: int) -> np.ndarray:
    n, c, h, wd = x.shape
    cin, og, kh, kw = w.shape
    assert cin == c
    o = og * group
    cg = c // group
    ho = stride * (h - 1) + kh - 2 * pad
    wo = stride * (wd - 1) + kw - 2 * pad
    full = np.zeros((n, o, stride * (h - 1) + kh, stride * (wd - 1) + kw), dtype=np.float32)
    for g in range(group):
        for ci in range(cg):
            cc = g * cg + ci
            for oo in range(og):
                for ky in range(kh):
                    for kx in range(kw):
                        full[:, g * og + oo, ky:ky + stride * (h - 1) + 1:stride,
                             kx:kx + stride * (wd - 1) + 1:stride] += x[:, cc] * w[cc, oo, ky, kx]
    y = full[:, :, pad:pad + ho, pad:pad + wo].copy()
    if b is not None:
        y += b.reshape(1, o, 1, 1)
    return y


# ----------------------------------------------------------------------------- torch back-end


def _conv2d_torch(x, w, b, stride, pad, group):
    import torch
    import torch.nn.functional as F
    y = F.conv2d(torch.from_numpy(x), torch.from_numpy(w),
                 None if b is None else torch.from_numpy(b), stride=stride, padding=pad,
                 groups=group)
    return y.numpy()


def _deconv2d_torch(x, w, b, stride, pad, group):
    import torch
    import torch.nn.functional as F
    y = F.conv_transpose2d(torch.from_numpy(x), torch.from_numpy(w),
                           None if b is None else torch.from_numpy(b), stride=stride,
                           padding=pad, groups=group)
    return y.numpy()


# ----------------------------------------------------------------------------- interpreter


class CaffeNet:
    """Evaluate a NetSpec exactly as Caffe would: one layer at a time, fp32, nothing folded."""

    def __init__(self, net: NetSpec, backend: str = "torch"):
        self.net = net
        if backend == "torch":
            self._conv, self._deconv = _conv2d_torch, _deconv2d_torch
        elif backend == "numpy":
            self._conv, self._deconv = _conv2d_numpy, _deconv2d_numpy
        else:
            raise ValueError(backend)

    def forward(self, data: np.ndarray, keep: Optional[Iterable[str]] = None,
                keep_all: bool = False) -> Dict[str, np.ndarray]:
        """data: float32 NCHW.  Returns blobs by top name (all of them if keep_all, otherwise
        the net outputs -- tops never consumed -- plus the names in `keep`)."""
        assert data.dtype == np.float32 and data.ndim == 4
        blobs: Dict[str, np.ndarray] = {self.net.input_name: data}
        # remember values of in-place-overwritten tops under "<layer name>" too when keep_all
        named: Dict[str, np.ndarray] = {}
        for l in self.net.layers:
            ins = [blobs[b] for b in l.bottoms]
            out = self._layer(l, ins)
            blobs[l.tops[0]] = out
            if keep_all:
                named[l.name] = out
        if keep_all:
            res = dict(blobs)
            for k, v in named.items():
                res.setdefault(k, v)
            res["__by_layer__"] = named  # type: ignore[assignment]
            return res
        consumed = {b for l in self.net.layers for b in l.bottoms}
        outs = {t: blobs[t] for l in self.net.layers for t in l.tops if t not in consumed}
        for k in keep or ():
            outs[k] = blobs[k]
        return outs

    def _layer(self, l: LayerSpec, ins: List[np.ndarray]) -> np.ndarray:
        t = l.type
        if t == "Convolution":
            w = l.blobs[0]
            b = l.blobs[1].reshape(-1) if l.bias_term else None
            assert w.shape[0] == l.num_output
            return self._conv(ins[0], w, b, l.stride, l.pad, l.group)
        if t == "Deconvolution":
            w = l.blobs[0]
            b = l.blobs[1].reshape(-1) if l.bias_term else None
            return self._deconv(ins[0], w, b, l.stride, l.pad, l.group)
        if t == "BatchNorm":
            mean, var, sf = (bb.reshape(-1) for bb in l.blobs[:3])
            s = np.float32(0.0) if sf[0] == 0 else np.float32(1.0) / sf[0]
            mean = (mean * s).astype(np.float32)
            var = (var * s).astype(np.float32)
            denom = np.sqrt(var + np.float32(l.eps)).astype(np.float32)
            x = ins[0]
            return ((x - mean.reshape(1, -1, 1, 1)) / denom.reshape(1, -1, 1, 1)).astype(np.float32)
        if t == "Scale":
            y = ins[0] * l.blobs[0].reshape(1, -1, 1, 1)
            if l.scale_bias:
                y = y + l.blobs[1].reshape(1, -1, 1, 1)
            return y.astype(np.float32)
        if t == "ReLU":
            return np.maximum(ins[0], np.float32(0))
        if t == "Eltwise":
            assert l.eltwise_op == "SUM"
            y = ins[0]
            for o in ins[1:]:
                y = y + o
            return y
        if t == "Concat":
            return np.concatenate(ins, axis=l.axis)
        if t == "Crop":
            x, ref = ins
            axis = l.axis
            sl = [slice(None)] * x.ndim
            offs = l.crop_offsets
            for i in range(axis, x.ndim):
                if len(offs) == 0:
                    o = 0
                elif len(offs) == 1:
                    o = offs[0]
                else:
                    o = offs[i - axis]
                sl[i] = slice(o, o + ref.shape[i])
            return np.ascontiguousarray(x[tuple(sl)])
        if t == "Reshape":
            x = ins[0]
            start = l.reshape_axis if l.reshape_axis >= 0 else x.ndim + l.reshape_axis + 1
            end = x.ndim if l.reshape_num_axes == -1 else start + l.reshape_num_axes
            new_mid: List[int] = []
            for i, d in enumerate(l.reshape_dims):
                new_mid.append(x.shape[start + i] if d == 0 else d)
            shape = list(x.shape[:start]) + new_mid + list(x.shape[end:])
            return x.reshape(shape)
        if t == "Softmax":
            x = ins[0]
            m = x.max(axis=l.axis, keepdims=True)
            e = np.exp(x - m).astype(np.float32)
            return (e / e.sum(axis=l.axis, keepdims=True)).astype(np.float32)
        raise ValueError(f"unsupported layer type {t}")


HEAD_STRIDES = (32, 16, 8)


def head_names(stride: int):
    return (f"face_rpn_cls_prob_reshape_stride{stride}", f"face_rpn_bbox_pred_stride{stride}",
            f"face_rpn_landmark_pred_stride{stride}")
