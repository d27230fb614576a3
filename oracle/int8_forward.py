"""Integer-exact CPU oracle of the int8 engine (test infrastructure, see oracle/__init__.py).

What the reference has for this path: ``builder->setInt8Mode`` + the calibration cache
(retinaface/tensorrt/trtnetbase.cpp:295-311, cache reader :31-44, shipped table model/mnet-deconv-0517.table.int8,
format SURVEY.md App. B.7).  TensorRT 5.1's int8 kernels are closed source and absent, so their arithmetic cannot be
restated; the *scheme* they document can -- symmetric quantisation ``real ~= q * scale``, ``q in [-127, 127]``, activation
scales from the table, one weight scale per output channel -- and that is what the repo's int8 engine is defined by
(DESIGN.md section 5).  This module restates that definition from the Caffe model + table alone, independently of the HIP
code and of the C++ weight packer:

  fold     Convolution + BatchNorm(use_global_stats) + Scale -> (w, b), fp64 arithmetic, cast to fp32 once
           (the same fold tests/test_host.py pins against the graph compiler, bit for bit)
  GEMM     ws_in[o][k] = w[o][k] * s_in[k % cin]            (fp32)      per-input-channel activation scale folded into the weights
           s_w[o]  = max_k |ws_in[o][k]| / 127               (fp32)      per-output-channel weight scale (1 if the row is zero)
           w_q     = clamp(rne(ws_in / s_w), -127, 127)
           mult[o] = s_w[o] / s_out[o],  bias[o] = b[o] / s_out[o]       (s_out = 1 for the heads: real outputs)
  depthwise  wf = w * s_in[c] / s_mid[c] (fp32), s_t[c] = max_t |wf| / 16256, taps = clamp(rne(wf / s_t), -16256, 16256)
           (15-bit integer taps), mult = s_t, bias = b / s_mid
  epilogue q = rne(clamp(fmaf(acc, mult, bias), 0, 127))     (oracle/csrc/rf_int8_ref.c)
  mids     (round 6) a depthwise output is read by a 1x1 conv only (no zero padding), so it uses all 8 bits: s_mid = table scale *
           fp32(127/255), q = rne(clamp(y, 0, 255)) - 128; the pointwise conv sums w_q * (q - 128) and its bias becomes
           fmaf(mult, 128 * sum_k w_q, bias) -- exact integer algebra, one more bit on the tensors that carried most of the
           activation noise of the box deltas
  calibrated weights  (round 6) when the model carries `int8_qweights` for a fused op, w_q is taken from there (error-compensated
           rounding chosen by tools/calibrate_int8.py --gptq on the SAME grid s_w) and bias = (b + bias_delta) / s_out
  add      lateral + bilinear x2 upsample of the coarser level, requantised to the `_plus` scale (rfi8_upadd)

Every accumulation is an exact integer, every float step is a single IEEE operation in a fixed order, so the HIP engine must
reproduce each int8 activation BIT FOR BIT (tests/test_gpu_parity.py::test_int8_engine_is_bit_exact_against_the_integer_oracle).
The first layers (preprocess + conv0 [+ the first blocks]) are computed by the engine in fp16/fp32-grade arithmetic on the raw
0..255 frame and only their OUTPUT is int8: `forward_from` therefore starts at that blob (taken from the device as the pinned
input), and `quantise_blob` gives the fp32 oracle's version of it for the <= 1 LSB check of the float front end.

Two conv back-ends: "blas" = float64 torch conv (every partial sum is an integer below 2**53: exact in any order), "c" = the
plain loops of rfi8_conv.  tests/test_oracle.py requires them to agree exactly.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, List, Optional, Tuple

import numpy as np

from .caffe_io import NetSpec

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "csrc", "rf_int8_ref.c")
_OUT = os.path.join(_HERE, "_build", "librf_int8_ref.so")
_lib = None

F32 = np.float32
DW_RANGE = 127 * 128          # 15-bit depthwise taps
MID_U8 = np.float32(127.0 / 255.0)   # a depthwise output's quantum relative to its table scale (0..255 quanta, see the header)


def build(force: bool = False) -> str:
    if force or not os.path.exists(_OUT) or os.path.getmtime(_OUT) < os.path.getmtime(_SRC):
        os.makedirs(os.path.dirname(_OUT), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-ffp-contract=off", "-o", _OUT, _SRC, "-lm"])
    return _OUT


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        vp, f, i, l = C.c_void_p, C.c_float, C.c_int, C.c_long
        _lib.rfi8_requant.argtypes = [vp, vp, vp, l, i, i, vp]
        _lib.rfi8_dequant.argtypes = [vp, vp, vp, l, i, vp]
        _lib.rfi8_conv.argtypes = [vp, i, i, i, vp, i, i, i, i, i, vp]
        _lib.rfi8_upadd.argtypes = [vp, vp, i, i, i, f, f, i, vp]
        _lib.rfi8_bias_u8.argtypes = [vp, vp, i, vp]
        for fn in (_lib.rfi8_requant, _lib.rfi8_dequant, _lib.rfi8_conv, _lib.rfi8_upadd, _lib.rfi8_bias_u8):
            fn.restype = None
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data


# ----------------------------------------------------------------------------------------------- model -> quantised ops


def fold(net: NetSpec, conv: str, bn: Optional[str] = None) -> Tuple[np.ndarray, np.ndarray]:
    """(w [cout][k][k][cin/g] fp32, b [cout] fp32): BN (+ Scale) folded in fp64, as Caffe's inference arithmetic composes them
    (BatchNorm: (x - mean/sf) / sqrt(var/sf + eps); Scale: gamma x + beta)."""
    l = net.layer(conv)
    w = l.blobs[0].astype(np.float64)
    b = l.blobs[1].reshape(-1).astype(np.float64) if l.bias_term else np.zeros(w.shape[0])
    if bn is not None:
        bl, sl = net.layer(bn), net.layer(bn + "_scale")
        sf = bl.blobs[2].reshape(-1)[0]
        inv = F32(0) if sf == 0 else F32(1) / F32(sf)
        mean = (bl.blobs[0].reshape(-1) * inv).astype(np.float64)
        var = (bl.blobs[1].reshape(-1) * inv).astype(np.float64)
        k = sl.blobs[0].reshape(-1).astype(np.float64) / np.sqrt(var + np.float64(F32(bl.eps)))
        w = w * k[:, None, None, None]
        b = (b - mean) * k + (sl.blobs[1].reshape(-1).astype(np.float64) if sl.scale_bias else 0.0)
    return np.ascontiguousarray(w.transpose(0, 2, 3, 1)).astype(F32), b.astype(F32)


def _bn_of(net: NetSpec, conv: str) -> Optional[str]:
    """name of the BatchNorm layer that consumes `conv`'s top (None for the heads)"""
    top = net.layer(conv).tops[0]
    seen = False
    for l in net.layers:
        if l.name == conv:
            seen = True
            continue
        if seen and l.type == "BatchNorm" and l.bottoms and l.bottoms[0] == top:
            return l.name
    return None


class QGemm:
    """one quantised dense conv (1x1 or 3x3): int weights [cout][k][k][cin] + fp32 epilogue constants"""

    def __init__(self, w: np.ndarray, b: np.ndarray, s_in: np.ndarray, s_out: Optional[np.ndarray], relu: bool = True,
                 in_u8: bool = False, calibrated: Optional[Tuple[np.ndarray, np.ndarray]] = None):
        cout, k, _, cin = w.shape
        s_in = (s_in.astype(F32) * MID_U8).astype(F32) if in_u8 else s_in.astype(F32)
        ws_in = (w * s_in.reshape(1, 1, 1, cin)).astype(F32)
        amax = np.abs(ws_in).reshape(cout, -1).max(axis=1).astype(F32)
        s_w = np.where(amax > 0, amax / F32(127), F32(1)).astype(F32)
        q = np.rint((ws_in / s_w.reshape(cout, 1, 1, 1)).astype(F32))          # np.rint = round half to even = nearbyintf
        self.wq = np.clip(q, -127, 127).astype(np.int32)
        os_ = np.ones(cout, F32) if s_out is None else s_out.astype(F32)
        self.mult = (s_w / os_).astype(F32)
        b = b.astype(F32)
        if calibrated is not None:                       # integers + bias correction chosen by the calibration run, same grid
            cq, db = calibrated
            assert cq.shape == (cout, k * k * cin) and db.shape == (cout,)
            self.wq = cq.astype(np.int32).reshape(cout, k, k, cin)
            b = (b + db.astype(F32)).astype(F32)
        self.bias = (b / os_).astype(F32)
        if in_u8:
            qsum = np.ascontiguousarray(self.wq.reshape(cout, -1).sum(axis=1).astype(np.int64))
            bias = np.ascontiguousarray(self.bias)
            lib().rfi8_bias_u8(_p(np.ascontiguousarray(self.mult)), _p(qsum), cout, _p(bias))
            self.bias = bias
        self.k, self.relu, self.cout, self.cin = k, relu, cout, cin


class QDw:
    """one quantised depthwise 3x3: 15-bit integer taps [c][3][3]"""

    def __init__(self, w: np.ndarray, b: np.ndarray, s_in: np.ndarray, s_mid: np.ndarray, stride: int):
        c = w.shape[0]
        w9 = w.reshape(c, 9).astype(F32)
        s_mid = (s_mid.astype(F32) * MID_U8).astype(F32)                          # 0..255 quanta of amax / 255
        wf = ((w9 * s_in.astype(F32).reshape(c, 1)).astype(F32) / s_mid.reshape(c, 1)).astype(F32)
        amax = np.abs(wf).max(axis=1).astype(F32)
        s_t = np.where(amax > 0, amax / F32(DW_RANGE), F32(1)).astype(F32)
        wi = np.rint((wf / s_t.reshape(c, 1)).astype(F32))
        self.wq = np.clip(wi, -DW_RANGE, DW_RANGE).astype(np.int32).reshape(c, 3, 3, 1)
        self.mult = s_t
        self.bias = (b.astype(F32) / s_mid.astype(F32)).astype(F32)
        self.stride, self.c = stride, c


class Int8Net:
    """The MobileNet-0.25 + FPN + SSH graph of model/mnet-deconv-0517.prototxt / mnet25.prototxt in int8."""

    BLOCK_COUT = (16, 32, 32, 64, 64, 128, 128, 128, 128, 128, 128, 256, 256)
    BLOCK_STRIDE = (1, 2, 1, 2, 1, 2, 1, 1, 1, 1, 1, 2, 1)

    def __init__(self, net: NetSpec, backend: str = "blas"):
        if not net.int8_scales:
            raise ValueError("the model carries no calibration table")
        self.net, self.backend = net, backend
        self.table = net.int8_scales
        self.per_channel = "_plus0#0" in self.table
        S = self.scales
        cal = lambda *names: net.int8_qweights.get("+".join(names))               # noqa: E731  (fused-op name as in plan.h)
        # ---- backbone
        self.dw: List[QDw] = []
        self.pw: List[QGemm] = []
        self.block_blobs: List[Tuple[str, str]] = []
        s_prev = S("mobilenet0_relu2_fwd", 16)
        c = 8
        self.s_block: Dict[int, np.ndarray] = {0: s_prev}
        for i in range(13):
            dn, pn = f"mobilenet0_conv{2 * i + 1}_fwd", f"mobilenet0_conv{2 * i + 2}_fwd"
            mid_blob, out_blob = f"mobilenet0_relu{2 * i + 1}_fwd", f"mobilenet0_relu{2 * i + 2}_fwd"
            self.block_blobs.append((mid_blob, out_blob))
            cout = self.BLOCK_COUT[i]
            if i == 0:                       # block 0 lives in the engine's float front end; only its output scale matters here
                self.dw.append(None)
                self.pw.append(None)
                c = cout
                continue
            s_mid, s_out = S(mid_blob, c), S(out_blob, cout)
            wd, bd = fold(net, dn, _bn_of(net, dn))
            wp, bp = fold(net, pn, _bn_of(net, pn))
            self.dw.append(QDw(wd, bd, s_prev, s_mid, self.BLOCK_STRIDE[i]))
            self.pw.append(QGemm(wp, bp, s_mid, s_out, in_u8=True, calibrated=cal(pn)))
            s_prev, c = s_out, cout
            self.s_block[i] = s_out
        # ---- FPN
        lat_names = ("rf_c3_lateral", "rf_c2_lateral", "rf_c1_red_conv")
        s_tap = (self.s_block[12], self.s_block[10], self.s_block[4])
        s_lat = [S(n + "_relu", 64) for n in lat_names]
        s_plus = [S("_plus0", 64), S("_plus1", 64)]
        s_aggr = [S("rf_c2_aggr_relu", 64), S("rf_c1_aggr_relu", 64)]
        if self.per_channel:
            # the three tensors of each add share one scale per channel (the largest of their calibrated ones)
            m0 = np.maximum(s_lat[0], np.maximum(s_lat[1], s_plus[0]))
            s_lat[0] = s_lat[1] = s_plus[0] = m0
            m1 = np.maximum(s_aggr[0], np.maximum(s_lat[2], s_plus[1]))
            s_aggr[0] = s_lat[2] = s_plus[1] = m1
        self.lat = [QGemm(*fold(net, n, _bn_of(net, n)), s_tap[i], s_lat[i], calibrated=cal(n)) for i, n in enumerate(lat_names)]
        self.lat_blobs = [n + "_relu" for n in lat_names]
        s_feat = [s_lat[0], s_aggr[0], s_aggr[1]]
        self.aggr, self.a_lat, self.a_up = [], [], []
        for i, n in enumerate(("rf_c2_aggr", "rf_c1_aggr")):
            self.a_lat.append(F32(1) if self.per_channel else F32(s_lat[i + 1][0]) / F32(s_plus[i][0]))
            self.a_up.append(F32(1) if self.per_channel else F32(s_feat[i][0]) / F32(s_plus[i][0]))
            self.aggr.append(QGemm(*fold(net, n, _bn_of(net, n)), s_plus[i], s_aggr[i], calibrated=cal(n)))
        self.aggr_blobs = ["rf_c2_aggr_relu", "rf_c1_aggr_relu"]
        self.scale_of_blob: Dict[str, np.ndarray] = {b: s for b, s in zip(self.lat_blobs, s_lat)}
        self.scale_of_blob.update({b: s for b, s in zip(self.aggr_blobs, s_aggr)})
        for i in range(13):
            if i in self.s_block:
                self.scale_of_blob[self.block_blobs[i][1]] = self.s_block[i]
        # ---- SSH context modules + heads (strides 32, 16, 8)
        self.ssh = []
        for i in range(3):
            pre = f"rf_c{3 - i}_det_"
            st = f"stride{(32, 16, 8)[i]}"
            s_cat, s_c1, s_c31 = S(pre + "concat_relu", 64), S(pre + "context_conv1_relu", 16), S(pre + "context_conv3_1_relu", 16)

            def merged(names):
                parts = [fold(net, n, _bn_of(net, n)) for n in names]
                return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])

            wa, ba = merged([pre + "conv1", pre + "context_conv1"])
            wb, bb = merged([pre + "context_conv2", pre + "context_conv3_1"])
            wc, bc = fold(net, pre + "context_conv3_2", _bn_of(net, pre + "context_conv3_2"))
            wh, bh = merged([f"face_rpn_cls_score_{st}", f"face_rpn_bbox_pred_{st}", f"face_rpn_landmark_pred_{st}"])
            self.ssh.append(dict(
                a=QGemm(wa, ba, s_feat[i], np.concatenate([s_cat[:32], s_c1]), calibrated=cal(pre + "conv1", pre + "context_conv1")),
                b=QGemm(wb, bb, s_c1, np.concatenate([s_cat[32:48], s_c31]), calibrated=cal(pre + "context_conv2", pre + "context_conv3_1")),
                c=QGemm(wc, bc, s_c31, s_cat[48:64], calibrated=cal(pre + "context_conv3_2")),
                head=QGemm(wh, bh, s_cat, None, relu=False,
                           calibrated=cal(f"face_rpn_cls_score_{st}", f"face_rpn_bbox_pred_{st}", f"face_rpn_landmark_pred_{st}")),
                pre=pre, stride=(32, 16, 8)[i]))
            self.scale_of_blob[pre + "concat_relu"] = s_cat
            self.scale_of_blob[pre + "context_conv1_relu"] = s_c1
            self.scale_of_blob[pre + "context_conv3_1_relu"] = s_c31

    # ------------------------------------------------------------------------------------------- helpers
    def scales(self, blob: str, channels: int) -> np.ndarray:
        """per-channel scales of a blob: `blob#c` lines when the table has them (tools/calibrate_int8.py --per-channel),
        otherwise the per-tensor TensorRT value broadcast"""
        if blob + "#0" in self.table:
            return np.array([self.table[f"{blob}#{c}"] for c in range(channels)], F32)
        if blob not in self.table:
            raise KeyError(f"the calibration table has no scale for '{blob}'")
        return np.full(channels, self.table[blob], F32)

    def quantise_blob(self, blob: str, real_hwc: np.ndarray) -> np.ndarray:
        """fp32 activation (H, W, C) -> the int8 quanta a ReLU'd tensor of that scale has: rne(clamp(x / s, 0, 127))"""
        s = self.scale_of_blob[blob]
        y = (real_hwc.astype(F32) * (F32(1) / s).reshape(1, 1, -1)).astype(F32)
        return np.rint(np.clip(y, 0, 127)).astype(np.int8)

    def _conv(self, x: np.ndarray, wq: np.ndarray, stride: int, pad: int, group: int) -> np.ndarray:
        """x (H, W, Cin) int8, wq [cout][k][k][cin/g] int32 -> (Ho, Wo, Cout) int32, exact"""
        h, w, cin = x.shape
        cout, k = wq.shape[0], wq.shape[1]
        if self.backend == "c":
            ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
            out = np.empty((ho, wo, cout), np.int32)
            xc, wc = np.ascontiguousarray(x, np.int8), np.ascontiguousarray(wq, np.int32)
            lib().rfi8_conv(_p(xc), h, w, cin, _p(wc), cout, k, stride, pad, group, _p(out))
            return out
        import torch
        import torch.nn.functional as Fn
        xt = torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1))[None].astype(np.float64))
        wt = torch.from_numpy(np.ascontiguousarray(wq.transpose(0, 3, 1, 2)).astype(np.float64))
        y = Fn.conv2d(xt, wt, None, stride=stride, padding=pad, groups=group)[0].numpy()
        yi = np.rint(y).astype(np.int64)
        assert np.array_equal(yi, y) and np.abs(yi).max(initial=0) < 2 ** 31
        return np.ascontiguousarray(yi.transpose(1, 2, 0)).astype(np.int32)

    @staticmethod
    def _requant(acc: np.ndarray, mult: np.ndarray, bias: np.ndarray, relu: bool) -> np.ndarray:
        h, w, c = acc.shape
        acc = np.ascontiguousarray(acc, np.int32)
        out = np.empty((h, w, c), np.int8)
        m, b = np.ascontiguousarray(mult, F32), np.ascontiguousarray(bias, F32)
        lib().rfi8_requant(_p(acc), _p(m), _p(b), h * w, c, int(relu), _p(out))
        return out

    def gemm(self, x: np.ndarray, g: QGemm) -> np.ndarray:
        return self._requant(self._conv(x, g.wq, 1, g.k // 2, 1), g.mult, g.bias, g.relu)

    def depthwise(self, x: np.ndarray, d: QDw) -> np.ndarray:
        return self._requant(self._conv(x, d.wq, d.stride, 1, d.c), d.mult, d.bias, 2)         # 2: 0..255 quanta stored - 128

    def upadd(self, lat: np.ndarray, up: np.ndarray, i: int) -> np.ndarray:
        h, w, c = lat.shape
        assert up.shape == (h // 2, w // 2, c)
        out = np.empty_like(lat)
        lat, up = np.ascontiguousarray(lat, np.int8), np.ascontiguousarray(up, np.int8)
        lib().rfi8_upadd(_p(lat), _p(up), h, w, c, float(self.a_lat[i]), float(self.a_up[i]), 1 if self.per_channel else 0, _p(out))
        return out

    # ------------------------------------------------------------------------------------------- forward
    def forward_from(self, blob: str, q: np.ndarray) -> Dict[str, np.ndarray]:
        """Continue the network from int8 activation `blob` (one image, (H, W, C) int8 quanta; a backbone block output
        `mobilenet0_relu{2,4,..}_fwd`).  Returns every later int8 activation by reference blob name (H, W, C) int8 -- the
        depthwise intermediates (`relu{odd}`) included -- plus the 9 head blobs as fp32 (C, H, W): raw bbox / landmark outputs and
        the 2-class softmax of the scores (Caffe Softmax: subtract the max, exponentiate, normalise)."""
        names = [b[1] for b in self.block_blobs]
        if blob not in names:
            raise ValueError(f"forward_from: '{blob}' is not a backbone block output")
        start = names.index(blob)
        assert q.dtype == np.int8 and q.ndim == 3 and q.shape[2] == self.BLOCK_COUT[start]
        acts: Dict[str, np.ndarray] = {blob: q}
        x = q
        tap: Dict[int, np.ndarray] = {}
        if start in (4, 10, 12):
            tap[start] = x
        for i in range(start + 1, 13):
            mid = self.depthwise(x, self.dw[i])
            x = self.gemm(mid, self.pw[i])
            acts[self.block_blobs[i][0]], acts[self.block_blobs[i][1]] = mid, x
            if i in (4, 10, 12):
                tap[i] = x
        if set(tap) != {4, 10, 12}:
            raise ValueError("forward_from must start at or before block 4 (the stride-8 FPN tap)")
        lat = [self.gemm(tap[t], self.lat[i]) for i, t in enumerate((12, 10, 4))]
        for b, v in zip(self.lat_blobs, lat):
            acts[b] = v
        feat = [lat[0]]
        for i in range(2):
            plus = self.upadd(lat[i + 1], feat[i], i)
            acts[f"_plus{i}"] = plus
            feat.append(self.gemm(plus, self.aggr[i]))
            acts[self.aggr_blobs[i]] = feat[-1]
        heads: Dict[str, np.ndarray] = {}
        for i, m in enumerate(self.ssh):
            ya = self.gemm(feat[i], m["a"])
            ctx1 = np.ascontiguousarray(ya[:, :, 32:48])
            yb = self.gemm(ctx1, m["b"])
            ctx31 = np.ascontiguousarray(yb[:, :, 16:32])
            yc = self.gemm(ctx31, m["c"])
            cat = np.concatenate([ya[:, :, :32], yb[:, :, :16], yc], axis=2)
            acts[m["pre"] + "context_conv1_relu"], acts[m["pre"] + "context_conv3_1_relu"] = ctx1, ctx31
            acts[m["pre"] + "concat_relu"] = cat
            hq = m["head"]
            acc = np.ascontiguousarray(self._conv(cat, hq.wq, 1, 0, 1), np.int32)
            h, w, c = acc.shape
            y = np.empty((h, w, c), F32)
            lib().rfi8_dequant(_p(acc), _p(np.ascontiguousarray(hq.mult)), _p(np.ascontiguousarray(hq.bias)), h * w, c, _p(y))
            a = c // 16                                    # anchors per cell
            y = y.transpose(2, 0, 1)                       # (16A, H, W): scores 2A (background A | foreground A), bbox 4A, landmarks 10A
            sc = y[:2 * a].reshape(2, a, h, w)
            mx = sc.max(axis=0, keepdims=True)
            e = np.exp(sc - mx).astype(F32)
            prob = (e / e.sum(axis=0, keepdims=True)).astype(F32).reshape(2 * a, h, w)
            s = m["stride"]
            heads[f"face_rpn_cls_prob_reshape_stride{s}"] = prob
            heads[f"face_rpn_bbox_pred_stride{s}"] = np.ascontiguousarray(y[2 * a:6 * a])
            heads[f"face_rpn_landmark_pred_stride{s}"] = np.ascontiguousarray(y[6 * a:])
        acts["__heads__"] = heads          # type: ignore[assignment]
        return acts
