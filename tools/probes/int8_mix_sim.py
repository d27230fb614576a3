#!/usr/bin/env python3
"""CPU design probe for the int8 engine's distance to the fp32 oracle (round 6): a fake-quantisation replay of
the fused-op sequence in which every STORED tensor is either int8 (table scale, per tensor or per channel), fp16 or fp32, and
every contraction takes the arithmetic of its input tensor (int8 input -> int8 weights with one scale per output channel;
fp16 input -> fp16 weights).  It answers, without a GPU, which tensors have to leave int8 for the engine to meet
`worst per-face IoU >= 0.97, anchor agreement >= 0.95` and what a recalibrated table buys.  Not bit-exact to the engine (the
integer oracle oracle/int8_forward.py is); it agrees with it to the last quantum on all but the rounding ties.

usage: python tools/probes/int8_mix_sim.py --model mnet25 --frames 64 --variants all_i8,heads_f16,...
"""
import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as Fn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import build as obuild                                     # noqa: E402
from oracle.caffe_forward import HEAD_STRIDES, head_names              # noqa: E402
from oracle.caffe_io import read_int8_table, read_rfw                   # noqa: E402
from oracle.int8_forward import _bn_of, fold                            # noqa: E402
from oracle.retinaface_post import iou_plus1                            # noqa: E402
from retinaface_amd.frames import padded_base_frame, synth_frames       # noqa: E402

BLOCK_COUT = (16, 32, 32, 64, 64, 128, 128, 128, 128, 128, 128, 256, 256)
BLOCK_STRIDE = (1, 2, 1, 2, 1, 2, 1, 1, 1, 1, 1, 2, 1)
DW_RANGE = 127 * 128


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


class Sim:
    """tensor names (the stored ones): relu{1..26}, lat3, lat2, lat1, plus0, P2, plus1, P1, and per stride s in 3,2,1:
    c{s}_a (conv1 | ctx1: 48 ch), c{s}_b (ctx2 | ctx3_1: 32 ch), c{s}_c (16 ch).  mode[name] in {"i8", "f16", "f32"}."""

    WEIGHT_BITS = 8          # 0: weights stay real (isolates the activation quantisation)
    W16 = ()                 # key prefixes (pw, lat, aggr, ssh, head; "dw" for the stencils) whose weights carry 16 bits (two int8 limbs): taken as real

    def __init__(self, net, table):
        self.net, self.table = net, table
        self.hess = None         # dict while a calibration pass collects the Gram matrices
        self.gram = {}           # key -> (G, sum, n) from the calibration frames
        self.gptq = None         # dict (cache) when error-compensated rounding is on
        self.bias_corr = True
        f = lambda n: fold(net, n, _bn_of(net, n))                    # noqa: E731
        self.conv0 = f("mobilenet0_conv0_fwd")
        self.dw = [f(f"mobilenet0_conv{2 * i + 1}_fwd") for i in range(13)]
        self.pw = [f(f"mobilenet0_conv{2 * i + 2}_fwd") for i in range(13)]
        self.lat = [f(n) for n in ("rf_c3_lateral", "rf_c2_lateral", "rf_c1_red_conv")]
        self.aggr = [f("rf_c2_aggr"), f("rf_c1_aggr")]
        self.ssh = []
        for c in (3, 2, 1):
            pre, st = f"rf_c{c}_det_", f"stride{ {3: 32, 2: 16, 1: 8}[c]}"
            cat = lambda names: (np.concatenate([f(n)[0] for n in names]), np.concatenate([f(n)[1] for n in names]))   # noqa: E731
            self.ssh.append(dict(a=cat([pre + "conv1", pre + "context_conv1"]), b=cat([pre + "context_conv2", pre + "context_conv3_1"]),
                                 c=f(pre + "context_conv3_2"),
                                 head=cat([f"face_rpn_cls_score_{st}", f"face_rpn_bbox_pred_{st}", f"face_rpn_landmark_pred_{st}"])))

    # ---- scales
    def scale(self, blob, ch):
        tb = self.table
        if blob + "#0" in tb:
            return np.array([tb[f"{blob}#{c}"] for c in range(ch)], np.float32)
        return np.full(ch, tb[blob], np.float32)

    def tensor_scales(self):
        """name -> per-channel scale vector, with the engine's sharing rules (oracle/int8_forward.py)"""
        S, s = self.scale, {}
        c = 8
        for i in range(13):
            s[f"relu{2 * i + 1}"] = S(f"mobilenet0_relu{2 * i + 1}_fwd", c)
            c = BLOCK_COUT[i]
            s[f"relu{2 * i + 2}"] = S(f"mobilenet0_relu{2 * i + 2}_fwd", c)
        lat = [S(n + "_relu", 64) for n in ("rf_c3_lateral", "rf_c2_lateral", "rf_c1_red_conv")]
        plus = [S("_plus0", 64), S("_plus1", 64)]
        ag = [S("rf_c2_aggr_relu", 64), S("rf_c1_aggr_relu", 64)]
        if "_plus0#0" in self.table:
            m0 = np.maximum(lat[0], np.maximum(lat[1], plus[0]))
            lat[0] = lat[1] = plus[0] = m0
            m1 = np.maximum(ag[0], np.maximum(lat[2], plus[1]))
            ag[0] = lat[2] = plus[1] = m1
        s["lat3"], s["lat2"], s["lat1"], s["plus0"], s["plus1"], s["P2"], s["P1"] = lat[0], lat[1], lat[2], plus[0], plus[1], ag[0], ag[1]
        for c in (3, 2, 1):
            pre = f"rf_c{c}_det_"
            cat, c1, c31 = S(pre + "concat_relu", 64), S(pre + "context_conv1_relu", 16), S(pre + "context_conv3_1_relu", 16)
            s[f"c{c}_a"] = np.concatenate([cat[:32], c1])
            s[f"c{c}_b"] = np.concatenate([cat[32:48], c31])
            s[f"c{c}_c"] = cat[48:64]
        return s

    # ---- arithmetic
    @staticmethod
    def store(x, mode, s, relu=True):
        """x real (1, C, H, W) fp32 -> the value the next op reads"""
        if relu:
            x = torch.clamp(x, min=0)
        if mode in ("f32", "w8"):                                    # w8: real activations, int8 weights (isolates the weight quantisation)
            return x
        if mode == "f16":
            return x.half().float()
        if mode == "i16":                                            # ReLU'd tensor on 0..32767 quanta (two int8 limbs in an engine), int8 weights
            sv = t(s).view(1, -1, 1, 1) * (127.0 / 32767.0)
            return torch.round(torch.clamp(x / sv, 0, 32767)) * sv
        if mode == "u8":                                             # ReLU'd tensor on 0..255 quanta of half the size (stored as q - 128)
            sv = t(s).view(1, -1, 1, 1) * (127.0 / 255.0)
            return torch.round(torch.clamp(x / sv, 0, 255)) * sv
        sv = t(s).view(1, -1, 1, 1)
        q = torch.round(torch.clamp(x / sv, 0 if relu else -127, 127))
        return q * sv

    def gemm(self, x, wb, mode_in, s_in, pad, key=None):
        """dense conv of a stored tensor; the weights take the arithmetic of the input"""
        w, b = wb
        wt = t(w.transpose(0, 3, 1, 2).copy())
        if self.hess is not None and key is not None:                  # calibration pass: Gram matrix of the im2col'd input, in real units
            cols = Fn.unfold(x, w.shape[1], padding=pad)[0].double()    # (cin*k*k, L), rows ordered (cin, ky, kx) = wt.flatten(1)'s columns
            g, m, n = self.hess.get(key, (0, 0, 0))
            self.hess[key] = (g + cols @ cols.T, m + cols.sum(dim=1), n + cols.shape[1])
        if mode_in in ("i8", "u8", "w8", "i16"):
            si = t(s_in).view(1, -1, 1, 1) * (127.0 / 255.0 if mode_in == "u8" else 1.0)
            ws = wt * si                                                # per-input-channel scale folded into the weights
            amax = ws.abs().flatten(1).max(dim=1).values
            sw = torch.where(amax > 0, amax / 127, torch.ones_like(amax)).view(-1, 1, 1, 1)
            w16 = key is not None and any(key.startswith(p_) for p_ in Sim.W16)
            wq = torch.clamp(torch.round(ws / sw), -127, 127) if (Sim.WEIGHT_BITS == 8 and not w16) else torch.round(ws / sw * 128) / 128
            if w16:
                pass
            elif self.gptq is not None and key in self.gram and Sim.WEIGHT_BITS == 8:
                ck = (key, mode_in, s_in.tobytes())
                if ck not in self.gptq:
                    self.gptq[ck] = self._gptq(key, ws, sw, si)
                wq, db = self.gptq[ck]
                b = b + db
            y = Fn.conv2d((x / si).double(), wq.double(), None, padding=pad).float() * sw.view(1, -1, 1, 1)
        elif mode_in == "f16":
            y = Fn.conv2d(x, wt.half().float(), None, padding=pad)
        else:
            y = Fn.conv2d(x, wt, None, padding=pad)
        return y + t(np.asarray(b, np.float32)).view(1, -1, 1, 1)

    def _gptq(self, key, ws, sw, si):
        """Error-compensated rounding (GPTQ: Frantar et al. 2022) of one layer's scaled weights ws (cout, cin, k, k) on the row
        scales sw, against the Gram matrix of the calibration inputs; + the bias correction for the residual mean error."""
        g, m, n = self.gram[key]
        sif = si.flatten().double().repeat_interleave(ws.shape[2] * ws.shape[3])           # real -> quanta per im2col row
        H = (g / n) / (sif[:, None] * sif[None, :])
        mean_q = (m / n) / sif
        K = H.shape[0]
        W = (ws.double().flatten(1) / sw.double().view(-1, 1)).clone()                    # in weight quanta
        W0 = W.clone()
        dead = torch.diag(H) <= 0
        H[dead, dead] = 1.0
        H = H + torch.eye(K, dtype=torch.float64) * (0.01 * torch.diag(H).mean())
        U = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(H)), upper=True)
        Q = torch.zeros_like(W)
        for i in range(K):
            q = torch.clamp(torch.round(W[:, i]), -127, 127)
            Q[:, i] = q
            err = (W[:, i] - q) / U[i, i]
            W[:, i + 1:] -= err[:, None] * U[i, i + 1:][None, :]
        db = -((Q - W0) @ mean_q) * sw.double().flatten() if self.bias_corr else torch.zeros(W.shape[0], dtype=torch.float64)
        return Q.view_as(ws).float(), db.float().numpy()

    @staticmethod
    def depthwise(x, wb, mode_in, s_in, s_mid, stride, mode_out):
        w, b = wb
        c = w.shape[0]
        wt = t(w.transpose(0, 3, 1, 2).copy())                           # (c, 1, 3, 3)
        if mode_out == "i16":                                            # a 15-bit mid: the stencil itself is exact enough to be taken as real
            mode_out = "f32"
        if mode_in in ("i8", "u8", "w8") and mode_out in ("i8", "u8", "w8"):
            s_in = s_in * np.float32(127.0 / 255.0 if mode_in == "u8" else 1.0)
            s_mid = s_mid * np.float32(127.0 / 255.0 if mode_out == "u8" else 1.0)
            wf = wt * t(s_in).view(-1, 1, 1, 1) / t(s_mid).view(-1, 1, 1, 1)
            amax = wf.abs().flatten(1).max(dim=1).values
            st = torch.where(amax > 0, amax / DW_RANGE, torch.ones_like(amax)).view(-1, 1, 1, 1)
            wq = torch.clamp(torch.round(wf / st), -DW_RANGE, DW_RANGE) if Sim.WEIGHT_BITS == 8 else wf / st
            acc = Fn.conv2d((x / t(s_in).view(1, -1, 1, 1)).double(), wq.double(), None, stride=stride, padding=1, groups=c).float()
            return (acc * st.view(1, -1, 1, 1) + t(b / s_mid).view(1, -1, 1, 1)) * t(s_mid).view(1, -1, 1, 1)
        if mode_in == "f16" or mode_out == "f16":
            wt = wt.half().float()
        return Fn.conv2d(x, wt, None, stride=stride, padding=1, groups=c) + t(b).view(1, -1, 1, 1)

    @staticmethod
    def upsample2(x):
        n, c, h, w = x.shape
        xp = Fn.pad(x, (1, 1, 1, 1))
        rows = torch.empty((n, c, 2 * h, w + 2))
        rows[:, :, 0::2] = 0.75 * xp[:, :, 1:-1] + 0.25 * xp[:, :, :-2]
        rows[:, :, 1::2] = 0.75 * xp[:, :, 1:-1] + 0.25 * xp[:, :, 2:]
        out = torch.empty((n, c, 2 * h, 2 * w))
        out[:, :, :, 0::2] = 0.75 * rows[:, :, :, 1:-1] + 0.25 * rows[:, :, :, :-2]
        out[:, :, :, 1::2] = 0.75 * rows[:, :, :, 1:-1] + 0.25 * rows[:, :, :, 2:]
        return out

    def forward(self, frame, mode, scales, collect=None):
        """frame (H, W, 3) BGR u8 at net size; mode: name -> "i8"/"f16"/"f32" (default "i8"); returns the 9 head blobs fp32 (C, H, W).
        collect: dict that receives every stored tensor's REAL pre-storage value (for calibration)."""
        M = lambda n: mode.get(n, mode.get("*", "i8"))                  # noqa: E731
        sc = scales
        rec = (lambda n, v: collect.__setitem__(n, torch.clamp(v, min=0)[0].numpy())) if collect is not None else (lambda n, v: None)
        x = t(frame[:, :, ::-1].transpose(2, 0, 1).astype(np.float32))[None]
        w0, b0 = self.conv0
        x = torch.clamp(Fn.conv2d(x, t(w0.transpose(0, 3, 1, 2).copy()), t(b0), stride=2, padding=1), min=0)
        # block 0 belongs to the float front end (fp32-grade); only relu2 is stored
        y = self.depthwise(x, self.dw[0], "f32", None, None, 1, "f32")
        rec("relu1", y)
        y = torch.clamp(y, min=0)
        y = self.gemm(y, self.pw[0], "f32", None, 0, "pw0")
        rec("relu2", y)
        x = self.store(y, M("relu2"), sc["relu2"])
        taps = {}
        for i in range(1, 13):
            nin, nm, no = f"relu{2 * i}", f"relu{2 * i + 1}", f"relu{2 * i + 2}"
            y = self.depthwise(x, self.dw[i], M(nin), sc[nin], sc[nm], BLOCK_STRIDE[i], M(nm))
            rec(nm, y)
            mid = self.store(y, M(nm), sc[nm])
            y = self.gemm(mid, self.pw[i], M(nm), sc[nm], 0, f"pw{i}")
            rec(no, y)
            x = self.store(y, M(no), sc[no])
            if i in (4, 10, 12):
                taps[i] = (x, no)
        lat = []
        for j, (ti, n) in enumerate(((12, "lat3"), (10, "lat2"), (4, "lat1"))):
            xin, nin = taps[ti]
            y = self.gemm(xin, self.lat[j], M(nin), sc[nin], 0, f"lat{j}")
            rec(n, y)
            lat.append(self.store(y, M(n), sc[n]))
        feat, fname = [lat[0]], ["lat3"]
        for j, (pn, an) in enumerate((("plus0", "P2"), ("plus1", "P1"))):
            y = lat[j + 1] + self.upsample2(feat[j])
            rec(pn, y)
            plus = self.store(y, M(pn), sc[pn])
            y = self.gemm(plus, self.aggr[j], M(pn), sc[pn], 1, f"aggr{j}")
            rec(an, y)
            feat.append(self.store(y, M(an), sc[an]))
            fname.append(an)
        heads = {}
        for j, c in enumerate((3, 2, 1)):
            m = self.ssh[j]
            na, nb, nc = f"c{c}_a", f"c{c}_b", f"c{c}_c"
            # the concat slices (read by the 1x1 heads only) and the two context tensors (read by 3x3 convs) may differ in storage type:
            # mode keys c{c}_cat / c{c}_ctx override the per-op keys c{c}_a/b/c
            mcat = lambda n: mode.get(f"c{c}_cat", M(n))                  # noqa: E731
            mctx = lambda n: mode.get(f"c{c}_ctx", M(n))                  # noqa: E731
            ya = self.gemm(feat[j], m["a"], M(fname[j]), sc[fname[j]], 1, f"ssh{c}a")
            rec(na, ya)
            sa_cat, sa_ctx = self.store(ya[:, :32], mcat(na), sc[na][:32]), self.store(ya[:, 32:48], mctx(na), sc[na][32:48])
            yb = self.gemm(sa_ctx, m["b"], mctx(na), sc[na][32:48], 1, f"ssh{c}b")
            rec(nb, yb)
            sb_cat, sb_ctx = self.store(yb[:, :16], mcat(nb), sc[nb][:16]), self.store(yb[:, 16:32], mctx(nb), sc[nb][16:32])
            yc = self.gemm(sb_ctx, m["c"], mctx(nb), sc[nb][16:32], 1, f"ssh{c}c")
            rec(nc, yc)
            scc = self.store(yc, mcat(nc), sc[nc])
            cat = torch.cat([sa_cat, sb_cat, scc], dim=1)
            s_cat = np.concatenate([sc[na][:32], sc[nb][:16], sc[nc]])
            modes = {mcat(na), mcat(nb), mcat(nc)}
            if len(modes) != 1:
                raise ValueError("the three concat slices must share one storage type")
            hm = modes.pop()
            y = self.gemm(cat, m["head"], hm, s_cat, 0, f"head{c}")[0].numpy()
            a = y.shape[0] // 16
            scs = y[:2 * a].reshape(2, a, *y.shape[1:])
            e = np.exp(scs - scs.max(axis=0, keepdims=True)).astype(np.float32)
            s = (32, 16, 8)[j]
            pn_, bn_, ln_ = head_names(s)
            heads[pn_] = (e / e.sum(axis=0, keepdims=True)).astype(np.float32).reshape(2 * a, *y.shape[1:])
            heads[bn_] = np.ascontiguousarray(y[2 * a:6 * a])
            heads[ln_] = np.ascontiguousarray(y[6 * a:])
        return heads


def detect(heads, hw, thr=0.5, with_cands=False):
    h9 = [heads[n] for s in HEAD_STRIDES for n in head_names(s)]
    cand, cidx, kept, kidx = obuild.decode_nms(h9, hw[0], hw[1], thr, 0.4)
    return (kept, kidx, cand, cidx) if with_cands else (kept, kidx)


def stats(results, refs):
    """results / refs: per frame (rows (n, 15), anchor indices).  -> dict"""
    ious, anch, agree, faces, ds, same = [], [], 0, 0, 0.0, 0
    for gg, rr in zip(results, refs):
        (g, gi), (r, ri) = gg[:2], rr[:2]
        same += len(g) == len(r)
        cand = {int(i): c for c, i in zip(rr[2], rr[3])} if len(rr) > 2 else {}
        for row, idx in zip(r, ri):
            faces += 1
            if len(g) == 0:
                ious.append(0.0)
                continue
            iou = [iou_plus1(x[1:5], row[1:5]) for x in g]
            k = int(np.argmax(iou))
            ious.append(iou[k])
            agree += int(gi[k]) == int(idx)
            ds = max(ds, abs(float(g[k][0]) - float(row[0])))
            if int(gi[k]) in cand:                                    # the same ANCHOR's box in the oracle (pre-NMS candidate): regression error alone
                anch.append(iou_plus1(g[k][1:5], cand[int(gi[k])][1:5]))
    ious, anch = np.array(ious), np.array(anch if anch else [0.0])
    return dict(frames=len(refs), faces=faces, same_count=same, worst=float(ious.min()), p01=float(np.quantile(ious, 0.01)), mean=float(ious.mean()),
                below97=int((ious < 0.97).sum()), agree=agree / max(faces, 1), dscore=ds, anchor_worst=float(anch.min()),
                anchor_p01=float(np.quantile(anch, 0.01)), anchor_below97=int((anch < 0.97).sum()))


VARIANTS = {
    "all_i8": {},
    "all_f16": {"*": "f16"},
    "all_f32": {"*": "f32"},
    "cat_f16": {f"c{c}_{p}": "f16" for c in (3, 2, 1) for p in "abc"},
    "ssh_f16": {**{f"c{c}_{p}": "f16" for c in (3, 2, 1) for p in "abc"}, "lat3": "f16", "P2": "f16", "P1": "f16"},
    "fpn_f16": {**{f"c{c}_{p}": "f16" for c in (3, 2, 1) for p in "abc"}, "lat3": "f16", "P2": "f16", "P1": "f16", "lat2": "f16", "lat1": "f16",
                "plus0": "f16", "plus1": "f16"},
    "backbone_only_i8": {**{f"c{c}_{p}": "f16" for c in (3, 2, 1) for p in "abc"}, "lat3": "f16", "P2": "f16", "P1": "f16", "lat2": "f16", "lat1": "f16",
                         "plus0": "f16", "plus1": "f16", "relu10": "f16", "relu22": "f16", "relu26": "f16"},
    "dwmid_f16": {f"relu{2 * i + 1}": "f16" for i in range(1, 13)},
    "dwmid_u8": {f"relu{2 * i + 1}": "u8" for i in range(1, 13)},
    "dwmid_i16": {f"relu{2 * i + 1}": "i16" for i in range(1, 13)},
    "dwmid_w8": {f"relu{2 * i + 1}": "w8" for i in range(1, 13)},
}


def calibrate(sim, frames, rule="amax", per_channel=True, floor=1 / 64.0, margin=1.0):
    """table (blob-name -> scale, incl. `#c` lines) from the fp32 replay of `frames`"""
    amax, store = {}, []
    ones = {k: np.ones(1, np.float32) for k in ()}
    sim.hess = {}
    for f in frames:
        col = {}
        sim.forward(f, {"*": "f32"}, _Any(), collect=col)
        store.append(col) if rule != "amax" else None
        for n, a in col.items():
            m = np.abs(a).reshape(a.shape[0], -1).max(axis=1)
            amax[n] = np.maximum(amax.get(n, 0), m)
    sim.gram, sim.hess = sim.hess, None
    thr = {}
    for n, m in amax.items():
        if rule == "amax":
            tt, tc = float(m.max()), m.astype(np.float64)
        else:
            q = float(rule[1:])
            allv = np.concatenate([c[n].reshape(c[n].shape[0], -1) for c in store], axis=1)
            tc = np.array([np.quantile(v[v > 0], q) if (v > 0).any() else 0.0 for v in allv])
            vv = allv[allv > 0]
            tt = float(np.quantile(vv, q)) if vv.size else 1.0
        thr[n] = (tt * margin, np.maximum(tc, tt * floor) * margin)
    return sim_table(thr, per_channel)


class _Any(dict):
    def __getitem__(self, k):
        return np.ones(256, np.float32)


def sim_table(thr, per_channel):
    """sim tensor names -> reference blob names (what the engine's table reader looks up)"""
    out = {"data": 255.0 / 127.0}

    def put(blob, tt, tc):
        out[blob] = tt / 127.0
        if per_channel:
            for c, v in enumerate(tc):
                out[f"{blob}#{c}"] = float(v) / 127.0
    for i in range(1, 27):
        if f"relu{i}" in thr:
            put(f"mobilenet0_relu{i}_fwd", *thr[f"relu{i}"])
    for n, b in (("lat3", "rf_c3_lateral_relu"), ("lat2", "rf_c2_lateral_relu"), ("lat1", "rf_c1_red_conv_relu"), ("plus0", "_plus0"), ("plus1", "_plus1"),
                 ("P2", "rf_c2_aggr_relu"), ("P1", "rf_c1_aggr_relu")):
        put(b, *thr[n])
    for c in (3, 2, 1):
        pre = f"rf_c{c}_det_"
        (ta, ca), (tb, cb), (tcc, cc) = thr[f"c{c}_a"], thr[f"c{c}_b"], thr[f"c{c}_c"]
        put(pre + "concat_relu", max(float(ca[:32].max()), float(cb[:16].max()), float(cc.max())), np.concatenate([ca[:32], cb[:16], cc]))
        put(pre + "context_conv1_relu", float(ca[32:].max()), ca[32:])
        put(pre + "context_conv3_1_relu", float(cb[16:].max()), cb[16:])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="mnet25")
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--config", type=int, default=300)
    ap.add_argument("--variants", default="all_f32,all_f16,all_i8,cat_f16,ssh_f16,fpn_f16,backbone_only_i8")
    ap.add_argument("--table", default=None, help="calibration table file (default: the one in assets/<model>.rfw)")
    ap.add_argument("--recal", default=None, help="recalibrate on CPU with this rule (amax, p0.9999, ...) on --cal-frames frames of faces 0,2,4")
    ap.add_argument("--cal-frames", type=int, default=24)
    ap.add_argument("--per-tensor", action="store_true")
    ap.add_argument("--margin", type=float, default=1.0, help="head-room factor on the recalibrated thresholds (tools/calibrate_int8.py --margin)")
    ap.add_argument("--gptq", action="store_true", help="error-compensated weight rounding + bias correction on the recalibration frames")
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--w16", default="", help="comma list of contraction-key prefixes whose weights carry 16 bits: pw, pw5 (one block), lat, aggr, ssh, head")
    ap.add_argument("--weight-bits", type=int, default=8)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    Sim.W16 = tuple(k for k in args.w16.split(",") if k)
    Sim.WEIGHT_BITS = args.weight_bits
    net = read_rfw(os.path.join(ROOT, "assets", args.model + ".rfw"))
    table = read_int8_table(args.table) if args.table else dict(net.int8_scales)
    sim = Sim(net, table)
    if args.recal:
        cal = synth_frames(448, 448, args.cal_frames, config=77, faces=[0, 2, 4])
        sim.table = calibrate(sim, cal, args.recal, per_channel=not args.per_tensor, margin=args.margin)
        if args.gptq:
            sim.gptq = {}
    scales = sim.tensor_scales()
    frames = synth_frames(448, 448, args.frames, config=args.config, faces=[1, 3, 5])
    refs = []
    t0 = time.time()
    for f in frames:
        refs.append(detect(sim.forward(f, {"*": "f32"}, scales), (448, 448)))
    print(f"fp32 replay: {time.time() - t0:.1f} s for {len(frames)} frames, {sum(len(r[1]) for r in refs)} faces", flush=True)
    for v in args.variants.split(","):
        t0 = time.time()
        res = [detect(sim.forward(f, VARIANTS[v], scales), (448, 448)) for f in frames]
        s = stats(res, refs)
        print(f"{args.model:18s} {v:18s} same {s['same_count']}/{s['frames']} worst {s['worst']:.4f} p01 {s['p01']:.4f} mean {s['mean']:.4f} "
              f"<0.97: {s['below97']}/{s['faces']} agree {s['agree']:.3f} dscore {s['dscore']:.4f}  ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
