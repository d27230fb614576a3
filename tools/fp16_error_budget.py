#!/usr/bin/env python3
"""fp16 error budget of the fused engine, simulated on the CPU (no GPU needed).

The fp16 engine keeps fp32 accumulators and rounds to fp16 exactly where an activation is STORED (LDS tile or HBM)
and where a weight is packed.  This tool replays the engine's fused-op sequence (same BN folding: the host plan via
rf_plan_folded, same fusion boundaries as engine.cpp::build_lane) with PyTorch-CPU fp32 convolutions and a switchable
`round to fp16` at every one of those points, decodes with the oracle's literal decode/NMS and reports, per face,
1 - IoU against the all-fp32 run.  It answers: which rounding points carry the box error of the fp16 engine, and which
cheap subset has to be kept wider to get every face inside north_star's 1e-3 IoU.

    python tools/fp16_error_budget.py --frames 16               # baseline + one-at-a-time sensitivities
    python tools/fp16_error_budget.py --wide w:all              # try a configuration (see --help)
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.retinaface_post import decode, iou_plus1, nms  # noqa: E402
from oracle.caffe_forward import HEAD_STRIDES, head_names  # noqa: E402
from retinaface_amd import _lib  # noqa: E402
from retinaface_amd.frames import synth_frames, padded_base_frame  # noqa: E402


def folded(stem: str, op: str):
    lib = _lib.load_library()
    dims = (C.c_int * 4)()
    _lib.check(lib.rf_plan_folded(os.path.join(ROOT, "assets").encode(), stem.encode(), op.encode(), None, 0, None, 0, dims))
    co, k, _, ci = list(dims)
    w = np.empty(co * k * k * ci, np.float32)
    b = np.empty(co, np.float32)
    _lib.check(lib.rf_plan_folded(os.path.join(ROOT, "assets").encode(), stem.encode(), op.encode(),
                                  w.ctypes.data_as(C.POINTER(C.c_float)), w.size, b.ctypes.data_as(C.POINTER(C.c_float)), b.size, dims))
    # [cout][ky][kx][cin/g] -> torch OIHW
    return torch.from_numpy(w.reshape(co, k, k, ci).transpose(0, 3, 1, 2).copy()), torch.from_numpy(b)


def h16(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.float16).to(torch.float32)


class Sim:
    """`narrow` = set of rounding points that are fp16 (activation points 'a:<name>', weight points 'w:<name>')."""

    def __init__(self, stem: str):
        self.stem = stem
        self.W = {}
        names = ["conv0"] + [f"dw{i}" for i in range(13)] + [f"pw{i}" for i in range(13)] + [f"lateral{i}" for i in range(3)]
        names += [f"aggr{i}" for i in range(2)] + [f"ssh{i}.{t}" for i in range(3) for t in ("a", "b", "c", "head")]
        for n in names:
            self.W[n] = folded(stem, n)
        self.points = []

    def conv(self, x, name, narrow, stride=1, pad=0, groups=1, relu=True, relu_from=0):
        w, b = self.W[name]
        if f"w:{name}" in narrow and name != "conv0":      # conv0 weights are an fp16 hi+lo pair: fp32-grade
            w = h16(w)
        y = F.conv2d(x, w, b, stride=stride, padding=pad, groups=groups)
        if relu:
            if relu_from:
                y = torch.cat([y[:, :relu_from], F.relu(y[:, relu_from:])], 1)
            else:
                y = F.relu(y)
        return y

    def act(self, y, name, narrow):
        if name not in self.points:
            self.points.append(name)
        y = h16(y) if f"a:{name}" in narrow else y
        if self.keep is not None:
            self.keep[name] = y
        return y

    keep = None        # set to a dict to record every storage point's tensor of the next forward (predicted_layer_errors)

    def forward(self, frame_bgr: np.ndarray, narrow) -> dict:
        x = torch.from_numpy(frame_bgr[:, :, ::-1].copy()).permute(2, 0, 1)[None].float()     # RGB planar, raw 0..255
        x = self.act(self.conv(x, "conv0", narrow, stride=2, pad=1), "conv0", narrow)
        lat = {}
        for i in range(13):
            wdw, _ = self.W[f"dw{i}"]
            c = wdw.shape[0]
            s = 2 if i in (1, 3, 5, 11) else 1
            x = self.act(self.conv(x, f"dw{i}", narrow, stride=s, pad=1, groups=c), f"dw{i}", narrow)
            x = self.act(self.conv(x, f"pw{i}", narrow), f"pw{i}", narrow)
            li = {4: 2, 10: 1, 12: 0}.get(i)
            if li is not None:
                lat[li] = self.act(self.conv(x, f"lateral{li}", narrow), f"lateral{li}", narrow)
        feat = [lat[0], None, None]
        k1 = torch.tensor([0.25, 0.75, 0.75, 0.25])
        for i in range(2):
            up_in = feat[i]
            c = up_in.shape[1]
            wk = (k1[:, None] * k1[None, :])[None, None].repeat(c, 1, 1, 1)
            up = F.conv_transpose2d(up_in, wk, stride=2, padding=1, groups=c)
            plus = self.act(lat[i + 1] + up, f"plus{i}", narrow)
            feat[i + 1] = self.act(self.conv(plus, f"aggr{i}", narrow, pad=1), f"aggr{i}", narrow)
        heads = {}
        for i, s in enumerate(HEAD_STRIDES):
            ya = self.conv(feat[i], f"ssh{i}.a", narrow, pad=1)                 # det_conv1 (32) | context_conv1 (16): all ReLU'd on store
            ya = self.act(ya, f"ssh{i}.a", narrow)
            yb = self.act(self.conv(ya[:, 32:], f"ssh{i}.b", narrow, pad=1), f"ssh{i}.b", narrow)
            yc = self.act(self.conv(yb[:, 16:], f"ssh{i}.c", narrow, pad=1), f"ssh{i}.c", narrow)
            cat = torch.cat([ya[:, :32], yb[:, :16], yc], 1)
            o = self.conv(cat, f"ssh{i}.head", narrow, relu=False)[0]
            sc = o[0:4]
            m = torch.maximum(sc[0:2], sc[2:4])
            e0, e1 = torch.exp(sc[0:2] - m), torch.exp(sc[2:4] - m)
            prob = torch.cat([e0 / (e0 + e1), e1 / (e0 + e1)], 0)
            n_cls, n_box, n_lmk = head_names(s)
            heads[n_cls] = prob[None].numpy()
            heads[n_box] = o[4:12][None].numpy()
            heads[n_lmk] = o[12:32][None].numpy()
        return heads

    def all_points(self, frame):
        self.forward(frame, set())
        acts = [f"a:{p}" for p in self.points]
        wts = [f"w:{n}" for n in self.W if n != "conv0"]
        return acts, wts


# storage point of the simulation -> the reference blob the engine's debug accessor serves (engine.cpp act() names)
def blob_of_point(p: str):
    if p == "conv0":
        return "mobilenet0_relu0_fwd", None
    if p.startswith("pw"):
        return f"mobilenet0_relu{2 * int(p[2:]) + 2}_fwd", None
    if p.startswith("lateral"):
        return ["rf_c3_lateral_relu", "rf_c2_lateral_relu", "rf_c1_red_conv_relu"][int(p[7:])], None
    if p.startswith("aggr"):
        return ["rf_c2_aggr_relu", "rf_c1_aggr_relu"][int(p[4:])], None
    if p.startswith("ssh") and p.endswith(".a"):
        return f"rf_c{3 - int(p[3])}_det_context_conv1_relu", slice(32, 48)
    return None, None


def predicted_layer_errors(stem: str, frame_bgr: np.ndarray) -> dict:
    """What plain fp16 storage (every activation and weight point narrow, fp32 accumulation) costs at every tensor the engine can
    show: {blob name: (max |fp16 run - fp32 run|, max |fp32 run|)} from two forwards of the simulated fused-op sequence.  The -m gpu
    per-layer test holds the real engine to a small multiple of these instead of a flat fraction of the range."""
    sim = Sim(stem)
    acts, wts = sim.all_points(frame_bgr)
    runs = []
    for narrow in (set(), set(acts) | set(wts)):
        sim.keep = {}
        sim.forward(frame_bgr, narrow)
        runs.append(sim.keep)
    sim.keep = None
    out = {}
    for p, wide in runs[0].items():
        blob, sl = blob_of_point(p)
        if blob is None:
            continue
        a, b = wide[0], runs[1][p][0]
        if sl is not None:
            a, b = a[sl], b[sl]
        out[blob] = (float((a - b).abs().max()), float(a.abs().max()))
    for i in range(3):       # the concat tensor: det_conv1 | context_conv2 | context_conv3_2 slices, all ReLU'd
        parts = [[r[f"ssh{i}.a"][0][:32], r[f"ssh{i}.b"][0][:16], r[f"ssh{i}.c"][0]] for r in runs]
        a, b = torch.cat(parts[0]), torch.cat(parts[1])
        out[f"rf_c{3 - i}_det_concat_relu"] = (float((a - b).abs().max()), float(a.abs().max()))
    return out


def detect(sim, frame, narrow, thr=0.5):
    heads = sim.forward(frame, narrow)
    h, w = frame.shape[:2]
    return nms(list(decode(heads, h, w, thr)), 0.4)


def worst_err(ref, got):
    if [d.anchor_index for d in ref] != [d.anchor_index for d in got]:
        return None
    return max([1.0 - iou_plus1(a.rect, b.rect) for a, b in zip(ref, got)], default=0.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stem", default="mnet25")
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--config", type=int, default=1)
    ap.add_argument("--wide", default="", help="comma list of points kept WIDE (fp32) while everything else is fp16, e.g. "
                                               "'w:all,a:ssh0.a' ('w:all' / 'a:all' = every weight / activation point); empty = "
                                               "run the baseline and the one-at-a-time sensitivity table")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 8)
    sim = Sim(args.stem)
    frames = synth_frames(448, 448, args.frames, config=args.config)
    acts, wts = sim.all_points(frames[0])
    every = set(acts) | set(wts)
    refs = [detect(sim, f, set()) for f in frames]

    def run(narrow):
        errs = []
        for f, r in zip(frames, refs):
            e = worst_err(r, detect(sim, f, narrow))
            errs.append(float("nan") if e is None else e)
        return errs

    def expand(spec):
        wide = set()
        for tok in filter(None, spec.split(",")):
            if tok == "w:all":
                wide |= set(wts)
            elif tok == "a:all":
                wide |= set(acts)
            elif tok.endswith("*"):
                wide |= {p for p in every if p.startswith(tok[:-1])}
            else:
                assert tok in every, tok
                wide.add(tok)
        return wide

    res = {}
    if args.wide:
        e = run(every - expand(args.wide))
        res[args.wide] = e
        print(f"wide = {args.wide}: worst 1-IoU {np.nanmax(e):.2e}  mean {np.nanmean(e):.2e}  per frame {['%.1e' % v for v in e]}")
    else:
        base = run(every)
        print(f"all fp16: worst 1-IoU {np.nanmax(base):.2e}  mean {np.nanmean(base):.2e}  nan (anchor set differs) {int(np.isnan(base).sum())}")
        res["all_fp16"] = base
        for label, sub in (("activations only", set(acts)), ("weights only", set(wts))):
            e = run(sub)
            res[label] = e
            print(f"{label:18s}: worst {np.nanmax(e):.2e}  mean {np.nanmean(e):.2e}")
        print("one point narrow at a time (rms over frames of the worst-face 1-IoU; squared contributions add up):")
        rows = []
        for p in acts + wts:
            e = np.array(run({p}))
            rows.append((float(np.sqrt(np.nanmean(e ** 2))), float(np.nanmax(e)), p))
            res[p] = e.tolist()
        tot = sum(r[0] ** 2 for r in rows)
        for rms, mx, p in sorted(rows, reverse=True):
            print(f"  {p:14s} rms {rms:.2e}  max {mx:.2e}  share of variance {rms * rms / tot:5.1%}")
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
