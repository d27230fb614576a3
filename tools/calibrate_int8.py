#!/usr/bin/env python3
"""INT8 calibration-table generator: writes the per-tensor activation scales the int8 engine needs
in the reference's own text format (`model/mnet-deconv-0517.table.int8`: header line, then `tensor-name: <big-endian hex of
the f32 scale>`, real ~= q * scale; reader: trtnetbase.cpp:31-44, retinaface_amd/csrc/model.cpp).

The reference's INT8-Calibration-Tool only feeds batches to TensorRT's IInt8EntropyCalibrator2; the calibration itself is
inside TensorRT (closed source).  This tool implements the published entropy calibration (NVIDIA, "8-bit inference with
TensorRT", GTC 2017): per tensor a 2048-bin histogram of |x| over the calibration set, and the clipping threshold T among
bins 128..2048 that minimises KL(P || Q) between the clipped distribution and its 128-level quantisation; scale = T / 127.
PARITY STATUS: unpinned (no reference implementation to compare with); sanity-checked against the shipped 0517 table
(`--compare`), and by the parity of the int8 engine that consumes the result (tests/test_gpu_parity.py).

Activations come from THIS repo's fp32 HIP engine (debug_activation of every fused op's output); the tensors the fused
kernels never materialise are rebuilt on the host from those: the depthwise outputs (3x3 stencil with the BN-folded
weights from rf_plan_folded) and the upsample+add tensors `_plus0/_plus1` (closed-form bilinear x2).  Needs a GPU.

Round 6: `--margin` widens every threshold (per-channel amax statistics of a few dozen frames under-estimate unseen data by
10-25 % at the 90th percentile: tools/probes/int8_mix_sim.py), and `--gptq` also calibrates the WEIGHTS: error-compensated
rounding + bias correction per fused dense conv on the calibration activations (retinaface_amd/calibrate.py), written to
`<out stem>.qweights.int8`; `--out-rfw` packs model + table + weights into one .rfw (rf_attach_calibration).

usage: python tools/calibrate_int8.py --model mnet25 --out gpurun_out/mnet25.table.int8 [--frames 48] [--compare table]
                                      [--per-channel --rule amax --margin 1.25 --gptq --out-rfw gpurun_out/mnet25.rfw]
"""
import argparse
import ctypes as C
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NBINS, NQ = 2048, 128


def entropy_threshold(hist: np.ndarray, bin_width: float, step: int = 8) -> float:
    """Clipping threshold minimising KL(P || Q): P = the histogram clipped at bin i with the outliers folded into the last
    bin, Q = the bins below i (without the outliers) merged into NQ levels of i // NQ bins each (remainder into the last
    level) and spread back uniformly over the non-empty bins of each level."""
    hist = hist.astype(np.float64)
    if hist.sum() == 0:
        return bin_width * NBINS
    best_i, best_kl = NBINS, np.inf
    for i in range(NQ, NBINS + 1, step):
        p = hist[:i].copy()
        p[i - 1] += hist[i:].sum()                       # P: clipped values saturate into the last bin
        nz = p > 0
        m = i // NQ
        src = hist[:i]                                   # Q is built from the bins below the clip WITHOUT the saturated mass
        level = src[:NQ * m].reshape(NQ, m).sum(axis=1)
        level[-1] += src[NQ * m:].sum()
        cnt = nz[:NQ * m].reshape(NQ, m).sum(axis=1).astype(np.float64)
        cnt[-1] += nz[NQ * m:].sum()
        per = np.divide(level, cnt, out=np.zeros(NQ), where=cnt > 0)
        q = np.concatenate([np.repeat(per, m), np.full(i - NQ * m, per[-1])])
        q = np.where(nz, q, 0.0)
        if q.sum() <= 0 or np.any(q[nz] <= 0):
            continue
        ps, qs = p / p.sum(), q / q.sum()
        kl = float(np.sum(ps[nz] * np.log(ps[nz] / qs[nz])))
        if kl < best_kl:
            best_kl, best_i = kl, i
    return (best_i + 0.5) * bin_width


def depthwise(x: np.ndarray, w: np.ndarray, b: np.ndarray, stride: int) -> np.ndarray:
    """x (H, W, C) fp32, w (C, 3, 3, 1) BN-folded, pad 1 -> ReLU(dw(x)) (Ho, Wo, C)."""
    h, wd, c = x.shape
    xp = np.zeros((h + 2, wd + 2, c), np.float32)
    xp[1:-1, 1:-1] = x
    ho, wo = (h + 2 - 3) // stride + 1, (wd + 2 - 3) // stride + 1
    acc = np.broadcast_to(b.astype(np.float32), (ho, wo, c)).copy()
    for ky in range(3):
        for kx in range(3):
            acc += xp[ky:ky + stride * (ho - 1) + 1:stride, kx:kx + stride * (wo - 1) + 1:stride] * w[:, ky, kx, 0]
    return np.maximum(acc, 0)


def upsample2(x: np.ndarray) -> np.ndarray:
    """Deconvolution k4 s2 p1 with the fixed bilinear kernel, zero boundary: (h, w, c) -> (2h, 2w, c)."""
    h, w, c = x.shape
    xp = np.zeros((h + 2, w + 2, c), np.float32)
    xp[1:-1, 1:-1] = x
    rows = np.empty((2 * h, w + 2, c), np.float32)
    rows[0::2] = 0.75 * xp[1:-1] + 0.25 * xp[:-2]
    rows[1::2] = 0.75 * xp[1:-1] + 0.25 * xp[2:]
    out = np.empty((2 * h, 2 * w, c), np.float32)
    out[:, 0::2] = 0.75 * rows[:, 1:-1] + 0.25 * rows[:, :-2]
    out[:, 1::2] = 0.75 * rows[:, 1:-1] + 0.25 * rows[:, 2:]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="mnet25")
    ap.add_argument("--out", default=None)
    ap.add_argument("--frames", type=int, default=48)
    ap.add_argument("--height", type=int, default=448)
    ap.add_argument("--width", type=int, default=448)
    ap.add_argument("--compare", default=None, help="an existing table to print side by side")
    ap.add_argument("--rule", default="kl", help="threshold rule: kl (entropy), kl_keep0, p0.999 / p0.9999 / p0.99999 (percentiles), kl_or_p0.9999, amax")
    ap.add_argument("--batches", default=None, help="directory of the reference tool's .batch files (int[4] header + f32 planar RGB, "
                                                    "INT8-Calibration-Tool/calibrationtable.cpp:432-440) to calibrate on")
    ap.add_argument("--images", default=None, help="directory of images to calibrate on (what the reference tool's input_dir holds)")
    ap.add_argument("--faces", default="0,2,4", help="fixture faces the built-in calibration set may use (the held-out int8 parity "
                                                     "test uses the others); 'all' = every face")
    ap.add_argument("--config", type=int, default=77, help="seed block of the built-in synthetic calibration frames")
    ap.add_argument("--margin", type=float, default=1.0, help="multiply every threshold by this (head-room for data the set did not show)")
    ap.add_argument("--gptq", action="store_true", help="also calibrate the weights: error-compensated rounding + bias correction per fused "
                                                        "dense conv, written next to --out as <stem>.qweights.int8")
    ap.add_argument("--out-rfw", default=None, help="pack <model> + the new table (+ weights) into this .rfw")
    ap.add_argument("--per-channel", action="store_true", help="also write per-channel scales as `tensor#<c>: hex` lines (an extension "
                    "the engine understands; TensorRT-style readers ignore them): needs a percentile or amax --rule")
    args = ap.parse_args()

    import retinaface_amd
    from retinaface_amd import _lib
    from retinaface_amd.frames import padded_base_frame, synth_frames
    assets = os.path.join(ROOT, "assets")
    lib = _lib.load_library()

    def folded(op):
        dims = (C.c_int * 4)()
        assert lib.rf_plan_folded(assets.encode(), args.model.encode(), op.encode(), None, 0, None, 0, dims) == 0
        n = dims[0] * dims[1] * dims[2] * dims[3]
        w, b = np.empty(n, np.float32), np.empty(dims[0], np.float32)
        assert lib.rf_plan_folded(assets.encode(), args.model.encode(), op.encode(), w.ctypes.data_as(C.POINTER(C.c_float)), n,
                                  b.ctypes.data_as(C.POINTER(C.c_float)), dims[0], dims) == 0
        return w.reshape(dims[0], dims[1], dims[2], dims[3]), b

    H, W = args.height, args.width
    base = padded_base_frame()
    rng = np.random.default_rng(2026)
    # calibration set: a quarter synthetic face-bearing frames (what bench.py feeds), the rest augmented crops of the one real
    # photo the reference ships (random position, flip, 0.5x..1.5x nearest-neighbour rescale): natural backgrounds and a
    # spread of face sizes -- with synthetic grey backgrounds only, the entropy thresholds come out ~2x tighter than TensorRT's
    from retinaface_amd import calib_io
    from retinaface_amd.frames import FACE_BOXES
    user_frames = []
    if args.batches:
        user_frames += calib_io.read_batch_dir(args.batches)
    if args.images:
        user_frames += calib_io.read_image_dir(args.images)
    faces = None if args.faces == "all" else [int(x) for x in args.faces.split(",")]
    frames = user_frames if user_frames else synth_frames(H, W, args.frames // 4, config=args.config, faces=faces)
    real = base[:886].copy()                                   # without the zero padding rows
    if faces is not None:
        # held-out discipline: the photo's faces that are NOT in the calibration subset are greyed out, so no calibration frame
        # (synthetic or crop) shows a face the held-out parity test uses
        for i, (x1, y1, x2, y2) in enumerate(FACE_BOXES):
            if i not in faces:
                cx, cy, bw, bh = (x1 + x2) // 2, (y1 + y2) // 2, int((x2 - x1) * 0.8), int((y2 - y1) * 0.8)
                real[max(0, cy - bh):cy + bh, max(0, cx - bw):cx + bw] = 128
    while not user_frames and len(frames) < args.frames:
        sc = float(rng.choice([0.5, 0.75, 1.0, 1.0, 1.5]))
        ys = (np.arange(int(real.shape[0] * sc)) / sc).astype(int)
        xs = (np.arange(int(real.shape[1] * sc)) / sc).astype(int)
        img = real[ys][:, xs]
        if rng.random() < 0.5:
            img = img[:, ::-1]
        canvas = np.zeros((max(H, img.shape[0]), max(W, img.shape[1]), 3), np.uint8)
        canvas[:img.shape[0], :img.shape[1]] = img
        y, x = int(rng.integers(0, canvas.shape[0] - H + 1)), int(rng.integers(0, canvas.shape[1] - W + 1))
        frames.append(np.ascontiguousarray(canvas[y:y + H, x:x + W]))

    det = retinaface_amd.RetinaFace(assets, "net3", 0.4, precision=retinaface_amd.PRECISION_FP32, net_hw=(H, W),
                                    model_stem=args.model, keep_outputs=True, use_graph=False, lanes=1, coalesce=1)
    dws = [folded(f"dw{i}") for i in range(13)]
    strides = [1, 2, 1, 2, 1, 2, 1, 1, 1, 1, 1, 2, 1]
    pw_names = [f"mobilenet0_relu{2 * i + 2}_fwd" for i in range(13)]
    dw_names = [f"mobilenet0_relu{2 * i + 1}_fwd" for i in range(13)]

    def tensors(frame):
        """name -> activation for every tensor the int8 engine has a scale lookup for (engine.cpp upload_weights)."""
        det.detect(frame, 0.5)
        t = {}
        conv0 = det.debug_activation("mobilenet0_relu0_fwd")
        prev = conv0
        for i in range(13):
            w, b = dws[i]
            t[dw_names[i]] = depthwise(prev, w, b, strides[i])
            prev = det.debug_activation(pw_names[i])
            t[pw_names[i]] = prev
        lat = {c: det.debug_activation(f"rf_c{c}_{'red_conv' if c == 1 else 'lateral'}_relu") for c in (3, 2, 1)}
        for c in (3, 2, 1):
            t[f"rf_c{c}_{'red_conv' if c == 1 else 'lateral'}_relu"] = lat[c]
        aggr2 = det.debug_activation("rf_c2_aggr_relu")
        aggr1 = det.debug_activation("rf_c1_aggr_relu")
        t["rf_c2_aggr_relu"], t["rf_c1_aggr_relu"] = aggr2, aggr1
        t["_plus0"] = lat[2] + upsample2(lat[3])
        t["_plus1"] = lat[1] + upsample2(aggr2)
        for c in (3, 2, 1):
            for n in ("context_conv1_relu", "context_conv3_1_relu", "concat_relu"):
                t[f"rf_c{c}_det_{n}"] = det.debug_activation(f"rf_c{c}_det_{n}")
        return t

    # pass 1: ranges; pass 2: histograms (per tensor, and per channel when asked for)
    amax, amax_c = {}, {}
    for f in frames:
        for n, a in tensors(f).items():
            amax[n] = max(amax.get(n, 0.0), float(np.abs(a).max()))
            if args.per_channel:
                m = np.abs(a).reshape(-1, a.shape[-1]).max(axis=0)
                amax_c[n] = np.maximum(amax_c.get(n, 0.0), m)
    hist = {n: np.zeros(NBINS, np.int64) for n in amax}
    hist_c = {n: np.zeros((len(amax_c[n]), NBINS), np.int64) for n in amax_c}
    # fused dense convs of the int8 plan -> (input tensor, kernel size): what --gptq needs the Gram matrices of
    gemm_inputs = {}
    for i in range(1, 13):
        gemm_inputs[f"mobilenet0_conv{2 * i + 2}_fwd"] = (dw_names[i], 1)
    for nm, tap in (("rf_c3_lateral", 12), ("rf_c2_lateral", 10), ("rf_c1_red_conv", 4)):
        gemm_inputs[nm] = (pw_names[tap], 1)
    gemm_inputs["rf_c2_aggr"], gemm_inputs["rf_c1_aggr"] = ("_plus0", 3), ("_plus1", 3)
    for c, feat, st in ((3, "rf_c3_lateral_relu", 32), (2, "rf_c2_aggr_relu", 16), (1, "rf_c1_aggr_relu", 8)):
        pre = f"rf_c{c}_det_"
        gemm_inputs[f"{pre}conv1+{pre}context_conv1"] = (feat, 3)
        gemm_inputs[f"{pre}context_conv2+{pre}context_conv3_1"] = (pre + "context_conv1_relu", 3)
        gemm_inputs[f"{pre}context_conv3_2"] = (pre + "context_conv3_1_relu", 3)
        gemm_inputs[f"face_rpn_cls_score_stride{st}+face_rpn_bbox_pred_stride{st}+face_rpn_landmark_pred_stride{st}"] = (pre + "concat_relu", 1)
    grams = {}
    from retinaface_amd import calibrate as cal
    for f in frames:
        acts = tensors(f)
        if args.gptq:
            for op, (src, k) in gemm_inputs.items():
                a = acts[src]
                if op not in grams:
                    grams[op] = cal.Gram(k * k * a.shape[-1])
                grams[op].add(a, k)
        for n, a in acts.items():
            h, _ = np.histogram(np.abs(a), bins=NBINS, range=(0.0, max(amax[n], 1e-12)))
            hist[n] += h
            if args.per_channel:
                c = a.shape[-1]
                flat = np.abs(a).reshape(-1, c)
                idx = np.minimum((flat / np.maximum(amax_c[n], 1e-12) * NBINS).astype(np.int64), NBINS - 1)
                hist_c[n] += np.bincount((np.arange(c)[None, :] * NBINS + idx).ravel(), minlength=c * NBINS).reshape(c, NBINS)
    scales = {"data": 255.0 / 127.0}                            # raw u8 pixels; the stem reads them exactly anyway
    scales_c = {}
    variants = {}
    for n in hist:
        bw = amax[n] / NBINS
        h0 = hist[n].copy()
        hist[n][0] = 0                                          # ReLU zeros carry no information about the range
        t_kl = entropy_threshold(hist[n], bw)
        cdf = np.cumsum(hist[n]) / max(hist[n].sum(), 1)
        pct = {q: (int(np.searchsorted(cdf, q)) + 1) * bw for q in (0.999, 0.9999, 0.99999)}
        variants[n] = {"kl": t_kl, "kl_keep0": entropy_threshold(h0, bw), **{f"p{q}": v for q, v in pct.items()},
                       "kl_or_p0.9999": max(t_kl, pct[0.9999]), "amax": amax[n]}
        scales[n] = variants[n][args.rule] * args.margin / 127.0
        if args.per_channel:
            if args.rule == "amax":
                t_c = amax_c[n].astype(np.float64)
            elif args.rule.startswith("p0."):
                q = float(args.rule[1:])
                hc = hist_c[n].copy()
                hc[:, 0] = 0
                cdf_c = np.cumsum(hc, axis=1) / np.maximum(hc.sum(axis=1, keepdims=True), 1)
                t_c = ((cdf_c < q).sum(axis=1) + 1) * (amax_c[n] / NBINS)
            else:
                raise SystemExit("--per-channel needs --rule amax or p0.xxx")
            # a channel that is (almost) dead in the calibration set must not get a vanishing quantum: floor at 1/64 of the tensor's
            scales_c[n] = np.maximum(t_c, variants[n][args.rule] / 64.0) * args.margin / 127.0

    other = {}
    if args.compare:
        for line in open(args.compare).read().splitlines()[1:]:
            if ": " in line:
                k, v = line.rsplit(": ", 1)
                other[k] = struct.unpack(">f", bytes.fromhex(v))[0]
    if other:
        print("rule vs shipped table: median / min / max of (threshold / shipped threshold) over", len(variants), "tensors")
        for rule in next(iter(variants.values())):
            r = np.array([variants[n][rule] / 127.0 / other[n] for n in variants if n in other])
            print(f"  {rule:16s} median {np.median(r):.2f}  min {r.min():.2f}  max {r.max():.2f}  mean |log2| {np.abs(np.log2(r)).mean():.3f}")
    for n in sorted(scales):
        extra = f"   shipped {other[n]:.5f}  ratio {scales[n] / other[n]:.2f}" if n in other else ""
        print(f"{n:44s} amax {amax.get(n, 255.0):9.3f}  scale {scales[n]:.5f}  (T = {scales[n] * 127:8.3f}){extra}")
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as fo:
            fo.write("TRT-5102-EntropyCalibration2\n")        # the header both readers (and TensorRT) expect; provenance: this tool
            for n, s in scales.items():
                fo.write(f"{n}: {struct.pack('>f', np.float32(s)).hex()}\n")
            for n, sc in scales_c.items():
                for c, s in enumerate(sc):
                    fo.write(f"{n}#{c}: {struct.pack('>f', np.float32(s)).hex()}\n")
        print("wrote", args.out)
    if args.gptq or args.out_rfw:
        if not args.out:
            raise SystemExit("--gptq / --out-rfw need --out (the table the weights are calibrated under)")
        qpath = None
        if args.gptq:
            qpath = args.out[:-len(".table.int8")] + ".qweights.int8" if args.out.endswith(".table.int8") else args.out + ".qweights.int8"
            qw = calibrate_weights(lib, assets, args.model, args.out, grams)
            cal.write_qweights(qw, qpath)
            print("wrote", qpath)
        if args.out_rfw:
            st = lib.rf_attach_calibration(assets.encode(), args.model.encode(), args.out.encode(), qpath.encode() if qpath else None, args.out_rfw.encode())
            if st != 0:
                raise SystemExit(f"rf_attach_calibration failed: {st} {lib.rf_last_error(None)}")
            print("wrote", args.out_rfw)


def calibrate_weights(lib, model_dir, stem, table_path, grams):
    """Error-compensated rounding of every fused dense conv of the int8 plan under the table at `table_path` (retinaface_amd/calibrate.py)."""
    from retinaface_amd import calibrate as cal
    F = C.POINTER(C.c_float)
    names = []
    while True:
        dims = (C.c_int * 4)()
        buf = np.zeros(64, np.float32)
        if lib.rf_plan_int8_gemm(model_dir.encode(), stem.encode(), table_path.encode(), f"?{len(names)}".encode(), buf.ctypes.data_as(F), buf.size,
                                 None, 0, None, None, 0, dims) != 0:
            break
        names.append(buf.tobytes()[:dims[0]].decode())
    missing = [n for n in names if n not in grams]
    if missing:
        raise SystemExit(f"no calibration inputs collected for the fused ops {missing}")
    out = {}
    tot_rtn = tot_cal = 0.0
    for op in names:
        dims = (C.c_int * 4)()
        assert lib.rf_plan_int8_gemm(model_dir.encode(), stem.encode(), table_path.encode(), op.encode(), None, 0, None, 0, None, None, 0, dims) == 0
        cout, ktot, cin, _ = dims
        quanta, s_in = np.empty(cout * ktot, np.float32), np.empty(cin, np.float32)
        s_row, s_out = np.empty(cout, np.float32), np.empty(cout, np.float32)
        assert lib.rf_plan_int8_gemm(model_dir.encode(), stem.encode(), table_path.encode(), op.encode(), quanta.ctypes.data_as(F), quanta.size,
                                     s_in.ctypes.data_as(F), cin, s_row.ctypes.data_as(F), s_out.ctypes.data_as(F), cout, dims) == 0
        quanta = quanta.reshape(cout, ktot)
        g = grams[op]
        if g.g.shape[0] != ktot:
            raise SystemExit(f"{op}: Gram matrix of size {g.g.shape[0]} for K = {ktot}")
        sk = np.tile(s_in.astype(np.float64), ktot // cin)                     # k = tap * cin + c
        hq, mq = (g.g / g.n) / np.outer(sk, sk), (g.s / g.n) / sk               # real units -> input quanta
        q, mean_err = cal.gptq_round(quanta, hq, mq)
        rtn = np.clip(np.rint(quanta), -127, 127)
        e_rtn, e_cal = cal.output_error(quanta, rtn, hq).sum(), cal.output_error(quanta, q, hq).sum()
        tot_rtn += e_rtn
        tot_cal += e_cal
        out[op] = (q, (-mean_err * s_row.astype(np.float64)).astype(np.float32))        # row-grid units -> real units
        print(f"  {op[:60]:60s} {cout:3d} x {ktot:4d}: output error on the calibration set  rtn {e_rtn:10.3f} -> {e_cal:10.3f}  "
              f"({(q != rtn).mean() * 100:4.1f} % of the weights moved)")
    print(f"weights calibrated: {len(out)} fused convs, summed output error (grid units^2) {tot_rtn:.1f} -> {tot_cal:.1f}")
    return out


if __name__ == "__main__":
    main()
