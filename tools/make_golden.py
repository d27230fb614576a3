#!/usr/bin/env python3
"""Freeze golden vectors from the CPU oracle into tests/golden/ (run in the dev container).

The reference ships no golden vectors (it has no tests at all), so these are minted from the
oracle after it has been pinned by its own cross-checks (tests/test_oracle.py).  They are what the `-m gpu`
parity tests compare the HIP path with on the GPU box, where /root/reference does not exist.

  fixture_<stem>.npz      the reference's one image (data/img.jpg, padded to 1280x896, Caffe path):
                          pre-NMS candidates, final detections, global anchor indices, stride-32/16 head blobs,
                          checksums of the stride-8 blobs
  crop448_<stem>.npz      448x448 crop of it (x0=440, y0=30): all 9 head blobs + candidates + detections
  synth448_<stem>.npz     8 seeded synthetic frames (retinaface_amd.frames.synth_frames(448,448,8,config=1)):
                          detections + anchor indices + candidate counts at thr 0.5 and 0.9
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.caffe_forward import HEAD_STRIDES, head_names  # noqa: E402
from oracle.caffe_io import read_rfw  # noqa: E402
from oracle.pipeline import OracleDetector  # noqa: E402
from retinaface_amd.frames import padded_base_frame, synth_frames  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def rows(dets):
    return np.stack([d.as_row() for d in dets]).astype(np.float32) if dets else np.zeros((0, 15), np.float32)


def idx(dets):
    return np.array([d.anchor_index for d in dets], np.int32)


def main():
    os.makedirs(OUT, exist_ok=True)
    frame = padded_base_frame()
    crop = np.ascontiguousarray(frame[30:478, 440:888])
    synth = synth_frames(448, 448, 8, config=1)
    for stem in ("mnet-deconv-0517", "mnet25"):
        det = OracleDetector(read_rfw(os.path.join(ROOT, "assets", stem + ".rfw")))
        r = det.detect(frame, 0.5, 0.4)
        r9 = det.detect(frame, 0.9, 0.4)
        d = {"cand": rows(r.candidates), "cand_idx": idx(r.candidates), "det": rows(r.detections),
             "det_idx": idx(r.detections), "det09": rows(r9.detections), "det09_idx": idx(r9.detections),
             "ncand09": np.int32(len(r9.candidates))}
        for s in (32, 16):
            for n in head_names(s):
                d[n] = r.heads[n][0]
        for n in head_names(8):
            d[n + "_sum"] = np.float64(r.heads[n].astype(np.float64).sum())
            d[n + "_abs"] = np.float64(np.abs(r.heads[n].astype(np.float64)).sum())
        np.savez_compressed(os.path.join(OUT, f"fixture_{stem}.npz"), **d)
        print(stem, "fixture:", len(r.candidates), "cand", len(r.detections), "det")

        c = det.detect(crop, 0.5, 0.4, net_hw=(448, 448))
        d = {"cand": rows(c.candidates), "cand_idx": idx(c.candidates), "det": rows(c.detections), "det_idx": idx(c.detections)}
        for s in HEAD_STRIDES:
            for n in head_names(s):
                d[n] = c.heads[n][0]
        np.savez_compressed(os.path.join(OUT, f"crop448_{stem}.npz"), **d)
        print(stem, "crop448:", len(c.candidates), "cand", len(c.detections), "det")

        d = {}
        for i, f in enumerate(synth):
            for thr, tag in ((0.5, "05"), (0.9, "09")):
                s_ = det.detect(f, thr, 0.4, net_hw=(448, 448))
                d[f"det{tag}_{i}"] = rows(s_.detections)
                d[f"idx{tag}_{i}"] = idx(s_.detections)
                d[f"ncand{tag}_{i}"] = np.int32(len(s_.candidates))
        np.savez_compressed(os.path.join(OUT, f"synth448_{stem}.npz"), **d)
        print(stem, "synth448:", [len(d[f'det05_{i}']) for i in range(8)])


SCORE_NOISE = 2e-3      # fp16 engine: bound on |score - oracle score| asserted by tests/test_gpu_parity.py (TOL[FP16]["score"])


def band(heads, thr, noise=SCORE_NOISE):
    """Anchors whose foreground probability lies within `noise` of the threshold: the only ones an engine whose scores are within
    `noise` of the oracle's can move across `conf > thr` (RetinaFace.cpp:693).  The foreground maps are the second half of each
    cls_prob blob."""
    n = 0
    for s in HEAD_STRIDES:
        p = heads[head_names(s)[0]]
        a = p.shape[1] // 2
        n += int((np.abs(p[:, a:] - np.float32(thr)) <= noise).sum())
    return n


def predicted_logit_noise(frames):
    """What PLAIN fp16 storage (every activation and weight rounded to fp16 where the engine stores one, fp32 accumulation) is predicted to
    cost the classification output, in LOGIT space: max |logit(p_fp16) - logit(p_fp32)| over the anchors whose foreground probability is not
    saturated (0.02 < p < 0.98, where the logit is well conditioned), from tools/fp16_error_budget.py's replay of the fused-op sequence.  A
    score moves by p (1 - p) times the logit error, so this one number bounds the fp16 engine's score error at every p:
    |score - oracle| <= logit_noise * p (1 - p).  The -m gpu tests hold the real engine to 1.0 x this prediction, as they do per layer."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from fp16_error_budget import Sim
    worst = 0.0
    for stem in ("mnet-deconv-0517", "mnet25"):
        sim = Sim(stem)
        acts, wts = sim.all_points(frames[0])
        for f in frames:
            wide, narrow = sim.forward(f, set()), sim.forward(f, set(acts) | set(wts))
            for s_ in HEAD_STRIDES:
                p0, p1 = np.asarray(wide[head_names(s_)[0]], np.float64), np.asarray(narrow[head_names(s_)[0]], np.float64)
                a = p0.shape[1] // 2
                f0, f1 = p0[:, a:].ravel(), np.clip(p1[:, a:].ravel(), 1e-9, 1 - 1e-9)
                m = (f0 > 0.02) & (f0 < 0.98)
                if m.any():
                    worst = max(worst, float(np.abs(np.log(f0[m] / (1 - f0[m])) - np.log(f1[m] / (1 - f1[m]))).max()))
    return worst


def bands():
    """tests/golden/threshold_bands.npz: for every golden frame and threshold, how many anchors sit inside the fp16 score-noise band
    around the threshold -- the tolerance the fp16 candidate-count assertions use instead of a flat +-4 (round 4) -- and (round 5,
    `<key>/twins`) the oracle's own near-tie TWIN anchors: pairs (kept W, suppressed T) on one face whose oracle scores differ by at
    most twice the score noise, with T's oracle row -- the only way an fp16 engine's kept anchor set may differ (tests/anchor_twins.py)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from anchor_twins import twins_of_result
    frame = padded_base_frame()
    crop = np.ascontiguousarray(frame[30:478, 440:888])
    synth = synth_frames(448, 448, 8, config=1)
    d = {"score_noise": np.float32(SCORE_NOISE)}
    for stem in ("mnet-deconv-0517", "mnet25"):
        det = OracleDetector(read_rfw(os.path.join(ROOT, "assets", stem + ".rfw")))
        for thr, tag in ((0.5, "05"), (0.9, "09")):
            cases = [(f"{stem}/fixture/{tag}", det.detect(frame, thr, 0.4)), (f"{stem}/crop448/{tag}", det.detect(crop, thr, 0.4, net_hw=(448, 448)))]
            cases += [(f"{stem}/synth448_{i}/{tag}", det.detect(f, thr, 0.4, net_hw=(448, 448))) for i, f in enumerate(synth)]
            for key, r in cases:
                d[key] = np.int32(band(r.heads, thr))
                d[key + "/twins"] = twins_of_result(r, 0.4, SCORE_NOISE)
    d["logit_noise"] = np.float32(predicted_logit_noise([crop] + list(synth)))
    np.savez_compressed(os.path.join(OUT, "threshold_bands.npz"), **d)
    print("logit noise of plain fp16 storage (max over the golden 448x448 frames, both models):", float(d["logit_noise"]))
    print({k: int(v) for k, v in d.items() if k not in ("score_noise", "logit_noise") and not k.endswith("/twins")})
    print("twin pairs:", {k: [(int(t[0]), int(t[1])) for t in v] for k, v in d.items() if k.endswith("/twins") and len(v)})


if __name__ == "__main__":
    if "--bands" in sys.argv:
        bands()
    else:
        main()
