#!/usr/bin/env python3
"""Mint tests/golden/ref_pin.npz from the REFERENCE ITSELF (run in the dev container, where /root/reference exists).

oracle/build_ref.py compiles the reference's own retinaface/RetinaFace.cpp (unmodified, from where it lies) against
stand-in third-party headers; this script drives that build and freezes its outputs, so that on machines without
/root/reference (the GPU box, CI) the oracle is still checked against vectors the reference produced:

  anchors      RetinaFace::_anchors as built by the constructor (RetinaFace.cpp:293-301) for a 448x448 net (sha256 + sums)
               and anchors_plane() (:127) on a 5x7 map per level (full arrays)
  preprocess   the tensor detect() (:576-660) hands the engine for a 37x53 frame in a 64x96 net and for a net-sized frame
  decode+NMS   RetinaFace::postProcess (:495-574) on seeded random head tensors of a 96x128 net and on the heads of the
               448x448 crop of data/img.jpg (tests/golden/crop448_*.npz), at several thresholds
  NMS          RetinaFace::nms (:439-492) on seeded random boxes at several IoU thresholds, incl. a tie case
  regression   bbox_pred (:378-398) / landmark_pred (:418-432) on seeded anchors / deltas (exercises expf rounding)
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import build_ref  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "ref_pin.npz")
POST_THRESHOLDS = (0.5, 0.9, 0.1, 0.02)
NMS_THRESHOLDS = (0.3, 0.4, 0.6)


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def random_heads(rng, net_h, net_w, fg_rate):
    """9 head blobs (C,H,W f32) of one image, HEAD_BLOBS order: fg probabilities are uniform, a share `fg_rate` of them
    above 0.5; deltas are wide enough to hit both clip branches."""
    heads = []
    for s in (32, 16, 8):
        h, w = net_h // s, net_w // s
        fg = rng.random((2, h, w), dtype=np.float32) ** np.float32(np.log(0.5) / np.log(1 - fg_rate))
        heads.append(np.concatenate([1 - fg, fg]).astype(np.float32))
        heads.append((rng.standard_normal((8, h, w)) * 0.6).astype(np.float32))
        heads.append((rng.standard_normal((20, h, w)) * 0.8).astype(np.float32))
    return heads


def random_faces(rng, n, size, ties=False):
    """n x 15 rows of overlapping boxes around a few cluster centres."""
    centres = rng.uniform(0.2 * size, 0.8 * size, (max(n // 12, 1), 2))
    c = centres[rng.integers(0, len(centres), n)] + rng.normal(0, 6, (n, 2))
    wh = rng.uniform(20, 60, (n, 2))
    rows = np.zeros((n, 15), np.float32)
    rows[:, 0] = np.round(rng.random(n) * 8) / 8 if ties else rng.random(n)
    rows[:, 1:3] = c - wh / 2
    rows[:, 3:5] = c + wh / 2
    rows[:, 5:] = rng.uniform(0, size, (n, 10))
    return rows


def main():
    if not build_ref.can_build():
        sys.exit("needs /root/reference (dev container)")
    build_ref.build(force=True)
    rng = np.random.default_rng(20260925)
    d = {}

    # --- anchors -----------------------------------------------------------------------------------------------------
    ref = build_ref.ReferenceRetinaFace(448, 448)
    for lvl, s in enumerate((32, 16, 8)):
        a = ref.anchors(s)
        d[f"anchors448_s{s}_sha"] = np.array(sha(a))
        d[f"anchors448_s{s}_sum"] = a.astype(np.float64).sum(axis=0)
        d[f"anchors_plane_5x7_s{s}"] = ref.anchors_plane(5, 7, lvl)

    # --- regression helpers ------------------------------------------------------------------------------------------
    anchors = np.concatenate([ref.anchors(32)[::37], ref.anchors(16)[::149], ref.anchors(8)[::601]])[:24].astype(np.float32)
    reg = (rng.standard_normal((len(anchors), 4)) * 0.7).astype(np.float32)
    pts = (rng.standard_normal((len(anchors), 10)) * 0.9).astype(np.float32)
    d["reg_anchors"], d["reg_deltas"], d["reg_pts"] = anchors, reg, pts
    d["reg_boxes"] = np.stack([ref.bbox_pred(a, r) for a, r in zip(anchors, reg)])
    d["reg_landmarks"] = np.stack([ref.landmark_pred(a, p) for a, p in zip(anchors, pts)])      # x[5], y[5]

    # --- NMS ---------------------------------------------------------------------------------------------------------
    for name, n, ties in (("nms40", 40, False), ("nms300", 300, False), ("nms12ties", 12, True)):
        faces = random_faces(rng, n, 448, ties)
        d[f"{name}_in"] = faces
        for t in NMS_THRESHOLDS:
            d[f"{name}_out_{t}"] = ref.nms(faces, t)

    # --- decode + NMS on the crop's real head tensors ----------------------------------------------------------------
    for stem in ("mnet-deconv-0517", "mnet25"):
        g = np.load(os.path.join(ROOT, "tests", "golden", f"crop448_{stem}.npz"))
        ref.set_heads(0, [g[n] for n in build_ref.HEAD_BLOBS])
        for t in POST_THRESHOLDS:
            d[f"crop448_{stem}_post_{t}"] = ref.postprocess(0, t)
    ref.close()

    # --- decode + NMS on random heads, small net, image slots 0 and 3 ------------------------------------------------
    ref = build_ref.ReferenceRetinaFace(96, 128)
    for case, rate in (("dense", 0.5), ("sparse", 0.03)):
        heads = random_heads(rng, 96, 128, rate)
        for n, a in zip(build_ref.HEAD_BLOBS, heads):
            d[f"rand_{case}_{n}"] = a
        ref.set_heads(3, heads)
        for t in POST_THRESHOLDS:
            d[f"rand_{case}_post_{t}"] = ref.postprocess(3, t)
    ref.close()

    # --- preprocess --------------------------------------------------------------------------------------------------
    ref = build_ref.ReferenceRetinaFace(64, 96)
    small = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    full = rng.integers(0, 256, (64, 96, 3), dtype=np.uint8)
    ref.detect(small, 0.5)
    d["pre_small_frame"], d["pre_small_input"] = small, ref.last_input()[0]
    ref.detect_batch([full, small], 0.5)
    x = ref.last_input()
    d["pre_full_frame"], d["pre_batch_input_sha"] = full, np.array(sha(x))
    assert x.shape == (2, 3, 64, 96) and np.array_equal(x[1], d["pre_small_input"])
    ref.close()

    np.savez_compressed(OUT, **d)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(d), "arrays")


if __name__ == "__main__":
    main()
