"""Probe: worst per-face IoU / score error of the int8 engine vs the golden fp32-oracle detections, per model."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import retinaface_amd
from oracle.retinaface_post import iou_plus1
from retinaface_amd.frames import synth_frames, padded_base_frame
assets = os.path.join(ROOT, "assets")
for stem in ("mnet-deconv-0517", "mnet25"):
    det = retinaface_amd.RetinaFace(assets, "net3", 0.4, precision=2, net_hw=(448, 448), model_stem=stem)
    g = np.load(os.path.join(ROOT, "tests", "golden", f"synth448_{stem}.npz"))
    res = det.detectBatchImages(synth_frames(448, 448, 8, config=1), 0.5)
    worst, ds, same = 1.0, 0.0, True
    for i in range(8):
        ref = g[f"det05_{i}"]
        same &= len(res[i]) == len(ref)
        for r in ref:
            best = max((iou_plus1(a.rect, r[1:5]), -abs(a.score - r[0])) for a in res[i]) if res[i] else (0, 0)
            worst = min(worst, best[0]); ds = max(ds, -best[1])
    big = retinaface_amd.RetinaFace(assets, "net3", 0.4, precision=2, net_hw=(896, 1280), max_batch=2, model_stem=stem)
    got = big.detect(padded_base_frame(), 0.5)
    ref = np.load(os.path.join(ROOT, "tests", "golden", f"fixture_{stem}.npz"))["det"]
    wf = min(max(iou_plus1(a.rect, r[1:5]) for a in got) for r in ref)
    print(f"{stem}: synth same-count {same} worst IoU {worst:.4f} max |dscore| {ds:.4f}; fixture {len(got)} faces worst IoU {wf:.4f}")
