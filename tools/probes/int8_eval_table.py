"""Probe: int8 parity of a model with a given calibration table (repacked into a temporary .rfw), on
  (a) the golden synthetic frames (config 1, all six fixture faces),
  (b) HELD-OUT frames: synthetic frames built only from fixture faces 1, 3, 5 with seeds no calibration run uses -- the built-in
      calibration set of tools/calibrate_int8.py uses faces 0, 2, 4 only (and greys the others out of its photo crops),
  (c) the reference photo at 1280 x 896.
Reports, per set: same face count, worst / mean per-face IoU vs the fp32 CPU oracle, anchor-index agreement rate, max |dscore|.
usage: int8_eval_table.py MODEL [TABLE]      (TABLE omitted = the scales already inside assets/MODEL.rfw)"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import retinaface_amd
from oracle.caffe_io import read_int8_table, read_rfw, write_rfw
from oracle.pipeline import OracleDetector
from oracle.retinaface_post import iou_plus1
from retinaface_amd.frames import padded_base_frame, synth_frames

stem = sys.argv[1]
net = read_rfw(os.path.join(ROOT, "assets", stem + ".rfw"))
d = os.path.join(ROOT, "assets")
if len(sys.argv) > 2:
    net.int8_scales = read_int8_table(sys.argv[2])
    d = tempfile.mkdtemp()
    write_rfw(net, os.path.join(d, stem + ".rfw"))
orc = OracleDetector(net)


def report(name, frames, hw, **kw):
    det = retinaface_amd.RetinaFace(d, "net3", 0.4, precision=2, net_hw=hw, model_stem=stem, **kw)
    res = det.detectBatchImages(frames, 0.5)
    ious, same_anchor, faces, dscore, same_count = [], 0, 0, 0.0, True
    for f, got in zip(frames, res):
        ref = orc.detect(f, 0.5, 0.4, net_hw=hw).detections
        same_count &= len(ref) == len(got)
        for r in ref:
            faces += 1
            if not got:
                ious.append(0.0)
                continue
            best = max(got, key=lambda a: iou_plus1(a.rect, r.rect))
            ious.append(iou_plus1(best.rect, r.rect))
            same_anchor += best.anchor_index == r.anchor_index
            dscore = max(dscore, abs(best.score - r.score))
    print(f"{stem} {name:10s}: {len(frames)} frames, {faces} faces, same count {same_count}, IoU worst {min(ious):.4f} mean {np.mean(ious):.4f}, "
          f"anchor agreement {same_anchor}/{faces} = {same_anchor / max(faces, 1):.3f}, max |dscore| {dscore:.4f}")
    det.close()


report("golden", synth_frames(448, 448, 8, config=1), (448, 448))
report("held-out", synth_frames(448, 448, 32, config=300, faces=[1, 3, 5]), (448, 448), max_batch=32)
report("photo", [padded_base_frame()], (896, 1280), max_batch=1)
