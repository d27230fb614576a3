"""Probe: worst 1 - IoU of the fp16 engine against the fp32 oracle over the frames the parity tests use (+ extra synthetic frames),
so that a kernel change can be judged by its margin to north_star's 1e-3 and not by one pass / fail bit.
usage: [RETINAFACE_AMD_LIB=...] python tools/probes/iou_margin.py [nsynth]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import retinaface_amd
from oracle.caffe_io import read_rfw
from oracle.pipeline import OracleDetector
from oracle.retinaface_post import iou_plus1
from retinaface_amd.frames import synth_frames, padded_base_frame
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 8
assets = os.path.join(ROOT, "assets")
base = padded_base_frame()
def cases():
    hw = (352, 608)
    yield "0517 odd352x608 full", "mnet-deconv-0517", hw, np.ascontiguousarray(base[100:100 + hw[0], 400:400 + hw[1]])
    yield "0517 odd352x608 small", "mnet-deconv-0517", hw, np.ascontiguousarray(base[100:100 + 301, 400:400 + 517])
    for stem in ("mnet-deconv-0517", "mnet25"):
        yield stem + " 1280x896", stem, (896, 1280), base
        for cfg in (1, 2, 3):
            for i, f in enumerate(synth_frames(448, 448, ns, config=cfg)):
                yield f"{stem} synth cfg{cfg} #{i}", stem, (448, 448), f
dets, oras = {}, {}
rows = []
for name, stem, hw, frame in cases():
    if stem not in oras: oras[stem] = OracleDetector(read_rfw(os.path.join(assets, stem + ".rfw")))
    key = (stem, hw)
    if key not in dets: dets[key] = retinaface_amd.RetinaFace(assets, "net3", 0.4, precision=1, net_hw=hw, model_stem=stem)
    ref = oras[stem].detect(frame, 0.5, 0.4, net_hw=hw)
    got = dets[key].detect(frame, 0.5)
    # faces are matched by global anchor index: two faces whose scores differ by less than the fp16 score noise may swap places
    by_anchor = {d.anchor_index: d for d in got}
    same = sorted(by_anchor) == sorted(d.anchor_index for d in ref.detections) and len(by_anchor) == len(got)
    worst = max([1 - iou_plus1(by_anchor[r.anchor_index].rect, r.rect) for r in ref.detections], default=0.0) if same else float("nan")
    rows.append((worst, name, len(got), same if [d.anchor_index for d in got] == [d.anchor_index for d in ref.detections] else "same set, order differs" if same else False))
rows.sort(key=lambda r: -(r[0] if r[0] == r[0] else 9))
print("lib:", retinaface_amd.lib_path())
for w, n, k, same in rows[:12]: print(f"  {w:.3e}  {n}  faces {k} anchors_same {same}")
ws = np.array([r[0] for r in rows]); print(f"cases {len(rows)}  worst {np.nanmax(ws):.3e}  mean {np.nanmean(ws):.3e}  >1e-3: {(ws > 1e-3).sum()}  nan: {np.isnan(ws).sum()}")
