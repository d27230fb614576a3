"""Probe: calibrate with tools/calibrate_int8.py (given rule), attach the table to a copy of the .rfw in /tmp and measure the
int8 engine's parity against the golden fp32-oracle detections.  usage: int8_recal_eval.py MODEL RULE [FRAMES]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle.caffe_io import read_int8_table, read_rfw, write_rfw      # probe only
model, rule = sys.argv[1], sys.argv[2]
nframes = sys.argv[3] if len(sys.argv) > 3 else "48"
tmp = f"/tmp/recal_{model}_{rule}"; os.makedirs(tmp, exist_ok=True)
table = os.path.join(tmp, model + ".table.int8")
cmp_ = ["--compare", os.path.join(ROOT, "assets", "mnet-deconv-0517.table.int8")] if model == "mnet-deconv-0517" else []
out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "calibrate_int8.py"), "--model", model, "--rule", rule, "--frames", nframes,
                      "--out", table] + cmp_, capture_output=True, text=True).stdout
print("\n".join(l for l in out.splitlines() if l.strip().startswith(rule + " ") or "rule vs" in l))
net = read_rfw(os.path.join(ROOT, "assets", model + ".rfw"))
net.int8_scales = read_int8_table(table)
write_rfw(net, os.path.join(tmp, model + ".rfw"))
import retinaface_amd
from oracle.retinaface_post import iou_plus1
from retinaface_amd.frames import synth_frames, padded_base_frame
det = retinaface_amd.RetinaFace(tmp, "net3", 0.4, precision=2, net_hw=(448, 448), model_stem=model)
worst, ds, same = 1.0, 0.0, True
for cfg in (1, 5, 9):
    from oracle.pipeline import OracleDetector
    if cfg == 1:
        g = np.load(os.path.join(ROOT, "tests", "golden", f"synth448_{model}.npz")); refs = [g[f"det05_{i}"] for i in range(8)]
    else:
        od = OracleDetector(read_rfw(os.path.join(ROOT, "assets", model + ".rfw")))
        refs = [od.detect(f, 0.5, 0.4, net_hw=(448, 448)).rows() for f in synth_frames(448, 448, 8, config=cfg)]
    res = det.detectBatchImages(synth_frames(448, 448, 8, config=cfg), 0.5)
    for i in range(8):
        same &= len(res[i]) == len(refs[i])
        for r in refs[i]:
            best = max((iou_plus1(a.rect, r[1:5]), -abs(a.score - r[0])) for a in res[i]) if res[i] else (0, 0)
            worst = min(worst, best[0]); ds = max(ds, -best[1])
big = retinaface_amd.RetinaFace(tmp, "net3", 0.4, precision=2, net_hw=(896, 1280), max_batch=2, model_stem=model)
got = big.detect(padded_base_frame(), 0.5)
ref = np.load(os.path.join(ROOT, "tests", "golden", f"fixture_{model}.npz"))["det"]
wf = min(max(iou_plus1(a.rect, r[1:5]) for a in got) for r in ref)
print(f"{model} rule {rule}: 24 synth frames same-count {same} worst IoU {worst:.4f} max |dscore| {ds:.4f}; fixture {len(got)} faces worst IoU {wf:.4f}")
