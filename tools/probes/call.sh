cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python tools/probes/iou_margin.py 6 2>&1 | grep -v "compute time\|amdgpu.ids" | tail -15
python - <<'P' 2>&1 | grep -v "compute time\|amdgpu.ids"
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, retinaface_amd
from oracle.caffe_io import read_rfw
from oracle.pipeline import OracleDetector
from retinaface_amd.frames import synth_frames
f = synth_frames(448, 448, 6, config=3)[2]
o = OracleDetector(read_rfw("assets/mnet-deconv-0517.rfw")).detect(f, 0.5, 0.4, net_hw=(448, 448))
for prec in (0, 1):
    d = retinaface_amd.RetinaFace("assets", "net3", 0.4, precision=prec, net_hw=(448, 448), model_stem="mnet-deconv-0517")
    g = d.detect(f, 0.5)
    print("prec", prec, [(x.anchor_index, round(x.score, 4), [round(v, 1) for v in x.rect]) for x in g])
print("oracle", [(x.anchor_index, round(x.score, 4), [round(v, 1) for v in x.rect]) for x in o.detections])
lo = OracleDetector(read_rfw("assets/mnet-deconv-0517.rfw")).detect(f, 0.45, 0.4, net_hw=(448, 448))
print("oracle thr .45", [(x.anchor_index, round(x.score, 4)) for x in lo.detections])
P
