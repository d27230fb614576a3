"""Workload for rocprofv3 --pmc passes: eager launches (PMC collection faults under hipGraph replay) of one engine configuration
on distinct frames.  usage: pmc_probe.py IMAGES_PER_LAUNCH [precision model H W batch]   (defaults: fp16 mnet25 448 448 8)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import retinaface_amd
from retinaface_amd.frames import synth_frames
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prec = {"fp16": 1, "fp32": 0, "int8": 2}[sys.argv[2] if len(sys.argv) > 2 else "fp16"]
model = sys.argv[3] if len(sys.argv) > 3 else "mnet25"
H, W = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (448, 448)
B = int(sys.argv[6]) if len(sys.argv) > 6 else 8
B = min(B, n)
frames = synth_frames(H, W, n, config=1)
det = retinaface_amd.RetinaFace(os.path.join(ROOT, "assets"), "net3", 0.4, precision=prec, net_hw=(H, W), max_batch=B, model_stem=model,
                                lanes=1, coalesce=max(n // B, 1), use_graph=False)
d = torch.from_numpy(np.stack(frames)).cuda(); torch.cuda.synchronize()
for it in range(4):
    tickets = [det.enqueue_device([d[k * B + i].data_ptr() for i in range(B)], [H] * B, [W] * B, 0.5) for k in range(max(n // B, 1))]
    r = [det.wait(t, B) for t in tickets]
print("launches of", n, "images ok", [len(x) for x in r[0]], flush=True)
