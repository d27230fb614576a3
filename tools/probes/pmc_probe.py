"""Workload for rocprofv3 --pmc passes: eager launches (PMC collection faults under hipGraph replay) of the fp16 mnet25
engine at 448x448; argv[1] = images per launch (8 = one uncoalesced batch, 128 = the bench's coalesced launch)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import retinaface_amd
from retinaface_amd.frames import synth_frames
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
frames = synth_frames(448, 448, 8, config=1)
det = retinaface_amd.RetinaFace(os.path.join(ROOT, "assets"), "net3", 0.4, net_hw=(448, 448), model_stem="mnet25", lanes=1,
                                coalesce=max(n // 8, 1), use_graph=False)
d = torch.from_numpy(np.stack(frames)).cuda(); torch.cuda.synchronize()
ptrs = [d[i].data_ptr() for i in range(8)]
for it in range(4):
    tickets = [det.enqueue_device(ptrs, [448] * 8, [448] * 8, 0.5) for _ in range(max(n // 8, 1))]
    r = [det.wait(t, 8) for t in tickets]
print("launches of", n, "images ok", [len(x) for x in r[0]], flush=True)
