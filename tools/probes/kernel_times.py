"""Probe: per-launch HIP-event timings at 8 and 32 images per launch (fp16), one line per kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, retinaface_amd
from retinaface_amd.frames import synth_frames
prec = {"fp16": 1, "int8": 2, "fp32": 0}[sys.argv[1] if len(sys.argv) > 1 else "fp16"]
frames = torch.from_numpy(np.stack(synth_frames(448, 448, 8, config=1))).cuda(); torch.cuda.synchronize()
det = retinaface_amd.RetinaFace(os.path.join(ROOT, "assets"), "net3", 0.4, precision=prec, net_hw=(448, 448), model_stem="mnet-deconv-0517", lanes=1, use_graph=False)
p32 = det.profile([frames[i % 8].data_ptr() for i in range(32)], iters=20)
p8 = det.profile([frames[i].data_ptr() for i in range(8)], iters=20)
for a, b in zip(p32, p8):
    print(f"{a['kernel']:26s} n32 {a['ms']*1e3:7.2f} us   n8 {b['ms']*1e3:7.2f} us")
print(f"TOTAL n32 {sum(a['ms'] for a in p32)*1e3:.1f} us   n8 {sum(b['ms'] for b in p8)*1e3:.1f} us")
