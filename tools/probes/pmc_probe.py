import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
print("torch ok", flush=True)
import retinaface_amd
from retinaface_amd.frames import synth_frames
frames = synth_frames(448, 448, 8, config=1)
det = retinaface_amd.RetinaFace(os.path.join(ROOT, "assets"), "net3", 0.4, net_hw=(448, 448), model_stem="mnet25", lanes=1, use_graph=False)
print("engine ok", flush=True)
r = det.detectBatchImages(frames, 0.5)
print("host-frame detect ok", [len(x) for x in r], flush=True)
d = torch.from_numpy(np.stack(frames)).cuda(); torch.cuda.synchronize()
for i in range(5):
    r = det.detect_device([d[i].data_ptr() for i in range(8)], [448]*8, [448]*8, 0.5)
print("device-frame detect ok", [len(x) for x in r], flush=True)
