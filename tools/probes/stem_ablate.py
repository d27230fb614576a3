import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, retinaface_amd
from retinaface_amd.frames import synth_frames
frames = torch.from_numpy(np.stack(synth_frames(448, 448, 8, config=1))).cuda(); torch.cuda.synchronize()
ptrs = [frames[i % 8].data_ptr() for i in range(32)]
det = retinaface_amd.RetinaFace(os.path.join(ROOT, "assets"), "net3", 0.4, net_hw=(448, 448), model_stem="mnet25", lanes=1, use_graph=False)
p = det.profile(ptrs, iters=10)
print("ABLATE", os.environ.get("RF_STEM_ABLATE", "0"), " ".join(f"{x['kernel']}={x['ms']*1e3:.1f}" for x in p[:2]))
