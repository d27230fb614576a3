"""Probe: host-side cost of ONE synchronous rf_detect_batch_device call (batch 8, 448 x 448, fp16), timed at the C ABI with
prebuilt argument arrays; with RF_HOST_TRACE=1 the engine prints where the host time goes when it is destroyed.
usage: [RF_HOST_TRACE=1] python tools/probes/sync_latency.py [batch] [use_graph 0/1]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import retinaface_amd
from retinaface_amd._lib import rf_face
from retinaface_amd.frames import synth_frames
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
graph = (int(sys.argv[2]) if len(sys.argv) > 2 else 1) != 0
det = retinaface_amd.RetinaFace(os.path.join(ROOT, "assets"), "net3", 0.4, precision=1, net_hw=(448, 448), max_batch=B, model_stem="mnet25", use_graph=graph)
d = torch.from_numpy(np.stack(synth_frames(448, 448, B, config=1))).cuda(); torch.cuda.synchronize()
pa = (C.c_void_p * B)(*[d[i].data_ptr() for i in range(B)]); ra = (C.c_int * B)(*[448] * B); sa = (C.c_int * B)(*[448 * 3] * B)
out, cnt = (rf_face * (B * 256))(), (C.c_int * B)()
lat = []
for it in range(3000):
    t = time.perf_counter()
    det._lib.rf_detect_batch_device(det._h, pa, ra, ra, sa, B, C.c_float(0.5), out, 256, cnt)
    lat.append(time.perf_counter() - t)
lat = np.array(lat[200:]) * 1e6
print(f"batch {B} graph {graph}: sync call median {np.median(lat):.1f} us  p10 {np.quantile(lat, .1):.1f}  p90 {np.quantile(lat, .9):.1f}  faces {list(cnt)}", flush=True)
det.close()
