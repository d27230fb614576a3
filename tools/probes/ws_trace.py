"""Probe: phase stamps of the warp-specialised aggregation conv (conv3x3_up_ws_kernel, trace build): consumer wave 0 (slots 0-4: interval start,
blend done, past barrier M, GEMM + epilogue done, past barrier E) and the producer wave (slots 5-10: interval start, past barrier M, stores issued,
DMA issued, counted wait done, past barrier E), last tile of every workgroup.  usage: RF_CONV3UPWS=3 python tools/probes/ws_trace.py [images]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["RETINAFACE_AMD_LIB"] = os.path.join(ROOT, "retinaface_amd", "lib", "libretinaface_amd_trace.so")
import numpy as np, torch, retinaface_amd
from retinaface_amd.frames import synth_frames
lib = retinaface_amd.load_library()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = 8
frames = torch.from_numpy(np.stack(synth_frames(448, 448, B, config=1))).cuda(); torch.cuda.synchronize()
ptrs = [frames[i % B].data_ptr() for i in range(B)]
det = retinaface_amd.RetinaFace(os.path.join(ROOT, "assets"), "net3", 0.4, net_hw=(448, 448), model_stem="mnet25", lanes=1, max_batch=B, coalesce=n // B, use_graph=False)
lib.rf_trace_select.argtypes = [C.c_int, C.c_uint]; lib.rf_trace_read.argtypes = [C.c_void_p, C.c_int]
def run():
    t = [det.enqueue_device(ptrs, [448] * B, [448] * B, 0.5) for _ in range(n // B)]
    for x in t: det.wait(x, B)
for _ in range(3): run()
lib.rf_trace_select(6, 0); torch.cuda.synchronize()
run()
NB, NS, GHZ = 8192, 12, 2.4
buf = np.zeros(NB * NS, dtype=np.uint64)
lib.rf_trace_read(buf.ctypes.data, NB)
tr = buf.reshape(NB, NS).astype(np.int64)
tr = tr[(tr[:, 0] > 0) & (tr[:, 5] > 0)]
print(f"{len(tr)} workgroups with stamps (the launch that ran last: rf_c1_aggr; all times ns at a nominal {GHZ} GHz, last tile of each workgroup)")
def d(a, b, label):
    x = (tr[:, b] - tr[:, a]) / GHZ
    x = x[(x > -1e6) & (x < 1e6)]
    print(f"  {label:58s} mean {x.mean():8.0f}  p10 {np.percentile(x, 10):8.0f}  p90 {np.percentile(x, 90):8.0f}")
d(0, 1, "consumer: blend pass"); d(1, 2, "consumer: wait at barrier M"); d(2, 3, "consumer: GEMM + epilogue"); d(3, 4, "consumer: wait at barrier E")
d(0, 4, "consumer: whole interval")
d(5, 6, "producer: wait at barrier M"); d(6, 7, "producer: previous tile's stores (LDS reads + issue)"); d(7, 8, "producer: DMA issue (halo + coarse patch)")
d(8, 9, "producer: counted vmcnt wait"); d(9, 10, "producer: wait at barrier E"); d(5, 10, "producer: whole interval")
d(0, 5, "offset producer interval start - consumer interval start")
