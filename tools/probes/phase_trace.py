"""Probe: phase timeline inside the dwpw / conv3x3 workgroups (needs `make -C retinaface_amd/csrc trace`).
usage: phase_trace.py {dwpw|conv3} IMAGES GRID [GRID...]
Runs eager passes at IMAGES per launch; for each GRID (workgroups of the launch to trace: picks one launch of the kernel
family) prints, per phase boundary, the mean time since the workgroup's first stamp (s_memtime = shader cycles, shown at
a nominal 2.4 GHz), the launch span and when workgroups started (rounds show up as steps)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["RETINAFACE_AMD_LIB"] = os.path.join(ROOT, "retinaface_amd", "lib", "libretinaface_amd_trace.so")
import numpy as np, torch, retinaface_amd
from retinaface_amd.frames import synth_frames
lib = retinaface_amd.load_library()
family = sys.argv[1]; n = int(sys.argv[2]); grids = [int(g) for g in sys.argv[3:]]
frames = torch.from_numpy(np.stack(synth_frames(448, 448, 8, config=1))).cuda(); torch.cuda.synchronize()
ptrs = [frames[i % 8].data_ptr() for i in range(8)]
det = retinaface_amd.RetinaFace(os.path.join(ROOT, "assets"), "net3", 0.4, net_hw=(448, 448), model_stem="mnet25", lanes=1,
                                coalesce=n // 8, use_graph=False)
kid = {"dwpw": 2, "conv3": 3}[family]
lib.rf_trace_select.argtypes = [C.c_int, C.c_uint]; lib.rf_trace_read.argtypes = [C.c_void_p, C.c_int]
def run():
    tickets = [det.enqueue_device(ptrs, [448] * 8, [448] * 8, 0.5) for _ in range(n // 8)]
    for t in tickets: det.wait(t, 8)
for _ in range(3): run()
NB, NS, GHZ = 8192, 12, 2.4
names = ["start", "loads issued+LDS written", "barrier 1", "stencil/GEMM done", "barrier 2", "epilogue->LDS", "barrier 3", "stored"]
for grid in grids:
    lib.rf_trace_select(kid, grid); torch.cuda.synchronize()
    run()
    buf = np.zeros(NB * NS, dtype=np.uint64)
    lib.rf_trace_read(buf.ctypes.data, NB)
    tr = buf.reshape(NB, NS).astype(np.int64)
    tr = tr[tr[:, 0] > 0]
    if not len(tr):
        print(f"{family} grid {grid}: no stamps"); continue
    tr = tr[(np.diff(tr[:, :8], axis=1) >= 0).all(axis=1)]
    rel = (tr[:, :8] - tr[:, :1]) / GHZ
    span = (tr[:, 7].max() - tr[:, 0].min()) / GHZ
    print(f"{family} grid {grid}: {len(tr)} workgroups traced (first {NB} of the launch); span {span / 1e3:.2f} us")
    for i, nme in enumerate(names):
        print(f"  {nme:28s} mean {rel[:, i].mean():7.0f} ns   p10 {np.percentile(rel[:, i], 10):7.0f}   p90 {np.percentile(rel[:, i], 90):7.0f}")
    starts = (tr[:, 0] - tr[:, 0].min()) / GHZ
    print("  workgroup start deciles (ns):", " ".join(f"{np.percentile(starts, p):.0f}" for p in range(0, 101, 10)))
