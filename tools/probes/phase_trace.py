"""Probe: phase timeline inside the dwpw / conv3x3 workgroups (needs `make -C retinaface_amd/csrc trace`).
usage: phase_trace.py {dwpw|conv3|stem2|dwpw2} IMAGES KEY [KEY...]   (KEY = tile count of the launch to trace; conv3: tiles of its first
level + output channels, e.g. 1024+48 for the merged SSH 64->48 conv at 256 x 448^2; 0 = any launch of the family)
optional env: RF_TRACE_HW="H W" (net size, default 448 448), RF_TRACE_BATCH (frames per enqueue, default 8)
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
family = sys.argv[1]; n = int(sys.argv[2]); grids = [int(g) for g in sys.argv[3:]]   # GRID = workgroups of the launch (persistent grid size)
H, W = [int(v) for v in os.environ.get("RF_TRACE_HW", "448 448").split()]
B = int(os.environ.get("RF_TRACE_BATCH", "8"))
frames = torch.from_numpy(np.stack(synth_frames(H, W, B, config=1))).cuda(); torch.cuda.synchronize()
ptrs = [frames[i % B].data_ptr() for i in range(B)]
det = retinaface_amd.RetinaFace(os.path.join(ROOT, "assets"), "net3", 0.4, net_hw=(H, W), model_stem="mnet25", lanes=1,
                                max_batch=B, coalesce=n // B, use_graph=False)
kid = {"dwpw": 2, "conv3": 3, "stem2": 4, "dwpw2": 5}[family]
lib.rf_trace_select.argtypes = [C.c_int, C.c_uint]; lib.rf_trace_read.argtypes = [C.c_void_p, C.c_int]
def run():
    tickets = [det.enqueue_device(ptrs, [H] * B, [W] * B, 0.5) for _ in range(n // B)]
    for t in tickets: det.wait(t, B)
for _ in range(3): run()
NB, NS, GHZ = 8192, 12, 2.4
# stamp slots, in program order inside the LAST tile a workgroup walked (slot 0 = workgroup start, 7 = after the loop / lateral)
ORDER = {"dwpw": [8, 9, 10, 1, 2, 3, 4, 5, 6, 7], "conv3": [8, 9, 10, 1, 2, 3, 5, 6], "stem2": [0, 1, 2, 3, 4, 5, 6, 7], "dwpw2": [0, 1, 2, 3, 4, 5, 6]}
LABEL = {8: "tile loop top", 9: "prefetch landed + staged to LDS", 10: "next tile's loads issued", 1: "previous tile's stores issued",
         2: "barrier", 3: "stencil (dwpw) / GEMM (conv3) done", 4: "barrier", 5: "GEMM + epilogue -> LDS", 6: "barrier", 7: "lateral / end"}
for grid in grids:
    lib.rf_trace_select(kid, grid); torch.cuda.synchronize()
    run()
    buf = np.zeros(NB * NS, dtype=np.uint64)
    lib.rf_trace_read(buf.ctypes.data, NB)
    tr = buf.reshape(NB, NS).astype(np.int64)
    nz = np.nonzero(tr[:, 0] > 0)[0]
    if len(nz):     # workgroup lifetime by block-index decile (multi-level conv3x3 launches: the levels own consecutive block ranges)
        lifes = (tr[nz][:, ORDER[family][-1]] - tr[nz][:, 0]) / GHZ / 1e3
        start = (tr[nz][:, 0] - tr[nz][:, 0].min()) / GHZ / 1e3
        edges = np.linspace(0, len(nz), 11).astype(int)
        print("  blocks: lifetime us by decile of block index " + " ".join(f"{lifes[a:b].mean():.1f}" for a, b in zip(edges[:-1], edges[1:])) +
              f" | first start {start.min():.1f} last start {start.max():.1f} last end {(start + lifes).max():.1f} us")
    tr = tr[tr[:, 0] > 0]
    if not len(tr):
        print(f"{family} grid {grid}: no stamps"); continue
    order = ORDER[family]
    if family == "dwpw2":
        LABEL.update({1: "1 halo -> LDS", 2: "2 depthwise A", 3: "3 pointwise A", 4: "4 depthwise B", 5: "5 pointwise B", 6: "6 store"})
    if family == "stem2":
        LABEL.update({1: "1 stage patch (arrival at barrier)", 2: "2 conv0 MFMA", 3: "3 depthwise conv1", 4: "4 pointwise conv2 MFMA",
                      5: "5 depthwise conv3 s2", 6: "6 pointwise conv4 MFMA", 7: "7 store"})
    seq = tr[:, order]
    okm = (np.diff(seq, axis=1) >= 0).all(axis=1)
    seq = seq[okm]
    life = (tr[okm][:, order[-1]] - tr[okm][:, 0]) / GHZ
    print(f"{family} grid {grid}: {len(seq)} workgroups; workgroup lifetime mean {life.mean() / 1e3:.2f} us (p90 {np.percentile(life, 90) / 1e3:.2f})")
    d = np.diff(seq, axis=1) / GHZ
    for i in range(d.shape[1]):
        print(f"  -> {LABEL[order[i + 1]]:36s} mean {d[:, i].mean():7.0f} ns   p10 {np.percentile(d[:, i], 10):7.0f}   p90 {np.percentile(d[:, i], 90):7.0f}")
    print(f"  last tile total {d.sum(axis=1).mean():7.0f} ns")
