"""Probe: pipelined throughput (device-resident ring, as bench.py times it) vs lanes x coalesce, one engine per setting."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import retinaface_amd
from retinaface_amd.frames import synth_frames
prec = {"fp16": 1, "int8": 2}[sys.argv[1]]; model = sys.argv[2]; B = int(sys.argv[3])
H = W = 448
nfr = 256
fr = synth_frames(H, W, 64, config=1)
frames = torch.from_numpy(np.stack([fr[i % 64] for i in range(nfr)])).cuda()
for lanes, co in [tuple(int(v) for v in a.split("x")) for a in sys.argv[4:]]:
    det = retinaface_amd.RetinaFace(os.path.join(ROOT, "assets"), "net3", 0.4, precision=prec, net_hw=(H, W), max_batch=B, model_stem=model,
                                    lanes=lanes, coalesce=co)
    ring = [det.prepare_device_batch([frames[(k * B + i) % nfr].data_ptr() for i in range(B)], [H] * B, [W] * B) for k in range(nfr // B)]
    slots = det.num_slots()
    def run(n):
        q = []
        for s in range(n):
            if len(q) == slots: det.wait_counts(q.pop(0), B)
            q.append(det.enqueue_prepared(ring[s % len(ring)], 0.5))
        while q: det.wait_counts(q.pop(0), B)
    run(4 * slots); torch.cuda.synchronize()
    n = max(slots * 20, 2000 // B * 8); t = time.perf_counter(); run(n); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"{sys.argv[1]} B={B} lanes {lanes} x coalesce {co:2d}: {n * B / dt / 1e3:7.1f} k img/s", flush=True)
    det.close()
