"""Probe: host-frame pipeline rate (rf_enqueue_batch, pageable caller memory) vs the number of staging threads."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import retinaface_amd
from retinaface_amd.frames import synth_frames
B, H, W = 8, 448, 448
frames = synth_frames(H, W, 64, config=1)
host = np.stack(frames).reshape(8, B, H, W, 3)
for th in [int(x) for x in sys.argv[1:]] or [1, 4, 8, 12, 24, 48]:
    det = retinaface_amd.RetinaFace(os.path.join(ROOT, "assets"), "net3", 0.4, net_hw=(H, W), model_stem="mnet25", copy_threads=th)
    ring = [det.prepare_host_batch([host[k, i] for i in range(B)]) for k in range(8)]
    slots = det.num_slots()
    def run(n):
        inflight = []
        for s in range(n):
            if len(inflight) == slots:
                det.wait_counts(inflight.pop(0), B)
            inflight.append(det.enqueue_prepared_host(ring[s % 8], 0.5))
        while inflight:
            det.wait_counts(inflight.pop(0), B)
    run(2 * slots)
    t = time.perf_counter(); n = 40 * slots; run(n); dt = time.perf_counter() - t
    print(f"copy_threads {th:3d}: {n * B / dt / 1e3:7.1f} k img/s  {n * B * H * W * 3 / dt / 1e9:6.1f} GB/s", flush=True)
    det.close()
