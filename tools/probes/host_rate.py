"""Probe: host-frame pipeline rate (rf_enqueue_batch, batch 8, 448 x 448, fp16) with pageable and with registered caller memory.
usage: [RF_COPY_STREAMS=2] python tools/probes/host_rate.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import retinaface_amd
from retinaface_amd.frames import synth_frames
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
B, H, W = 8, 448, 448
det = retinaface_amd.RetinaFace(os.path.join(ROOT, "assets"), "net3", 0.4, precision=1, net_hw=(H, W), max_batch=B, model_stem="mnet25")
slots = det.num_slots()
nb = 2 * slots
host = np.stack(synth_frames(H, W, nb * B, config=1)).reshape(nb, B, H, W, 3)
def run(steps, ring):
    infl = []
    for s in range(steps):
        if len(infl) == slots: det.wait_counts(infl.pop(0), B)
        infl.append(det.enqueue_prepared_host(ring[s % len(ring)], 0.5))
    while infl: det.wait_counts(infl.pop(0), B)
for label, reg in (("pageable", False), ("registered", True)):
    buf = host.copy()
    if reg: det.host_register(buf)
    ring = [det.prepare_host_batch([buf[k, i] for i in range(B)]) for k in range(nb)]
    run(2 * slots, ring); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < secs:
        run(slots, ring); n += slots
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"RF_COPY_STREAMS={os.environ.get('RF_COPY_STREAMS', '1')} {label}: {n * B / dt:.0f} images/s  {n * B * H * W * 3 / dt / 1e9:.1f} GB/s", flush=True)
    if reg: det.host_unregister(buf)
det.close()
