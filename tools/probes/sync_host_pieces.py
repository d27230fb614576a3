#!/usr/bin/env python3
"""Probe (round 6): ONE synchronous rf_detect_batch of pageable host frames at the C ABI, for the piece count in RF_SYNC_PIECES
(engine.cpp submit(), pipelined staging).  usage: RF_SYNC_PIECES=n python tools/probes/sync_host_pieces.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import retinaface_amd                                                  # noqa: E402
from retinaface_amd._lib import rf_face                                 # noqa: E402
from retinaface_amd.frames import synth_frames                          # noqa: E402

out = []
for (H, W, B) in ((448, 448, 8), (448, 448, 32), (896, 1280, 1)):
    det = retinaface_amd.RetinaFace(os.path.join(ROOT, "assets"), "net3", 0.4, precision=1, net_hw=(H, W), max_batch=B, model_stem="mnet25")
    fr = np.stack(synth_frames(H, W, min(B, 8), config=5))
    host = np.ascontiguousarray(fr[np.arange(B) % len(fr)])
    pa = (C.c_void_p * B)(*[host[i].ctypes.data for i in range(B)])
    ra, ca, sa = (C.c_int * B)(*([H] * B)), (C.c_int * B)(*([W] * B)), (C.c_int * B)(*([3 * W] * B))
    ob, cnt = (rf_face * (B * det.max_detections))(), (C.c_int * B)()
    lat = []
    t0 = time.perf_counter()
    while len(lat) < 30 or time.perf_counter() - t0 < 0.4:
        t = time.perf_counter()
        det._lib.rf_detect_batch(det._h, pa, ra, ca, sa, B, C.c_float(0.5), ob, det.max_detections, cnt)
        lat.append(time.perf_counter() - t)
    out.append(f"{W}x{H} b{B}: {np.median(lat[len(lat) // 6:]) * 1e3:.4f} ms")
    det.close()
print(f"RF_SYNC_PIECES={os.environ.get('RF_SYNC_PIECES', 'default')}: " + "; ".join(out))
