#!/usr/bin/env python3
"""Probe: throughput and host-side enqueue cost vs lanes / branch-parallel graphs / eager (448x448, batch 8, fp16)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import retinaface_amd
from retinaface_amd.frames import synth_frames
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
frames = torch.from_numpy(np.stack(synth_frames(448, 448, B, config=1))).cuda()
torch.cuda.synchronize()
ptrs = [frames[i].data_ptr() for i in range(B)]
rows = cols = [448] * B
for lanes in [int(x) for x in os.environ.get("LANES", "1,2,3,4").split(",")]:
    for par in [int(x) for x in os.environ.get("COALESCE", "1,2,3,4").split(",")]:
        for graph in (True,):
            det = retinaface_amd.RetinaFace(os.path.join(ROOT, "assets"), "net3", 0.4, net_hw=(448, 448), max_batch=B,
                                            model_stem="mnet25", lanes=lanes, coalesce=par, use_graph=graph)
            def run(steps):
                infl = []; tenq = 0.0
                for _ in range(steps):
                    if len(infl) == det.num_slots():
                        det.wait_counts(infl.pop(0), B)
                    t = time.perf_counter(); infl.append(det.enqueue_device(ptrs, rows, cols, 0.5)); tenq += time.perf_counter() - t
                while infl: det.wait_counts(infl.pop(0), B)
                return tenq
            run(30)
            torch.cuda.synchronize(); t0 = time.perf_counter(); tenq = run(300); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print(f"lanes={lanes} coalesce={par!s:3} graph={graph!s:5}: {dt/300*1e3:.4f} ms/step  {300*B/dt:9.0f} img/s   host enqueue {tenq/300*1e6:.1f} us/step", flush=True)
            det.close()
