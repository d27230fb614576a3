"""Workload for rocprofv3 --pmc passes: eager launches (PMC collection faults under hipGraph replay) of one or several engine
configurations on distinct frames.
usage: pmc_probe.py IMAGES_PER_LAUNCH [precision model H W batch]   (defaults: fp16 mnet25 448 448 8)
       pmc_probe.py --multi '[{"n": 256, "precision": "int8", "model": "mnet25", "H": 448, "W": 448, "B": 32}, ...]'   (one process, the
       configurations one after the other: bench.py tells their kernels apart by element type and grid)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import retinaface_amd
from retinaface_amd.frames import synth_frames


def run(n, precision, model, H, W, B):
    prec = {"fp16": 1, "fp32": 0, "int8": 2}[precision]
    B = min(B, n)
    nd = min(n, 64 if H * W <= 512 * 512 else 8)            # distinct frames (cycled): frame generation is the slow part of a pass
    frames = synth_frames(H, W, nd, config=1)
    det = retinaface_amd.RetinaFace(os.path.join(ROOT, "assets"), "net3", 0.4, precision=prec, net_hw=(H, W), max_batch=B, model_stem=model,
                                    lanes=1, coalesce=max(n // B, 1), use_graph=False)
    d = torch.from_numpy(np.stack(frames)).cuda(); torch.cuda.synchronize()
    for it in range(4):
        tickets = [det.enqueue_device([d[(k * B + i) % nd].data_ptr() for i in range(B)], [H] * B, [W] * B, 0.5) for k in range(max(n // B, 1))]
        r = [det.wait(t, B) for t in tickets]
    print("launches of", n, "images ok", precision, model, H, W, [len(x) for x in r[0]], flush=True)
    det.close()


if len(sys.argv) > 2 and sys.argv[1] == "--multi":
    for c in json.loads(sys.argv[2]):
        run(int(c["n"]), c["precision"], c["model"], int(c["H"]), int(c["W"]), int(c["B"]))
else:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    precision = sys.argv[2] if len(sys.argv) > 2 else "fp16"
    model = sys.argv[3] if len(sys.argv) > 3 else "mnet25"
    H, W = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (448, 448)
    B = int(sys.argv[6]) if len(sys.argv) > 6 else 8
    run(n, precision, model, H, W, B)
