#!/usr/bin/env python3
"""Per-kernel HIP-event timing table of one engine configuration (rf_profile), for A/B runs of kernel variants selected by
environment knobs (RF_CONV3, RF_PERSIST_MIN_ROUNDS, ...).  Usage on the GPU box:
    python tools/kbench.py [--precision fp16|int8|fp32] [--model mnet25] [--hw 448 448] [--n 128] [--iters 30] [--tag name]
Writes gpurun_out/kbench_<tag>.json and prints the table."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="fp16")
    ap.add_argument("--model", default="mnet25")
    ap.add_argument("--hw", type=int, nargs=2, default=[448, 448])
    ap.add_argument("--n", type=int, default=128)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--tag", default="run")
    a = ap.parse_args()
    import numpy as np
    import torch
    import retinaface_amd
    from retinaface_amd.frames import synth_frames
    H, W = a.hw
    prec = {"fp16": 1, "fp32": 0, "int8": 2}[a.precision]
    coalesce = max(1, a.n // a.batch)
    det = retinaface_amd.RetinaFace(os.path.join(ROOT, "assets"), "net3", 0.4, precision=prec, net_hw=(H, W), max_batch=a.batch,
                                    model_stem=a.model, use_graph=False, lanes=1, coalesce=coalesce)
    nd = min(a.n, 64)
    frames = torch.from_numpy(np.stack(synth_frames(H, W, nd, config=1))).cuda()
    ptrs = [frames[i % nd].data_ptr() for i in range(a.n)]
    prof = det.profile(ptrs, iters=a.iters)
    tot = sum(p["ms"] for p in prof)
    print(f"== {a.tag}: {a.model} {a.precision} {W}x{H} n={a.n}  total {tot * 1e3:.1f} us  -> {a.n / tot / 1e-3 / 1e3:.1f} k img/s on one lane")
    for p in prof:
        print(f"  {p['kernel']:26s} {p['ms'] * 1e3:7.1f} us   alg {p['alg_bytes'] / p['ms'] / 1e6:7.0f} GB/s   {2 * p['macs'] / p['ms'] / 1e9:7.1f} TFLOP/s")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump({"tag": a.tag, "config": vars(a), "env": {k: v for k, v in os.environ.items() if k.startswith("RF_")},
               "total_us": tot * 1e3, "kernels": prof}, open(os.path.join(ROOT, "gpurun_out", f"kbench_{a.tag}.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
