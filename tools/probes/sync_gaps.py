"""Probe: where the GPU time of ONE synchronous call goes -- kernel durations vs the gaps between dependent launches.
Step 1 (under rocprofv3 --kernel-trace): `python tools/probes/sync_gaps.py run [batch] [H W] [precision]` makes 300 synchronous
rf_detect_batch_device calls.  Step 2: `python tools/probes/sync_gaps.py report <rocpd .db>` reads the kernel dispatch records (start / end
timestamps) and prints, per call: sum of kernel durations, sum of the gaps between consecutive kernels of the call, first-kernel-start to
last-kernel-end, and the per-kernel average duration and the average gap BEFORE each kernel.  The gap is what a single persistent launch
(or fewer launches) could remove; the durations are what it could not."""
import os, sqlite3, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run(B, H, W, precision):
    import ctypes as C
    import numpy as np, torch
    import retinaface_amd
    from retinaface_amd._lib import rf_face
    from retinaface_amd.frames import synth_frames
    prec = {"fp16": 1, "int8": 2, "fp32": 0}[precision]
    det = retinaface_amd.RetinaFace(os.path.join(ROOT, "assets"), "net3", 0.4, precision=prec, net_hw=(H, W), max_batch=B, model_stem="mnet25")
    d = torch.from_numpy(np.stack(synth_frames(H, W, B, config=1))).cuda(); torch.cuda.synchronize()
    pa = (C.c_void_p * B)(*[d[i].data_ptr() for i in range(B)]); ra = (C.c_int * B)(*[H] * B); ca = (C.c_int * B)(*[W] * B); sa = (C.c_int * B)(*[W * 3] * B)
    out, cnt = (rf_face * (B * 256))(), (C.c_int * B)()
    for it in range(300):
        det._lib.rf_detect_batch_device(det._h, pa, ra, ca, sa, B, C.c_float(0.5), out, 256, cnt)
    print("300 synchronous calls ok", B, H, W, precision, list(cnt), flush=True)
    det.close()


def report(db_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_summary
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    rows = [(pmc_summary.descriptor(n), s, e) for n, s, e in rows if "rf::" in n or n.startswith("_ZN2rf")]
    # a call = a run of kernels ending with nms
    calls, cur = [], []
    for r in rows:
        cur.append(r)
        if r[0] == "nms":
            calls.append(cur); cur = []
    calls = [c for c in calls[50:] if len(c) == len(calls[-1])]          # steady state, graph replays only
    n = len(calls[0])
    dur = [0.0] * n; gap = [0.0] * n
    tot_d = tot_g = tot_span = 0.0
    for c in calls:
        for k, (name, s, e) in enumerate(c):
            dur[k] += (e - s) / 1e3
            if k: gap[k] += (s - c[k - 1][2]) / 1e3
        tot_d += sum(e - s for _, s, e in c) / 1e3
        tot_g += sum(c[k][1] - c[k - 1][2] for k in range(1, n)) / 1e3
        tot_span += (c[-1][2] - c[0][1]) / 1e3
    m = len(calls)
    print(f"# {db_path}: {m} synchronous calls of {n} launches each (steady state)")
    print(f"per call: first kernel start -> last kernel end {tot_span / m:7.1f} us = kernels {tot_d / m:7.1f} us + gaps between dependent launches {tot_g / m:7.1f} us "
          f"({tot_g / m / (n - 1):.2f} us per launch boundary)")
    print(f"{'launch':28s} {'avg us':>8s} {'gap before us':>14s}")
    for k in range(n):
        print(f"{calls[0][k][0]:28s} {dur[k] / m:8.2f} {gap[k] / m:14.2f}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
        H, W = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (448, 448)
        run(B, H, W, sys.argv[5] if len(sys.argv) > 5 else "fp16")
    else:
        report(sys.argv[2])
