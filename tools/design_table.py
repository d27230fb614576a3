#!/usr/bin/env python3
"""Print DESIGN.md section 4's kernel table from a bench run's gpurun_out/bench_kernels.json (per-launch HIP-event times + the counter
passes of the same run: HBM bytes, VALU / MFMA busy fractions) and, optionally, its gpurun_out/bench_pipeline_trace.json (the same kernels'
average durations in the single-lane rocprofv3 kernel trace of the timed loop).
usage: design_table.py kernels_and_counters.json [pipeline_trace.json]"""
import json
import sys

HBM_PEAK, MFMA_PEAK = 8000.0, 2500.0      # GB/s, dense fp16 TFLOP/s (bench.py)
j = json.load(open(sys.argv[1]))
trace = json.load(open(sys.argv[2])) if len(sys.argv) > 2 else None
n = j["per_launch_images"]
cnt = {}
for e in j.get("counters") or []:
    cnt.setdefault(e["kernel"], []).append(e)
tr = {}
for g in (trace or {}).get("kernels", []):
    tr.setdefault(g["kernel"], []).append(g)
for v in tr.values():
    v.sort(key=lambda g: g["avg_ms"])          # two kernels under one descriptor (the aggregation convs): the smaller map first, as in launch order
seen = {}
print(f"| kernel instance | reference layers | µs / {n} img (HIP events) | µs in the pipeline | HBM MB measured | HBM TB/s measured (of 8) | useful HBM (compulsory bytes / time / 8 TB/s) | VALU-active | MFMA-busy | useful MFMA (layer MACs / time / 2.5 PFLOP/s) |")
print("|---|---|---|---|---|---|---|---|---|---|")
tot_ms = tot_hbm = tot_pipe = tot_cb = tot_macs = 0.0
for k in j["kernels"]:
    name = k["kernel"]
    idx = seen.get(name, 0)
    seen[name] = idx + 1
    es = sorted(cnt.get(name, []), key=lambda e: e["hbm_bytes"])
    e = None
    if es:
        # two launches of one instance (the aggregation convs): the smaller map comes first in launch order
        e = es[min(idx, len(es) - 1)] if len(es) > 1 and es[0]["launches_per_pass"] == 1 else es[-1]
    hbm = e["hbm_bytes"] if e else None
    valu = 4 * e["valu_quad"] / (e["gpu_cycles"] * 1024) if e and e.get("valu_quad") else None
    mfma = e["mfma_cycles"] / (e["gpu_cycles"] * 1024) if e and e.get("mfma_cycles") else None
    gs = tr.get(name, [])
    pipe = None
    if gs:
        g = gs[min(idx, len(gs) - 1)] if len(gs) > 1 and gs[0]["launches_per_sequence"] == 1 else gs[-1]
        pipe = g["avg_ms"]
    tot_ms += k["ms"]
    tot_hbm += hbm or 0
    tot_pipe += pipe or k["ms"]
    cb, macs = k.get("compulsory_bytes"), k["macs"]
    tot_cb += cb or 0
    tot_macs += macs
    layers = k["name"].replace("mobilenet0_", "").replace("_fwd", "")
    if len(layers) > 70:
        layers = layers[:67] + "..."
    layers = layers.replace(" | ", " ; ")
    f2 = lambda v: "-" if v is None else f"{v:.2f}"      # noqa: E731
    f3 = lambda v: "-" if v is None else f"{v:.3f}"      # noqa: E731
    ucb = cb / k["ms"] / 1e6 / HBM_PEAK if cb else None
    umf = 2 * macs / k["ms"] / 1e9 / MFMA_PEAK if macs else None
    ms = k["ms"]
    c_pipe = "-" if pipe is None else "%.1f" % (pipe * 1e3)
    c_mb = "-" if hbm is None else "%.0f" % (hbm / 1e6)
    c_tb = "-" if hbm is None else "%.2f (%.2f)" % (hbm / ms / 1e9, hbm / ms / 1e9 / 8)
    print("| `%s` | %s | %.1f | %s | %s | %s | %s | %s | %s | %s |" % (name, layers, ms * 1e3, c_pipe, c_mb, c_tb, f2(ucb), f2(valu), f2(mfma), f3(umf)))
print(f"\nsum {tot_ms * 1e3:.1f} us per {n} images by HIP events, {tot_pipe * 1e3:.1f} us in the single-lane pipeline trace; HBM {tot_hbm / 1e6:.0f} MB = {tot_hbm / n / 1e6:.2f} MB per "
      f"image measured ({tot_cb / n / 1e6:.2f} MB compulsory); {tot_hbm / tot_ms / 1e9:.2f} TB/s inside the kernels; useful over the path: HBM {tot_cb / tot_ms / 1e6 / HBM_PEAK:.2f}, "
      f"MFMA {2 * tot_macs / tot_ms / 1e9 / MFMA_PEAK:.3f}")
