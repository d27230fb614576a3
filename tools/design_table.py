#!/usr/bin/env python3
"""Print DESIGN.md section 4's kernel table from a bench run's gpurun_out/bench_kernels.json (per-launch HIP-event times + the counter
passes of the same run: HBM bytes, VALU / MFMA busy fractions).  usage: design_table.py kernels_and_counters.json"""
import json
import sys

j = json.load(open(sys.argv[1]))
n = j["per_launch_images"]
cnt = {}
for e in j.get("counters") or []:
    cnt.setdefault(e["kernel"], []).append(e)
seen = {}
print(f"| kernel instance | reference layers | µs / {n} img | alg GB/s (layer-wise credit) | HBM MB measured | HBM TB/s measured (of 8) | VALU-active | MFMA-busy |")
print("|---|---|---|---|---|---|---|---|")
tot_ms = tot_hbm = 0.0
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
    tot_ms += k["ms"]
    tot_hbm += hbm or 0
    layers = k["name"].replace("mobilenet0_", "").replace("_fwd", "")
    if len(layers) > 90:
        layers = layers[:87] + "..."
    layers = layers.replace(" | ", " ; ")
    f2 = lambda v: "-" if v is None else f"{v:.2f}"      # noqa: E731
    print(f"| `{name}` | {layers} | {k['ms'] * 1e3:.1f} | {k['alg_bytes'] / k['ms'] / 1e6:.0f} | {hbm / 1e6:.0f} | {hbm / k['ms'] / 1e9:.2f} ({hbm / k['ms'] / 1e9 / 8:.2f}) | {f2(valu)} | {f2(mfma)} |"
          if hbm else f"| `{name}` | {layers} | {k['ms'] * 1e3:.1f} | - | - | - | - | - |")
print(f"\nsum {tot_ms * 1e3:.1f} us per {n} images; HBM {tot_hbm / 1e6:.0f} MB = {tot_hbm / n / 1e6:.2f} MB per image; {tot_hbm / tot_ms / 1e9:.2f} TB/s inside the kernels")
