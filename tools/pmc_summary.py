#!/usr/bin/env python3
"""Combine the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- they do not fit one pass, MI355X_MICROARCH.md
"rocprofv3 PMC slots") into per-kernel HBM bytes per launch.  gfx950 correction from the same guide: FETCH_SIZE
reports exactly half the bytes of a wide coalesced read stream, so it is doubled; WRITE_SIZE is used as reported
(uncalibrated per the guide).  Usage: pmc_summary.py fetch.db write.db out.json [images_per_launch [workload_key, e.g. 448x448_fp16]]"""
import json
import re
import sqlite3
import sys


def descriptor(mangled: str) -> str:
    """Same kernel-instance string as rf_profile reports (engine.cpp OpInfo.kernel); template tails added in later rounds
    (padded-row / wave-split booleans) are ignored."""
    m = re.search(r"dwpw_kernelI(?:DF16_|f|a)Li(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELi(\d+)ELi(\d+)ELb(\d)E", mangled)
    if m:
        cin, cout, st, dw, th, tw, lat = m.groups()
        return f"dwpw<{cin},{cout},s{st}{',lat' if lat == '1' else ''}>"
    # K_b(8), the 8-wave weights-stationary form of the same op (round 5): dwpw_wide_kernel<T, CIN, COUT, TH, TW, LAT, PADROW, OCC>, always stride 1
    m = re.search(r"dwpw_wide_kernelI(?:DF16_|f|a)Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)E", mangled)
    if m:
        cin, cout, th, tw, lat = m.groups()
        return f"dwpw<{cin},{cout},s1{',lat' if lat == '1' else ''}>"
    m = re.search(r"dwpw_wide_kernel<[^,]+, (\d+), (\d+), (\d+), (\d+), (true|false)", mangled)
    if m:
        cin, cout, th, tw, lat = m.groups()
        return f"dwpw<{cin},{cout},s1{',lat' if lat == 'true' else ''}>"
    m = re.search(r"conv3x3_kernelI(?:DF16_|f|a)Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)E", mangled)
    if m:
        cin, cout, th, tw, up = m.groups()
        return f"conv3x3<{cin},{cout},{th}x{tw}{',up' if up == '1' else ''}>"
    # the same instances when the profiler reports them demangled ("void rf::dwpw_kernel<signed char, 128, 128, 1, true, 4, 8, false, true>(...)")
    m = re.search(r"dwpw_kernel<[^,]+, (\d+), (\d+), (\d+), (?:true|false), (\d+), (\d+), (true|false)", mangled)
    if m:
        cin, cout, st, th, tw, lat = m.groups()
        return f"dwpw<{cin},{cout},s{st}{',lat' if lat == 'true' else ''}>"
    m = re.search(r"conv3x3_kernel<[^,]+, (\d+), (\d+), (\d+), (\d+), (true|false)", mangled)
    if m:
        cin, cout, th, tw, up = m.groups()
        return f"conv3x3<{cin},{cout},{th}x{tw}{',up' if up == 'true' else ''}>"
    if "conv3x3_up_ws_kernel" in mangled or "conv3x3_up_dma_kernel" in mangled:    # the warp-specialised aggregation conv keeps the op name of the lock-step instance
        return "conv3x3<64,64,4x8,up>"
    if "conv3x3_ws_kernel" in mangled:       # the warp-specialised instance of the merged SSH conv (same op name as the lock-step one)
        return "conv3x3<64,48,8x8>"
    if "dwpw2_kernel" in mangled:
        return "dwpw2<32,32,64>"
    if "ssh_tail_kernel" in mangled:
        return "ssh_tail<16,32,16>"
    for k in ("stem2", "stem", "conv0", "head", "nms", "resize_area", "resize_bilinear"):
        if k + "_kernel" in mangled:
            return k
    return mangled


def dtype_of(symbol: str) -> str:
    """Element type a kernel instance was built for, from its (mangled or demangled) symbol: 'fp16' | 'int8' | 'fp32' | '' (untyped:
    nms, resize)."""
    if "DF16_" in symbol or "_Float16" in symbol:
        return "fp16"
    if re.search(r"_kernelIa|<signed char", symbol):
        return "int8"
    if re.search(r"_kernelIf|<float", symbol):
        return "fp32"
    if "stem2_kernel" in symbol or "dwpw2_kernel" in symbol:
        return "fp16"
    return ""


def per_kernel(db_path, counter):
    db = sqlite3.connect(db_path)
    rows = db.execute("select kernel_name, grid_size, count(*), avg(value) from counters_collection where counter_name=? "
                      "group by kernel_name, grid_size", (counter,)).fetchall()
    return {(r[0], r[1]): (r[2], r[3]) for r in rows}


def main(fetch_db, write_db, out, images_per_launch=8, workload_key="448x448_fp16"):
    f = per_kernel(fetch_db, "FETCH_SIZE")
    w = per_kernel(write_db, "WRITE_SIZE")
    res = []
    for key in sorted(set(f) | set(w)):
        name, grid = key
        if not name.startswith("_ZN2rf") and "rf::" not in name:
            continue
        fk = f.get(key, (0, 0.0))[1] * 1024.0
        wk = w.get(key, (0, 0.0))[1] * 1024.0
        res.append({"kernel": descriptor(name), "symbol": name, "grid_threads": grid, "launches_sampled": f.get(key, (0, 0))[0],
                    "fetch_size_bytes_raw": fk, "fetch_bytes_corrected_x2": 2 * fk, "write_size_bytes": wk,
                    "hbm_bytes_per_launch": 2 * fk + wk})
    # a kernel instance that runs several times per pass (the four plain 128-channel blocks) shows up once with an average:
    # weight it by how many times it ran per engine pass
    base = min((r["launches_sampled"] for r in res if r["launches_sampled"]), default=1)
    for r in res:
        r["launches_per_pass"] = max(1, round(r["launches_sampled"] / base)) if r["launches_sampled"] else 1
    total = sum(r["hbm_bytes_per_launch"] * r["launches_per_pass"] for r in res)
    json.dump({"note": f"per launch of {images_per_launch} images, {workload_key}, eager launches (tools/probes/pmc_probe.py; PMC collection "
                       "faults under hipGraph replay); FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM",
               "workload_key": workload_key, "images_per_launch": int(images_per_launch),
               "hbm_bytes_all_kernels_per_launch": total, "hbm_bytes_per_image": total / int(images_per_launch),
               "kernels": res}, open(out, "w"), indent=1)
    print(f"total {total / 1e6:.1f} MB per launch of {images_per_launch} images = {total / int(images_per_launch) / 1e6:.2f} MB per image")
    for r in res:
        print(f"{r['kernel'][:90]:90s} grid {r['grid_threads']:8d}  fetch*2 {r['fetch_bytes_corrected_x2'] / 1e6:8.3f} MB  "
              f"write {r['write_size_bytes'] / 1e6:8.3f} MB")


if __name__ == "__main__":
    main(*sys.argv[1:6])
