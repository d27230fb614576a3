#!/usr/bin/env python3
"""Cross-compile retinaface_amd/csrc/kernels.hip for gfx950 (device only, no GPU needed) and print every kernel's register / spill /
occupancy figures from -Rpass-analysis=kernel-resource-usage; optionally keep the ISA (--asm out.s) for reading a loop's schedule.
usage: python tools/kernel_resources.py [--asm /tmp/kernels.s] [--filter substring]"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm", default="/tmp/rf_kernels.s")
    ap.add_argument("--filter", default="")
    a = ap.parse_args()
    src = os.path.join(ROOT, "retinaface_amd", "csrc", "kernels.hip")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-x", "hip", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.dirname(src), "--cuda-device-only", "-mllvm", "--amdgpu-mfma-vgpr-form", "-S", src, "-o", a.asm, "-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr[-3000:])
    cur, rows = None, {}
    for l in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", l)
        if m:
            cur = m.group(1)
            rows[cur] = {}
        for k in ("VGPRs:", "AGPRs:", "VGPRs Spill:", "SGPRs Spill:", "TotalSGPRs:", "Occupancy [waves/SIMD]:", "ScratchSize [bytes/lane]:"):
            if cur and k in l:
                rows[cur][k] = l.split(k)[1].split()[0]
    names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
    for k, d in zip(rows, names):
        d = d.replace("rf::", "").replace("_Float16", "f16").replace("signed char", "i8")
        d = re.sub(r"\(.*\)$", "", d)
        if a.filter and a.filter not in d:
            continue
        v = rows[k]
        print(f"{d[:78]:78s} vgpr {v.get('VGPRs:'):>4} spill {v.get('VGPRs Spill:'):>3} sgpr {v.get('TotalSGPRs:'):>4} occ {v.get('Occupancy [waves/SIMD]:')} "
              f"scratch {v.get('ScratchSize [bytes/lane]:')}")


if __name__ == "__main__":
    main()
