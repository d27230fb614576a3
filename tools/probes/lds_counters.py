#!/usr/bin/env python3
"""Probe: is the LDS data path what bounds the depthwise-pointwise blocks?  Summarises one rocprofv3 --pmc pass with LDS counters per kernel instance.
usage: lds_counters.py pmc.db COUNTER [COUNTER ...]   (run after:  rocprofv3 --pmc <counters> GRBM_GUI_ACTIVE -d out -o pmc -- python tools/probes/pmc_probe.py 256)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pmc_summary


def main(db, counters):
    per = {c: pmc_summary.per_kernel(db, c) for c in counters + ["GRBM_GUI_ACTIVE"]}
    keys = sorted(per["GRBM_GUI_ACTIVE"], key=lambda k: -per["GRBM_GUI_ACTIVE"][k][1])
    print("%-30s %8s %12s " % ("kernel", "grid", "gpu_cycles") + " ".join("%22s" % c for c in counters))
    for k in keys:
        g = per["GRBM_GUI_ACTIVE"][k]
        g = g[1]
        row = []
        for c in counters:
            v = per[c].get(k, (0, 0.0))[1]
            row.append("%12.0f (%6.3f)" % (v, v / max(g, 1) / 256.0 * 8))     # per CU and GPU-active cycle: GRBM_GUI_ACTIVE is summed over the 8 XCDs
        print("%-30s %8s %12.0f " % (pmc_summary.descriptor(k[0])[:30], k[1], g) + " ".join("%22s" % r for r in row))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
