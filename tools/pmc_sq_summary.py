#!/usr/bin/env python3
"""Summarise rocprofv3 SQ counter passes (collected separately, eager launches: tools/probes/pmc_probe.py) into per-kernel
issue statistics: how busy each wave keeps the VALU / LDS / MFMA pipes and how long it waits.
Usage: pmc_sq_summary.py pass1.db pass2.db out.json images_per_launch
  pass 1: SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
  pass 2: SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles summed over waves (MI355X_MICROARCH.md), so their ratios are
per-wave fractions; multiply `valu_active_per_wave` by the resident waves per SIMD for the SIMD's VALU utilisation."""
import collections
import json
import sqlite3
import sys

from pmc_summary import descriptor


def main(db1, db2, out, n):
    d = collections.defaultdict(dict)
    for path in (db1, db2):
        db = sqlite3.connect(path)
        for k, g, c, v in db.execute("select kernel_name, grid_size, counter_name, avg(value) from counters_collection "
                                     "group by kernel_name, grid_size, counter_name"):
            if "rf" in k:
                d[(k, g)][c] = v
    res = []
    for (k, g), c in sorted(d.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        wc = c.get("SQ_WAVE_CYCLES") or 1.0
        waves = g / 64.0
        res.append({
            "kernel": descriptor(k), "workgroups": g // 256, "waves": waves,
            "valu_active_per_wave": c.get("SQ_ACTIVE_INST_VALU", 0) / wc,
            "lds_active_per_wave": c.get("SQ_ACTIVE_INST_LDS", 0) / wc,
            "wait_any_per_wave": c.get("SQ_WAIT_INST_ANY", 0) / wc,
            "wait_lds_per_wave": c.get("SQ_WAIT_INST_LDS", 0) / wc,
            # MFMA utilisation against the chip's matrix-pipe capacity while the kernel ran: busy cycles summed over every SIMD
            # (SQ_VALU_MFMA_BUSY_CYCLES counts cycles, MI355X_MICROARCH.md) / (cycles the GPU was active x 256 CUs x 4 SIMDs).
            # rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (checked: conv3x3<64,48> issues 1.78 M v_mfma_16x16x32 per
            # 128-image launch = 28.5 M busy cycles against 39.5 us x 1024 SIMDs = ~29 %; the counter ratio gives 27 % with the /8)
            "mfma_busy_frac_of_chip": (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)) if c.get("GRBM_GUI_ACTIVE") else None,
            "mfma_busy_cycles": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), "gpu_active_cycles": c.get("GRBM_GUI_ACTIVE"),
            "sq_busy_cycles": c.get("SQ_BUSY_CYCLES"),
            "valu_insts_per_wave": c.get("SQ_INSTS_VALU", 0) / waves, "salu_insts_per_wave": c.get("SQ_INSTS_SALU", 0) / waves,
            "lds_insts_per_wave": c.get("SQ_INSTS_LDS", 0) / waves,
            "vmem_rd_insts_per_wave": c.get("SQ_INSTS_VMEM_RD", 0) / waves, "vmem_wr_insts_per_wave": c.get("SQ_INSTS_VMEM_WR", 0) / waves,
            "lds_bank_conflict_over_lds_active": c.get("SQ_LDS_BANK_CONFLICT", 0) / (c.get("SQ_ACTIVE_INST_LDS") or 1.0),
        })
    json.dump({"note": f"per launch of {n} images, 448x448, fp16, eager launches; persistent kernels: one wave walks many tiles",
               "images_per_launch": int(n), "kernels": res}, open(out, "w"), indent=1)
    print(f"{'kernel':24s} {'wgs':>6s} {'VALU/w':>7s} {'LDS/w':>6s} {'wait/w':>6s} {'iVALU/w':>8s} {'iSALU/w':>8s} {'iLDS/w':>7s} {'bankcf':>6s} {'MFMA%':>6s}")
    for r in res:
        print(f"{r['kernel'][:24]:24s} {r['workgroups']:6d} {r['valu_active_per_wave']:7.3f} {r['lds_active_per_wave']:6.3f} "
              f"{r['wait_any_per_wave']:6.2f} {r['valu_insts_per_wave']:8.0f} {r['salu_insts_per_wave']:8.0f} {r['lds_insts_per_wave']:7.0f} "
              f"{r['lds_bank_conflict_over_lds_active']:6.2f} {100 * (r['mfma_busy_frac_of_chip'] or 0):6.2f}")


if __name__ == "__main__":
    main(*sys.argv[1:5])
