#!/usr/bin/env python3
"""Dynamic instruction mix per kernel from ONE rocprofv3 --pmc pass of the SQ_INSTS_* counters over tools/probes/pmc_probe.py
("VALU instructions per wave", "an instruction-mix split for the dominant kernel").  SQ_INSTS_* count wave-instructions
issued, SQ_WAVES the waves launched: the quotient is instructions per wave, measured, loops and divergent branches included.
usage: pmc_insts.py pass.db [out.json]        (the counters present in the pass are reported; kernels of namespace rf only)"""
import json
import sqlite3
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from pmc_summary import descriptor, dtype_of          # noqa: E402


def main(db_path, out=None):
    db = sqlite3.connect(db_path)
    names = [r[0] for r in db.execute("select distinct counter_name from counters_collection").fetchall()]
    rows = db.execute("select kernel_name, grid_size, counter_name, count(*), avg(value) from counters_collection group by kernel_name, grid_size, counter_name").fetchall()
    per = {}
    for k, g, c, n, v in rows:
        if not (k.startswith("_ZN2rf") or "rf::" in k):
            continue
        per.setdefault((k, g), {})[c] = v
    res = []
    for (k, g), c in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
        waves = c.get("SQ_WAVES") or 0
        e = {"kernel": descriptor(k), "dtype": dtype_of(k), "grid_threads": g, "waves": waves}
        for n in names:
            if n != "SQ_WAVES" and waves:
                e[n.lower() + "_per_wave"] = c.get(n, 0.0) / waves
        res.append(e)
    for e in res[:24]:
        print(f"{e['kernel'][:28]:28s} {e['dtype']:5s} waves {e['waves']:9.0f}  " + "  ".join(f"{k[8:-9]} {v:7.1f}" for k, v in e.items() if k.endswith("_per_wave")))
    if out:
        json.dump({"counters": names, "note": "wave-instructions issued per wave launched (SQ_INSTS_* / SQ_WAVES), one eager launch sequence of 256 images", "kernels": res},
                  open(out, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:3])
