#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (what `rocprofv3 --kernel-trace --stats` writes on ROCm 7.2) into
the text table committed under profiles/: per kernel (name + grid) calls, average / min / max duration, share."""
import sqlite3
import sys


def main(db_path, out_path=None):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, grid_x, workgroup_x, lds_size, vgpr_count, count(*), avg(duration), min(duration), "
                      "max(duration), sum(duration) from kernels group by name, grid_x order by sum(duration) desc").fetchall()
    total = sum(r[9] for r in rows) or 1
    lines = [f"# source: {db_path}", f"# total kernel time {total / 1e6:.3f} ms over {sum(r[5] for r in rows)} dispatches",
             f"{'kernel':110s} {'grid':>8s} {'wg':>5s} {'lds':>7s} {'vgpr':>5s} {'calls':>6s} {'avg_us':>8s} {'min_us':>8s} {'max_us':>8s} {'share%':>7s}"]
    for name, gx, wx, lds, vg, n, avg, mn, mx, s in rows:
        short = name.replace("rf::", "")
        lines.append(f"{short[:110]:110s} {gx // max(wx, 1):8d} {wx:5d} {lds:7d} {vg:5d} {n:6d} {avg / 1e3:8.2f} {mn / 1e3:8.2f} {mx / 1e3:8.2f} {100 * s / total:7.2f}")
    text = "\n".join(lines) + "\n"
    if out_path:
        open(out_path, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
