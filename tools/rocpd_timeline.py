#!/usr/bin/env python3
"""Timeline of ONE dense factorisation from a rocprofv3 rocpd SQLite database (dev tool).
usage: rocpd_timeline.py results.db [which=3] > timeline.csv
Selects the kernels between the `which`-th k_yty_semisep (Y^T Y update, runs right before the factorisation)
and the next k_bwd_step, and prints start (us, relative), duration (us), queue, short name, grid."""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    which = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
    sys.stderr.write("kernels columns: " + ",".join(cols) + "\n")
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    gcol = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else "0")
    rows = db.execute(f"select name, start, end, {qcol}, {gcol} from kernels order by start").fetchall()
    yty = [i for i, r in enumerate(rows) if "k_yty_semisep" in r[0]]
    i0 = yty[which]
    i1 = next(i for i in range(i0, len(rows)) if "k_bwd_step" in rows[i][0])
    t0 = rows[i0][2]
    print("start_us,dur_us,queue,name,grid")
    for name, st, en, q, g in rows[i0 + 1:i1 + 1]:
        short = re.sub(r"covgpu::|void |\(.*", "", name)
        print(f"{(st - t0) / 1e3:.1f},{(en - st) / 1e3:.1f},{q},{short},{g}")


if __name__ == "__main__":
    main()
