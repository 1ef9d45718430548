#!/usr/bin/env python3
"""Timeline of ONE trust-region iteration from a rocprofv3 rocpd SQLite database (dev tool).
usage: rocpd_iter_timeline.py results.db [which=12] > timeline.csv
Prints every kernel between the `which`-th and the next k_lm_lin launch: start (us, relative), duration (us), queue, short
name, grid. Structure only: the profiler inflates dispatch gaps."""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    which = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    gcol = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else "0")
    rows = db.execute(f"select name, start, end, {qcol}, {gcol} from kernels order by start").fetchall()
    lin = [i for i, r in enumerate(rows) if "k_lm_lin" in r[0]]
    i0, i1 = lin[which], lin[which + 1]
    # back up to the first kernel of the build (zero fills precede the linearisation)
    while i0 > 0 and rows[i0][1] - rows[i0 - 1][2] < 30000 and i0 > lin[which - 1] + 1 and "k_accept" not in rows[i0 - 1][0]:
        i0 -= 1
    t0 = rows[i0][1]
    print("start_us,dur_us,queue,name,grid")
    for name, st, en, q, g in rows[i0:i1]:
        short = re.sub(r"covgpu::|void |\(.*", "", name)
        print(f"{(st - t0) / 1e3:.1f},{(en - st) / 1e3:.1f},{q},{short},{g}")


if __name__ == "__main__":
    main()
