#!/usr/bin/env python3
"""Kernel statistics (the `--stats` summary) from a rocprofv3 rocpd SQLite database.
usage: rocpd_stats.py results.db [out.csv]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                      "group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs"]
    for n, c, s, a, mn, mx in rows:
        lines.append(f"\"{n}\",{c},{s},{a:.1f},{100.0 * s / total:.2f},{mn},{mx}")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    sys.stdout.write(out)


if __name__ == "__main__":
    main()
