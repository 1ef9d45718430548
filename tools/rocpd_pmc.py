#!/usr/bin/env python3
"""Per-kernel PMC totals from rocprofv3 rocpd databases (one database per counter pass).
usage: rocpd_pmc.py FETCH_SIZE=fetch.db WRITE_SIZE=write.db [out.csv]
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB. gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE
counts 128-B requests as 64 B for wide coalesced streaming reads -> the `fetch_x2` column doubles it."""
import sqlite3
import sys


def main():
    dbs = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a)
    out = [a for a in sys.argv[1:] if "=" not in a]
    per = {}
    for ctr, path in dbs.items():
        db = sqlite3.connect(path)
        for name, n, tot, dur in db.execute("select name, count(*), sum(counter_value), sum(duration) from pmc_events where counter_name=? group by name", (ctr,)):
            per.setdefault(name, {})[ctr] = (n, tot, dur)
    lines = ["Name,Calls,FETCH_KiB_per_call,FETCH_x2_MB_per_call,WRITE_KiB_per_call,WRITE_MB_per_call,AvgDurationUs_in_pmc_pass"]
    for name, d in sorted(per.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE", (0, 0, 0))[1] or 0)):
        n, f, dur = d.get("FETCH_SIZE", (0, 0.0, 0))
        n2, w, dur2 = d.get("WRITE_SIZE", (0, 0.0, 0))
        calls = max(n, n2, 1)
        lines.append(f"\"{name}\",{calls},{(f or 0) / calls:.1f},{2 * (f or 0) / calls / 1024:.3f},{(w or 0) / calls:.1f},{(w or 0) / calls / 1024:.3f},{(dur or dur2 or 0) / calls / 1e3:.1f}")
    txt = "\n".join(lines) + "\n"
    if out:
        open(out[0], "w").write(txt)
    sys.stdout.write(txt)


if __name__ == "__main__":
    main()
