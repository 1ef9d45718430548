#!/usr/bin/env python3
"""Per-kernel totals of arbitrary PMC counters from one rocprofv3 rocpd database (dev tool).
usage: rocpd_counters.py results.db [out.csv]
rocpd stores one row per counter instance (XCD / shader engine) and dispatch; rows are summed per dispatch.
MfmaBusy_pct = sum of SQ_VALU_MFMA_BUSY_CYCLES over all SIMDs / (kernel duration x 2.4 GHz x 1024 SIMDs): the share
of SIMD-cycles the matrix pipe was busy, to be read next to the flop-based roofline fraction."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, counter_name, count(distinct dispatch_id), sum(counter_value), "
                      "sum(duration) * 1.0 / count(*) * count(distinct dispatch_id) from pmc_events group by name, counter_name").fetchall()
    per = {}
    for name, ctr, n, tot, dur in rows:
        per.setdefault(name, {})[ctr] = (n, tot or 0.0, dur or 0)
    ctrs = sorted({r[1] for r in rows})
    lines = ["Name,Calls,AvgDurationUs," + ",".join(c + "_per_call" for c in ctrs) + ",MfmaBusy_pct"]
    for name, d in sorted(per.items(), key=lambda kv: -max(v[2] for v in kv[1].values())):
        n = max(v[0] for v in d.values())
        dur = max(v[2] for v in d.values()) / n / 1e3
        vals = [d.get(c, (0, 0.0, 0))[1] / n for c in ctrs]
        util = ""
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and dur > 0:
            util = f"{100.0 * d['SQ_VALU_MFMA_BUSY_CYCLES'][1] / n / (dur * 1e-6 * 2.4e9 * 1024):.1f}"
        lines.append(f"\"{name}\",{n},{dur:.1f}," + ",".join(f"{v:.1f}" for v in vals) + "," + util)
    txt = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
    sys.stdout.write(txt)


if __name__ == "__main__":
    main()
