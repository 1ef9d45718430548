#!/usr/bin/env python3
"""Per-kernel PMC totals from rocprofv3 rocpd databases (one database per counter pass).
usage: rocpd_pmc.py FETCH_SIZE=fetch.db WRITE_SIZE=write.db [out.csv [out.json workload]]
The optional JSON is what bench.py reads for `roofline.traffic` (profiles/r01_pmc_traffic.json).
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
    if len(out) >= 3:
        import json
        alias = {"k_gemm_abt<0, 128, 128>": "k_gemm_abt<SYRK_TRI>", "k_pair_blocks": "k_pair_blocks", "k_lm_lin": "k_lm_lin",
                 "k_yty_semisep": "k_yty_semisep", "k_sb_chain_cols": "k_sb_chain_cols", "k_kf_reduce": "k_kf_reduce"}
        kernels = {}
        for name, d in per.items():
            for pat, key in alias.items():
                if pat in name:
                    n, f, _ = d.get("FETCH_SIZE", (0, 0.0, 0))
                    n2, w, _ = d.get("WRITE_SIZE", (0, 0.0, 0))
                    calls = max(n, n2, 1)
                    kernels[key] = {"fetch_bytes_x2": 2.0 * (f or 0) * 1024 / calls, "write_bytes": (w or 0) * 1024 / calls, "calls": calls}
        json.dump({"workload": out[2],
                   "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace), bench.py --steps 1 --warmup 0 "
                             "--no-cpu-baseline --no-e2e; FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM (gfx950 counts 128-B requests "
                             "as 64 B); bytes per launch, averaged over all launches of the kernel",
                   "kernels": kernels}, open(out[1], "w"), indent=1)
    sys.stdout.write(txt)


if __name__ == "__main__":
    main()
