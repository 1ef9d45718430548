#!/usr/bin/env python3
"""Generates tests/golden/pgo_{mh123,mh12345}.npz: the CPU oracle's PoseGraphOptimization result (optimization_be.cpp:833-1086)
on the BASELINE.json configs[2] / configs[3] maps at FULL size — the ~6K-edge graph of :947-1021 (successor + five-neighbour
edges measured from the VIO poses, loop edges with the keyframe weight and Cauchy(0.5)), dogleg (reference) and LM, 10
iterations (opt.pgo_iteration_limit). Inputs are regenerated from the seeded generator; `in_digest` pins them.
Run in this container:  python tools/make_golden_pgo.py [names...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from covins_amd import capi, mapdata, synth  # noqa: E402
from oracle import covo  # noqa: E402
from make_golden_full import digest  # noqa: E402


def main():
    names = sys.argv[1:] or ["mh123", "mh12345"]
    covo.use_sparse_solver(min_n=3000)
    for name in names:
        m = synth.make_map(synth.config_named(name))
        p, _ = mapdata.flatten_pgo(m, {}, mapdata.PgoParams())
        out = {"in_digest": np.array(digest(p)), "sizes": np.array([p.K, p.E]), "truth_xyz": m.truth["kf_pose"][:, 4:]}
        for sname, strat in (("dogleg", capi.COVGPU_DOGLEG), ("lm", capi.COVGPU_LM)):
            t0 = time.perf_counter()
            q, res = covo.gba_solve(p, covo.default_options(strategy=strat, max_iterations=10), pgo=True)
            dt = time.perf_counter() - t0
            n = res.iterations
            out[f"{sname}_pose"] = q.kf_pose
            out[f"{sname}_trace"] = np.array(res.cost_trace[:n]); out[f"{sname}_acc"] = np.array(res.accepted_trace[:n])
            out[f"{sname}_radius"] = np.array(res.radius_trace[:n]); out[f"{sname}_cost"] = np.array([res.initial_cost, res.final_cost])
            out[f"{sname}_term"] = np.array(res.termination)
            print(f"pgo {name} {sname}: K={p.K} E={p.E} {n} iterations (accepted {res.accepted}, termination {res.termination}) {dt:.1f} s "
                  f"cost {res.initial_cost:.6e} -> {res.final_cost:.6e} ATE {synth.ate_rmse(m.kf_pose[:, 4:], m.truth['kf_pose'][:, 4:]):.4f} -> "
                  f"{synth.ate_rmse(q.kf_pose[:, 4:], m.truth['kf_pose'][:, 4:]):.4f} m", flush=True)
        dst = os.path.join(ROOT, "tests", "golden", f"pgo_{name}.npz")
        np.savez_compressed(dst, **out)
        print("wrote", dst, os.path.getsize(dst), "bytes", flush=True)


if __name__ == "__main__":
    main()
