#!/usr/bin/env python3
"""Generates tests/golden/{mh01,mh123,mh12345}.npz: the CPU oracle's result on the BASELINE.json configurations at
FULL size (configs[1], configs[2] GBA part, and the metric's 5-agent map), for dogleg (the reference's strategy,
optimization_be.cpp:564) and Levenberg-Marquardt (north-star mode), 10 iterations each (opt.gba_iteration_limit).

The reduced camera systems (15 K = 8k .. 33k unknowns) are solved by scipy's SuperLU through oracle/covo.py — a
library that shares no code with the oracle's own dense Cholesky or with the HIP kernels. Inputs are NOT stored (the
5-agent map has 878k observations): they are regenerated from the seeded generator, and `in_digest` pins them.
Stored per strategy: final poses, speed-bias, every 8th landmark, cost / acceptance / radius trace. Plus, for the
single-linearisation test at full size: ||S x - b|| inputs are regenerated on the fly by the test itself.

Run in this container (about four minutes on 8 cores):  python tools/make_golden_full.py [names...]

BASELINE configs[4] shape (`--one-iter a12x1000 a12`: 12 000 / 20 004 keyframes): tests/golden/<name>_it1.npz holds ONE
trust-region iteration of the oracle (optimization_be.cpp:560-567 with max_num_iterations = 1) — initial / final cost, the
accept flag, radius, and every KF_STRIDE-th pose and speed-bias. At this size the reduced system (15 K = 180k / 300k
unknowns) goes through the threaded CPU port of the multifrontal solve (oracle/covo_mf.py: LAPACK per front), SuperLU does
not finish; about 3 / 8 minutes on 8 cores. `--iters N a12x1000` (round 6): N iterations the same way -> tests/golden/<name>_itN.npz (cost trace,
accept sequence, radii, strided poses and speed-bias after the N-th iteration; the seconds it took and the thread count are stored).
"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from covins_amd import capi, mapdata, synth  # noqa: E402
from oracle import covo  # noqa: E402

LM_STRIDE = 8
KF_STRIDE = 4


def digest(p: capi.FlatProblem) -> str:
    h = hashlib.sha256()
    for k in sorted(p.__dict__):
        v = getattr(p, k)
        if v is not None:
            h.update(k.encode()); h.update(np.ascontiguousarray(v).tobytes())
    return h.hexdigest()


def one_iteration(names, iters=1):
    from covins_amd import backend
    from oracle import covo_mf
    threads = int(covo.lib().covo_num_threads())
    for name in names:
        m = synth.make_map(synth.config_named(name))
        p, _ = mapdata.flatten_gba(m, False, True)
        covo_mf.set_problem(p, backend.default_options(), threads)
        covo.use_sparse_solver(min_n=3000, kind="multifrontal")
        t0 = time.perf_counter()
        q, res = covo.gba_solve(p, covo.default_options(max_iterations=iters))
        dt = time.perf_counter() - t0
        out = {"in_digest": np.array(digest(p)), "sizes": np.array([p.K, p.L, p.O, p.I, p.E]), "kf_stride": np.array(KF_STRIDE),
               "pose": q.kf_pose[::KF_STRIDE], "sb": q.kf_speed_bias[::KF_STRIDE], "cost": np.array([res.initial_cost, res.final_cost]),
               "trace": np.array(res.cost_trace[:res.iterations]), "acc": np.array(res.accepted_trace[:res.iterations]),
               "radius": np.array(res.radius_trace[:res.iterations]), "iterations": np.array(res.iterations),
               "solver": np.array("oracle/covo_mf.py (multifrontal, LAPACK per front)")}
        out["seconds"] = np.array(dt); out["threads"] = np.array(threads)
        print(f"{name} {iters} iteration(s): K={p.K} L={p.L} O={p.O} {dt:.1f} s on {threads} threads cost {res.initial_cost:.9e} -> {res.final_cost:.9e} acc {list(res.accepted_trace[:res.iterations])}", flush=True)
        dst = os.path.join(ROOT, "tests", "golden", f"{name}_it{iters}.npz")
        np.savez_compressed(dst, **out)
        print("wrote", dst, os.path.getsize(dst), "bytes", flush=True)
    covo.use_sparse_solver(min_n=3000)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--one-iter":
        return one_iteration(sys.argv[2:] or ["a12x1000"])
    if len(sys.argv) > 2 and sys.argv[1] == "--iters":   # round 6: several trust-region iterations at configs[4]'s shape -> tests/golden/<name>_it<N>.npz
        return one_iteration(sys.argv[3:] or ["a12x1000"], int(sys.argv[2]))
    names = sys.argv[1:] or ["mh01", "mh123", "mh12345"]
    covo.use_sparse_solver(min_n=3000)
    for name in names:
        m = synth.make_map(synth.config_named(name))
        p, _ = mapdata.flatten_gba(m, False, True)
        out = {"in_digest": np.array(digest(p)), "sizes": np.array([p.K, p.L, p.O, p.I, p.E]), "lm_stride": np.array(LM_STRIDE),
               "truth_xyz": m.truth["kf_pose"][:, 4:]}
        for sname, strat in (("dogleg", capi.COVGPU_DOGLEG), ("lm", capi.COVGPU_LM)):
            t0 = time.perf_counter()
            q, res = covo.gba_solve(p, covo.default_options(strategy=strat, max_iterations=10))
            dt = time.perf_counter() - t0
            n = res.iterations
            out[f"{sname}_pose"] = q.kf_pose; out[f"{sname}_sb"] = q.kf_speed_bias; out[f"{sname}_lm"] = q.lm_pos[::LM_STRIDE]
            out[f"{sname}_trace"] = np.array(res.cost_trace[:n]); out[f"{sname}_acc"] = np.array(res.accepted_trace[:n])
            out[f"{sname}_radius"] = np.array(res.radius_trace[:n])
            out[f"{sname}_cost"] = np.array([res.initial_cost, res.final_cost])
            # conditioning of the stored landmarks at the final estimate (parity criterion, tests/test_gpu_full.py)
            H = covo.landmark_hessians(q, covo.default_options())[::LM_STRIDE]
            out[f"{sname}_lm_cond"] = np.linalg.cond(H)
            print(f"{name} {sname}: K={p.K} L={p.L} O={p.O} {n} iterations {dt:.1f} s cost {res.initial_cost:.6e} -> {res.final_cost:.6e} "
                  f"ATE {synth.ate_rmse(q.kf_pose[:, 4:], m.truth['kf_pose'][:, 4:]):.4f} m", flush=True)
        dst = os.path.join(ROOT, "tests", "golden", f"{name}.npz")
        np.savez_compressed(dst, **out)
        print("wrote", dst, os.path.getsize(dst), "bytes", flush=True)


if __name__ == "__main__":
    main()
