#!/usr/bin/env python3
"""bench.py — BASELINE.json metric on MI355X: GBA iterations/sec (+ keyframes-optimized/sec) on the
5-agent EuRoC-shaped merged map.

A "step" = one pass of the hot path over one map = one `covgpu_solve_resident` call, i.e. the whole
ceres::Solve replacement of GlobalBundleAdjustment (optimization_be.cpp:560-567) with the reference's
iteration cap (opt.gba_iteration_limit = 10): preintegration, then per trust-region iteration
linearise + landmark Schur + multifrontal MFMA Cholesky of the reduced camera system + trust-region step + cost evaluation.
Inputs are uploaded to HBM before the timed region; every step restarts from the same uploaded initial estimate. With
--gpus N (one process per GPU) the ONE map is sharded by subtree of the elimination tree and every rank runs the same
iterations (strong scaling); the collectives are issued by libcovgpu over RCCL.

The timed region runs WITHOUT profiling events; one extra, un-timed step with profiling on supplies the phase times
and the per-launch kernel durations of the `roofline` object. `cpu_baseline` runs the oracle (CPU port of the same
algorithm on all host threads; reduced system by the threaded multifrontal port) on THE SAME workload in the same run, three
repetitions, median, and
`delta_ate_gpu_cpu_m` compares the two final trajectories (north star: within 1e-3 m).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MATRIX_PEAK_TFLOPS = 78.6   # MI355X FP64 vector/matrix peak
HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md


def cpu_baseline(prob, strategy: int, iterations: int, truth_xyz, reps: int = 3):
    """Oracle (CPU port of the same algorithm) on THE SAME problem, timed on this box's host cores in this run, `reps`
    repetitions, median. The reference binary itself cannot be built here (Ceres / robopt / aslam absent,
    SURVEY.md §8c). Linearisation and the landmark Schur complement run on all OpenMP threads; the block-sparse reduced
    camera system (what Ceres hands to CHOLMOD with opt threads, optimization_be.cpp:258-262) is solved by the threaded CPU
    port of the multifrontal solve (oracle/covo_mf.py: fronts of one tree level concurrently, LAPACK per front)."""
    from covins_amd import backend, synth
    from oracle import covo, covo_mf
    threads = int(covo.lib().covo_num_threads())
    covo_mf.set_problem(prob, backend.default_options(), threads)
    covo.use_sparse_solver(min_n=3000, kind="multifrontal")
    o = covo.default_options(max_iterations=iterations, strategy=strategy)
    times, solver_s = [], []
    q = res = None
    for _ in range(reps):
        covo_mf.stats.update(calls=0, seconds=0.0)
        t0 = time.perf_counter()
        q, res = covo.gba_solve(prob, o)
        times.append(time.perf_counter() - t0); solver_s.append(covo_mf.stats["seconds"])
        if len(times) == 2 and sum(times) > 20.0:   # a bounded sample: ~30 s of CPU work on the corrected 5-agent map (14.5 s per solve)
            break
    reps = len(times)
    covo.use_sparse_solver(min_n=3000)   # back to SuperLU (parity legs)
    dt = float(np.median(times))
    return q, {
        "value": res.iterations / dt, "unit": "GBA iterations/s", "cores": threads, "reps": reps, "times_s": [round(t, 3) for t in times],
        "kind": "port", "solver": f"multifrontal Cholesky, {covo_mf.stats['fronts']} fronts in {covo_mf.stats['levels']} levels", "solver_threads": threads,
        "threads_per_phase": {"linearise+landmark Schur (OpenMP)": threads, "reduced system (thread pool x LAPACK)": threads},
        "sample": f"the timed workload itself: K={prob.K} L={prob.L} O={prob.O} (15K={15 * prob.K}), all {res.iterations} "
                  f"trust-region iterations, median of {reps}: {dt:.1f} s of which {float(np.median(solver_s)):.1f} s in the reduced-system solve",
        "kf_per_s": prob.K * res.iterations / dt, "final_cost": res.final_cost,
        "ate_rmse_m_final": synth.ate_rmse(q.kf_pose[:, 4:], truth_xyz),
    }


def cpu_baseline_pgo(pgo_prob, iterations: int, reps: int = 3):
    """Oracle PoseGraphOptimization on the same pose graph (sparse system through scipy's SuperLU), median of `reps`."""
    from oracle import covo
    covo.use_sparse_solver(min_n=3000)
    o = covo.default_options(max_iterations=iterations)
    times = []
    q = res = None
    for _ in range(reps):
        t0 = time.perf_counter()
        q, res = covo.gba_solve(pgo_prob, o, pgo=True)
        times.append(time.perf_counter() - t0)
    return q, res, {"t_call_s": float(np.median(times)), "reps": reps, "cores": 1, "kind": "port",
                    "sample": f"the same pose graph (K={pgo_prob.K}, E={pgo_prob.E}), all {res.iterations} iterations; 6K-order sparse system by SuperLU (serial)"}


def _kernels_digest():
    """sha256 over the HIP sources: a counter file collected on other kernels is stale."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "covins_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def run_a12_leg(args, rank, local_rank, world, strategy, store):
    """BASELINE configs[4] (synthetic 12-agent map at its stated 20 004 keyframes / 2.02 M landmarks) on the ranks of this run: the ONE map
    sharded by subtree exactly like the metric's (world 1: unsharded), one warm-up and --a12-steps (5) timed steps of 10 iterations, barrier +
    max-over-ranks timing. Every rank generates the map itself (seeded generator) and computes the same plan."""
    from covins_amd import backend, distrib, mapdata, synth
    t_gen = time.perf_counter()
    m = synth.make_map(synth.config_named("a12"))
    full, _ = mapdata.flatten_gba(m, visual_only=False, loop_loss=True)
    t_gen = time.perf_counter() - t_gen
    opt = backend.default_options(strategy=strategy, max_iterations=args.iterations, device=local_rank)
    ctx = backend.Context(local_rank)
    try:
        sharded = world > 1
        prob, keep = full, None
        if sharded:
            plan = distrib.shard_plan(full, opt, world)
            if plan is None:
                return {"value": None, "error": "the map does not split"}
            prob = distrib.shard_problem(full, plan, rank)
            keep = distrib.attach(ctx, plan, rank, world, store=store, tag="a12")
        t_up = time.perf_counter()
        ctx.upload(prob, opt)
        t_up = time.perf_counter() - t_up
        ctx.solve_resident(opt)
        distrib.barrier(ctx, sharded)
        t0 = time.perf_counter(); iters = 0; steps = max(2, int(args.a12_steps))
        for _ in range(steps):
            res = ctx.solve_resident(opt); iters += res.iterations
        distrib.barrier(ctx, sharded)
        dt, iters_all = distrib.aggregate(time.perf_counter() - t0, iters, ctx, sharded)
        lay = ctx.layout(); st = ctx.shard_stats()
        # ---- CPU baseline of this leg: ONE trust-region iteration of the oracle at the stated size on the box's host threads — a bounded
        # sample (ten iterations would be minutes) — and the HIP path's distance from it after that iteration. Rank 0 of a one-GPU run only.
        cpu = None
        if world == 1 and not args.no_cpu_baseline and not args.no_a12_cpu:
            from oracle import covo, covo_mf
            threads = int(covo.lib().covo_num_threads())
            covo_mf.set_problem(full, backend.default_options(), threads)
            covo.use_sparse_solver(min_n=3000, kind="multifrontal")
            covo_mf.stats.update(calls=0, seconds=0.0)
            tc = time.perf_counter()
            q1, r1 = covo.gba_solve(full, covo.default_options(max_iterations=1, strategy=strategy))
            tc = time.perf_counter() - tc
            covo.use_sparse_solver(min_n=3000)
            o1 = backend.default_options(strategy=strategy, max_iterations=1, device=local_rank)
            g1 = ctx.solve_resident(o1)
            s1 = ctx.download()
            cpu = {"value": r1.iterations / tc, "unit": "GBA iterations/s", "cores": threads, "kind": "port",
                   "sample": f"ONE trust-region iteration of the oracle on this map at its stated size (K={full.K} L={full.L} O={full.O}, 15K={15 * full.K}): "
                             f"{tc:.1f} s, of which {covo_mf.stats['seconds']:.1f} s in the reduced-system solve (oracle/covo_mf.py, LAPACK per front)",
                   "cost_after_one_iteration": {"cpu": r1.cost_trace[0], "gpu": g1.cost_trace[0]}, "accepted": {"cpu": int(r1.accepted_trace[0]), "gpu": int(g1.accepted_trace[0])},
                   "max_pose_diff_gpu_cpu_m": float(np.abs(s1.kf_pose[:, 4:] - q1.kf_pose[:, 4:]).max()),
                   "max_speed_bias_diff_gpu_cpu": float(np.abs(s1.kf_speed_bias - q1.kf_speed_bias).max())}
        # one more, un-timed step with HIP events on: phase times and the per-launch figures of this leg's own roofline objects
        ctx.set_profiling(True)
        ctx.solve_resident(opt)
        prof = ctx.profile()
        ctx.set_profiling(False)
        n_lin = max(prof["n_factor"], 1)
        nnzS = prof["offdiag_blocks"] + full.K
        b_build = 32.0 * full.O + 128.0 * full.K + 24.0 * full.L + 2304.0 * full.I + 8.0 * (135.0 * full.K + 9.0 * full.L) + 288.0 * nnzS
        b_iter = (2.0 * (32.0 * full.O + 128.0 * full.K + 24.0 * full.L) + 2304.0 * full.I + 8.0 * (135.0 * full.K + 9.0 * full.L)
                  + 2.0 * 288.0 * nnzS + (128.0 * full.K + 24.0 * full.L))
        t_iter_ms = dt / max(iters_all, 1) * 1e3
        t_ideal_ms = (b_iter / (HBM_PEAK_GBS * 1e9) + prof["plan_flops"] / (FP64_MATRIX_PEAK_TFLOPS * 1e12)) * 1e3
        syrk_tf = prof["syrk_flops"] / (prof["syrk_ms"] * 1e-3) / 1e12 if prof["syrk_ms"] > 0 else 0.0
        potrf_tf = prof["potrf_flops"] / (prof["potrf_ms"] * 1e-3) / 1e12 if prof["potrf_ms"] > 0 else 0.0
        build_ms = prof["build_ms"] / max(prof["n_build"], 1)
        return {"workload": f"a12: 12-agent synthetic map (BASELINE configs[4] at its stated size), K={full.K} L={full.L} O={full.O} I={full.I} E={full.E}",
                "value": iters_all / dt, "unit": "iterations/s", "n_gpus": world,
                "steps": steps, "warmup": 1, "ms_per_step": dt / steps * 1e3, "scaling": "strong", "final_cost": res.final_cost, "initial_cost": res.initial_cost,
                "kf_per_s": int(full.K - full.kf_fixed.sum()) * iters_all / dt,
                "phase_ms_per_iteration": {"linearise+schur": build_ms, "factor+solve": prof["factor_ms"] / n_lin},
                "roofline_iteration": {"algorithmic_bytes": b_iter, "factorisation_flops": prof["plan_flops"], "ideal_ms": t_ideal_ms, "measured_ms": t_iter_ms,
                                       "frac": t_ideal_ms / t_iter_ms if t_iter_ms > 0 else 0.0,
                                       "phase_flop_rate_tflops": prof["plan_flops"] / (prof["factor_ms"] / n_lin * 1e-3) / 1e12 if prof["factor_ms"] > 0 else 0.0},
                "roofline_syrk": {"kernel": "k_gemm_abt<SYRK_TRI> (rank-256 trailing update, full 128x128 tiles)", "bound": "mfma", "achieved": syrk_tf,
                                  "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": syrk_tf / FP64_MATRIX_PEAK_TFLOPS, "traffic": None,
                                  "launches": prof["n_syrk"], "avg_launch_ms": prof["syrk_ms"] / max(prof["n_syrk"], 1)},
                "roofline_potrf": {"kernel": "k_potrf_panel", "bound": "mfma", "achieved": potrf_tf, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": potrf_tf / FP64_MATRIX_PEAK_TFLOPS, "traffic": None, "launches_per_linear_solve": prof["n_potrf"] / n_lin,
                                   "avg_launch_ms": prof["potrf_ms"] / max(prof["n_potrf"], 1), "chain_ms_per_iteration": prof["potrf_ms"] / n_lin},
                "roofline_build": {"bound": "hbm", "achieved": b_build / (build_ms * 1e-3) / 1e9 if build_ms > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": (b_build / (build_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if build_ms > 0 else 0.0, "traffic": None, "algorithmic_bytes": b_build},
                "layout": lay, "allreduce_mib_per_linear_solve": lay["allreduce_kib"] / 1024.0, "collectives_rank0": st["collectives"],
                "map_generation_s": t_gen, "upload_s": t_up, "cpu_baseline": cpu}
    finally:
        del keep
        ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="mh12345", help="mh12345 (BASELINE metric) | mh123 | mh01 | small | a12 (configs[4] at its stated 12 x 1667 keyframes / 2M landmarks) | a12x1000 | a12x500")
    ap.add_argument("--strategy", default="dogleg", choices=["dogleg", "lm"])
    ap.add_argument("--iterations", type=int, default=10, help="trust-region iteration cap per step (reference: 10)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-shard", action="store_true", help="run the agent-sharded path (plan, sub-problem, RCCL collectives) even on ONE GPU")
    ap.add_argument("--no-e2e", action="store_true", help="skip the whole-call timing after the timed region (profiling runs)")
    ap.add_argument("--a12-steps", type=int, default=5, help="timed steps of the a12 leg (after one warm-up)")
    ap.add_argument("--no-a12-cpu", action="store_true", help="skip the a12 leg's one-iteration CPU baseline (about a minute of host time)")
    ap.add_argument("--a12-leg", type=int, default=-1, help="after the metric's workload, also time BASELINE configs[4] at its stated size (12 x 1667 keyframes, 2M landmarks) "
                    "on the same ranks and report it as `a12_leg` with rooflines of its own (1 / 0; default: on for the metric's workload at any --gpus — the "
                    "configuration the north star names for the scaling curve; ~70 s of map generation + upload, 3 + 1 steps)")
    ap.add_argument("--sustain-s", type=float, default=8.0, help="after the timed region, repeat the same steps back to back for about this many seconds, "
                    "un-timed in `value` and reported as `sustained`: long enough for an external GPU-utilisation sampler to see the device busy (0: off)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: launch the ranks ourselves, exactly as the driver does (one process per GPU over RCCL)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
                                  "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        os.environ.setdefault("COVGPU_COLL_TIMEOUT_S", "120")   # a collective that never completes ends the run with an error after two minutes, not a hang
    import torch
    torch.cuda.init()  # torch's bundled HIP runtime must come up BEFORE libcovgpu loads /opt/rocm's (else "No HIP GPUs"); torch is used for nothing else here
    from covins_amd import backend, capi, distrib, mapdata, synth
    rank, local_rank, world = distrib.env_ranks()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    strategy = capi.COVGPU_DOGLEG if args.strategy == "dogleg" else capi.COVGPU_LM
    cfg = synth.config_named(args.workload)   # ONE map, the same on every rank (seeded generator); N > 1 shards it by agent
    m = synth.make_map(cfg)
    full, _ = mapdata.flatten_gba(m, visual_only=False, loop_loss=True)
    pgo_prm = mapdata.PgoParams()
    pgo_prob, _ = mapdata.flatten_pgo(m, {}, pgo_prm)  # the pose graph of the same (still drifted) map, for the side figure below
    opt = backend.default_options(strategy=strategy, max_iterations=args.iterations, device=local_rank)
    ctx = backend.Context(local_rank)
    plan, keep = None, None
    prob = full
    sharded = world > 1 or args.force_shard
    if sharded:
        # sub-map-sharded solve of the ONE map: same plan on every rank, each keeps the residuals of its subtrees
        # of the elimination tree; the collectives (top fronts + gradient rows once per linear solve, three scalar exchanges)
        # are issued by libcovgpu itself over RCCL on its own stream — no Python, no torch on the data path
        plan = distrib.shard_plan(full, opt, world)
        assert plan is not None, "this map does not split"
        prob = distrib.shard_problem(full, plan, rank)
        keep = distrib.attach(ctx, plan, rank, world, force_single=args.force_shard)
    t_up = time.perf_counter()
    ctx.upload(prob, opt)  # inputs resident in HBM before the timed region
    t_up = time.perf_counter() - t_up

    for _ in range(args.warmup):
        ctx.solve_resident(opt)
    distrib.barrier(ctx, sharded)
    t0 = time.perf_counter()
    iters = 0
    res = None
    for _ in range(args.steps):
        res = ctx.solve_resident(opt)  # returns after its stream has drained
        iters += res.iterations        # (sharded: every rank executes the SAME iterations of the one solve)
    distrib.barrier(ctx, sharded)
    dt = time.perf_counter() - t0
    dt, iters_all = distrib.aggregate(dt, iters, ctx, sharded)
    shard_stats = ctx.shard_stats()
    # sustained leg (not `value`): the same step repeated for --sustain-s seconds, so that a sampler outside this process (rocm-smi every
    # few seconds) sees the GPU busy — the timed region above is 20 steps x ~35 ms and falls between two samples
    sustained = None
    if args.sustain_s > 0 and dt > 0:
        n_sus = max(1, int(args.sustain_s / (dt / max(args.steps, 1))))   # same count on every rank (dt is the max over ranks)
        distrib.barrier(ctx, sharded)
        ts = time.perf_counter(); it_s = 0
        for _ in range(n_sus):
            it_s += ctx.solve_resident(opt).iterations
        distrib.barrier(ctx, sharded)
        ts = time.perf_counter() - ts
        sustained = {"steps": n_sus, "seconds": ts, "iterations_per_s": it_s / ts,
                     "what": "the timed step repeated back to back after the timed region (not `value`): corroboration for an external GPU-busy sampler"}
    # one more step, NOT timed, with HIP events around the build pass, the factor+solve and every trailing-update launch
    ctx.set_profiling(True)
    ctx.solve_resident(opt)
    prof = ctx.profile()
    ctx.set_profiling(False)
    lay = ctx.layout()
    sol = ctx.download()
    if sharded:  # assemble the optimised map from the ranks' pieces (small: poses, speed-bias, landmark positions)
        pieces = distrib.gather_solutions(sol, rank, world, keep[0]) if world > 1 else [(sol.kf_pose, sol.kf_speed_bias, sol.lm_pos)]
        parts = []
        for r, (kp, ks, lp) in enumerate(pieces):
            q = distrib.shard_problem(full, plan, r) if r != rank else prob
            q = q.copy(); q.kf_pose[:] = kp; q.kf_speed_bias[:] = ks; q.lm_pos[:] = lp
            parts.append(q)
        sol = distrib.merge_solution(full, plan, parts)
    prob = full

    if rank == 0:
        # HBM traffic of the dominant kernel cannot be read inside this process (PMC counters need rocprofv3):
        # tools/pmc_pass.sh runs the counter passes of THIS command and stamps the result with the git SHA and the kernel
        # name; a stale or foreign file is refused and `traffic` stays null.
        traffic, mfma_busy, traffic_src, build_traffic, build_parts, potrf_traffic, potrf_busy = None, None, None, None, None, None, None
        try:
            import subprocess
            sha = subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
            pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_current.json")))
            if pj.get("workload") == args.workload and pj.get("kernels_sha256") == _kernels_digest() and "k_gemm_abt" in pj.get("kernel", ""):
                traffic = pj["fetch_bytes_x2"] + pj["write_bytes"]; mfma_busy = pj.get("mfma_busy_frac"); traffic_src = pj.get("git_sha", sha)
                build_traffic = pj.get("build_bytes_per_iteration"); build_parts = pj.get("build_bytes_by_kernel")
                potrf_traffic = pj.get("potrf_bytes_per_launch"); potrf_busy = pj.get("potrf_mfma_busy_frac")
        except Exception:
            pass
        truth = m.truth["kf_pose"][:, 4:]
        n = 15 * prob.K
        k_free = int(prob.K - prob.kf_fixed.sum())
        syrk_tflops = prof["syrk_flops"] / (prof["syrk_ms"] * 1e-3) / 1e12 if prof["syrk_ms"] > 0 else 0.0
        # algorithmic HBM bytes of one linearise+Schur pass (first, second and third terms)
        nnzS = prof["offdiag_blocks"] + prob.K  # covisible keyframe pairs incl. the diagonal
        b_build = (32.0 * prob.O + 128.0 * prob.K + 24.0 * prob.L + 2304.0 * prob.I + 8.0 * (135.0 * prob.K + 9.0 * prob.L)
                   + 288.0 * nnzS)  # inputs once, H/g blocks, S blocks written once (their read-back belongs to the solve)
        # SURVEY.md 8(d) in full: algorithmic bytes of one trust-region iteration (linearise + Schur, candidate evaluation, state update)
        b_iter = (2.0 * (32.0 * prob.O + 128.0 * prob.K + 24.0 * prob.L) + 2304.0 * prob.I + 8.0 * (135.0 * prob.K + 9.0 * prob.L)
                  + 2.0 * 288.0 * nnzS + (128.0 * prob.K + 24.0 * prob.L))
        t_iter_ms = dt / max(iters_all, 1) * 1e3
        potrf_tflops = prof["potrf_flops"] / (prof["potrf_ms"] * 1e-3) / 1e12 if prof["potrf_ms"] > 0 else 0.0
        n_lin = max(prof["n_factor"], 1)
        t_ideal_ms = (b_iter / (HBM_PEAK_GBS * 1e9) + prof["plan_flops"] / (FP64_MATRIX_PEAK_TFLOPS * 1e12)) * 1e3
        out = {
            "metric": "GBA iterations/sec, 5-agent EuRoC merged map" if args.workload == "mh12345" else f"GBA iterations/sec, {args.workload}",
            "value": iters_all / dt, "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": ("synthetic map on the EuRoC ground-truth trajectories MH01-05 (orb_slam3/evaluation/Ground_truth/EuRoC_left_cam), EuRoC calibration; "
                     "IMU of agents 3-5: the RECORDED 200 Hz samples (EuRoC_IMU/MH03-05.txt), agents 1-2: synthesised from the trajectory (no recording in the "
                     "tree); landmarks, pixel measurements (1 px noise), drift of the initial estimate and loop constraints synthetic") if args.workload == "mh12345"
                    else "synthetic (covins_amd/synth.py; recorded EuRoC IMU where an agent runs MH03-05 at its own keyframe times)",
            "timed_region": "profiling events off; phase / kernel figures below come from one extra un-timed step",
            "config": {"workload": f"{args.workload}: {len(cfg.agents)}-agent EuRoC MH-shaped merged map, visual-inertial GBA "
                                   f"(K={prob.K} keyframes, L={prob.L} landmarks, O={prob.O} observations, I={prob.I} IMU factors, "
                                   f"E={prob.E} loop edges; reduced camera system 15K={n}: "
                                   + (f"multifrontal MFMA Cholesky over a nested-dissection tree of {lay['nd_fronts']} fronts in {lay['nd_levels']} levels, "
                                      f"{lay['nd_serial_panels']} serial 256-column panels, root front of order {lay['nd_root_order']})" if lay["nd_fronts"] else
                                      (f"speed-bias chains eliminated block-tridiagonally, pose system 6K={6 * prob.K} as {lay['blocks']} agent blocks + "
                                       f"{lay['border_kf']} shared keyframes)" if lay["arrow"] else f"dense MFMA Cholesky on the 6K={6 * prob.K} pose system)")),
                       "layout": lay,
                       "nnzS_fill": (2 * prof["offdiag_blocks"] + prob.K) / float(prob.K) ** 2,
                       "strategy": args.strategy, "iterations_per_step": args.iterations,
                       "sharding": ("none (1 GPU)" if not sharded else
                                    f"ONE map sharded over {world} ranks: {plan.subtrees} subtrees of the elimination tree dealt to the ranks, "
                                    f"{lay['top_unknowns']} scalar unknowns in the replicated top ({int((plan.pose_rank < 0).sum())} shared keyframe poses); "
                                    f"{'RCCL' if world > 1 or keep == ('rccl',) else 'in-process one-rank group'} all-reduce (issued by libcovgpu on its own stream) of the top fronts + right-hand sides + gradient rows ({lay['allreduce_kib'] / 1024:.1f} MiB per linear solve) "
                                    f"and 3 scalar exchanges per iteration: {shard_stats['collectives']} collectives, {shard_stats['bytes'] / 1e6:.1f} MB on rank 0 "
                                    f"in the timed steps + warm-up")},
            "kf_per_s": k_free * iters_all / dt,
            "iterations_executed": iters_all,
            "final_cost": res.final_cost, "initial_cost": res.initial_cost,
            "upload_s_not_in_value": t_up,  # host layout build (chains, covisible-pair lists) + H2D; PCIe-inclusive cost of one call
            "ate_rmse_m": {"initial": synth.ate_rmse(prob.kf_pose[:, 4:], truth), "final": synth.ate_rmse(sol.kf_pose[:, 4:], truth),
                           "initial_sim3": synth.ate_rmse(prob.kf_pose[:, 4:], truth, with_scale=True),
                           "final_sim3": synth.ate_rmse(sol.kf_pose[:, 4:], truth, with_scale=True)},  # evo_ape -va / -vas
            "phase_ms_per_iteration": {"linearise+schur": prof["build_ms"] / max(prof["n_build"], 1),
                                       "factor+solve": prof["factor_ms"] / max(prof["n_factor"], 1)},
            # the serial panel chain: second row of profiles/r05z_kernel_stats.csv (16.6 % of GPU time), and on the critical path of every linear solve
            "roofline_potrf": {"kernel": "k_potrf_panel (ONE workgroup per front factors a 256x256 diagonal block of the batched front factorisation: "
                                   "the serial panel chain; v_mfma_f64_16x16x4_f64 for the in-block updates, wave 0 carries the pivot chain)",
                         "bound": "mfma", "achieved": potrf_tflops, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": potrf_tflops / FP64_MATRIX_PEAK_TFLOPS, "traffic": potrf_traffic, "mfma_busy_frac_pmc": potrf_busy,
                         "launches": prof["n_potrf"], "launches_per_linear_solve": prof["n_potrf"] / n_lin,
                         "avg_launch_ms": prof["potrf_ms"] / max(prof["n_potrf"], 1),
                         "flops_per_launch": prof["potrf_flops"] / max(prof["n_potrf"], 1),
                         "chain_ms_per_iteration": prof["potrf_ms"] / n_lin,
                         "note": "algorithmic flops per launch = sum over the fronts of the batch of n^3/3 + n^2 (n = the front's real columns in the "
                                 "panel), over the HIP-event duration of the launch on its stream (un-timed profiling step). One workgroup per "
                                 "front: a latency-bound pivot chain, far below the matrix peak by construction — its launches are serial, so "
                                 "chain_ms_per_iteration is on the critical path of every linear solve (DESIGN.md 4.5-4.6)"},
            # the kernel with the largest share of GPU time (first row of profiles/r05z_kernel_stats.csv: 22.0 % on the corrected 5-agent map, whose
            # fronts carry borders of 2 000 unknowns; on the round-4 map the panel chain led): the rank-256 trailing update
            "roofline": {"kernel": "k_gemm_abt<SYRK_TRI> / k_gemm_abt_q<SYRK_TRI,64,64> (rank-256 trailing update of the batched front factorisation, "
                                        "v_mfma_f64_16x16x4_f64; 128x128 tiles, or 64x64 quadrants for tile lists of <= 1024 entries)",
                              "bound": "mfma", "achieved": syrk_tflops, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": syrk_tflops / FP64_MATRIX_PEAK_TFLOPS, "traffic": traffic,
                              "mfma_busy_frac_pmc": mfma_busy,
                              "traffic_note": ("HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE) from tools/pmc_pass.sh at " + str(traffic_src)) if traffic is not None
                                              else "null: no counter pass of these kernel sources (tools/pmc_pass.sh writes profiles/pmc_traffic_current.json)",
                              "launches": prof["n_syrk"], "avg_launch_ms": prof["syrk_ms"] / max(prof["n_syrk"], 1),
                              "note": "algorithmic flops per launch = 2 x 128 x 128 x K per live tile pair (K = the front's real columns in the panel), over the HIP-event "
                                      "duration of the launch on its stream (un-timed profiling step). The launches of this map are short — ~1 000 quarter-tile workgroups "
                                      "whose waves live ~7 us: dispatch and tail, not the matrix pipe (DESIGN.md 5); configs[4]'s full-tile launches: a12_leg.roofline_syrk"},
            "roofline_iteration": {"what": "one whole trust-region iteration against the two rooflines it could be bound by: SURVEY.md 8(d)'s B_iter at the "
                                           "HBM peak plus the plan's factorisation flops at the FP64 matrix peak",
                                   "algorithmic_bytes": b_iter, "factorisation_flops": prof["plan_flops"], "ideal_ms": t_ideal_ms,
                                   "measured_ms": t_iter_ms, "frac": t_ideal_ms / t_iter_ms if t_iter_ms > 0 else 0.0,
                                   "phase_flop_rate_tflops": prof["plan_flops"] / (prof["factor_ms"] / n_lin * 1e-3) / 1e12 if prof["factor_ms"] > 0 else 0.0},
            "roofline_build": {"kernel": "linearise + landmark Schur pass (k_lm_lin, k_kf_reduce, k_pair_blocks, k_imu_*, k_edge_*)", "bound": "hbm",
                               "achieved": b_build / (prof["build_ms"] / max(prof["n_build"], 1) * 1e-3) / 1e9 if prof["build_ms"] > 0 else 0.0,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "traffic": build_traffic, "traffic_mb_by_kernel": build_parts,
                               "traffic_note": "HBM bytes per iteration of the pass's kernels (2 x FETCH_SIZE + WRITE_SIZE, tools/pmc_pass.sh); null without a counter pass of these sources",
                               "nnzS_blocks": nnzS,
                               "note": "whole linearise+Schur pass (a dozen kernels; the clearing of the fronts' live tiles runs on a second stream "
                                       "underneath) against SURVEY.md 8(d)'s algorithmic bytes (inputs once, H/g blocks, S blocks once). The pass "
                                       "really moves more: the fill and the per-observation records of the atomic-free, bit-reproducible build "
                                       "(DESIGN.md 4.1, 6)"},
        }
        out["roofline_build"]["frac"] = out["roofline_build"]["achieved"] / HBM_PEAK_GBS
        if sustained is not None:
            out["sustained"] = sustained
        # whole Optimization::GlobalBundleAdjustment call as backend.cpp:141-156 issues it (outlier round of 5 iterations +
        # main round, flatten + H2D + solves + D2H + write-back + Map::Clean) — PCIe- and host-inclusive, never `value`
        from covins_amd.optimization import Optimization, OptParams
        if sharded:
            args.no_e2e = True; args.no_cpu_baseline = True   # side figures belong to the single-GPU line
        if not args.no_e2e:
            # (three calls, each on its own copy of the map — a call erases observations —; the median is reported with all three: a
            #  single host-side call of 0.1 s is at the mercy of one page-fault storm or scheduler hiccup)
            calls = []
            for _ in range(3):
                mc = m.copy()
                t_call = time.perf_counter()
                info = Optimization.GlobalBundleAdjustment(mc, args.iterations, -1.0, False, True, False, params=OptParams(strategy=strategy), ctx=ctx)
                calls.append((time.perf_counter() - t_call, info))
            t_call, info = sorted(calls, key=lambda c: c[0])[1]
            out["e2e_call"] = {"t_call_s": t_call, "t_calls_s": [round(c[0], 4) for c in calls], "kf_per_s_e2e": k_free / t_call,
                               "iterations": info["round1"].iterations + info["round2"].iterations,
                               "outliers_removed": info["outliers_removed"],
                               "stages_s": {k: round(v, 4) for k, v in info["stages_s"].items()},
                               "what": "covins_amd.optimization.Optimization.GlobalBundleAdjustment(map, 10, outlier_removal=True) "
                                       "on the same map (median of three calls; stages of that call), host flattening in numpy; ONE flatten and ONE upload per call: the second round's "
                                       "problem is derived on the device (covgpu_gba_two_round)"}
        if not args.no_e2e:
            # side figure: one PoseGraphOptimization solve of the same map (multifrontal solve on the pose graph's own elimination tree); not `value`
            popt = backend.default_options(max_iterations=pgo_prm.pgo_iteration_limit, device=local_rank)
            ctx.pgo_solve(pgo_prob, popt)
            t_p = time.perf_counter()
            psol, pres = ctx.pgo_solve(pgo_prob, popt)
            out["pgo_call"] = {"t_call_s": time.perf_counter() - t_p, "iterations": pres.iterations, "edges": int(pgo_prob.E), "keyframes": int(pgo_prob.K),
                               "initial_cost": pres.initial_cost, "final_cost": pres.final_cost,
                               "phase_s": {"upload (plan + H2D)": pres.t_upload_s, "solve": pres.t_solve_s,
                                           "download": pres.t_download_s},
                               "critical_path": "one panel chain per level of the pose graph's elimination tree (6-dof blocks, chains read from the edge graph; k_front.hip): latency-bound, no roofline",
                               "what": "covgpu_pgo_solve: whole PoseGraphOptimization solve (optimization_be.cpp:1024-1031) incl. plan, H2D, D2H"}
            if not args.no_cpu_baseline:
                pq, pr, out["pgo_call"]["cpu_baseline"] = cpu_baseline_pgo(pgo_prob, pgo_prm.pgo_iteration_limit)
                out["pgo_call"]["max_pose_diff_gpu_cpu_m"] = float(np.abs(psol.kf_pose[:, 4:] - pq.kf_pose[:, 4:]).max())
                out["pgo_call"]["gpu_over_cpu"] = out["pgo_call"]["cpu_baseline"]["t_call_s"] / out["pgo_call"]["t_call_s"]
        if not args.no_cpu_baseline and prob.K > 5000:
            out["cpu_baseline"] = {"value": None, "unit": "GBA iterations/s", "cores": 0, "kind": "port",
                                   "sample": f"not run: the CPU port does not finish K={prob.K} within minutes (its 5-agent time is in the default bench line)"}
        elif not args.no_cpu_baseline:
            qc, out["cpu_baseline"] = cpu_baseline(prob, strategy, args.iterations, truth)
            out["delta_ate_gpu_cpu_m"] = abs(out["ate_rmse_m"]["final"] - out["cpu_baseline"]["ate_rmse_m_final"])
            out["max_pose_diff_gpu_cpu_m"] = float(np.abs(sol.kf_pose[:, 4:] - qc.kf_pose[:, 4:]).max())
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    distrib.barrier(ctx, sharded)
    ctx.close()
    if rank == 0 and not args.no_e2e:
        # the same call through the C++ facade (include/covins_gpu/optimization_gpu.hpp — the maintainer-facing drop-in) on stand-in
        # Map / Keyframe / Landmark objects of the same map (tests/cpp: the real COVINS classes need ROS / Eigen / OpenCV); the
        # first call creates the thread's context and warms the library up. Behind ctx.close(): with the bench's own context alive the
        # facade's streams share the runtime's four hardware queues with it (0.19 instead of 0.17 s)
        try:
            from tests.facade_util import StandinMap
            ts, stg = [], []
            for _ in range(4):
                smap = StandinMap(m)
                t_c = time.perf_counter(); smap.gba(args.iterations); ts.append(time.perf_counter() - t_c)
                stg.append({k: round(v * 1e-3, 4) for k, v in StandinMap.last_stages().items()})
                smap.close()
            mid = sorted(range(1, 4), key=lambda i: ts[i])[1]   # median of the three calls after the warm-up call
            stages_cpp = stg[mid]
            out["e2e_call_cpp"] = {"t_call_s": ts[mid], "t_calls_s": [round(t, 4) for t in ts[1:]], "t_first_call_s": ts[0], "kf_per_s_e2e": int(full.K - full.kf_fixed.sum()) / ts[mid], "stages_s": stages_cpp,
                                   "note": "Map::Clean is the reference's own method (map_be.cpp:448-454, 698-743: it copies every landmark's observation "
                                           "map to read its size; the stand-in mirrors that) — not part of what this build replaces",
                                   "what": "covins_gpu::Optimization::GlobalBundleAdjustment(map, 10) through the C++ facade on stand-in map "
                                           "objects (one Map -> IR walk, one upload, both rounds on the device, erase, write-back, Map::Clean)"}
            # the same with Params::device_clean (opt-in): the call's end from its own counts instead of map->Clean()
            from tests.facade_util import lib as _shim
            _shim().shim_set_device_clean(1)
            try:
                ts2, stg2 = [], []
                for _ in range(3):
                    smap = StandinMap(m)
                    t_c = time.perf_counter(); smap.gba(args.iterations); ts2.append(time.perf_counter() - t_c)
                    stg2.append({k: round(v * 1e-3, 4) for k, v in StandinMap.last_stages().items()})
                    smap.close()
                mid2 = sorted(range(3), key=lambda i: ts2[i])[1]
                out["e2e_call_cpp"]["device_clean"] = {"t_call_s": ts2[mid2], "t_calls_s": [round(t, 4) for t in ts2], "stages_s": stg2[mid2],
                                                       "what": "Params::device_clean = 1 (opt-in): landmarks left with < 2 observations erased from the counts of the call "
                                                               "(map_be.cpp:698-717) instead of map->Clean()'s second copy of every observation map"}
            finally:
                _shim().shim_set_device_clean(0)
                _shim().shim_shutdown()   # the facade's cached context (its streams share the runtime's hardware queues with the a12 leg's: 15.6 instead of 20.4 it/s)
        except Exception as e:   # (the shim needs g++ on the box; never fatal for the metric line)
            out["e2e_call_cpp"] = {"t_call_s": None, "error": repr(e)[:200]}
    # ---- configs[4] leg (all ranks, behind everything else and with the metric's context closed: eight live streams share the runtime's
    #      four hardware queues and the chain and bulk streams of the second context then serialise — measured: 32.8 instead of 43.7 it/s)
    want_leg = args.a12_leg == 1 or (args.a12_leg < 0 and args.workload == "mh12345" and not args.force_shard)
    if want_leg:
        try:
            a12_leg = run_a12_leg(args, rank, local_rank, world, strategy, keep[0] if (world > 1 and keep) else None)
        except Exception as e:   # (never fatal for the metric line; a hang is bounded by COVGPU_COLL_TIMEOUT_S)
            a12_leg = {"value": None, "error": repr(e)[:300]}
        if rank == 0:
            out["a12_leg"] = a12_leg
    if rank == 0:
        # RCCL prints its version banner through C stdio, which would otherwise be flushed AFTER this line at exit: flush it
        # first so that the JSON line is the last thing on stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
