"""End-to-end runs of the BASELINE.json configurations that fit one GPU, through the Optimization facade
(Python mirror of covins::Optimization), checked with size-independent properties: monotone cost, ATE against the
synthetic ground truth, idempotence, agreement between the dogleg (reference) and LM (north-star) modes."""
import numpy as np
import pytest

from covins_amd import backend, capi, mapdata, synth
from covins_amd.optimization import Optimization, OptParams
from tests.util import truth_map

pytestmark = pytest.mark.gpu


def ate(m):
    ok = ~m.kf_invalid
    return synth.ate_rmse(m.kf_pose[ok, 4:], m.truth["kf_pose"][ok, 4:])


def test_config3_three_agents_pgo_then_gba():
    """configs[2]: 3-agent merged map with loop constraints, PoseGraphOptimization then GlobalBundleAdjustment(action=1)."""
    cfg = synth.config_named("mh123"); cfg.max_kf_per_agent = 150; cfg.drift_trans = 0.04; cfg.drift_yaw_deg = 0.3
    m = synth.make_map(cfg)
    a0 = ate(m)
    lm_err0 = np.linalg.norm(m.lm_pos - m.truth["lm_pos"], axis=1).mean()
    info = Optimization.PoseGraphOptimization(m, {})
    assert info["result"].final_cost < info["result"].initial_cost
    a1 = ate(m)
    # landmarks were re-anchored rigidly with their reference keyframes: they follow the pose correction
    lm_err1 = np.linalg.norm(m.lm_pos - m.truth["lm_pos"], axis=1).mean()
    assert lm_err1 < lm_err0 * 1.5
    m.kf_gba_optimized[:] = False
    info = Optimization.GlobalBundleAdjustment(m, 10, -1.0, False, True, False)
    a2 = ate(m)
    # (round 5: cameras look where the ground truth says — at the hall's far walls — and agent 3 carries the recorded IMU, which agrees with the
    #  ground-truth file to a few centimetres only: the oracle ends at 0.030 m on this map, 0.021 / 0.027 / 0.031 per agent)
    print(f"config3: ATE start {a0:.4f} after PGO {a1:.4f} after GBA {a2:.4f} m")
    # (bounds pinned to the oracle's value with a 20 % margin — observed 0.0293)
    assert info["outliers_removed"] >= 0 and a2 < a0 and a2 < 1.2 * 0.030 and a2 <= a1 * 1.05
    tr = np.array(info["round2"].cost_trace[:info["round2"].iterations])
    assert np.all(np.diff(tr) <= 1e-9 * tr[:-1])
    assert m.kf_gba_optimized.all()
    # PGO after GBA: every keyframe is fixed (opt.pgo_fix_kfs_after_gba) -> nothing moves
    before = m.kf_pose.copy()
    Optimization.PoseGraphOptimization(m, {})
    assert np.abs(m.kf_pose - before).max() < 1e-12


def test_gba_outlier_round_removes_gross_outliers():
    cfg = synth.config_named("small"); cfg.outlier_frac = 0.04
    m = synth.make_map(cfg)
    n_obs = m.O
    info = Optimization.GlobalBundleAdjustment(m, 10, -1.0, False, True, False)
    removed = info["outliers_removed"]
    assert 0.03 * n_obs < removed < 0.12 * n_obs and m.O == n_obs - removed
    print(f"outlier round: ATE {ate(m):.4f} m")
    assert ate(m) < 1.2 * 0.027      # observed 0.0285 (what 15 s of trajectory per agent and landmarks 5-12 m away determine: the oracle ends at 0.027 m)
    # without the outlier round the Cauchy loss still keeps the estimate sane, but worse
    m2 = synth.make_map(cfg)
    Optimization.GlobalBundleAdjustment(m2, 10, -1.0, False, False, False)
    print(f"without the outlier round: ATE {ate(m2):.4f} m")
    assert m2.O == n_obs and ate(m2) < 0.04   # observed 0.0267


def test_dogleg_and_lm_agree_at_convergence():
    m = synth.make_map(synth.config_named("small"))
    p = mapdata.flatten_gba(m, False, True)[0]
    ctx = backend.Context(0)
    sols = {}
    for name, strat in (("dogleg", capi.COVGPU_DOGLEG), ("lm", capi.COVGPU_LM)):
        sols[name] = ctx.gba_solve(p, backend.default_options(strategy=strat, max_iterations=40))
    (sd, rd), (sl, rl) = sols["dogleg"], sols["lm"]
    assert rd.termination in (1, 2, 3) and rl.termination in (1, 2, 3)       # both converge well inside 40 iterations
    # both stop on the 1e-6 function tolerance; along the weakly observed directions (biases, far landmarks) that
    # leaves them ~1 % apart in cost and millimetres apart in pose
    assert abs(rd.final_cost - rl.final_cost) < 0.03 * rd.final_cost
    assert np.abs(sd.kf_pose[:, 4:] - sl.kf_pose[:, 4:]).max() < 5e-3
    ctx.close()


def test_visual_only_gba_and_equidistant_camera():
    cfg = synth.config_named("small")
    m = synth.make_map(cfg)
    mv = m.copy()
    info = Optimization.GlobalBundleAdjustment(mv, 10, -1.0, True, False, False)   # action 4/5 of covins_gba: visual only
    assert info["round2"].final_cost < info["round2"].initial_cost
    assert np.array_equal(mv.kf_velocity, m.kf_velocity) and np.array_equal(mv.kf_bias_a, m.kf_bias_a)  # untouched (opt_be.cpp:591-593)
    # equidistant model end to end: re-project the true landmarks through an equidistant camera, then optimise
    from oracle import covo
    me = m.copy()
    me.cam_dist_type[:] = capi.COVGPU_DIST_EQUIDISTANT
    me.cam_dist[:] = [-0.01, 0.02, -0.005, 0.001]
    t = truth_map(me)
    pt, idx = mapdata.flatten_gba(t, True, True)
    pt.obs_uv[:] = 0
    r, _, _, _ = covo.linearize_reprojection(pt, covo.default_options(visual_only=1, reproj_loss_a=0.0))
    rng = np.random.default_rng(0)
    me.obs_uv[idx.obs_rows] = (r * pt.obs_sigma[:, None] + rng.normal(0, 1.0, r.shape)).astype(np.float32)
    Optimization.GlobalBundleAdjustment(me, 10, -1.0, False, True, False)
    print(f"equidistant: ATE {ate(me):.4f} m")
    assert ate(me) < 0.03   # observed 0.0243


@pytest.mark.parametrize("fix_loaded", [False, True])
def test_device_second_round_equals_the_literal_two_flatten_sequence(small_map, fix_loaded):
    """covgpu_gba_two_round derives the second round's problem on the device from the resident first round (observation stream
    compacted by the erase flags, landmarks left with fewer than two dropped, loop loss switched on, optionally more constant poses).
    It must leave the map where the reference's literal sequence leaves it: flatten, solve 5, erase, flatten AGAIN, solve 10
    (optimization_be.cpp:62-293 then :296-610)."""
    from covins_amd.optimization import OptParams
    prm = OptParams(gba_fix_poses_loaded_maps=fix_loaded)
    a, b = small_map.copy(), small_map.copy()
    if fix_loaded:
        for m in (a, b):
            m.kf_loaded[: m.K // 3] = True
    ia = Optimization.GlobalBundleAdjustment(a, 10, -1.0, False, True, False, params=prm, device_second_round=True)
    ib = Optimization.GlobalBundleAdjustment(b, 10, -1.0, False, True, False, params=prm, device_second_round=False)
    assert ia["outliers_removed"] == ib["outliers_removed"] > 0
    assert ia["problem"] == ib["problem"], (ia["problem"], ib["problem"])
    assert ia["round2"].iterations == ib["round2"].iterations
    assert list(ia["round2"].accepted_trace[:10]) == list(ib["round2"].accepted_trace[:10])
    assert np.allclose(np.array(ia["round2"].cost_trace[:ia["round2"].iterations]), np.array(ib["round2"].cost_trace[:ib["round2"].iterations]), rtol=1e-12)
    assert np.array_equal(a.lm_invalid, b.lm_invalid) and np.array_equal(a.lm_gba_optimized, b.lm_gba_optimized)
    assert np.abs(a.kf_pose - b.kf_pose).max() < 1e-10 and np.abs(a.kf_velocity - b.kf_velocity).max() < 1e-10
    assert np.abs(a.lm_pos - b.lm_pos).max() < 1e-9
    print(f"device second round: {ia['outliers_removed']} observations erased, problem {ia['problem']}, "
          f"stages {dict((k, round(v, 4)) for k, v in ia['stages_s'].items())} vs literal {dict((k, round(v, 4)) for k, v in ib['stages_s'].items())}")


def test_twelve_agent_map_runs_on_one_gpu():
    """BASELINE configs[4] at its stated 12 x 1667 keyframes (2M landmarks, 8M observations; 12 x 1000 with COVGPU_TEST_A12=0) on ONE GPU (VERDICT r01 item 6 /
    row J1): no K^2 allocation is left (the system lives in the fronts of the nested-dissection tree), so the footprint reported by the
    allocator stays far below the 115 + 173 GB the dense layout would need at the stated 20k-keyframe size. Size-independent
    properties instead of an oracle run (the CPU oracle does not finish at this size): monotone accepted steps, ATE drops."""
    import os
    name = "a12x1000" if os.environ.get("COVGPU_TEST_A12") == "0" else "a12"   # a12 = the stated 20k keyframes / 2M landmarks
    from tests.util import cached_problem
    m, p = cached_problem(name)   # (shared with tests/test_gpu_full.py's oracle-parity checks at this size)
    assert p.K >= 12000 and p.L > 1_000_000
    ctx = backend.Context(0)
    o = backend.default_options(max_iterations=6)
    ctx.upload(p, o)
    lay = ctx.layout()
    assert lay["nd_fronts"] > 100 and lay["nd_levels"] >= 4
    dense_gb = (6.0 * p.K) ** 2 * 8e-9 * (1 + 1.5)   # dense C + K-major Y of the round-1 layout
    print(f"{name}: K={p.K} L={p.L} O={p.O} layout {lay} (round-1 layout would need {dense_gb:.0f} GB for C and Y alone)")
    assert lay["device_mib"] / 1024.0 < 0.5 * dense_gb
    res = ctx.solve_resident(o)
    sol = ctx.download()
    ctx.close()
    tr = np.array(res.cost_trace[:res.iterations]); acc = np.array(res.accepted_trace[:res.iterations])
    assert acc.sum() >= 4 and np.all(np.diff(tr) <= 1e-9 * tr[:-1]) and res.final_cost < 0.2 * res.initial_cost
    truth = m.truth["kf_pose"][:, 4:]
    a0, a1 = synth.ate_rmse(p.kf_pose[:, 4:], truth), synth.ate_rmse(sol.kf_pose[:, 4:], truth)
    print(f"{name}: {res.iterations} iterations in {res.t_solve_s:.2f} s, cost {res.initial_cost:.4e} -> {res.final_cost:.4e}, ATE {a0:.4f} -> {a1:.4f} m")
    assert a1 < 0.5 * a0


def test_gba_on_loaded_saved_map(small_map, tmp_path):
    """BASELINE configs[0]/[1] plumbing: a map saved in the covins_backend on-disk format (Map::SaveToFile layout) is read back
    by covins_amd.mapio and bundle-adjusted; the result equals the GBA of the in-memory original."""
    from covins_amd import mapio
    p = str(tmp_path / "saved")
    mapio.save_map(p, small_map)
    a, b = small_map.copy(), mapio.load_map(p)
    ia = Optimization.GlobalBundleAdjustment(a, 10, -1.0, False, True, False)
    ib = Optimization.GlobalBundleAdjustment(b, 10, -1.0, False, True, False)
    assert ia["outliers_removed"] == ib["outliers_removed"] and ia["round2"].iterations == ib["round2"].iterations
    assert np.abs(a.kf_pose - b.kf_pose).max() < 1e-8 and np.abs(a.kf_velocity - b.kf_velocity).max() < 1e-8
    assert b.kf_gba_optimized.all() and b.kf_loaded.all()
