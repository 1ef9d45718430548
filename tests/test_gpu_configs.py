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
    assert info["outliers_removed"] >= 0 and a2 < a0 and a2 < 0.02 and a2 <= a1 * 1.05
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
    assert ate(m) < 0.01
    # without the outlier round the Cauchy loss still keeps the estimate sane, but worse
    m2 = synth.make_map(cfg)
    Optimization.GlobalBundleAdjustment(m2, 10, -1.0, False, False, False)
    assert m2.O == n_obs and ate(m2) < 0.05


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
    assert ate(me) < 0.01
