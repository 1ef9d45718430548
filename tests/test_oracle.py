"""Pins the CPU oracle's self-consistency (SURVEY.md §8c golden checks 1-5).

The reference ships no tests or golden vectors for this path and its arithmetic lives in un-vendored
dependencies, so the oracle is "parity unpinned" w.r.t. the reference; what CAN be pinned is checked here:
analytic Jacobians against central differences, preintegration against closed forms, the Schur path
against a dense solve of the full normal equations, and known-answer problems.
"""
import ctypes as C

import numpy as np
import pytest
from scipy.spatial.transform import Rotation as R

from covins_amd import capi, mapdata, synth
from covins_amd.capi import dptr
from oracle import covo
from tests.util import truth_map, rel_err, rot_angle

RNG = np.random.default_rng(7)


def rand_pose(scale=1.0):
    q = R.random(random_state=RNG.integers(1 << 30)).as_quat()
    return np.concatenate([q, RNG.normal(0, scale, 3)])


def pose_plus(pose, d):
    out = np.zeros(7)
    covo.lib().covo_pose_plus(dptr(np.ascontiguousarray(pose)), dptr(np.ascontiguousarray(d)), dptr(out))
    return out


def num_jac(f, n_in, n_out, h=1e-6):
    J = np.zeros((n_out, n_in))
    for k in range(n_in):
        d = np.zeros(n_in); d[k] = h
        J[:, k] = (f(d) - f(-d)) / (2 * h)
    return J


def test_pose_plus_is_right_perturbation():
    T = rand_pose()
    d = np.array([0.01, -0.02, 0.03, 0.1, 0.2, -0.3])
    out = pose_plus(T, d)
    Rn = R.from_quat(T[:4]) * R.from_rotvec(d[:3])
    assert np.allclose(R.from_quat(out[:4]).as_matrix(), Rn.as_matrix(), atol=1e-12)
    assert np.allclose(out[4:], T[4:] + d[3:])
    assert abs(np.linalg.norm(out[:4]) - 1) < 1e-14


@pytest.mark.parametrize("dist_type", [0, 1])
def test_reprojection_jacobians_vs_central_differences(dist_type):
    L = covo.lib()
    intr = synth.INTR.copy()
    dist = synth.DIST.copy() if dist_type == 0 else np.array([-0.01, 0.02, -0.005, 0.001])
    extr = np.concatenate([R.from_matrix(synth.TBC[:3, :3]).as_quat(), synth.TBC[:3, 3]])
    for _ in range(20):
        pose = rand_pose(2.0)
        Rwc = R.from_quat(pose[:4]) * R.from_quat(extr[:4])
        pc = pose[4:] + R.from_quat(pose[:4]).apply(extr[4:])
        lc = np.array([RNG.uniform(-1, 1), RNG.uniform(-0.6, 0.6), RNG.uniform(1, 8)])
        lm = pc + Rwc.apply(lc * np.array([lc[2], lc[2], 1.0]) * np.array([0.5, 0.5, 1]))
        kp = np.array([300.0, 200.0]); sigma = 2.0
        r = np.zeros(2); Jp = np.zeros(12); Jl = np.zeros(6)
        L.covo_reproj_residual(dptr(pose), dptr(extr), dptr(lm), dptr(intr), dptr(dist), dist_type, dptr(kp), sigma, dptr(r), dptr(Jp), dptr(Jl))

        def fp(d):
            out = np.zeros(2)
            L.covo_reproj_residual(dptr(pose_plus(pose, d)), dptr(extr), dptr(lm), dptr(intr), dptr(dist), dist_type, dptr(kp), sigma, dptr(out), None, None)
            return out

        def fl(d):
            out = np.zeros(2)
            L.covo_reproj_residual(dptr(pose), dptr(extr), dptr(lm + d), dptr(intr), dptr(dist), dist_type, dptr(kp), sigma, dptr(out), None, None)
            return out

        Jp_n, Jl_n = num_jac(fp, 6, 2), num_jac(fl, 3, 2)
        assert rel_err(Jp.reshape(2, 6), Jp_n) < 1e-6
        assert rel_err(Jl.reshape(2, 3), Jl_n) < 1e-6


def test_reprojection_behind_camera_is_zeroed():
    L = covo.lib()
    pose = np.array([0, 0, 0, 1.0, 0, 0, 0]); extr = pose.copy()
    r = np.ones(2); Jp = np.ones(12); Jl = np.ones(6)
    L.covo_reproj_residual(dptr(pose), dptr(extr), dptr(np.array([0.1, 0.1, -2.0])), dptr(synth.INTR.copy()), dptr(synth.DIST.copy()), 0,
                           dptr(np.array([1.0, 2.0])), 2.0, dptr(r), dptr(Jp), dptr(Jl))
    assert not r.any() and not Jp.any() and not Jl.any()


def test_radtan_matches_sympy_closed_form():
    sp = pytest.importorskip("sympy")
    X, Y, Z = sp.symbols("X Y Z")
    fx, fy, cx, cy = synth.INTR; k1, k2, p1, p2 = synth.DIST
    x, y = X / Z, Y / Z
    r2 = x * x + y * y
    xd = x * (1 + k1 * r2 + k2 * r2 ** 2) + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * (1 + k1 * r2 + k2 * r2 ** 2) + 2 * p2 * x * y + p1 * (r2 + 2 * y * y)
    proj = sp.Matrix([fx * xd + cx, fy * yd + cy])
    Jsym = sp.lambdify((X, Y, Z), proj.jacobian([X, Y, Z]))
    fsym = sp.lambdify((X, Y, Z), proj)
    L = covo.lib()
    ident = np.array([0, 0, 0, 1.0, 0, 0, 0])
    for _ in range(10):
        lc = np.array([RNG.uniform(-2, 2), RNG.uniform(-1.5, 1.5), RNG.uniform(2, 9)])
        r = np.zeros(2); Jl = np.zeros(6)
        L.covo_reproj_residual(dptr(ident), dptr(ident), dptr(lc), dptr(synth.INTR.copy()), dptr(synth.DIST.copy()), 0,
                               dptr(np.zeros(2)), 1.0, dptr(r), None, dptr(Jl))
        assert np.allclose(r, np.array(fsym(*lc)).ravel(), rtol=1e-13)
        assert np.allclose(Jl.reshape(2, 3), np.array(Jsym(*lc)), rtol=1e-11, atol=1e-11)


def test_between_jacobians_vs_central_differences():
    L = covo.lib()
    for _ in range(20):
        T1, T2 = rand_pose(1.0), rand_pose(1.0)
        # measurement near the true relative pose so that the error quaternion is small but non-zero
        Ra, Rb = R.from_quat(T1[:4]), R.from_quat(T2[:4])
        qm = (Ra.inv() * Rb * R.from_rotvec(RNG.normal(0, 0.05, 3))).as_quat()
        tm = Ra.inv().apply(T2[4:] - T1[4:]) + RNG.normal(0, 0.05, 3)
        meas = np.concatenate([qm, tm])
        S = np.triu(RNG.normal(0, 1, (6, 6))) + np.diag([10, 10, 10, 5, 5, 5.0])
        S = np.ascontiguousarray(S)
        r = np.zeros(6); J1 = np.zeros(36); J2 = np.zeros(36)
        L.covo_between_residual(dptr(T1), dptr(T2), dptr(meas), dptr(S), dptr(r), dptr(J1), dptr(J2))

        def f1(d):
            o = np.zeros(6); L.covo_between_residual(dptr(pose_plus(T1, d)), dptr(T2), dptr(meas), dptr(S), dptr(o), None, None); return o

        def f2(d):
            o = np.zeros(6); L.covo_between_residual(dptr(T1), dptr(pose_plus(T2, d)), dptr(meas), dptr(S), dptr(o), None, None); return o

        assert rel_err(J1.reshape(6, 6), num_jac(f1, 6, 6)) < 1e-6
        assert rel_err(J2.reshape(6, 6), num_jac(f2, 6, 6)) < 1e-6
    # zero residual at the exact measurement
    meas = np.concatenate([(Ra.inv() * Rb).as_quat(), Ra.inv().apply(T2[4:] - T1[4:])])
    L.covo_between_residual(dptr(T1), dptr(T2), dptr(meas), dptr(S), dptr(r), None, None)
    assert np.abs(r).max() < 1e-12


NOISE = np.array([synth.SIG_A, synth.SIG_G, synth.SIG_AW, synth.SIG_GW, synth.GRAVITY])


def imu_eval(first, samples, ba, bg, Ti, sbi, Tj, sbj, whiten=0, jac=True):
    r = np.zeros(15); J = np.zeros(450); d = np.zeros(11)
    covo.lib().covo_imu_residual(dptr(first), dptr(samples), samples.shape[0], dptr(ba), dptr(bg), dptr(NOISE), dptr(Ti), dptr(sbi),
                                 dptr(Tj), dptr(sbj), whiten, dptr(r), dptr(J) if jac else None, dptr(d))
    return r, J.reshape(15, 30), d


def test_preintegration_closed_form_constant_motion():
    # constant body-frame acceleration, zero rotation: dv = a t, dp = a t^2 / 2, dq = identity
    n, dt = 60, 0.005
    a = np.array([0.3, -0.2, 9.9])
    samples = np.concatenate([np.full((n, 1), dt), np.tile(a, (n, 1)), np.zeros((n, 3))], 1)
    first = np.concatenate([a, np.zeros(3)])
    I7 = np.array([0, 0, 0, 1.0, 0, 0, 0]); z9 = np.zeros(9)
    _, _, d = imu_eval(first, samples, np.zeros(3), np.zeros(3), I7, z9, I7, z9, jac=False)
    T = n * dt
    assert np.allclose(d[0:3], 0.5 * a * T * T, atol=1e-13)
    assert np.allclose(d[7:10], a * T, atol=1e-13)
    assert np.allclose(d[3:7], [0, 0, 0, 1], atol=1e-15) and abs(d[10] - T) < 1e-15
    # constant rate about z, zero acceleration: dq = Exp(w T)
    w = np.array([0, 0, 0.8])
    samples = np.concatenate([np.full((n, 1), dt), np.zeros((n, 3)), np.tile(w, (n, 1))], 1)
    _, _, d = imu_eval(np.concatenate([np.zeros(3), w]), samples, np.zeros(3), np.zeros(3), I7, z9, I7, z9, jac=False)
    assert np.allclose(R.from_quat(d[3:7]).as_rotvec(), w * T, atol=1e-6)  # first-order quaternion step, renormalised
    # bias subtraction: measuring a + ba with linearisation bias ba gives the same deltas
    ba = np.array([0.05, -0.03, 0.02])
    s2 = np.concatenate([np.full((n, 1), dt), np.tile(a + ba, (n, 1)), np.zeros((n, 3))], 1)
    _, _, d2 = imu_eval(np.concatenate([a + ba, np.zeros(3)]), s2, ba, np.zeros(3), I7, z9, I7, z9, jac=False)
    assert np.allclose(d2[0:3], 0.5 * a * T * T, atol=1e-13)


def test_imu_factor_jacobians_and_bias_jacobian():
    n, dt = 50, 0.005
    t = np.arange(1, n + 1) * dt
    acc = np.stack([0.5 * np.sin(3 * t), 0.3 * np.cos(2 * t), 9.81 + 0.2 * np.sin(5 * t)], 1)
    gyr = np.stack([0.4 * np.cos(2 * t), -0.3 * np.sin(3 * t), 0.5 * np.cos(t)], 1)
    samples = np.ascontiguousarray(np.concatenate([np.full((n, 1), dt), acc, gyr], 1))
    first = np.array([0.0, 0.3, 9.81, 0.4, 0.0, 0.5])
    ba_lin, bg_lin = np.array([0.02, -0.01, 0.03]), np.array([0.002, 0.001, -0.003])
    Ti, Tj = rand_pose(1.0), rand_pose(1.0)
    Tj[:4] = (R.from_quat(Ti[:4]) * R.from_rotvec([0.05, -0.02, 0.1])).as_quat()
    Tj[4:] = Ti[4:] + [0.1, 0.05, -0.02]
    # evaluate AT the linearisation bias: every analytic block is then exact to first order
    sbi = np.concatenate([RNG.normal(0, 0.5, 3), ba_lin, bg_lin])
    sbj = np.concatenate([RNG.normal(0, 0.5, 3), ba_lin + 1e-3, bg_lin - 1e-4])
    r, J, _ = imu_eval(first, samples, ba_lin, bg_lin, Ti, sbi, Tj, sbj)

    def f(which):
        def g(d):
            a, b, c, e = Ti, sbi, Tj, sbj
            if which == 0: a = pose_plus(Ti, d)
            if which == 1: b = sbi + d
            if which == 2: c = pose_plus(Tj, d)
            if which == 3: e = sbj + d
            return imu_eval(first, samples, ba_lin, bg_lin, a, b, c, e, jac=False)[0]
        return g

    cols = [(0, 6), (6, 15), (15, 21), (21, 30)]
    for w, (c0, c1) in enumerate(cols):
        Jn = num_jac(f(w), c1 - c0, 15, h=1e-6)
        assert rel_err(J[:, c0:c1], Jn) < 2e-6, w
    # bias Jacobian of the preintegration itself vs re-integration at a shifted bias (first-order scheme:
    # SURVEY.md A.4 quotes ~7e-4 relative for J[R,BG])
    h = 1e-4
    I7 = np.array([0, 0, 0, 1.0, 0, 0, 0]); z9 = np.zeros(9)
    _, _, d0 = imu_eval(first, samples, ba_lin, bg_lin, I7, z9, I7, z9, jac=False)
    # r_p sensitivity to ba_i equals -J[P,BA]: compare against re-integration
    for k in range(3):
        e = np.zeros(3); e[k] = h
        _, _, d1 = imu_eval(first, samples, ba_lin + e, bg_lin, I7, z9, I7, z9, jac=False)
        assert np.allclose((d1[0:3] - d0[0:3]) / h, -J[0:3, 9 + k], rtol=5e-3, atol=1e-6)
        _, _, d2 = imu_eval(first, samples, ba_lin, bg_lin + e, I7, z9, I7, z9, jac=False)
        assert np.allclose((d2[0:3] - d0[0:3]) / h, -J[0:3, 12 + k], rtol=2e-2, atol=2e-4)
        assert np.allclose((d2[7:10] - d0[7:10]) / h, -J[6:9, 12 + k], rtol=2e-2, atol=2e-3)


def test_imu_whitening_reproduces_information():
    m = synth.make_map(synth.config_named("tiny"))
    p, _ = mapdata.flatten_gba(m, False, True)
    o = covo.default_options()
    d, J, P = covo.preintegrate(p, o)
    P0 = P[0].reshape(15, 15)
    assert np.allclose(P0, P0.T, rtol=1e-9, atol=1e-30)
    sd = np.sqrt(np.diag(P0))
    # SURVEY.md A.4 sanity numbers for ~0.25-0.3 s of 200 Hz EuRoC noise
    assert 5e-5 < sd[0] < 3e-4 and 3e-5 < sd[3] < 1e-4 and 3e-4 < sd[6] < 1.5e-3 and 3e-6 < sd[9] < 2e-5 and 1e-8 < sd[12] < 2e-7
    r_w, _ = covo.linearize_imu(p, o)
    # compare ||W r||^2 with r^T P^-1 r computed independently in numpy
    f = 0
    i, j = p.imu_kf_i[f], p.imu_kf_j[f]
    s0, s1 = p.imu_sample_ptr[f], p.imu_sample_ptr[f + 1]
    r_u, _, _ = imu_eval(np.ascontiguousarray(p.imu_first[f]), np.ascontiguousarray(p.imu_samples[s0:s1]),
                         p.kf_speed_bias[j, 3:6].copy(), p.kf_speed_bias[j, 6:9].copy(), p.kf_pose[i].copy(),
                         p.kf_speed_bias[i].copy(), p.kf_pose[j].copy(), p.kf_speed_bias[j].copy(), whiten=0, jac=False)
    sc = 1.0 / sd
    Pn = P0 * sc[:, None] * sc[None, :]
    chi2 = (r_u * sc) @ np.linalg.solve(Pn, r_u * sc)
    assert abs(chi2 - (r_w[0] ** 2).sum()) / chi2 < 1e-6


def test_dense_vs_schur_step(tiny_map):
    for visual_only in (1, 0):
        p, _ = mapdata.flatten_gba(tiny_map, bool(visual_only), True)
        o = covo.default_options(visual_only=visual_only)
        for mu in (1e-8, 1e-3):
            dp_d, dl_d = covo.step(p, o, mu, dense=True)
            dp_s, dl_s = covo.step(p, o, mu, dense=False)
            # two elimination orders of one system agree to rounding x condition: the visual-only system of this 28-keyframe map at
            # mu = 1e-8 has cond(S) ~ 3e11 (weakly damped scale / gauge directions), the others stay below the 1e-7 floor
            S, _, _ = covo.schur(p, o, mu)
            live = np.nonzero(np.diag(S) != 0)[0]
            tol = max(1e-7, 1e-16 * np.linalg.cond(S[np.ix_(live, live)]))
            assert rel_err(dp_s, dp_d) < tol
            assert rel_err(dl_s, dl_d) < tol
    # the gauge keyframe's pose does not move, its speed-bias block does
    g = np.nonzero(p.kf_fixed)[0][0]
    assert not dp_s[15 * g:15 * g + 6].any() and dp_s[15 * g + 6:15 * g + 15].any()


def test_reduced_solver_matches_numpy():
    n = 150
    A = RNG.normal(0, 1, (n, n)); S = A @ A.T + n * np.eye(n); b = RNG.normal(0, 1, n)
    rc, x = covo.solve_reduced(S, b)
    assert rc == 0 and np.allclose(x, np.linalg.solve(S, b), rtol=1e-10, atol=1e-12)
    S[5, 5] = -1.0
    rc, _ = covo.solve_reduced(S, b)
    assert rc == capi.COVGPU_OK + 4  # COVGPU_ERR_NUMERIC


def test_known_answer_zero_noise_is_stationary():
    cfg = synth.config_named("tiny")
    cfg.px_noise = 0.0; cfg.imu_noise = False; cfg.loop_noise_t = 0.0; cfg.loop_noise_deg = 0.0
    m = synth.make_map(cfg)
    t = truth_map(m)
    t.obs_uv = t.obs_uv.astype(np.float64)  # no float32 rounding for the exact-zero case
    p, _ = mapdata.flatten_gba(t, True, True)
    p.obs_uv[:] = _exact_pixels(p)
    o = covo.default_options(visual_only=1)
    assert covo.cost(p, o) < 1e-18
    q, res = covo.gba_solve(p, o)
    # gradient (3) or parameter (2) tolerance before any step is taken
    assert res.termination in (2, 3) and res.accepted == 0
    assert np.array_equal(q.kf_pose, p.kf_pose) and np.array_equal(q.lm_pos, p.lm_pos)


def _exact_pixels(p):
    """Noise-free pixels through the oracle's own camera model (residual + kp)."""
    o = covo.default_options(visual_only=1, reproj_loss_a=0.0)
    z = p.copy(); z.obs_uv[:] = 0
    r, _, _, _ = covo.linearize_reprojection(z, o)
    return r * z.obs_sigma[:, None]


@pytest.mark.parametrize("strategy", [capi.COVGPU_DOGLEG, capi.COVGPU_LM])
def test_known_answer_perturbed_converges_to_truth(strategy):
    cfg = synth.config_named("tiny")
    cfg.px_noise = 0.0; cfg.loop_noise_t = 0.0; cfg.loop_noise_deg = 0.0; cfg.lm_noise = 0.01
    cfg.drift_trans = 0.005; cfg.vel_noise = 0.0; cfg.imu_noise = False
    m = synth.make_map(cfg)
    pt, _ = mapdata.flatten_gba(truth_map(m), True, True)
    exact = _exact_pixels(pt)
    p, _ = mapdata.flatten_gba(m, True, True)
    p.obs_uv[:] = exact
    # anchor scale and gauge: monocular visual-only BA has a 7-dof gauge; fix a second keyframe
    p.kf_fixed[1] = 1; p.kf_pose[1] = pt.kf_pose[1]; p.kf_pose[0] = pt.kf_pose[0]
    o = covo.default_options(visual_only=1, strategy=strategy, max_iterations=30)
    q, res = covo.gba_solve(p, o)
    assert res.final_cost < 1e-12 * max(res.initial_cost, 1.0)
    assert np.abs(q.kf_pose[:, 4:] - pt.kf_pose[:, 4:]).max() < 1e-6
    assert np.abs(q.lm_pos - pt.lm_pos).max() < 1e-5


def test_vi_gba_reduces_cost_and_stays_near_truth(tiny_map):
    p, _ = mapdata.flatten_gba(tiny_map, False, True)
    pt, _ = mapdata.flatten_gba(truth_map(tiny_map), False, True)
    for strategy in (capi.COVGPU_DOGLEG, capi.COVGPU_LM):
        o = covo.default_options(strategy=strategy)
        q, res = covo.gba_solve(p, o)
        assert res.iterations == 10 or res.termination in (1, 2, 3)
        assert res.final_cost < 1e-2 * res.initial_cost  # (initial bias estimates follow the random-walk prior now: no 1e10 start)
        assert res.final_cost < 1.2 * covo.cost(pt, o)
        assert np.linalg.norm(q.kf_pose[:, 4:] - pt.kf_pose[:, 4:], axis=1).mean() < 0.02
        tr = np.array(res.cost_trace[:res.iterations])
        assert np.all(np.diff(tr) <= 1e-9 * tr[:-1])  # monotone


def test_outlier_rule_threshold(tiny_map):
    cfg = synth.config_named("tiny"); cfg.outlier_frac = 0.05
    m = synth.make_map(cfg)
    p, _ = mapdata.flatten_gba(truth_map(m), True, True)
    o = covo.default_options(visual_only=1)
    nrm = covo.residual_norms(p, o)
    # loss-corrected norm r/sqrt(1+r^2) > 0.92  <=>  raw whitened norm > 2.347
    o0 = covo.default_options(visual_only=1, reproj_loss_a=0.0)
    raw = covo.residual_norms(p, o0)
    assert np.array_equal(nrm > 0.92, raw > 0.92 / np.sqrt(1 - 0.92 ** 2))
    frac = (nrm > 0.92).mean()
    assert 0.03 < frac < 0.09


def test_pgo_pulls_drifted_graph_together():
    cfg = synth.config_named("tiny"); cfg.drift_trans = 0.05; cfg.drift_yaw_deg = 0.5
    m = synth.make_map(cfg)
    prm = mapdata.PgoParams()
    p, _ = mapdata.flatten_pgo(m, {}, prm)
    assert p.L == 0 and p.I == 0 and p.E > 5 * p.K * 0.5
    o = covo.default_options()
    c0 = covo.cost(p, o, pgo=True)
    q, res = covo.gba_solve(p, o, pgo=True)
    assert res.final_cost < c0
    g = np.nonzero(p.kf_fixed)[0]
    assert np.array_equal(q.kf_pose[g], p.kf_pose[g])


def test_pgo_reanchor_is_rigid():
    K, Lm = 6, 40
    po = np.stack([rand_pose() for _ in range(K)]); pn = np.stack([rand_pose() for _ in range(K)])
    lm = RNG.normal(0, 3, (Lm, 3)); ref = RNG.integers(-1, K, Lm).astype(np.int32)
    vel = RNG.normal(0, 1, (K, 3))
    v2, lm2 = covo.pgo_reanchor(po, pn, vel, ref, lm)
    for l in range(Lm):
        k = ref[l]
        if k < 0:
            assert np.array_equal(lm2[l], lm[l]); continue
        ps = R.from_quat(po[k, :4]).inv().apply(lm[l] - po[k, 4:])
        assert np.allclose(lm2[l], R.from_quat(pn[k, :4]).apply(ps) + pn[k, 4:], atol=1e-12)
    assert np.allclose(np.linalg.norm(v2, axis=1), np.linalg.norm(vel, axis=1))


def test_landmark_sensitivity_is_a_property_of_the_problem(small_map):
    """Two CORRECT linear solvers inside the same oracle (own dense Cholesky vs scipy's SuperLU on the block-sparse
    system) agree on poses to 1e-8 m and on cost to 1e-10, yet a few landmarks with near-degenerate parallax land far
    apart (cond(H_ll) >= 1e6, along the viewing ray). This is why the parity criterion for landmarks is conditioned on
    cond(H_ll) (tests/util.landmark_parity) — and it also checks the sparse-solver path against the dense one."""
    from tests.util import landmark_parity
    p, _ = mapdata.flatten_gba(small_map, False, True)
    o = covo.default_options()
    covo.use_sparse_solver(enable=False)
    qd, rd = covo.gba_solve(p, o)
    covo.use_sparse_solver(min_n=0)
    try:
        qs, rs = covo.gba_solve(p, o)
    finally:
        covo.use_sparse_solver(enable=False)
    assert rd.iterations == rs.iterations and abs(rd.final_cost - rs.final_cost) < 1e-9 * rd.final_cost
    assert np.abs(qd.kf_pose - qs.kf_pose).max() < 1e-8
    n_ill, d_good, d_white = landmark_parity(qs.lm_pos, qd)
    assert d_good < 1e-6 and d_white < 1e-4 and n_ill <= max(5, p.L // 12), (n_ill, d_good, d_white)
    # thread-count independence of the oracle (fixed-order sums): 1 thread vs the default team gives identical bits
    nt = covo.lib().covo_num_threads()
    covo.lib().covo_set_num_threads(1)
    try:
        q1, _ = covo.gba_solve(p, covo.default_options(max_iterations=3))
    finally:
        covo.lib().covo_set_num_threads(nt)
    qn, _ = covo.gba_solve(p, covo.default_options(max_iterations=3))
    assert np.array_equal(q1.kf_pose, qn.kf_pose) and np.array_equal(q1.lm_pos, qn.lm_pos)


def test_relative_pose_oracle():
    """Oracle of Optimization::OptimizeRelativePose (oracle/covo_relpose.cpp): analytic Jacobian of the kNormal / kInverse
    residual pair vs central differences through the pose (+), and recovery of the true relative pose with the gross
    keypoint outliers flagged."""
    from tests.util import make_relpose_batch
    for dist_type in (0, 1):
        bt = make_relpose_batch(6, seed=3 + dist_type, dist_type=dist_type)
        i = 5
        T = bt["T0"][0]
        # (sigma = 1e4 px: |r| ~ 1e-4, so the Cauchy corrector sqrt(rho') is 1 to 1e-8 — the corrector scales J but, by Ceres'
        #  definition, is not differentiated, which a finite difference of the corrected residual would do)
        args = (bt["pB"][i], bt["pA"][i], bt["kpA"][i], bt["kpB"][i], 1e4, 1e4, bt["camA"][0], dist_type, bt["camB"][0], dist_type)
        r, J = covo.relpose_residual(T, *args)
        Jn = np.zeros((4, 6))
        for k in range(6):
            d = np.zeros(6); d[k] = 1e-6
            Tp, Tm = np.zeros(7), np.zeros(7)
            covo.lib().covo_pose_plus(covo._d(np.ascontiguousarray(T)), covo._d(d), covo._d(Tp))
            covo.lib().covo_pose_plus(covo._d(np.ascontiguousarray(T)), covo._d(-d), covo._d(Tm))
            Jn[:, k] = (covo.relpose_residual(Tp, *args, jac=False)[0] - covo.relpose_residual(Tm, *args, jac=False)[0]) / 2e-6
        assert np.abs(J - Jn).max() < 2e-5 * np.abs(J).max()
        for b in range(6):
            s = slice(bt["ptr"][b], bt["ptr"][b + 1])
            n_in, T, out = covo.relpose(bt["pB"][s], bt["pA"][s], bt["kpA"][s], bt["kpB"][s], bt["sigA"][s], bt["sigB"][s], bt["camA"][b], dist_type,
                                        bt["camB"][b], dist_type, bt["T0"][b], th_outlier=0.9)
            assert n_in == (~out).sum() >= 12 and out.sum() > 0
            assert np.abs(T[4:] - bt["Ttrue"][b][4:]).max() < 0.05 and rot_angle(T[None, :4], bt["Ttrue"][b][None, :4])[0] < 0.02
            assert np.abs(bt["T0"][b][4:] - bt["Ttrue"][b][4:]).max() > np.abs(T[4:] - bt["Ttrue"][b][4:]).max()
    # too few inliers: returns 0 and leaves the pose untouched (optimization_be.cpp:813-815)
    bt = make_relpose_batch(1, seed=9, nmin=13, nmax=13, outlier_frac=0.5)
    n_in, T, out = covo.relpose(bt["pB"], bt["pA"], bt["kpA"], bt["kpB"], bt["sigA"], bt["sigB"], bt["camA"][0], 0, bt["camB"][0], 0, bt["T0"][0], th_outlier=0.9)
    assert n_in == 0 and np.array_equal(T, bt["T0"][0]) and out.sum() >= 2
    # reference quirk, kept: problem.Evaluate returns LOSS-CORRECTED residuals, |r| / sqrt(1 + |r|^2) < 1 under Cauchy(1), so the
    # configured opt.th_outlier_align = 1.3 (config_backend.yaml:117) can never be exceeded and nothing is ever removed
    n_in, T, out = covo.relpose(bt["pB"], bt["pA"], bt["kpA"], bt["kpB"], bt["sigA"], bt["sigB"], bt["camA"][0], 0, bt["camB"][0], 0, bt["T0"][0], th_outlier=1.3)
    assert n_in == 13 and not out.any()


def test_threaded_multifrontal_solver_equals_superlu(small_map):
    """oracle/covo_mf.py (the cpu_baseline leg's threaded reduced-system solve over the nested-dissection plan) against the
    SuperLU path the goldens were made with: same trust-region trajectory, poses to round-off."""
    from covins_amd import backend
    from oracle import covo_mf
    p = mapdata.flatten_gba(small_map, False, True)[0]
    o = covo.default_options(max_iterations=4)
    try:
        covo.use_sparse_solver(min_n=100)
        q0, r0 = covo.gba_solve(p, o)
        covo_mf.set_problem(p, backend.default_options(), 4)
        covo.use_sparse_solver(min_n=100, kind="multifrontal")
        covo_mf.stats.update(calls=0)
        q1, r1 = covo.gba_solve(p, o)
    finally:
        covo.use_sparse_solver(enable=False)
    assert covo_mf.stats["calls"] >= 4 and covo_mf.stats["fronts"] >= 1
    assert r0.iterations == r1.iterations and list(r0.accepted_trace[:4]) == list(r1.accepted_trace[:4])
    assert abs(r0.final_cost - r1.final_cost) <= 1e-9 * r0.final_cost
    assert np.abs(q0.kf_pose - q1.kf_pose).max() < 1e-9 and np.abs(q0.kf_speed_bias - q1.kf_speed_bias).max() < 1e-8
