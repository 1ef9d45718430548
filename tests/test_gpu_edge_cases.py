"""Ragged / degenerate inputs through the C ABI against the oracle ("empty and ragged inputs,
maximum sizes, erasures"): long tracks (multi-chunk landmark groups), keyframes without observations or IMU
factors, several constant poses, mixed camera models, landmark-free and edge-free problems."""
import numpy as np
import pytest

from covins_amd import backend, capi, mapdata, synth
from oracle import covo
from tests.util import rel_err, rot_angle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = backend.Context(0)
    yield c
    c.close()


def both(**kw):
    return backend.default_options(**kw), covo.default_options(**kw)


def check(ctx, p, g, o, pgo=False, pos_tol=1e-6):
    sol, res = (ctx.pgo_solve if pgo else ctx.gba_solve)(p, g)
    ref, rres = covo.gba_solve(p, o, pgo=pgo)
    assert res.iterations == rres.iterations and res.termination == rres.termination
    assert list(res.accepted_trace[:res.iterations]) == list(rres.accepted_trace[:rres.iterations])
    assert abs(res.final_cost - rres.final_cost) <= 1e-7 * rres.final_cost + 1e-12 * rres.initial_cost
    assert np.abs(sol.kf_pose[:, 4:] - ref.kf_pose[:, 4:]).max() < pos_tol
    assert rot_angle(sol.kf_pose[:, :4], ref.kf_pose[:, :4]).max() < 1e-7
    return sol, ref


def test_long_tracks_multi_chunk(ctx):
    cfg = synth.config_named("small"); cfg.p_fuse = 0.5; cfg.max_fused_obs = 70; cfg.track_window = 12
    cfg.max_obs_per_kf = 3000; cfg.new_lm_per_kf = 16
    m = synth.make_map(cfg)
    p = mapdata.flatten_gba(m, False, True)[0]
    n = np.diff(p.lm_obs_ptr)
    assert n.max() > 48 and (n > 16).sum() > 100       # 2-, 3- and 4-chunk landmark groups
    g, o = both()
    S, b, c = ctx.schur(p, g, 1e-8); S0, b0, c0 = covo.schur(p, o, 1e-8)
    sc = np.sqrt(np.abs(np.diag(S0)))
    assert np.abs(S / sc[:, None] / sc[None, :] - S0 / sc[:, None] / sc[None, :]).max() < 1e-9
    assert abs(c - c0) < 1e-12 * c0
    check(ctx, p, g, o)


def test_keyframes_without_observations_or_imu(ctx, tiny_map):
    m = tiny_map.copy()
    # strip every observation of three keyframes; cut one IMU chain in two by invalidating nothing but removing samples
    drop = np.isin(m.obs_kf, [5, 6, 11])
    m.erase_observations(drop)
    k = 9
    s0, s1 = m.imu_ptr[k], m.imu_ptr[k + 1]
    keep = np.ones(len(m.imu_samples), bool); keep[s0:s1] = False
    m.imu_samples = m.imu_samples[keep]
    m.imu_ptr = m.imu_ptr.copy(); m.imu_ptr[k + 1:] -= (s1 - s0)
    p, _ = mapdata.flatten_gba(m, False, True)
    assert p.I == m.K - 2 - 1 and not np.isin(p.obs_kf, [5, 6, 11]).any()
    g, o = both()
    check(ctx, p, g, o)


def test_several_constant_poses_and_landmarks_seen_only_by_them(ctx, tiny_map):
    p = mapdata.flatten_gba(tiny_map, False, True)[0]
    p.kf_fixed[[0, 1, 2, 3, 4, 5]] = 1
    g, o = both()
    sol, ref = check(ctx, p, g, o)
    fx = np.nonzero(p.kf_fixed)[0]
    assert np.array_equal(sol.kf_pose[fx], p.kf_pose[fx])
    assert np.abs(sol.kf_speed_bias[fx] - ref.kf_speed_bias[fx]).max() < 1e-6   # speed-bias of a constant pose still moves
    g, o = both(visual_only=1)
    check(ctx, p, g, o)


def test_mixed_camera_models_per_agent(ctx, tiny_map):
    p = mapdata.flatten_gba(tiny_map, False, True)[0]
    p.cam_dist_type[1] = capi.COVGPU_DIST_EQUIDISTANT
    p.cam_dist[1] = [-0.01, 0.02, -0.005, 0.001]
    p.cam_intr[1] = [430.0, 431.0, 380.0, 240.0]
    g, o = both()
    r, Jp, Jl, c = ctx.linearize_reprojection(p, g); r0, Jp0, Jl0, c0 = covo.linearize_reprojection(p, o)
    assert rel_err(r, r0) < 1e-12 and rel_err(Jp, Jp0) < 1e-12 and rel_err(Jl, Jl0) < 1e-12
    # visual-inertial (scale observable): the visual-only problem with one constant pose has a free scale, along
    # which two correct solvers may legitimately drift apart once the measurements are inconsistent with the camera
    check(ctx, p, g, o)


def test_problem_without_landmarks_or_edges(ctx, tiny_map):
    p = mapdata.flatten_gba(tiny_map, False, True)[0]
    q = capi.FlatProblem(kf_pose=p.kf_pose, kf_speed_bias=p.kf_speed_bias, kf_fixed=p.kf_fixed, kf_cam=p.kf_cam, cam_extr=p.cam_extr,
                         cam_intr=p.cam_intr, cam_dist=p.cam_dist, cam_dist_type=p.cam_dist_type, imu_kf_i=p.imu_kf_i, imu_kf_j=p.imu_kf_j,
                         imu_sample_ptr=p.imu_sample_ptr, imu_samples=p.imu_samples, imu_first=p.imu_first)
    assert q.L == 0 and q.O == 0 and q.E == 0
    g, o = both()
    check(ctx, q, g, o, pos_tol=1e-5)   # inertial-only: poses are observable only through double integration


def test_single_keyframe_and_two_keyframe_problems(ctx, tiny_map):
    p = mapdata.flatten_pgo(tiny_map, {}, mapdata.PgoParams())[0]
    one = capi.FlatProblem(kf_pose=p.kf_pose[:1], kf_speed_bias=p.kf_speed_bias[:1], kf_fixed=[1], kf_cam=[0], cam_extr=p.cam_extr,
                           cam_intr=p.cam_intr, cam_dist=p.cam_dist, cam_dist_type=p.cam_dist_type)
    g, o = both()
    sol, res = ctx.pgo_solve(one, g)
    assert res.iterations == 0 and np.array_equal(sol.kf_pose, one.kf_pose)       # nothing to optimise: gradient tolerance at once
    two = capi.FlatProblem(kf_pose=p.kf_pose[:2], kf_speed_bias=p.kf_speed_bias[:2], kf_fixed=[1, 0], kf_cam=[0, 0], cam_extr=p.cam_extr,
                           cam_intr=p.cam_intr, cam_dist=p.cam_dist, cam_dist_type=p.cam_dist_type, edge_i=[0], edge_j=[1],
                           edge_meas=[[0, 0, 0, 1.0, 0.3, -0.2, 0.1]], edge_sqrt_info=np.eye(6).reshape(1, 36) * 10, edge_loss_a=[0.0])
    sol, ref = check(ctx, two, g, o, pgo=True)
    # the single edge can be satisfied exactly: relative pose equals the measurement
    from scipy.spatial.transform import Rotation as R
    Ra = R.from_quat(sol.kf_pose[0, :4])
    assert np.allclose(Ra.inv().apply(sol.kf_pose[1, 4:] - sol.kf_pose[0, 4:]), [0.3, -0.2, 0.1], atol=1e-9)


def test_two_round_call_degenerate_cases(small_map):
    """covgpu_gba_two_round at its corners: (i) a threshold nothing exceeds — the second round's problem is the first's with the loop
    loss switched on: equal to solving that problem directly; (ii) second round without the loop edges (opt.gba_use_map_loop_constraints
    = 0); (iii) visual-only; (iv) a threshold everything exceeds — every landmark is dropped, the second round is the inertial / loop
    problem alone and the landmark array comes back untouched."""
    import numpy as np
    from covins_amd import backend, mapdata
    ctx = backend.Context(0)
    try:
        p1, _ = mapdata.flatten_gba(small_map, False, loop_loss=False, use_loops=True)
        o = backend.default_options(max_iterations=6)
        # (i)
        sol, r1, r2, bad, left, (nb, ns) = ctx.gba_two_round(p1, o, 1e9)
        assert nb == 0 and not bad.any() and (left == np.diff(p1.lm_obs_ptr)).all()
        p2, _ = mapdata.flatten_gba(small_map, False, loop_loss=True, use_loops=True)
        ref, rr = ctx.gba_solve(p2, o)
        assert r1.iterations == 5 and r2.iterations == rr.iterations and list(r2.accepted_trace[:6]) == list(rr.accepted_trace[:6])
        assert np.allclose(np.array(r2.cost_trace[:r2.iterations]), np.array(rr.cost_trace[:rr.iterations]), rtol=1e-12)
        assert np.abs(sol.kf_pose - ref.kf_pose).max() < 1e-10 and np.abs(sol.lm_pos - ref.lm_pos).max() < 1e-9
        # (ii)
        sol, r1, r2, bad, left, _ = ctx.gba_two_round(p1, o, 1e9, use_loops_round2=False)
        p3, _ = mapdata.flatten_gba(small_map, False, loop_loss=True, use_loops=False)
        ref, rr = ctx.gba_solve(p3, o)
        assert p3.E == 0 and p1.E > 0
        assert np.allclose(np.array(r2.cost_trace[:r2.iterations]), np.array(rr.cost_trace[:rr.iterations]), rtol=1e-12)
        assert np.abs(sol.kf_pose - ref.kf_pose).max() < 1e-10
        # (iii)
        pv, _ = mapdata.flatten_gba(small_map, True, loop_loss=False, use_loops=True)
        ov = backend.default_options(max_iterations=6, visual_only=1)
        sol, r1, r2, bad, left, (nb, ns) = ctx.gba_two_round(pv, ov, 0.92)
        assert nb > 0 and r2.final_cost < r2.initial_cost and np.array_equal(sol.kf_speed_bias, pv.kf_speed_bias)
        # (iv)
        sol, r1, r2, bad, left, (nb, ns) = ctx.gba_two_round(p1, o, 0.0)
        assert bad.all() and (left == 0).all() and ns == p1.L
        assert np.array_equal(sol.lm_pos, p1.lm_pos) and r2.final_cost <= r2.initial_cost and np.isfinite(sol.kf_pose).all()
    finally:
        ctx.close()
