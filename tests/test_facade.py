"""The C++ facade (include/covins_gpu/optimization_gpu.hpp) on stand-in Map classes: its flattening must produce
the same IR as the Python host mirror (CPU), and its GlobalBundleAdjustment / PoseGraphOptimization must leave the
map in the same state as covins_amd.optimization.Optimization (GPU)."""
import numpy as np
import pytest

from covins_amd import mapdata, synth
from tests.facade_util import StandinMap
from tests.util import rot_angle


def _sorted_obs(ptr, kf, *cols):
    order = np.lexsort((kf, np.repeat(np.arange(len(ptr) - 1), np.diff(ptr))))
    return (kf[order],) + tuple(c[order] for c in cols)


@pytest.mark.parametrize("visual_only,round2,visitor", [(False, True, 1), (False, False, 1), (True, True, 1), (False, True, 0)])
def test_cpp_flatten_matches_python(tiny_map, visual_only, round2, visitor):
    # visitor 0: the walk over the copy Landmark::GetObservations() returns (an unmodified LandmarkBase); 1: Types::visit_observations
    from tests.facade_util import lib
    sm = StandinMap(tiny_map)
    try:
        lib().shim_use_visitor(visitor)
        f = sm.flatten_gba(visual_only, round2)
        lib().shim_use_visitor(1)
        p, _ = mapdata.flatten_gba(tiny_map, visual_only, loop_loss=round2)
        assert f["sizes"][:5] == (p.K, p.L, p.O, p.I, p.E)
        assert np.allclose(f["pose"][:, 4:], p.kf_pose[:, 4:], atol=1e-15)
        assert rot_angle(f["pose"][:, :4], p.kf_pose[:, :4]).max() < 1e-7  # quaternion -> 4x4 -> quaternion round trip
        assert np.array_equal(f["fixed"], p.kf_fixed) and np.array_equal(f["lm"], p.lm_pos)
        assert np.array_equal(f["obs_ptr"], p.lm_obs_ptr)
        # observation order inside a landmark follows std::map<KeyframePtr,...> (pointer order) in C++
        a = _sorted_obs(f["obs_ptr"], f["obs_kf"], f["uv"], f["sigma"])
        b = _sorted_obs(p.lm_obs_ptr, p.obs_kf, p.obs_uv, p.obs_sigma)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
        assert np.array_equal(f["imu_i"], p.imu_kf_i) and np.array_equal(f["imu_j"], p.imu_kf_j)
        assert np.array_equal(f["ei"], p.edge_i) and np.array_equal(f["ej"], p.edge_j) and np.array_equal(f["loss"], p.edge_loss_a)
    finally:
        sm.close()


@pytest.mark.gpu
def test_cpp_gba_matches_python_facade():
    from covins_amd.optimization import Optimization
    cfg = synth.config_named("tiny"); cfg.outlier_frac = 0.03
    m = synth.make_map(cfg)
    sm = StandinMap(m)
    try:
        sm.gba(10, visual_only=False, outlier_removal=True)
        st = sm.state()
    finally:
        sm.close()
    # the explicit release of the thread's cached context (Optimization::Shutdown): the next call builds a new one and gives the same result
    from tests.facade_util import lib
    lib().shim_shutdown()
    sm2 = StandinMap(m)
    try:
        sm2.gba(10, visual_only=False, outlier_removal=True)
        st2 = sm2.state()
    finally:
        sm2.close()
    lib().shim_shutdown()
    assert np.abs(st["pose"] - st2["pose"]).max() < 1e-9 and np.array_equal(st["lm_nobs"], st2["lm_nobs"])
    mp = m.copy()
    info = Optimization.GlobalBundleAdjustment(mp, 10, -1.0, False, True, False)
    assert info["outliers_removed"] > 0
    assert np.abs(st["pose"][:, 4:] - mp.kf_pose[:, 4:]).max() < 1e-6
    assert rot_angle(st["pose"][:, :4], mp.kf_pose[:, :4]).max() < 1e-6
    assert np.abs(st["vel"] - mp.kf_velocity).max() < 1e-6
    d = np.abs(st["lm"] - mp.lm_pos).max(axis=1)
    assert np.median(d) < 1e-7 and d.max() < 1e-2
    assert st["gba"].all() and mp.kf_gba_optimized.all()
    assert np.array_equal(st["lm_nobs"], np.diff(mp.lm_obs_ptr))          # same observations erased
    assert np.array_equal(st["lm_invalid"].astype(bool), mp.lm_invalid)   # same landmarks cleaned


@pytest.mark.gpu
def test_cpp_device_clean_leaves_the_map_where_map_clean_leaves_it():
    """Params::device_clean (opt-in): the end of the two-round call erases the landmarks Map::Clean's map check would erase
    (map_be.cpp:698-717: fewer than two entries in the observation map), from the counts of the call instead of a second copy of every
    observation map. Same map state as the call that ends with map->Clean(): a map with gross outliers (landmarks drop below two
    observations) and an invalid keyframe (observation-map entries outside the IR)."""
    from tests.facade_util import lib as shim_lib
    cfg = synth.config_named("small"); cfg.outlier_frac = 0.08; cfg.track_window = 2; cfg.min_parallax_cos = 1.0
    m = synth.make_map(cfg)
    states = []
    for on in (0, 1):
        shim_lib().shim_set_device_clean(on)
        sm = StandinMap(m)
        try:
            shim_lib().shim_set_invalid(sm.h, int(np.nonzero(m.kf_succ < 0)[0][0]))   # (an agent's LAST keyframe: an invalid predecessor is fatal, optimization_be.cpp:371-379)
            sm.gba(10, visual_only=False, outlier_removal=True)
            states.append((sm.state(), dict(StandinMap.last_stages())))
        finally:
            sm.close()
            shim_lib().shim_set_device_clean(0)
    (a, sa), (b, sb) = states
    assert "Map::Clean" in sa and "Map::Clean (from the call's counts)" in sb
    assert np.array_equal(a["lm_invalid"], b["lm_invalid"]) and np.array_equal(a["lm_nobs"], b["lm_nobs"])
    assert a["lm_invalid"].sum() > 0 and (a["lm_nobs"][a["lm_invalid"].astype(bool)] < 2).all()
    # (two object graphs: the observations of a landmark are walked in std::map<KeyframePtr> = pointer order, so the two solves round differently)
    assert np.abs(a["pose"] - b["pose"]).max() < 1e-6 and np.abs(a["vel"] - b["vel"]).max() < 1e-6   # (the pose criterion of every parity test)


@pytest.mark.gpu
def test_cpp_gba_sharded_over_two_in_process_ranks():
    """Params::n_gpus = 2: the C++ facade's GlobalBundleAdjustment through covgpu_gba_solve_multi — two contexts in this
    process (both on the one GPU of the test box: virtual ranks, the library's in-process collective), both rounds incl. the
    merged outlier decisions. Must leave the map where the single-context facade leaves it."""
    from tests.facade_util import lib as shim_lib
    cfg = synth.config_named("small"); cfg.outlier_frac = 0.03
    m = synth.make_map(cfg)
    out = []
    for n in (1, 2):
        shim_lib().shim_set_gpus(n, 0)
        sm = StandinMap(m)
        try:
            sm.gba(10, visual_only=False, outlier_removal=True)
            out.append(sm.state())
        finally:
            sm.close()
            shim_lib().shim_set_gpus(1, 0)
    a, b = out
    assert np.abs(a["pose"][:, 4:] - b["pose"][:, 4:]).max() < 1e-8 and rot_angle(a["pose"][:, :4], b["pose"][:, :4]).max() < 1e-7
    assert np.abs(a["vel"] - b["vel"]).max() < 1e-7
    assert np.array_equal(a["lm_nobs"], b["lm_nobs"]) and np.array_equal(a["lm_invalid"], b["lm_invalid"])   # same observations erased
    d = np.abs(a["lm"] - b["lm"]).max(axis=1)
    assert np.median(d) < 1e-9 and d.max() < 1e-4


@pytest.mark.gpu
def test_cpp_pgo_matches_python_facade():
    from covins_amd.optimization import Optimization
    cfg = synth.config_named("tiny"); cfg.drift_trans = 0.05; cfg.drift_yaw_deg = 0.5
    m = synth.make_map(cfg)
    corrected = {m.K - 1: m.truth["kf_pose"][m.K - 1].copy()}
    sm = StandinMap(m)
    try:
        sm.pgo(corrected)
        st = sm.state()
    finally:
        sm.close()
    mp = m.copy()
    Optimization.PoseGraphOptimization(mp, corrected)
    assert np.abs(st["pose"][:, 4:] - mp.kf_pose[:, 4:]).max() < 1e-6
    assert rot_angle(st["pose"][:, :4], mp.kf_pose[:, :4]).max() < 1e-6
    assert np.abs(st["vel"] - mp.kf_velocity).max() < 1e-6
    assert np.abs(st["lm"] - mp.lm_pos).max() < 1e-6


@pytest.mark.gpu
def test_cpp_optimize_relative_pose():
    """The facade's OptimizeRelativePose (reference signature, optimization_be.cpp:620-831) on two consecutive keyframes of the
    ground-truth map: starting 3 cm / 1.5 deg off, it lands on the true camera-to-camera transform within the pixel noise."""
    from scipy.spatial.transform import Rotation as R
    from tests.util import truth_map
    m = truth_map(synth.make_map(synth.config_named("small")))
    k1 = int(np.nonzero(m.kf_client == 0)[0][10]); k2 = int(m.kf_succ[k1])
    def T_w_c(k):
        Rws = R.from_quat(m.kf_pose[k, :4]); Rsc = R.from_quat(m.cam_extr[m.kf_cam[k], :4])
        return Rws * Rsc, m.kf_pose[k, 4:] + Rws.apply(m.cam_extr[m.kf_cam[k], 4:])
    (Ra, ta), (Rb, tb) = T_w_c(k1), T_w_c(k2)
    Rab = Ra.inv() * Rb; tab = Ra.inv().apply(tb - ta)            # camera B -> camera A
    q0 = (Rab * R.from_rotvec([0.02, -0.015, 0.01])).as_quat(); q0 = -q0 if q0[3] < 0 else q0
    T0 = np.concatenate([q0, tab + [0.03, -0.02, 0.01]])
    sm = StandinMap(m)
    try:
        n, T, removed = sm.relpose(k1, k2, T0)
    finally:
        sm.close()
    assert n >= 50 and not removed.any()        # th_outlier_align = 1.3 never removes anything (reference quirk)
    qt = Rab.as_quat(); qt = -qt if qt[3] < 0 else qt
    assert np.abs(T[4:] - tab).max() < 0.01 and rot_angle(T[None, :4], qt[None]) [0] < 3e-3
    assert np.abs(T0[4:] - tab).max() > 3 * np.abs(T[4:] - tab).max()
