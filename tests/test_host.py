"""CPU-side tests of the host logic: map flattening rules, the C-ABI library's exported surface, the
Optimization facade's bookkeeping against the oracle-driven reference flow (no GPU compute)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from covins_amd import backend, capi, mapdata, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    so = os.path.join(ROOT, "covins_amd", "libcovgpu.so")
    if not os.path.exists(so):
        backend.build()
    lib = C.CDLL(so)
    hdr = open(os.path.join(ROOT, "include", "covgpu.h")).read()
    names = sorted(set(re.findall(r"\b(covgpu_[a-z_0-9]+)\s*\(", hdr)))
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/covgpu.h but not exported"


def test_struct_layouts_match_header():
    # sizes are fixed by the field lists in include/covgpu.h (LP64)
    assert C.sizeof(capi.Options) == 4 * 4 + 12 * 8 + 2 * 4
    assert C.sizeof(capi.ProblemStruct) == 8 * 4 + 24 * 8
    assert C.sizeof(capi.Result) == 4 * 4 + 6 * 8 + 64 * 8 * 2 + 64 * 4


def test_default_options_match_reference_constants():
    o = backend.default_options()
    assert o.strategy == capi.COVGPU_DOGLEG and o.max_iterations == 10 and o.reproj_loss_a == 1.0
    assert o.initial_radius == 1e4 and o.min_relative_decrease == 1e-3 and o.function_tolerance == 1e-6
    assert abs(o.sigma_g - 1.7e-4 * np.sqrt(200)) < 1e-15 and abs(o.sigma_aw - 3e-3 / np.sqrt(200)) < 1e-15 and o.gravity == 9.81


def test_no_device_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(backend.CovGpuError, match="no HIP device"):
        backend.Context()


def test_flatten_gba_gating(tiny_map):
    m = tiny_map.copy()
    p, idx = mapdata.flatten_gba(m, False, True)
    assert p.K == m.K and p.I == m.K - 2 and p.E == len(m.loops)
    assert p.kf_fixed.sum() == 1 and p.kf_fixed[m.gauge_kf()] == 1
    assert np.all(p.obs_sigma == 2.0) and p.obs_uv.dtype == np.float64
    assert np.array_equal(p.obs_uv, m.obs_uv[idx.obs_rows].astype(np.float64))
    # invalidate a keyframe in the middle: its observations, IMU factor and loops disappear;
    # its successor has no valid predecessor -> fatal like the reference (opt_be.cpp:371-373)
    k = 9
    m.kf_invalid[k] = True
    with pytest.raises(RuntimeError, match="no predecessor"):
        mapdata.flatten_gba(m, False, True)
    p2, idx2 = mapdata.flatten_gba(m, True, True)
    assert p2.K == m.K - 1 and k not in idx2.kf_rows and p2.I == 0
    assert not np.any(m.obs_kf[idx2.obs_rows] == k)
    assert np.all(np.diff(p2.lm_obs_ptr) >= 2)
    # landmarks with a single remaining observation are dropped
    m2 = tiny_map.copy()
    l = 5
    o0, o1 = m2.lm_obs_ptr[l], m2.lm_obs_ptr[l + 1]
    mask = np.zeros(m2.O, bool); mask[o0 + 1:o1] = True
    m2.erase_observations(mask)
    p3, idx3 = mapdata.flatten_gba(m2, True, True)
    assert l not in idx3.lm_rows and p3.L == m2.L - 1
    assert m2.clean() == 1 and m2.lm_invalid[l]
    # round 1 has loop edges without loss, round 2 with Cauchy(1) (opt_be.cpp:253 vs :555)
    assert np.all(mapdata.flatten_gba(tiny_map, False, loop_loss=False)[0].edge_loss_a == 0)
    assert np.all(p.edge_loss_a == 1.0) and np.allclose(p.edge_sqrt_info[0].reshape(6, 6), np.diag([100] * 3 + [1e4] * 3))


def test_flatten_pgo_edges(tiny_map):
    m = tiny_map.copy()
    prm = mapdata.PgoParams()
    p, _ = mapdata.flatten_pgo(m, {}, prm)
    A = len(set(m.kf_client.tolist())); n_per = m.K // A
    n_succ = m.K - A
    # neighbours: KF with id k gets min(5, k - 1) extra previous-neighbour edges, the first duplicates nothing
    # (pairs are ordered (kf, other), successor edges are (pred, kf))
    n_nbr = sum(min(5, max(0, k - 1)) for k in range(n_per)) * A
    assert p.E == len(m.loops) + n_succ + n_nbr
    W1 = np.diag([100.0] * 3 + [10.0] * 3)
    assert np.allclose(p.edge_sqrt_info[len(m.loops)].reshape(6, 6), W1)
    assert p.edge_loss_a[0] == 0.5 and p.edge_loss_a[len(m.loops)] == 0.0
    infos = {tuple(np.round(np.diag(x.reshape(6, 6)), 6)) for x in p.edge_sqrt_info}
    assert infos == {(100.0,) * 3 + (10.0,) * 3, (50.0,) * 3 + (5.0,) * 3, tuple(np.round([100 / 3] * 3 + [10 / 3] * 3, 6))}
    # fixed-after-GBA rule
    m.kf_gba_optimized[:6] = True
    p2, _ = mapdata.flatten_pgo(m, {}, prm)
    assert p2.kf_fixed[:6].all() and p2.kf_fixed.sum() >= 6
    # corrected poses override the initial estimate
    T = m.kf_pose[7].copy(); T[4:] += 1.0
    p3, _ = mapdata.flatten_pgo(m, {7: T}, prm)
    assert np.array_equal(p3.kf_pose[7], T)
    # odometry measurements come from the VIO poses, not the current ones
    m.kf_pose[:, 4:] += 5.0
    p4, _ = mapdata.flatten_pgo(m, {}, prm)
    assert np.allclose(p4.edge_meas[len(m.loops):], p.edge_meas[len(m.loops):])


def test_pgo_partition_invariants():
    """The block partition behind the pose-graph solve (host-only part of k_pgo.hip): no edge joins two different
    blocks, the border is small, agents' trajectories are cut into short blocks, small graphs stay dense."""
    cfg = synth.config_named("mh123"); cfg.max_kf_per_agent = 220
    m = synth.make_map(cfg)
    p, _ = mapdata.flatten_pgo(m, {}, mapdata.PgoParams())
    blk, n = backend.pgo_partition(p.K, p.edge_i, p.edge_j)
    assert n >= 3  # at least the three agents
    bi, bj = blk[p.edge_i], blk[p.edge_j]
    inner = (bi >= 0) & (bj >= 0)
    assert np.all(bi[inner] == bj[inner])                       # blocks are independent given the border
    assert (blk < 0).sum() * 3 <= p.K                           # border stays a small part of the system
    sizes = np.bincount(blk[blk >= 0], minlength=n)
    assert sizes.min() > 0 and sizes.max() <= 0.7 * (blk >= 0).sum()
    loops = {lc.kf1 for lc in m.loops} | {lc.kf2 for lc in m.loops}
    remap = -np.ones(m.K, int); remap[np.nonzero(~m.kf_invalid)[0]] = np.arange(p.K)
    assert all(blk[remap[k]] == -1 for k in loops if remap[k] >= 0)  # loop-closure keyframes are border
    # a permutation of the keyframe numbering (the map hands them over agent-interleaved) changes nothing essential
    rng = np.random.default_rng(0)
    perm = rng.permutation(p.K).astype(np.int32)
    blk2, n2 = backend.pgo_partition(p.K, perm[p.edge_i], perm[p.edge_j])
    assert n2 >= 3 and (blk2 < 0).sum() * 3 <= p.K
    # too small for the arrow form: dense path
    tiny = synth.make_map(synth.config_named("tiny"))
    pt, _ = mapdata.flatten_pgo(tiny, {}, mapdata.PgoParams())
    blk3, n3 = backend.pgo_partition(pt.K, pt.edge_i, pt.edge_j)
    assert n3 == 0 and np.all(blk3 == -1)


def test_synthetic_map_statistics():
    m = synth.make_map(synth.config_named("small"))
    n = np.diff(m.lm_obs_ptr)
    assert n.min() >= 2 and 5 < n.mean() < 16
    assert np.bincount(m.obs_kf, minlength=m.K).max() <= 400
    assert m.obs_uv.dtype == np.float32
    # sorted by (kf_id, client): agents interleaved like the reference's std::map
    key = m.kf_id.astype(np.int64) * 100 + m.kf_client
    assert np.all(np.diff(key) > 0)
    # deterministic
    m2 = synth.make_map(synth.config_named("small"))
    assert np.array_equal(m.obs_uv, m2.obs_uv) and np.array_equal(m.kf_pose, m2.kf_pose)


def test_tum_writer(tmp_path, tiny_map):
    f = tmp_path / "kf.csv"
    mapdata.write_tum(str(f), tiny_map, client=0)
    rows = np.loadtxt(f)
    assert rows.shape == ((tiny_map.kf_client == 0).sum(), 8) and np.all(np.diff(rows[:, 0]) > 0)
