"""Nested-dissection plan of the multifrontal GBA solve (covgpu_nd_plan_*, covins_amd/csrc/nd_plan.hip): host-only.

The plan is replayed in numpy on the ORACLE's reduced camera system (oracle/covo.schur): fronts are assembled from the
system's entries exactly where the device kernels put them (entry between two variables -> front of the deeper owner),
factorised partially, extend-added into their parents and back-substituted top-down. The result must equal the dense
solve: this pins the separator property, the symbolic structure (no fill outside the fronts) and the batch levels."""
import ctypes as C

import numpy as np
import pytest

from covins_amd import backend, mapdata, synth
from oracle import covo


def _plan(prob, opt, leaf, pgo=False):
    lib = backend.lib()
    h = C.c_void_p()
    s = prob.as_struct()
    rc = (lib.covgpu_nd_plan_create_pgo if pgo else lib.covgpu_nd_plan_create)(C.byref(opt), C.byref(s), leaf, C.byref(h))
    assert rc == 0, lib.covgpu_last_error()
    info = (C.c_int64 * 16)()
    lib.covgpu_nd_plan_info(h, info)
    nn = info[0]
    parent = np.zeros(nn, np.int32); level = np.zeros(nn, np.int32)
    optr = np.zeros(nn + 1, np.int32); sptr = np.zeros(nn + 1, np.int32)
    ov = np.zeros(max(info[3], 1), np.int32); sv = np.zeros(max(info[4], 1), np.int32)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    lib.covgpu_nd_plan_arrays(h, ip(parent), ip(level), ip(optr), ip(ov), ip(sptr), ip(sv))
    lib.covgpu_nd_plan_destroy(h)
    own = [ov[optr[n]:optr[n + 1]] for n in range(nn)]
    st = [sv[sptr[n]:sptr[n + 1]] for n in range(nn)]
    return list(info), parent, level, own, st


def _scalars(vs, D):
    """IR rows of a list of variables (2 kf = pose rows 0..5, 2 kf + 1 = speed-bias rows 6..14)."""
    out = []
    for v in vs:
        kf, t = int(v) >> 1, int(v) & 1
        out.extend(range(D * kf + (6 if t else 0), D * kf + (15 if t else 6)))
    return np.array(out, int)


def _replay(S, b, parent, level, own, st, D):
    nn = len(parent)
    n = S.shape[0]
    node_of = -np.ones(n, int)
    rows_own = [_scalars(own[k], D) for k in range(nn)]
    rows_st = [_scalars(st[k], D) for k in range(nn)]
    for k in range(nn):
        assert np.all(node_of[rows_own[k]] == -1)
        node_of[rows_own[k]] = k
    assert np.all(node_of >= 0), "every unknown is owned by exactly one node"
    depth = np.zeros(nn, int)
    for k in range(nn):
        if parent[k] >= 0:
            assert parent[k] < k
            depth[k] = depth[parent[k]] + 1
    # every structural non-zero must be representable: between two own rows of one node, or own row x border row of the deeper owner
    covered = np.zeros_like(S, dtype=bool)
    fronts = []
    for k in range(nn):
        idx = np.r_[rows_own[k], rows_st[k]]
        assert np.all(depth[node_of[rows_st[k]]] < depth[k])
        F = np.zeros((len(idx), len(idx)))
        m = len(rows_own[k])
        F[:m, :m] = S[np.ix_(rows_own[k], rows_own[k])]
        F[m:, :m] = S[np.ix_(rows_st[k], rows_own[k])]
        F[:m, m:] = F[m:, :m].T
        covered[np.ix_(rows_own[k], idx)] = True
        covered[np.ix_(idx, rows_own[k])] = True
        fronts.append((idx, m, F, np.r_[b[rows_own[k]], np.zeros(len(rows_st[k]))]))
    assert not np.any((S != 0) & ~covered), "a non-zero of the system has no place in any front"
    x = np.zeros(n)
    order = np.argsort(level, kind="stable")
    facs = {}
    for k in order:                      # bottom-up: children are at lower levels
        idx, m, F, r = fronts[k]
        L11 = np.linalg.cholesky(F[:m, :m])
        L21 = np.linalg.solve(L11, F[:m, m:]).T
        y1 = np.linalg.solve(L11, r[:m])
        U = F[m:, m:] - L21 @ L21.T      # Schur complement on the border (original border-border entries live higher up: zero here)
        rb = r[m:] - L21 @ y1
        facs[k] = (L11, L21, y1)
        p = parent[k]
        if p >= 0:
            pidx = fronts[p][0]
            where = {int(g): i for i, g in enumerate(pidx)}
            loc = np.array([where[int(g)] for g in idx[m:]], dtype=int)   # KeyError here = fill outside the parent's front
            fronts[p][2][np.ix_(loc, loc)] += U
            fronts[p][3][loc] += rb
        else:
            assert len(idx) == m
    for k in order[::-1]:                # top-down
        idx, m, F, r = fronts[k]
        L11, L21, y1 = facs[k]
        x[idx[:m]] = np.linalg.solve(L11.T, y1 - L21.T @ x[idx[m:]])
    return x


@pytest.mark.parametrize("leaf", [90, 200, 100000, 0x3fffffff])
def test_replay_equals_dense_solve(small_map, leaf):
    prob, _ = mapdata.flatten_gba(small_map, visual_only=False, loop_loss=True)
    opt = covo.default_options()
    S, bvec, _ = covo.schur(prob, opt, 1e-4)
    info, parent, level, own, st = _plan(prob, backend.default_options(), leaf)
    if leaf >= 0x3fffffff:
        assert info[0] == 1 and info[1] == 1     # one front = the dense system (COVGPU_GBA_DENSE)
    elif leaf >= 100000:
        # nothing is cut: the pose front (+ the cut blocks of the speed-bias chains) and, below it, chain segments of <= 14 blocks
        assert info[1] == 2 and all(p == 0 for p in parent[1:])
        assert all(len(o) <= 14 and all(v & 1 for v in o) for o in own[1:])
        assert sum(len(o) for o in own) == 2 * prob.K
    else:
        assert info[0] > 3 and info[1] >= 2
    x = _replay(S, bvec, parent, level, own, st, 15)
    xd = np.linalg.solve(S, bvec)
    assert np.abs(x - xd).max() <= 1e-9 * np.abs(xd).max()


def test_visual_only_plan(tiny_map):
    prob, _ = mapdata.flatten_gba(tiny_map, visual_only=True, loop_loss=True)
    opt = covo.default_options(visual_only=1)
    S, bvec, _ = covo.schur(prob, opt, 1e-3)
    info, parent, level, own, st = _plan(prob, backend.default_options(visual_only=1), 40)
    assert all((v & 1) == 0 for o in own for v in o)
    x = _replay(S, bvec, parent, level, own, st, 6)
    xd = np.linalg.solve(S, bvec)
    assert np.abs(x - xd).max() <= 1e-9 * np.abs(xd).max()


def test_plan_is_deterministic_and_levels_are_heights(small_map):
    prob, _ = mapdata.flatten_gba(small_map, visual_only=False, loop_loss=True)
    a = _plan(prob, backend.default_options(), 150)
    b = _plan(prob, backend.default_options(), 150)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and all(np.array_equal(x, y) for x, y in zip(a[3], b[3]))
    parent, level = a[1], a[2]
    h = np.zeros(len(parent), int)
    for k in range(len(parent) - 1, -1, -1):
        if parent[k] >= 0:
            h[parent[k]] = max(h[parent[k]], h[k] + 1)
    assert np.array_equal(h, level)


def _replay_sparse(S, b, parent, level, own, st, D):
    """_replay on a scipy CSR system: the same multifrontal elimination, fronts gathered by sparse slicing (full-size maps)."""
    nn, n = len(parent), S.shape[0]
    rows_own = [_scalars(own[k], D) for k in range(nn)]
    rows_st = [_scalars(st[k], D) for k in range(nn)]
    node_of = -np.ones(n, int)
    for k in range(nn):
        assert np.all(node_of[rows_own[k]] == -1)
        node_of[rows_own[k]] = k
    assert np.all(node_of >= 0)
    depth = np.zeros(nn, int)
    for k in range(nn):
        if parent[k] >= 0:
            assert parent[k] < k
            depth[k] = depth[parent[k]] + 1
    # every structural non-zero between two nodes lies in the front of the deeper one
    C = S.tocoo()
    na, nb = node_of[C.row], node_of[C.col]
    deeper = np.where(depth[na] >= depth[nb], na, nb); other_row = np.where(depth[na] >= depth[nb], C.col, C.row)
    cross = (na != nb) & (C.data != 0)   # (the block-CSR source stores whole 15x15 blocks: explicit zeros are no couplings)
    mask = np.zeros(n, bool)
    for k in np.unique(deeper[cross]):
        mask[:] = False; mask[rows_st[k]] = True
        sel = cross & (deeper == k)
        assert mask[other_row[sel]].all(), "a non-zero of the system has no place in any front"
    fronts = []
    for k in range(nn):
        idx = np.r_[rows_own[k], rows_st[k]]
        m = len(rows_own[k])
        F = np.zeros((len(idx), len(idx)))
        F[:, :m] = S[idx][:, rows_own[k]].toarray()
        F[:m, m:] = F[m:, :m].T
        fronts.append([idx, m, F, np.r_[b[rows_own[k]], np.zeros(len(rows_st[k]))]])
    x = np.zeros(n)
    facs = {}
    for k in np.argsort(level, kind="stable"):
        idx, m, F, r = fronts[k]
        L11 = np.linalg.cholesky(F[:m, :m])
        L21 = np.linalg.solve(L11, F[:m, m:]).T
        y1 = np.linalg.solve(L11, r[:m])
        facs[k] = (L11, L21, y1)
        p = parent[k]
        if p >= 0:
            where = -np.ones(n, int); where[fronts[p][0]] = np.arange(len(fronts[p][0]))
            loc = where[idx[m:]]
            assert (loc >= 0).all(), "fill outside the parent's front"
            fronts[p][2][np.ix_(loc, loc)] += F[m:, m:] - L21 @ L21.T
            fronts[p][3][loc] += r[m:] - L21 @ y1
        else:
            assert len(idx) == m
        fronts[k][2] = None
    for k in np.argsort(level, kind="stable")[::-1]:
        idx, m = fronts[k][0], fronts[k][1]
        L11, L21, y1 = facs[k]
        x[idx[:m]] = np.linalg.solve(L11.T, y1 - L21.T @ x[idx[m:]])
    return x


def test_replay_at_three_agent_size():
    """The plan the GPU solves with on the 3-agent BASELINE map — speed-bias chains cut into segments, excess unknowns of a level's
    widest fronts moved into the parents, ~200 fronts in 7 levels (round 5, ground-truth orientations read correctly: root of 1 050 unknowns, 13 serial panels) — replayed on the
    oracle's sparse reduced camera system: the residual of the multifrontal solution is at rounding level."""
    import scipy.sparse as sp
    m = synth.make_map(synth.config_named("mh123"))
    prob, _ = mapdata.flatten_gba(m, visual_only=False, loop_loss=True)
    ptr, col, blocks, bvec, _ = covo.schur_sparse(prob, covo.default_options(), 1e-4)
    S = sp.bsr_matrix((np.asarray(blocks), np.asarray(col), np.asarray(ptr)), shape=(15 * prob.K, 15 * prob.K)).tocsr()
    info, parent, level, own, st = _plan(prob, backend.default_options(), 0)
    od = np.array([sum(9 if v & 1 else 6 for v in o) for o in own])
    assert info[0] > 150 and info[1] >= 6
    assert max(od[level == 0]) <= 126                       # bottom level: speed-bias segments of <= 14 blocks
    widest = [od[level == l].max() for l in range(info[1])]
    assert sum(-(-w // 256) for w in widest) <= 13          # serial 256-column panels of the whole factorisation
    x = _replay_sparse(S, np.asarray(bvec), parent, level, own, st, 15)
    r = S @ x - bvec
    assert np.linalg.norm(r) <= 1e-9 * np.linalg.norm(bvec)


@pytest.mark.parametrize("keep_every", [4, 6])
def test_many_short_imu_chains_plan_in_milliseconds(small_map, keep_every):
    """every keyframe without an IMU predecessor starts a chain, and the two-groups cut enumerated 2^(chains-1) bipartitions — 26 chains
    took 49 s, 33 were undefined behaviour. A map of short tracking sessions (here: every `keep_every`-th IMU factor of the small map dropped, 30+
    chains) must plan in well under a second through the local-search bipartition, and the plan must still replay to the dense solve."""
    import time
    prob, _ = mapdata.flatten_gba(small_map, visual_only=False, loop_loss=True)
    q = prob.copy()
    keep = np.ones(q.I, bool); keep[::keep_every] = False
    ptr = q.imu_sample_ptr
    sel = np.concatenate([np.arange(ptr[i], ptr[i + 1]) for i in np.nonzero(keep)[0]]) if keep.any() else np.zeros(0, int)
    q.imu_samples = q.imu_samples[sel]
    q.imu_sample_ptr = np.r_[0, np.cumsum((ptr[1:] - ptr[:-1])[keep])].astype(np.int32)
    q.imu_kf_i, q.imu_kf_j, q.imu_first = q.imu_kf_i[keep], q.imu_kf_j[keep], q.imu_first[keep]
    if q.imu_noise is not None:
        q.imu_noise = q.imu_noise[keep]
    nchains = q.K - q.I     # a chain per keyframe without a predecessor
    assert nchains >= 30
    t0 = time.perf_counter()
    info, parent, level, own, st = _plan(q, backend.default_options(), 0)      # leaf 0: the default candidates, two-groups cut included
    dt = time.perf_counter() - t0
    assert dt < 2.0, f"planning {nchains} chains took {dt:.2f} s"
    info2, parent2, _, own2, _ = _plan(q, backend.default_options(), 0)
    assert info == info2 and np.array_equal(parent, parent2) and all(np.array_equal(a, b) for a, b in zip(own, own2))   # deterministic
    S, bvec, _ = covo.schur(q, covo.default_options(), 1e-4)
    x = _replay(S, bvec, parent, level, own, st, 15)
    xd = np.linalg.solve(S, bvec)
    assert np.abs(x - xd).max() <= 1e-9 * np.abs(xd).max()


@pytest.mark.parametrize("name,kf,leaf", [("small", None, 60), ("mh123", 120, 0)])
def test_pose_graph_plan_replays_to_the_dense_solve(name, kf, leaf):
    """Round 6: PoseGraphOptimization's linear solve runs on the elimination tree too (covgpu_nd_plan_create_pgo: 6-dof blocks, an agent's time axis
    read from the edge graph — connected components of the graph without its bridges, each in breadth-first order). The plan replayed in numpy on
    the ORACLE's pose-graph system (optimization_be.cpp:846-1021 assembled by oracle/covo.schur(pgo=True)) must equal the dense solve: every edge's
    block has its place in a front, no fill outside the fronts; the chains must be few (one per agent, not one per keyframe) and the separators of
    the time axis narrow."""
    cfg = synth.config_named(name)
    if kf:
        cfg.max_kf_per_agent = kf
    cfg.drift_trans = 0.05; cfg.drift_yaw_deg = 0.5
    m = synth.make_map(cfg)
    p = mapdata.flatten_pgo(m, {}, mapdata.PgoParams())[0]
    S, bvec, _ = covo.schur(p, covo.default_options(), 1e-6, pgo=True)
    info, parent, level, own, st = _plan(p, backend.default_options(), leaf, pgo=True)
    assert all((v & 1) == 0 for o in own for v in o)
    x = _replay(S, bvec, parent, level, own, st, 6)
    xd = np.linalg.solve(S, bvec)
    assert np.abs(x - xd).max() <= 1e-8 * np.abs(xd).max()
    od = np.array([6 * len(o) for o in own])
    if info[0] > 3:
        # the root separates agents (a cover of the few loop edges) or halves of a time axis (~5 keyframes): far below the system's order
        assert od[np.asarray(parent) < 0].max() <= max(90, 6 * p.K // 8)
