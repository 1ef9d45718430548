"""Agent-sharded multi-GPU solve (covins_amd/distrib.py, covgpu_shard_plan / covgpu_set_shard) — CPU part.
  * the plan: every residual has exactly one owner, a landmark's interior observers all belong to its owner's blocks,
    sub-problems partition the full problem, block loads are balanced (LPT);
  * world_size-2 `gloo`: each rank linearises ITS share with the oracle, packs the shared-pose (border) Hessian blocks and
    gradient rows exactly as the device layout orders them, all-reduces the packed buffers with the reducer bench.py uses,
    and obtains the shared-pose blocks of the FULL problem — the identity the device path relies on."""
import os

import numpy as np
import torch.multiprocessing as mp

from covins_amd import backend, distrib, mapdata, synth
from oracle import covo


def _problem(name="small"):
    m = synth.make_map(synth.config_named(name))
    return mapdata.flatten_gba(m, False, True)[0]


def test_shard_plan_invariants():
    p = _problem("mh123")
    o = backend.default_options()
    for world in (2, 3, 4):
        pl = distrib.shard_plan(p, o, world)
        assert pl is not None and pl.num_blocks == 3 and pl.block_rank.max() < world
        # deterministic
        pl2 = distrib.shard_plan(p, o, world)
        assert np.array_equal(pl.block_of_kf, pl2.block_of_kf) and np.array_equal(pl.lm_rank, pl2.lm_rank)
        subs = [distrib.shard_problem(p, pl, r) for r in range(world)]
        assert sum(s.L for s in subs) == p.L and sum(s.O for s in subs) == p.O
        assert sum(s.I for s in subs) == p.I and sum(s.E for s in subs) == p.E
        pose_rank = pl.pose_owner()
        shared = (pl.block_of_kf < 0) | p.kf_fixed.astype(bool)   # (a constant keyframe has no pose block: it binds nobody)
        for r, s in enumerate(subs):
            assert s.K == p.K
            # my landmarks are seen from my interiors and from shared keyframes only
            seen = np.unique(s.obs_kf)
            assert np.all(shared[seen] | (pose_rank[seen] == r))
            # my IMU factors and between factors touch my interiors and shared keyframes only
            for arr in (s.imu_kf_i, s.imu_kf_j, s.edge_i, s.edge_j):
                assert np.all(shared[arr] | (pose_rank[arr] == r))
        # LPT: with as many ranks as blocks every rank gets exactly one
        if world == 3:
            assert sorted(pl.block_rank.tolist()) == [0, 1, 2]
    # merge puts every piece back where its owner had it
    pl = distrib.shard_plan(p, o, 2)
    parts = []
    for r in range(2):
        s = distrib.shard_problem(p, pl, r).copy()
        s.kf_pose[:] = r + 1.0; s.kf_speed_bias[:] = r + 1.0; s.lm_pos[:] = r + 1.0
        parts.append(s)
    mg = distrib.merge_solution(p, pl, parts)
    assert np.array_equal(mg.kf_pose[:, 0], pl.pose_owner() + 1.0)
    assert np.array_equal(mg.lm_pos[:, 0], pl.lm_rank + 1.0)
    assert np.array_equal(mg.kf_speed_bias[:, 0], distrib.chain_owner(p, pl) + 1.0)
    # a single agent does not split
    assert distrib.shard_plan(_problem("mh01"), o, 2) is None


def _pack_border(p, pl, ptr, col, blocks, b):
    """[C_b lower (nb x nb, border order = IR keyframe order) | b_b] like the device's contiguous border buffer; rows that
    no residual of this share touches carry the oracle's placeholder diagonal 1 — removed, as the device leaves them 0."""
    bk = np.nonzero(pl.block_of_kf < 0)[0]
    idx = -np.ones(p.K, np.int64); idx[bk] = np.arange(len(bk))
    nb = 6 * len(bk)
    Cb = np.zeros((nb, nb)); bb = np.zeros(nb)
    for i in bk:
        for q in range(ptr[i], ptr[i + 1]):
            j = col[q]
            if idx[j] < 0 or idx[j] > idx[i]:
                continue
            Cb[6 * idx[i]:6 * idx[i] + 6, 6 * idx[j]:6 * idx[j] + 6] = blocks[q][:6, :6]
        bb[6 * idx[i]:6 * idx[i] + 6] = b[15 * i:15 * i + 6]
    Cb = np.tril(Cb)
    d = np.diag(Cb).copy()
    untouched = (d == 1.0) & (np.abs(Cb - np.diag(d)).sum(1) == 0) & (bb == 0)
    Cb[np.diag_indices(nb)] = np.where(untouched, 0.0, d)
    return np.concatenate([Cb.reshape(-1), bb])


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import ctypes as C
    dist = distrib.init("gloo")
    assert dist is not None and dist.get_world_size() == world
    p = _problem("small")
    o = backend.default_options()
    pl = distrib.shard_plan(p, o, world)       # same plan on every rank, no communication
    mine = distrib.shard_problem(p, pl, rank)
    buf = _pack_border(p, pl, *covo.schur_sparse(mine, covo.default_options(), 0.0)[:4])
    red = distrib.TorchReducer(dist, "cpu")
    cb = red.callback()
    cb(None, buf.ctypes.data_as(C.POINTER(C.c_double)), len(buf), 0, 0)   # the call libcovgpu makes (stage_on_host form)
    mx = np.array([float(rank), 7.0 - rank])
    cb(None, mx.ctypes.data_as(C.POINTER(C.c_double)), 2, 1, 0)
    dt, its = distrib.aggregate(1.0 + rank, 10, dist, "cpu")
    q.put((rank, buf, mx, dt, its, mine.L))
    distrib.barrier(dist, "cpu")
    dist.destroy_process_group()


def test_two_rank_gloo_border_reduction():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs: pr.start()
    out = sorted((q.get(timeout=300) for _ in range(2)), key=lambda t: t[0])
    for pr in procs: pr.join(timeout=60)
    assert all(pr.exitcode == 0 for pr in procs)
    (_, b0, m0, dt0, it0, l0), (_, b1, m1, dt1, it1, l1) = out
    assert np.array_equal(b0, b1) and np.array_equal(m0, m1) and list(m0) == [1.0, 7.0]
    assert dt0 == dt1 == 2.0 and it0 == it1 == 10.0   # MAX of time; every rank ran the SAME 10 iterations
    p = _problem("small")
    pl = distrib.shard_plan(p, backend.default_options(), 2)
    assert l0 + l1 == p.L and l0 > 0 and l1 > 0
    full = _pack_border(p, pl, *covo.schur_sparse(p, covo.default_options(), 0.0)[:4])
    scale = np.abs(full).max()
    assert np.abs(b0 - full).max() <= 1e-12 * scale


def test_single_process_is_identity():
    assert distrib.aggregate(0.5, 7, None) == (0.5, 7.0)
    assert distrib.throughput(2.0, 10.0) == 5.0
