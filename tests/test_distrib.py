"""Sub-map-sharded multi-GPU solve (covins_amd/distrib.py, covgpu_shard_plan) — CPU part.
  * the plan: every residual has exactly one owner; what a rank's residuals touch lies in its own subtrees or in the replicated
    top of the elimination tree; the sub-problems partition the full problem; loads are balanced (LPT over subtrees);
  * world_size-2 `gloo`: each rank linearises ITS share with the oracle; the rows / blocks of the reduced camera system that
    belong to the TOP unknowns are summed across the ranks with a gloo all-reduce and must equal the full problem's, while
    every other block of the system is produced by exactly one rank — the identity the device path relies on (one
    all-reduce over the top fronts per linear solve, nothing else exchanged)."""
import os

import numpy as np
import torch.multiprocessing as mp

from covins_amd import backend, capi, distrib, mapdata, synth
from oracle import covo


def _problem(name="small"):
    m = synth.make_map(synth.config_named(name))
    return mapdata.flatten_gba(m, False, True)[0]


def test_shard_plan_invariants():
    p = _problem("mh123")
    o = backend.default_options()
    for world in (1, 2, 3, 4, 8):
        pl = distrib.shard_plan(p, o, world)
        assert pl is not None and pl.subtrees >= min(world, 3)
        assert pl.node_rank.max() < world and (pl.node_rank == -1).any()
        # deterministic
        pl2 = distrib.shard_plan(p, o, world)
        assert np.array_equal(pl.node_rank, pl2.node_rank) and np.array_equal(pl.lm_rank, pl2.lm_rank) and np.array_equal(pl.pose_rank, pl2.pose_rank)
        subs = [distrib.shard_problem(p, pl, r) for r in range(world)]
        assert sum(s.L for s in subs) == p.L and sum(s.O for s in subs) == p.O
        assert sum(s.I for s in subs) == p.I and sum(s.E for s in subs) == p.E
        fixed = p.kf_fixed.astype(bool)
        for r, s in enumerate(subs):
            assert s.K == p.K
            # my landmarks are seen from keyframes whose pose is mine or top (a constant keyframe has no pose unknown: it binds nobody)
            seen = np.unique(s.obs_kf)
            assert np.all(fixed[seen] | (pl.pose_rank[seen] == r) | (pl.pose_rank[seen] < 0))
            # my IMU factors touch poses and speed-bias blocks that are mine or top; my between factors likewise
            for arr in (s.imu_kf_i, s.imu_kf_j):
                assert np.all((pl.pose_rank[arr] == r) | (pl.pose_rank[arr] < 0)) and np.all((pl.sb_rank[arr] == r) | (pl.sb_rank[arr] < 0))
            for arr in (s.edge_i, s.edge_j):
                assert np.all(fixed[arr] | (pl.pose_rank[arr] == r) | (pl.pose_rank[arr] < 0))
        if 1 < world <= 4:   # balance: no rank carries more than 1.7x its fair share of the observations
            load = np.array([s.O for s in subs], float)
            assert load.max() <= 1.7 * load.sum() / world + 0.1 * load.sum()
    # merge puts every piece back where its owner had it
    pl = distrib.shard_plan(p, o, 2)
    parts = []
    for r in range(2):
        s = distrib.shard_problem(p, pl, r).copy()
        s.kf_pose[:] = r + 1.0; s.kf_speed_bias[:] = r + 1.0; s.lm_pos[:] = r + 1.0
        parts.append(s)
    mg = distrib.merge_solution(p, pl, parts)
    assert np.array_equal(mg.kf_pose[:, 0], np.where(pl.pose_rank < 0, 0, pl.pose_rank) + 1.0)
    assert np.array_equal(mg.lm_pos[:, 0], pl.lm_rank + 1.0)
    assert np.array_equal(mg.kf_speed_bias[:, 0], np.where(pl.sb_rank < 0, 0, pl.sb_rank) + 1.0)
    # a single agent splits too (the units are subtrees of the elimination tree, not agents)
    p1 = distrib.shard_plan(_problem("mh01"), o, 2)
    assert p1 is not None and p1.subtrees >= 2


def _rows(rank_of, D, want):
    """IR rows (15 per keyframe) of the unknowns whose owner satisfies `want`: rank_of = (pose_rank, sb_rank)."""
    pr, sr = rank_of
    out = []
    for k in range(len(pr)):
        if want(pr[k]): out.extend(range(D * k, D * k + 6))
        if want(sr[k]): out.extend(range(D * k + 6, D * k + 15))
    return np.array(out, int)


def _dense(p, prob):
    ptr, col, blocks, b, _ = covo.schur_sparse(prob, covo.default_options(), 0.0)
    n = 15 * p.K
    S = np.zeros((n, n))
    for i in range(p.K):
        for q in range(ptr[i], ptr[i + 1]):
            S[15 * i:15 * i + 15, 15 * col[q]:15 * col[q] + 15] = blocks[q]
    # rows no residual of this share touches carry the oracle's placeholder diagonal 1 (the device leaves them 0 until the
    # damping is applied after the exchange): removed
    d = np.diag(S).copy()
    untouched = (d == 1.0) & (np.abs(S - np.diag(d)).sum(1) == 0) & (b == 0)
    S[np.diag_indices(n)] = np.where(untouched, 0.0, d)
    return S, b


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    dist.init_process_group(backend="gloo")
    assert dist.get_world_size() == world
    p = _problem("small")
    pl = distrib.shard_plan(p, backend.default_options(), world)       # same plan on every rank, no communication
    mine = distrib.shard_problem(p, pl, rank)
    S, b = _dense(p, mine)
    top = _rows((pl.pose_rank, pl.sb_rank), 15, lambda r: r < 0)
    buf = np.concatenate([S[np.ix_(top, top)].reshape(-1), b[top]])   # what the device all-reduces: the top fronts' own blocks + right-hand side
    t = torch.from_numpy(buf)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    own = _rows((pl.pose_rank, pl.sb_rank), 15, lambda r: r == rank)
    q.put((rank, buf, S[np.ix_(own, own)], S[np.ix_(own, top)], b[own], mine.L))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_top_front_reduction():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs: pr.start()
    out = sorted((q.get(timeout=300) for _ in range(2)), key=lambda t: t[0])
    for pr in procs: pr.join(timeout=60)
    assert all(pr.exitcode == 0 for pr in procs)
    p = _problem("small")
    pl = distrib.shard_plan(p, backend.default_options(), 2)
    S, b = _dense(p, p)
    top = _rows((pl.pose_rank, pl.sb_rank), 15, lambda r: r < 0)
    full = np.concatenate([S[np.ix_(top, top)].reshape(-1), b[top]])
    scale = np.abs(full).max()
    assert len(top) > 0 and np.array_equal(out[0][1], out[1][1])            # both ranks hold the same reduced top blocks
    assert np.abs(out[0][1] - full).max() <= 1e-12 * scale                   # ... and they are the full problem's
    for r, _, Soo, Sot, bo, L in out:                                        # everything else is complete on its one owner
        own = _rows((pl.pose_rank, pl.sb_rank), 15, lambda x: x == r)
        assert L > 0 and len(own) > 0
        assert np.abs(Soo - S[np.ix_(own, own)]).max() <= 1e-12 * scale
        assert np.abs(Sot - S[np.ix_(own, top)]).max() <= 1e-12 * scale
        assert np.abs(bo - b[own]).max() <= 1e-12 * max(np.abs(b).max(), 1.0)
    # and unknowns of different ranks never couple
    o0 = _rows((pl.pose_rank, pl.sb_rank), 15, lambda x: x == 0); o1 = _rows((pl.pose_rank, pl.sb_rank), 15, lambda x: x == 1)
    assert not S[np.ix_(o0, o1)].any()


def test_store_rendezvous_under_torchrun(tmp_path):
    """The few bytes the ranks exchange outside the data path (RCCL unique id, solved pieces) travel through a TCPStore on
    MASTER_ADDR:MASTER_PORT. Under `python -m torch.distributed.run` — how the driver launches bench.py for N > 1 — the elastic agent
    already listens on that port: the ranks must connect as clients (a rank-0 server there fails with 'address already in use')."""
    import subprocess
    import sys
    port = 29500 + 2000 + (os.getpid() % 2000)
    out = str(tmp_path / "store")
    env = dict(os.environ, COVGPU_TEST_OUT=out)
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "store_worker.py")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), worker], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    h = [open(f"{out}.{k}").read() for k in range(2)]
    assert h[0] == h[1] and len(h[0]) == 64


def test_store_rendezvous_launched_by_hand(tmp_path):
    """Without the launcher (RANK / WORLD_SIZE / MASTER_* set by hand, no elastic agent) rank 0 hosts the store itself."""
    import subprocess
    import sys
    port = 29500 + 4000 + (os.getpid() % 2000)
    out = str(tmp_path / "store")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "store_worker.py")
    procs = []
    for r in range(2):
        env = dict(os.environ, COVGPU_TEST_OUT=out, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
        procs.append(subprocess.Popen([sys.executable, worker], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for pr in procs:
        _, err = pr.communicate(timeout=300)
        assert pr.returncode == 0, err[-2000:]
    h = [open(f"{out}.{k}").read() for k in range(2)]
    assert h[0] == h[1] and len(h[0]) == 64


def test_ranks_with_different_shard_plans_refuse_before_any_collective(tmp_path):
    """distrib.attach exchanges a digest of the shard plan every rank computed for itself: a rank whose plan differs (another map
    revision, another option) makes EVERY rank raise before a communicator exists — instead of hanging in the first all-reduce."""
    import subprocess
    import sys
    port = 29500 + 6000 + (os.getpid() % 2000)
    out = str(tmp_path / "store")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "store_worker.py")
    procs = []
    for r in range(2):
        env = dict(os.environ, COVGPU_TEST_OUT=out, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   COVGPU_TEST_BAD_DIGEST="1")
        env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
        procs.append(subprocess.Popen([sys.executable, worker], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for pr in procs:
        _, err = pr.communicate(timeout=300)
        assert pr.returncode == 0, err[-2000:]   # (the worker asserts that its rank refused)


def test_single_process_is_identity():
    assert distrib.aggregate(0.5, 7, None, False) == (0.5, 7.0)
    assert distrib.throughput(2.0, 10.0) == 5.0


def test_shard_plan_keeps_the_replicated_top_small_on_the_five_agent_map():
    """Round 5: with the ground truth read correctly ONE separator of all agents is 621 keyframes on the 5-agent map — 61 % of the factorisation flops on
    every rank, slower than one GPU. covgpu_shard_plan chooses among candidate trees by nd_shard_cost; what it keeps must replicate a small share and
    give at least two ranks real work, and round 4's policy (one separator of all agents, 48 MiB cap) must still be reachable through its switches."""
    import ctypes as C
    p = _problem("mh12345")
    o = backend.default_options()

    def shares(pl):
        lib = backend.lib()
        info = (C.c_int64 * 16)()
        lib.covgpu_nd_plan_info(pl.handle, info)
        nn = int(info[0])
        parent = np.zeros(nn, np.int32); level = np.zeros(nn, np.int32); optr = np.zeros(nn + 1, np.int32); sptr = np.zeros(nn + 1, np.int32)
        ov = np.zeros(max(int(info[3]), 1), np.int32); sv = np.zeros(max(int(info[4]), 1), np.int32)
        lib.covgpu_nd_plan_arrays(pl.handle, capi.iptr(parent), capi.iptr(level), capi.iptr(optr), capi.iptr(ov), capi.iptr(sptr), capi.iptr(sv))
        dim = lambda v: np.where(v & 1, 9, 6)
        own = np.array([dim(ov[optr[n]:optr[n + 1]]).sum() for n in range(nn)], float)
        st = np.array([dim(sv[sptr[n]:sptr[n + 1]]).sum() for n in range(nn)], float)
        fl = own ** 3 / 3.0 + own ** 2 * st + own * st ** 2
        top = pl.node_rank < 0
        per_rank = np.array([fl[pl.node_rank == r].sum() for r in range(pl.world)])
        return fl[top].sum() / fl.sum(), per_rank / fl.sum(), int(own[top].sum())

    for world in (2, 4):
        pl = distrib.shard_plan(p, o, world)
        top_share, rank_share, top_dims = shares(pl)
        assert top_share < 0.2 and top_dims < 2500, (top_share, top_dims)
        assert np.sort(rank_share)[-2] > 0.3            # two ranks carry real work
        assert top_share + rank_share.max() < 0.65      # what the busiest rank factorises: well below the whole
        pl.close()
    os.environ["COVGPU_SHARD_TREE"] = "0"; os.environ["COVGPU_SHARD_CAP_MIB"] = "48"
    try:
        pl = distrib.shard_plan(p, o, 4)
        top_share, rank_share, top_dims = shares(pl)
        assert pl.subtrees == 5 and top_dims > 3000 and top_share > 0.5
        pl.close()
    finally:
        del os.environ["COVGPU_SHARD_TREE"]; del os.environ["COVGPU_SHARD_CAP_MIB"]
