"""world_size-2 `gloo` test of the N > 1 path of bench.py (covins_amd/distrib.py) on CPU: per-rank map shards are
distinct, the timed-region aggregation is MAX over ranks for the wall time and SUM for the executed iterations."""
import os

import numpy as np
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from covins_amd import distrib, mapdata, synth
    dist = distrib.init("gloo")
    assert dist is not None and dist.get_world_size() == world
    m = synth.make_map(synth.config_named("tiny", seed=distrib.map_seed_for_rank(rank)))
    p, _ = mapdata.flatten_gba(m, False, True)
    distrib.barrier(dist, "cpu")
    dt, its = distrib.aggregate(1.0 + rank, 10 + rank, dist, "cpu")   # rank 1 is slower and did one more iteration
    q.put((rank, dt, its, float(p.obs_uv.sum()), p.K))
    distrib.barrier(dist, "cpu")
    dist.destroy_process_group()


def test_two_rank_gloo_aggregation():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    out = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    (r0, dt0, it0, s0, k0), (r1, dt1, it1, s1, k1) = out
    assert dt0 == dt1 == 2.0 and it0 == it1 == 21.0          # MAX of time, SUM of iterations on every rank
    assert k0 == k1 and s0 != s1                              # same configuration, different map per rank
    from covins_amd import distrib
    assert distrib.throughput(dt0, it0) == 10.5


def test_single_process_is_identity():
    from covins_amd import distrib
    assert distrib.aggregate(0.5, 7, None) == (0.5, 7.0)
    assert distrib.map_seed_for_rank(3, base_seed=10) == 13
