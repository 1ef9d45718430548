"""Agent-sharded solve on VIRTUAL ranks: R contexts on the one GPU of the test box, each
holding one rank's share of the SAME map, driven in lockstep by host threads; the library's three collectives per linear
solve go through distrib.ThreadReducer (host sum in rank order). The sharded result must equal the unsharded one:
one Gauss-Newton step to 1e-10 (relative, system metric) and the full 10-iteration solve to 1e-8 m."""
import threading

import numpy as np
import pytest

from covins_amd import backend, distrib, mapdata, synth

pytestmark = pytest.mark.gpu

_cache = {}


@pytest.fixture(autouse=True)
def _same_elimination_order(monkeypatch):
    # the sharded solve still runs the round-2 agent-block form; its unsharded reference uses the same form so that the comparison
    # isolates the sharding (the multifrontal form is compared with the oracle and the one-front solve in test_gpu_full.py)
    monkeypatch.setenv("COVGPU_GBA_LEGACY", "1")


def problem(name):
    if name not in _cache:
        m = synth.make_map(synth.config_named(name))
        _cache[name] = mapdata.flatten_gba(m, False, True)[0]
    return _cache[name]


def run_virtual_ranks(prob, plan, job):
    """job(ctx, sub_problem, rank) on every virtual rank, concurrently; returns the list of results."""
    red = distrib.ThreadReducer(plan.world)
    out, err = [None] * plan.world, []

    def work(r):
        try:
            ctx = backend.Context(0)
            cb = red.callback(r)
            ctx.set_shard(plan, r, cb, stage_on_host=True)
            out[r] = job(ctx, distrib.shard_problem(prob, plan, r), r)
            ctx.close()
        except Exception as e:  # a failing rank must not leave the others waiting at the barrier forever
            err.append(e)
            red._bar.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(plan.world)]
    for t in th: t.start()
    for t in th: t.join(timeout=600)
    assert not err, err
    return out, red


@pytest.mark.parametrize("name,world", [("mh123", 2), ("mh123", 3), ("mh123", 4), ("mh12345", 2), ("mh12345", 4), ("mh12345", 5), ("mh12345", 8)])
def test_sharded_gauss_newton_step_equals_unsharded(name, world):
    p = problem(name)
    o = backend.default_options()
    plan = distrib.shard_plan(p, o, world)
    assert plan is not None   # (world > number of agents: the surplus ranks own nothing and only take part in the reductions)
    ctx = backend.Context(0)
    dx0, dl0, c0 = ctx.gn_step(p, o, 1e-8)
    ctx.close()
    parts, red = run_virtual_ranks(p, plan, lambda ctx, sub, r: ctx.gn_step(sub, o, 1e-8))
    po, so = plan.pose_owner(), distrib.chain_owner(p, plan)
    dx = np.zeros_like(dx0); dl = np.zeros_like(dl0)
    X = dx.reshape(p.K, 15)
    cost = 0.0
    for r, (dxr, dlr, cr) in enumerate(parts):
        Xr = dxr.reshape(p.K, 15)
        X[po == r, :6] = Xr[po == r, :6]
        X[so == r, 6:] = Xr[so == r, 6:]
        dl[plan.lm_rank == r] = dlr
        assert abs(cr - c0) <= 1e-12 * c0          # the all-reduced cost is the full cost on every rank
    # shared keyframes: identical on every rank (the border system is solved redundantly from identical data)
    sh = plan.block_of_kf < 0
    for dxr, _, _ in parts[1:]:
        assert np.array_equal(dxr.reshape(p.K, 15)[sh, :6], parts[0][0].reshape(p.K, 15)[sh, :6])
    scale = np.abs(dx0).max()
    print(f"{name} world {world}: step difference {np.abs(dx - dx0).max() / scale:.2e} (relative), landmarks {np.abs(dl - dl0).max():.2e} m, "
          f"{red.calls} collectives, {red.bytes / 1e6:.1f} MB")
    assert np.abs(dx - dx0).max() <= 1e-9 * scale
    assert np.abs(dl - dl0).max() <= 1e-9 * max(np.abs(dl0).max(), 1.0)


@pytest.mark.parametrize("name,world", [("mh123", 3), ("mh12345", 5)])
def test_sharded_solve_equals_unsharded(name, world):
    p = problem(name)
    o = backend.default_options(max_iterations=10)
    plan = distrib.shard_plan(p, o, world)
    ctx = backend.Context(0)
    s0, r0 = ctx.gba_solve(p, o)
    ctx.close()
    parts, red = run_virtual_ranks(p, plan, lambda ctx, sub, r: ctx.gba_solve(sub, o))
    sol = distrib.merge_solution(p, plan, [q for q, _ in parts])
    for _, res in parts:   # every rank took the same decisions
        assert res.iterations == r0.iterations and list(res.accepted_trace[:10]) == list(r0.accepted_trace[:10])
        assert np.allclose(np.array(res.cost_trace[:res.iterations]), np.array(r0.cost_trace[:r0.iterations]), rtol=1e-9)
    dp = np.abs(sol.kf_pose - s0.kf_pose).max(); ds = np.abs(sol.kf_speed_bias - s0.kf_speed_bias).max(); dl = np.abs(sol.lm_pos - s0.lm_pos).max()
    print(f"{name} world {world}: pose {dp:.2e} speed-bias {ds:.2e} landmarks {dl:.2e}; {red.calls} collectives, {red.bytes / 1e6:.1f} MB per solve")
    assert dp < 1e-8 and ds < 1e-8 and dl < 1e-6
