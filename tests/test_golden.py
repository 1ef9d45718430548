"""Golden vectors (tests/golden/tiny_gba.npz, made by tools/make_golden.py): inputs = the flat IR of the seeded
`tiny` map, outputs = the CPU oracle's results at commit time. CPU: the generator, the flattening and the oracle
still reproduce them. GPU: the HIP path matches them through the C ABI."""
import os

import numpy as np
import pytest

from covins_amd import capi, mapdata, synth
from oracle import covo
from tests.util import rel_err, rot_angle

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_gba.npz"))


def golden_problem():
    return capi.FlatProblem(**{k[3:]: G[k] for k in G.files if k.startswith("in_")})


def test_generator_and_flattening_reproduce_golden_inputs(tiny_map):
    p, _ = mapdata.flatten_gba(tiny_map, False, True)
    g = golden_problem()
    for k, v in p.__dict__.items():
        assert np.array_equal(v, getattr(g, k)), k


def test_oracle_reproduces_golden_outputs():
    p = golden_problem()
    o = covo.default_options()
    r, Jp, Jl, c = covo.linearize_reprojection(p, o)
    assert rel_err(r, G["reproj_r"]) < 1e-13 and rel_err(Jp, G["reproj_Jp"]) < 1e-13 and rel_err(Jl, G["reproj_Jl"]) < 1e-13
    d, J, P = covo.preintegrate(p, o)
    assert rel_err(d, G["pre_delta"]) < 1e-13 and rel_err(J, G["pre_J"]) < 1e-12
    ri, Ji = covo.linearize_imu(p, o)
    assert rel_err(ri, G["imu_r"]) < 1e-9 and rel_err(Ji, G["imu_J"]) < 1e-9
    q, res = covo.gba_solve(p, o)
    assert np.abs(q.kf_pose - G["dogleg_pose"]).max() < 1e-8
    assert np.allclose(np.array(res.cost_trace[:res.iterations]), G["dogleg_trace"], rtol=1e-8)


@pytest.mark.gpu
def test_gpu_matches_golden():
    from covins_amd import backend
    ctx = backend.Context(0)
    p = golden_problem()
    o = backend.default_options()
    r, Jp, Jl, c = ctx.linearize_reprojection(p, o)
    assert rel_err(r, G["reproj_r"]) < 1e-12 and rel_err(Jp, G["reproj_Jp"]) < 1e-12 and rel_err(Jl, G["reproj_Jl"]) < 1e-12
    d, J, P = ctx.preintegrate(p, o)
    assert rel_err(d, G["pre_delta"]) < 1e-12 and rel_err(J, G["pre_J"]) < 1e-11
    ri, Ji = ctx.linearize_imu(p, o)
    assert rel_err(ri, G["imu_r"]) < 1e-8 and rel_err(Ji, G["imu_J"]) < 1e-8
    S, b, cost = ctx.schur(p, o, 1e-8)
    sc = np.sqrt(np.abs(np.diag(G["schur_S"])))
    assert np.abs(S / sc[:, None] / sc[None, :] - G["schur_S"] / sc[:, None] / sc[None, :]).max() < 1e-9
    assert abs(cost - G["schur_cost"][0]) < 1e-12 * cost
    for name, strat in (("dogleg", capi.COVGPU_DOGLEG), ("lm", capi.COVGPU_LM)):
        q, res = ctx.gba_solve(p, backend.default_options(strategy=strat))
        assert np.abs(q.kf_pose[:, 4:] - G[f"{name}_pose"][:, 4:]).max() < 1e-6
        assert rot_angle(q.kf_pose[:, :4], G[f"{name}_pose"][:, :4]).max() < 1e-7
        assert np.abs(q.kf_speed_bias - G[f"{name}_sb"]).max() < 1e-6
        assert np.allclose(np.array(res.cost_trace[:res.iterations]), G[f"{name}_trace"], rtol=1e-6)
        assert list(res.accepted_trace[:res.iterations]) == list(G[f"{name}_acc"])
    pg, _ = mapdata.flatten_pgo(synth.make_map(synth.config_named("tiny")), {}, mapdata.PgoParams())
    q, res = ctx.pgo_solve(pg, o)
    assert np.abs(q.kf_pose - G["pgo_pose"]).max() < 1e-6
    ctx.close()
