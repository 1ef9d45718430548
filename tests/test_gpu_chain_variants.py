"""The two generations of serial-chain kernels must agree: the 256-column panel chain (k_panel.hip), the single-product chain
sweeps and the sweep-based pose right-hand side against the round-2a kernels they replace (COVGPU_PANEL=0, COVGPU_SB_BACK=0,
COVGPU_POSE_RHS_Y=1 — read once per process, hence the subprocesses). Both are exact solves of the same system: dense solve to
1e-11 (relative), visual-inertial GBA of the `small` map to 1e-9 in cost and 1e-8 m in the poses after 10 iterations."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import json, sys
import numpy as np
sys.path.insert(0, %r)
import torch; torch.cuda.init()
from covins_amd import backend, mapdata, synth
ctx = backend.Context(0)
out = {}
for n in (129, 700, 1100):          # odd tile counts exercise the single-tile tail of the panel chain
    rng = np.random.default_rng(n)
    A = rng.normal(0, 1, (n, n + 5)); S = A @ A.T + 0.5 * n * np.eye(n); b = rng.normal(0, 1, n)
    rc, x = ctx.solve_reduced(S, b)
    assert rc == 0
    out["x%%d" %% n] = x.tolist()
m = synth.make_map(synth.config_named("small"))
p = mapdata.flatten_gba(m, False, True)[0]
opt = backend.default_options(max_iterations=10)
sol, res = ctx.gba_solve(p, opt)
out["cost"] = res.final_cost; out["iterations"] = res.iterations; out["accepted"] = res.accepted
out["pose"] = sol.kf_pose.tolist(); out["sb"] = sol.kf_speed_bias.tolist()
ctx.close()
print("RESULT" + json.dumps(out))
""" % ROOT


def _run(env_extra):
    env = dict(os.environ); env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1]
    return json.loads(line[len("RESULT"):])


def test_panel_chain_and_sweeps_match_round2a_kernels():
    new = _run({})
    old = _run({"COVGPU_PANEL": "0", "COVGPU_SB_BACK": "0", "COVGPU_POSE_RHS_Y": "1"})
    for n in (129, 700, 1100):
        a, b = np.array(new["x%d" % n]), np.array(old["x%d" % n])
        assert np.max(np.abs(a - b)) <= 1e-11 * np.max(np.abs(b))
    assert new["iterations"] == old["iterations"] and new["accepted"] == old["accepted"]
    assert abs(new["cost"] - old["cost"]) <= 1e-9 * abs(old["cost"])
    assert np.max(np.abs(np.array(new["pose"])[:, 4:] - np.array(old["pose"])[:, 4:])) < 1e-8
    assert np.max(np.abs(np.array(new["sb"]) - np.array(old["sb"]))) < 1e-7
