#!/usr/bin/env python3
"""Generates tests/golden/tiny_gba.npz: the seeded `tiny` synthetic map's flat IR (inputs) together with the CPU
oracle's outputs for it (per-kernel linearisations, Schur system, dogleg and LM solutions). The reference has no
golden vectors for this path, so these pin OUR oracle: tests/test_golden.py checks that the
oracle still reproduces them (CPU) and that the HIP path matches them (GPU)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from covins_amd import capi, mapdata, synth  # noqa: E402
from oracle import covo  # noqa: E402


def main():
    m = synth.make_map(synth.config_named("tiny"))
    p, _ = mapdata.flatten_gba(m, False, True)
    out = {f"in_{k}": v for k, v in p.__dict__.items()}
    o = covo.default_options()
    r, Jp, Jl, c = covo.linearize_reprojection(p, o)
    out.update(reproj_r=r, reproj_Jp=Jp, reproj_Jl=Jl, reproj_cost=c)
    d, J, P = covo.preintegrate(p, o)
    out.update(pre_delta=d, pre_J=J, pre_P=P)
    ri, Ji = covo.linearize_imu(p, o)
    out.update(imu_r=ri, imu_J=Ji)
    S, b, cost = covo.schur(p, o, 1e-8)
    out.update(schur_S=S, schur_b=b, schur_cost=np.array([cost]))
    for name, strat in (("dogleg", capi.COVGPU_DOGLEG), ("lm", capi.COVGPU_LM)):
        q, res = covo.gba_solve(p, covo.default_options(strategy=strat))
        out[f"{name}_pose"] = q.kf_pose; out[f"{name}_sb"] = q.kf_speed_bias; out[f"{name}_lm"] = q.lm_pos
        out[f"{name}_trace"] = np.array(res.cost_trace[:res.iterations]); out[f"{name}_acc"] = np.array(res.accepted_trace[:res.iterations])
    pg, _ = mapdata.flatten_pgo(m, {}, mapdata.PgoParams())
    q, res = covo.gba_solve(pg, o, pgo=True)
    out.update(pgo_pose=q.kf_pose, pgo_trace=np.array(res.cost_trace[:res.iterations]))
    dst = os.path.join(ROOT, "tests", "golden", "tiny_gba.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
