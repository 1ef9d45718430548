#!/usr/bin/env python3
"""Host-only: what the shard plan of a workload looks like at 2 / 4 / 8 ranks — subtrees, size of the replicated top, bytes of the one
all-reduce per linear solve (packed live lower 128x128 tiles of the top fronts + their right-hand sides), factorisation flops of the top
and of the busiest rank's subtrees. Usage: python tools/shard_stats.py [workload ...]   (no GPU needed: covgpu_shard_plan is host code)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from covins_amd import backend, capi, distrib, mapdata, synth  # noqa: E402


def plan_arrays(plan):
    lib = backend.lib()
    info = (C.c_int64 * 16)()
    lib.covgpu_nd_plan_info(plan.handle, info)
    nn = int(info[0])
    parent = np.zeros(nn, np.int32); level = np.zeros(nn, np.int32); optr = np.zeros(nn + 1, np.int32); sptr = np.zeros(nn + 1, np.int32)
    ov = np.zeros(max(int(info[3]), 1), np.int32); sv = np.zeros(max(int(info[4]), 1), np.int32)
    ip = capi.iptr
    lib.covgpu_nd_plan_arrays(plan.handle, ip(parent), ip(level), ip(optr), ip(ov), ip(sptr), ip(sv))
    dim = lambda v: np.where(v & 1, 9, 6)
    own = np.array([dim(ov[optr[n]:optr[n + 1]]).sum() for n in range(nn)])
    st = np.array([dim(sv[sptr[n]:sptr[n + 1]]).sum() for n in range(nn)])
    return parent, own, st


def main():
    for name in (sys.argv[1:] or ["mh12345", "a12"]):
        m = synth.make_map(synth.config_named(name))
        p, _ = mapdata.flatten_gba(m, False, True)
        opt = backend.default_options()
        for world in (2, 4, 8):
            plan = distrib.shard_plan(p, opt, world)
            if plan is None:
                print(f"{name} world {world}: does not split"); continue
            parent, own, st = plan_arrays(plan)
            fl = own ** 3 / 3.0 + own ** 2 * st + own * st ** 2          # NdHostPlan::flops per node
            top = plan.node_rank < 0
            t = lambda x: (x + 127) // 128
            tiles = 0
            for n in np.nonzero(top)[0]:
                lo2 = 2 * ((own[n] + 255) // 256); lb = t(st[n]); nl = lo2 + lb
                tiles += nl * (nl + 1) // 2
            rank_fl = np.array([fl[plan.node_rank == r].sum() for r in range(world)])
            print(f"{name} world {world}: {plan.subtrees} subtrees, top = {int(top.sum())} fronts / {int(own[top].sum())} unknowns, all-reduce "
                  f"{tiles * 128 * 128 * 8 / 1e6:.1f} MB per linear solve (packed live lower tiles), flops: total {fl.sum():.3e}, top {fl[top].sum():.3e} "
                  f"({fl[top].sum() / fl.sum():.0%}, replicated on every rank), busiest rank's subtrees {rank_fl.max():.3e} ({rank_fl.max() / fl.sum():.0%}), "
                  f"lightest {rank_fl.min():.3e}")
            plan.close()


if __name__ == "__main__":
    main()
