"""Host-only: statistics of the nested-dissection plan (covgpu_nd_plan_*) on the synthetic workloads, per leaf size."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from covins_amd import backend, mapdata, synth


def plan_info(prob, opt, leaf):
    lib = backend.lib()
    h = C.c_void_p()
    s = prob.as_struct()
    t0 = time.time()
    rc = lib.covgpu_nd_plan_create(C.byref(opt), C.byref(s), leaf, C.byref(h))
    dt = time.time() - t0
    assert rc == 0, lib.covgpu_last_error()
    out = (C.c_int64 * 16)()
    lib.covgpu_nd_plan_info(h, out)
    nn = out[0]
    parent = np.zeros(nn, np.int32); level = np.zeros(nn, np.int32)
    optr = np.zeros(nn + 1, np.int32); sptr = np.zeros(nn + 1, np.int32)
    ov = np.zeros(out[3], np.int32); sv = np.zeros(max(out[4], 1), np.int32)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    lib.covgpu_nd_plan_arrays(h, ip(parent), ip(level), ip(optr), ip(ov), ip(sptr), ip(sv))
    lib.covgpu_nd_plan_destroy(h)
    return list(out), dt, parent, level, optr, ov, sptr, sv


if __name__ == "__main__":
    names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["mh01", "mh12345"]
    leaves = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [600, 900, 1200]
    for name in names:
        m = synth.make_map(synth.config_named(name))
        prob, _ = mapdata.flatten_gba(m, visual_only=False, loop_loss=True)
        opt = backend.default_options()
        print(f"{name}: K={prob.K} L={prob.L} O={prob.O}")
        for leaf in leaves:
            info, dt, parent, level, optr, ov, sptr, sv = plan_info(prob, opt, leaf)
            dim = lambda v: np.where(v & 1, 9, 6)
            od = np.array([dim(ov[optr[n]:optr[n + 1]]).sum() for n in range(info[0])])
            sd = np.array([dim(sv[sptr[n]:sptr[n + 1]]).sum() for n in range(info[0])])
            print(f"  leaf {leaf}: nodes {info[0]} levels {info[1]} depth {info[2]} front elems {info[5] * 8e-9:.2f} GB flops {info[6]:.3e} "
                  f"max own {info[7]} max border {info[8]} root {info[9]}  ({dt:.2f} s)")
            panels = 0
            for l in range(info[1]):
                sel = level == l
                nI = max(256, -(-od[sel].max() // 256) * 256)
                panels += nI // 256
                print(f"    level {l}: {sel.sum():3d} nodes, own {od[sel].min()}..{od[sel].max()} border {sd[sel].min()}..{sd[sel].max()}")
            print(f"    serial 256-panels: {panels}")
