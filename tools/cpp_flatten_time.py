"""Host-only: time of the C++ facade's Map -> IR walk (covins_gpu::OptimizationT::FlattenGBA, include/covins_gpu/optimization_gpu.hpp)
on stand-in Map / Keyframe / Landmark objects of a BASELINE-sized map (the Python mirror's numpy flatten is
not what a covins_backend process would run). Usage: python tools/cpp_flatten_time.py [workload] [reps]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from covins_amd import mapdata, synth            # noqa: E402
from tests.facade_util import StandinMap, lib   # noqa: E402

if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[3] == "gba":   # (torch's HIP runtime must come up before libcovgpu loads /opt/rocm's: INTEGRATION.md §5)
        import torch
        torch.cuda.init()
    name = sys.argv[1] if len(sys.argv) > 1 else "mh12345"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    m = synth.make_map(synth.config_named(name))
    t0 = time.perf_counter(); p, _ = mapdata.flatten_gba(m, False, loop_loss=True); t_py = time.perf_counter() - t0
    t0 = time.perf_counter(); sm = StandinMap(m); t_build = time.perf_counter() - t0
    sizes = np.zeros(6, np.int32); ncam = C.c_int(0)
    out = {}
    for visitor in (1, 0):
        lib().shim_use_visitor(visitor)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            lib().shim_flatten_gba(sm.h, 0, 1, sizes.ctypes.data_as(C.POINTER(C.c_int)), *([None] * 14), C.byref(ncam))
            ts.append(time.perf_counter() - t0)
        out[visitor] = ts
    lib().shim_use_visitor(1)
    t_gba = None
    if len(sys.argv) > 3 and sys.argv[3] == "gba":   # needs a GPU: the whole C++ GlobalBundleAdjustment call on the stand-in map, twice (first: warm-up)
        t_gba = []
        for _ in range(3):
            sm2 = StandinMap(m)
            t0 = time.perf_counter(); sm2.gba(10); t_gba.append(time.perf_counter() - t0)
            sm2.close()
    sm.close()
    print(f"{name}: K={sizes[0]} L={sizes[1]} O={sizes[2]} I={sizes[3]} E={sizes[4]}; C++ FlattenGBA, {reps} reps, {os.cpu_count()} host threads available: "
          f"median {np.median(out[1]) * 1e3:.1f} ms (min {min(out[1]) * 1e3:.1f}) with Types::visit_observations, "
          f"{np.median(out[0]) * 1e3:.1f} ms (min {min(out[0]) * 1e3:.1f}) over Landmark::GetObservations() copies; "
          f"numpy mirror flatten {t_py * 1e3:.1f} ms; building the stand-in object graph {t_build:.2f} s (not part of a call)"
          + (f"; whole C++ covins_gpu::Optimization::GlobalBundleAdjustment(map, 10) on the stand-in map: {', '.join(f'{t * 1e3:.0f}' for t in t_gba)} ms "
             f"(first call includes context creation and library warm-up)" if t_gba else ""))
