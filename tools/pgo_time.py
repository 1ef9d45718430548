import time, sys, numpy as np
sys.path.insert(0, '/root/repo')
import torch; torch.cuda.init()
from covins_amd import backend, capi, mapdata, synth
from covins_amd.optimization import Optimization, OptParams
cfg = synth.config_named(sys.argv[1] if len(sys.argv) > 1 else "mh12345")
m = synth.make_map(cfg)
prm = mapdata.PgoParams()
prob, idx = mapdata.flatten_pgo(m, {}, prm)
print("PGO problem K", prob.K, "E", prob.E, "iteration limit", prm.pgo_iteration_limit)
ctx = backend.Context(0)
opt = backend.default_options(max_iterations=prm.pgo_iteration_limit)
for rep in range(3):
    t = time.perf_counter()
    sol, res = ctx.pgo_solve(prob, opt)
    dt = time.perf_counter() - t
    print(f"pgo_solve: {dt*1e3:.1f} ms, {res.iterations} iterations, cost {res.initial_cost:.4g} -> {res.final_cost:.4g}, t_solve {res.t_solve_s*1e3:.1f} ms t_upload {res.t_upload_s*1e3:.1f} ms")
