"""Worker of tests/test_gpu_schedule.py: one full GBA solve of a named synthetic map in THIS process (the library reads its
scheduling switches from the environment once per process), result saved as .npz."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

if __name__ == "__main__":
    name, out = sys.argv[1], sys.argv[2]
    import torch
    torch.cuda.init()
    from covins_amd import backend, mapdata, synth
    m = synth.make_map(synth.config_named(name))
    p, _ = mapdata.flatten_gba(m, visual_only=False, loop_loss=True)
    o = backend.default_options(max_iterations=6)
    ctx = backend.Context(0)
    sol, res = ctx.gba_solve(p, o)
    np.savez(out, pose=sol.kf_pose, sb=sol.kf_speed_bias, lm=sol.lm_pos, cost=np.array(res.cost_trace[:res.iterations]),
             acc=np.array(res.accepted_trace[:res.iterations]), final=res.final_cost, ordering=np.int64(ctx.layout()["stream_ordering"]))
