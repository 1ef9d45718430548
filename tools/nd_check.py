"""GPU: Gauss-Newton step of the multifrontal path vs the oracle on small maps, several leaf sizes (dev aid)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from covins_amd import backend, mapdata, synth
from oracle import covo

names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["tiny", "small"]
for name in names:
    m = synth.make_map(synth.config_named(name))
    prob, _ = mapdata.flatten_gba(m, visual_only=False, loop_loss=True)
    dxo, dlo = covo.step(prob, covo.default_options(), 1e-6, dense=False) if prob.K < 400 else (None, None)
    co = covo.cost(prob, covo.default_options()) if prob.K < 400 else None
    for leaf in ["100000", "150", "300", "600"]:
        os.environ["COVGPU_ND_LEAF"] = leaf
        ctx = backend.Context(0)
        try:
            t0 = time.time()
            dx, dl, c = ctx.gn_step(prob, backend.default_options(verbose=1), 1e-6)
            dt = time.time() - t0
            if dxo is not None:
                print(f"{name} leaf {leaf}: cost {c:.9e} (oracle {co:.9e})  max|dx-oracle| {np.abs(dx - dxo).max():.3e} / {np.abs(dxo).max():.3e}  "
                      f"max|dl-oracle| {np.abs(dl.reshape(-1) - dlo.reshape(-1)).max():.3e}  ({dt:.2f} s)", flush=True)
            else:
                print(f"{name} leaf {leaf}: cost {c:.9e} |dx| {np.abs(dx).max():.3e} ({dt:.2f} s)", flush=True)
        except Exception as e:
            print(f"{name} leaf {leaf}: FAILED {e}", flush=True)
        ctx.close()
