import sys, time
sys.path.insert(0,'.')
import torch; torch.cuda.init()
from covins_amd import synth, mapdata, backend
m = synth.make_map(synth.config_named("mh12345"))
p,_ = mapdata.flatten_gba(m, False, True)
ctx = backend.Context(0); o = backend.default_options()
for i in range(3):
    t0=time.perf_counter(); ctx.upload(p,o); print("upload total %.1f ms"%((time.perf_counter()-t0)*1e3), flush=True)
