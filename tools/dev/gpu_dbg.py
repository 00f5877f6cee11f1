import os
os.environ.setdefault("PC_DEBUG", "2")   # developer counters of the engine
import ctypes as C, sys
sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api
lib = api.load()
L, P, keep = api.make_problem("gaussian", 20, 2)
for B in (1000, 512, 256):
    s = api.Settings(); lib.pchip_settings_default(C.byref(s), 20, 2)
    s.nlive, s.num_repeats, s.seed, s.batch, s.profile, s.feedback = 2000, 40, 1, B, 1, 2
    g = api.run(s, L, P)
    k = g["kernel_time"]
    print(f"B={B}: t_total {g['t_total']*1e3:.1f} ms niter {g['niter']} ndead {g['ndead']} batches {g['nbatches']} rounds {g['nrounds']} " + " ".join(f"{n}={v['total_s']*1e3:.1f}" for n, v in k.items()), flush=True)
