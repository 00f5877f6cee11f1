"""dev tool: ablation timing of the contraction kernel (not a test)"""
import ctypes as C, sys
sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api
lib = api.load()
L, P, keep = api.make_problem("gaussian", 20, 2)
for ab in (0, 1, 2, 4, 6, 8, 16, 32, 64, 127):
    s = api.Settings(); lib.pchip_settings_default(C.byref(s), 20, 2)
    s.nlive, s.num_repeats, s.seed, s.batch, s.profile, s.max_ndead, s.ablate = 2000, 40, 1, 1000, 1, 9000, ab
    g = api.run(s, L, P)
    k = g["kernel_time"]["k_consume"]
    print(f"ablate {ab:3d}: consume {k['total_s']*1e3:8.2f} ms over {g['niter']} steps = {k['total_s']/g['niter']*1e6:.3f} us/step; slice {g['kernel_time']['k_slice']['total_s']*1e3:.2f} ms nhats {g['kernel_time']['k_nhats']['total_s']*1e3:.2f} ms batches {g['nbatches']}", flush=True)
