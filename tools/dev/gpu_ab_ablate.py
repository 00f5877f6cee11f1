"""dev: the metric configuration with settings.ablate = each of argv[1:] in turn, interleaved, five runs each: median wall per run (A/B of a code path on ONE box)"""
import ctypes as C, sys, time
sys.path.insert(0, ".")
import torch
from polychordlite_amd import _ctypes_api as api
lib = api.load()
L, P, keep = api.make_problem("gaussian", 20, 2)
abl = [int(a) for a in sys.argv[1:]] or [0]
ts = {a: [] for a in abl}
for rep in range(6):
    for a in abl:
        s = api.Settings(); lib.pchip_settings_default(C.byref(s), 20, 2)
        s.nlive, s.num_repeats, s.ablate, s.seed = 2000, 40, a, 10 + rep
        t0 = time.perf_counter(); r = api.run(s, L, P); dt = time.perf_counter() - t0
        if rep: ts[a].append(dt)
        r = None
for a in abl:
    v = sorted(ts[a]); print("ablate %6d: median %.3f ms  min %.3f  max %.3f" % (a, 1e3 * v[len(v) // 2], 1e3 * v[0], 1e3 * v[-1]))
