"""dev: the metric configuration by the closed form and as a general functor (settings.ablate bit 0): where a run's time goes
(t_setup, t_generate, t_loop, t_final, t_results, t_teardown of pchip_result; wall per call)"""
import ctypes as C, sys, time
sys.path.insert(0, ".")
import torch
from polychordlite_amd import _ctypes_api as api
lib = api.load()
L, P, keep = api.make_problem("gaussian", 20, 2)
for ab in (0, 1, 0, 1):
    s = api.Settings(); lib.pchip_settings_default(C.byref(s), 20, 2)
    s.nlive, s.num_repeats, s.ablate = 2000, 40, ab
    s.seed = 1; api.run(s, L, P)
    ts = []
    for k in range(5):
        s.seed = 10 + k
        t0 = time.perf_counter(); r = api.run(s, L, P); dt = time.perf_counter() - t0
        ts.append((dt, r["t_setup"], r["t_generate"], r["t_loop"], r["t_final"], r["t_results"], r["t_teardown"], r["t_total"], r["nbatches"], r["nupdates"]))
    m = sorted(ts)[2]
    print("ablate", ab, "wall %.2f ms; setup %.2f generate %.2f loop %.2f final %.2f results %.2f teardown %.2f total %.2f; batches %d updates %d" % tuple([1e3 * x for x in m[:8]] + list(m[8:])))
