"""dev: one run of BASELINE configs[2] on the CL_DBG build of the one-wave contraction (PCHIP_LIB=.../libpolychord_hip_cldbg.so PC_DEBUG=4):
cycles of its loop by section -- head (termination test, prefetch), identify (candidate lists of the chain's babies), death (evidence
update), order (next death, row prefetch), add (newcomer, sums over the clusters) -- deaths and chains consumed."""
import ctypes as C, sys
sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api
lib = api.load()
s = api.Settings(); lib.pchip_settings_default(C.byref(s), 10, 0)
s.nlive, s.num_repeats, s.do_clustering, s.seed = 1000, 30, 1, 7001
L, P, keep = api.make_problem("rastrigin", 10, 0, -5.12, 5.12)
g = api.run(s, L, P)
print(g["t_total"] * 1e3, "ms", g["ndead"], g["nlike"])
