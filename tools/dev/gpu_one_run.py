"""dev: one run of the metric configuration with settings.ablate = argv[1] (for rocprofv3 --kernel-trace --stats around it)"""
import ctypes as C, sys
sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api
lib = api.load()
s = api.Settings(); lib.pchip_settings_default(C.byref(s), 20, 2)
s.nlive, s.num_repeats, s.seed, s.ablate = 2000, 40, 1001, int(sys.argv[1])
L, P, keep = api.make_problem("gaussian", 20, 2)
api.run(s, L, P)
s.seed = 1002
r = api.run(s, L, P)
print(r["t_total"] * 1e3, "ms")
