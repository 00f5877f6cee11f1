import ctypes as C, sys, os
os.environ.setdefault("PC_DEBUG", os.environ.get("PC_FB", "3"))   # developer counters of the engine
sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api
lib = api.load()
s = api.Settings(); lib.pchip_settings_default(C.byref(s), 20, 2)
s.nlive, s.num_repeats, s.seed, s.feedback, s.profile = 2000, 40, 1, int(os.environ.get("PC_FB", "3")), 1
L, P, keep = api.make_problem("gaussian", 20, 2)
for i in range(2):
    g = api.run(s, L, P)
print({k: (round(v["total_s"] * 1e3, 2), v["launches"]) for k, v in g["kernel_time"].items()}, round(g["t_total"] * 1e3, 2))
