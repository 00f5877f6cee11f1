# C2 run time with and without the split k_nhats launch (bases on the side stream); prints logZ / ndead / nlike too
import ctypes as C, sys, os, time
sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api
lib = api.load()
s = api.Settings(); lib.pchip_settings_default(C.byref(s), 20, 2)
s.nlive, s.num_repeats, s.seed, s.feedback = 2000, 40, 1, 0
L, P, keep = api.make_problem("gaussian", 20, 2)
for i in range(6):
    s.seed = 1 + i % 3
    g = api.run(s, L, P)
    print(s.seed, round(g["t_total"] * 1e3, 2), round(g["t_loop"] * 1e3, 2), g["logZ"], g["ndead"], g["nlike"])
