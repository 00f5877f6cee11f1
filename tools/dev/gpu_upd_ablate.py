"""dev: metric config with pieces of k_upd_move switched off (results are wrong on purpose; timing only)"""
import ctypes as C, sys
sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api
lib = api.load()
s = api.Settings(); lib.pchip_settings_default(C.byref(s), 20, 2)
s.nlive, s.num_repeats, s.seed, s.max_ndead = 2000, 40, 7, 40000
s.ablate = int(sys.argv[1])
L, P, keep = api.make_problem("gaussian", 20, 2)
for i in range(2):
    r = api.run(s, L, P)
print("ablate", s.ablate, "ndead", r["ndead"], "t", r["t_total"])
