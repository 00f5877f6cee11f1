# C2 with the stopwatch on the contraction class: per-launch time of k_consume next to the run time
import ctypes as C, sys
sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api
lib = api.load()
s = api.Settings(); lib.pchip_settings_default(C.byref(s), 20, 2)
s.nlive, s.num_repeats, s.seed, s.feedback = 2000, 40, 1, 0
L, P, keep = api.make_problem("gaussian", 20, 2)
for i in range(5):
    s.profile = (1 << 3) if i >= 3 else 0
    g = api.run(s, L, P)
    kt = g["kernel_time"].get("k_consume")
    print(round(g["t_total"] * 1e3, 2), (round(kt["total_s"] / kt["launches"] * 1e6, 1) if kt else None))
