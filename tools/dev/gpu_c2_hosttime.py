import ctypes as C, sys
sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api
lib = api.load()
s = api.Settings(); lib.pchip_settings_default(C.byref(s), 20, 2)
s.nlive, s.num_repeats, s.seed, s.feedback = 2000, 40, 1, 0
L, P, keep = api.make_problem("gaussian", 20, 2)
for i in range(3):
    s.seed = i + 1; s.feedback = 5 if i == 2 else 0
    g = api.run(s, L, P)
    print({k: round(g[k] * 1e3, 3) for k in ("t_setup", "t_generate", "t_loop", "t_final", "t_results", "t_teardown", "t_total")})
