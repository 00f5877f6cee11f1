"""dev: one run of BASELINE configs[2] or [3] on a build of pc_contract.hip with -DNND_DBG (PCHIP_LIB=..., PC_DEBUG=3): cycles of chain 0's
workgroup of k_nn_lists_d by section (set-up, staging, scanning, merges; the whole kernel in the last counter).  usage: gpu_nnd_dbg.py c3|c4"""
import ctypes as C, sys
sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api
lib = api.load()
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
if which == "c3":
    s = api.Settings(); lib.pchip_settings_default(C.byref(s), 10, 0)
    s.nlive, s.num_repeats, s.do_clustering, s.seed = 1000, 30, 1, 7001
    L, P, keep = api.make_problem("rastrigin", 10, 0, -5.12, 5.12)
else:
    s = api.Settings(); lib.pchip_settings_default(C.byref(s), 30, 1)
    s.nlive, s.num_repeats, s.do_clustering, s.seed = 500, 40, 1, 7001
    L, P, keep = api.make_problem("twin_gaussian", 30, 1, -1.0, 1.0)
g = api.run(s, L, P)
print(which, g["t_total"] * 1e3, "ms", g["ndead"], g["nlike"], "nn_lists launches", g["path"]["nn_lists"])
