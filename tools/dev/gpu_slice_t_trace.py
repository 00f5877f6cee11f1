"""dev: first slices of chain 0 of the first nursery, traced by k_slice and k_slice_t (PC_TRACE_SLICE build in PCHIP_LIB); argv[1] = ablate"""
import ctypes as C, sys
sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api
lib = api.load()
s = api.Settings(); lib.pchip_settings_default(C.byref(s), 8, 0)
s.nlive, s.num_repeats, s.seed, s.ablate, s.max_ndead = 300, 16, 11, int(sys.argv[1]), 200
L, P, keep = api.make_problem("gaussian", 8, 0)
api.run(s, L, P)
