"""dev: section cycles of k_slice_t (SLICE_T_DBG build in PCHIP_LIB, PC_DEBUG=4): whitening, coefficients, bracket + stepping out, shrinkage, stores"""
import ctypes as C, sys
sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api
lib = api.load()
s = api.Settings(); lib.pchip_settings_default(C.byref(s), 20, 2)
s.nlive, s.num_repeats, s.seed, s.ablate = 2000, 40, 1001, 64
s.profile = 1 << 2
L, P, keep = api.make_problem("gaussian", 20, 2)
api.run(s, L, P)
s.seed = 1002
r = api.run(s, L, P)
kt = r["kernel_time"]["k_slice"]
print("k_slice_t us per launch", kt["total_s"] / kt["launches"] * 1e6, "launches", r["nbatches"], file=sys.stderr)
