"""Dev script: k_nhats variants at a given nDims.  usage: PC_NHATS_QUAD_MIN=<n> gpu_nhats_cmp.py D"""
import ctypes as C, sys
sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api
D = int(sys.argv[1]); lib = api.load()
s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, 0)
s.nlive, s.num_repeats, s.seed, s.batch, s.profile, s.max_ndead = 600, 2 * D, 3, 300, 1, 3000
L, P, keep = api.make_problem("gaussian", D, 0)
for _ in range(2):
    g = api.run(s, L, P)
k = g["kernel_time"]
print(D, {n: round(v["total_s"] / v["launches"] * 1e6, 1) for n, v in k.items()}, g["ndead"], g["nlike"])
