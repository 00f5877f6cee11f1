# wall time of a wide run (nDims > 128: k_nhats_big, Cholesky factor in HBM); usage: gpu_wide.py [D] [nlive] [nr]
import ctypes as C, sys
sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api
lib = api.load()
D = int(sys.argv[1]) if len(sys.argv) > 1 else 200
nlive = int(sys.argv[2]) if len(sys.argv) > 2 else 500
nr = int(sys.argv[3]) if len(sys.argv) > 3 else 2 * D
s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, 0)
s.nlive, s.num_repeats, s.seed, s.feedback, s.profile = nlive, nr, 1, 0, 1
L, P, keep = api.make_problem("gaussian", D, 0)
g = api.run(s, L, P)
print(D, nlive, nr, "t_total %.2f s" % g["t_total"], "logZ %.3f +/- %.3f" % (g["logZ"], g["logZerr"]), g["ndead"], g["nlike"], "batch", g["batch"])
print({k: (round(v["total_s"], 3), v["launches"]) for k, v in g["kernel_time"].items()})
