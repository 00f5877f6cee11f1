"""BASELINE configs[4] for a few nurseries (developer script, for rocprofv3 kernel traces)"""
import ctypes as C, sys
sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api
from bench import random_correlated_gaussian
lib = api.load(); D = 100
ic, mean, logdet = random_correlated_gaussian(D)
s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, 0)
s.nlive, s.num_repeats, s.seed, s.batch, s.max_ndead = 5000, 200, 3, 0, int(sys.argv[1]) if len(sys.argv) > 1 else 20000
L, P, keep = api.make_problem("corr_gaussian", D, 0, invcov=ic, mean=mean, logdet=logdet)
g = api.run(s, L, P)
print("ok", g["logZ"], g["nlike"], g["ndead"], g["t_total"], flush=True)
