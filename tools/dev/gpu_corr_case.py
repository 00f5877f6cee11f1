"""one small correlated-Gaussian run through the engine (developer script): D nlive nr B"""
import ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np
from polychordlite_amd import _ctypes_api as api
from bench import random_correlated_gaussian
D, nlive, nr, B = (int(x) for x in sys.argv[1:5])
lib = api.load()
ic, mean, logdet = random_correlated_gaussian(D)
s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, 0)
s.nlive, s.num_repeats, s.seed, s.batch, s.max_ndead = nlive, nr, 3, B, 6 * nlive
L, P, keep = api.make_problem("corr_gaussian", D, 0, invcov=ic, mean=mean, logdet=logdet)
g = api.run(s, L, P)
print("ok", D, nlive, nr, B, g["logZ"], g["nlike"], g["ndead"], flush=True)
