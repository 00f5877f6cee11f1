"""BASELINE configs[4] twice in one process (developer script): where the wall time of the first run goes"""
import ctypes as C, sys, time
sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api
from bench import random_correlated_gaussian
lib = api.load(); D = 100
ic, mean, logdet = random_correlated_gaussian(D)
for it in range(2):
    s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, 0)
    s.nlive, s.num_repeats, s.seed, s.batch, s.profile = 5000, 200, 3, 0, 0
    L, P, keep = api.make_problem("corr_gaussian", D, 0, invcov=ic, mean=mean, logdet=logdet)
    t0 = time.time(); g = api.run(s, L, P); dt = time.time() - t0
    print(it, "wall %.3f" % dt, {k: round(g[k], 3) for k in g if k.startswith("t_")}, g["nlike"], flush=True)
