"""BASELINE configs[4] over several seeds (developer script): evidence and posterior statistics.  usage: gpu_c5_stats.py [nseeds]"""
import ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np
from polychordlite_amd import _ctypes_api as api
from bench import random_correlated_gaussian
lib = api.load(); D = 100
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ic, mean, logdet = random_correlated_gaussian(D)
L, P, keep = api.make_problem("corr_gaussian", D, 0, invcov=ic, mean=mean, logdet=logdet)
z, e, pm = [], [], []
for i in range(n):
    s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, 0)
    s.nlive, s.num_repeats, s.seed, s.batch = 5000, 200, 900 + i, 0
    g = api.run(s, L, P)
    z.append(g["logZ"]); e.append(g["logZerr"]); pm.append(np.abs(g["post_mean"][:D] - 0.5).max())
    print(i, "logZ %.4f +- %.4f  t %.2f s  max |post mean - 0.5| %.4f" % (g["logZ"], g["logZerr"], g["t_total"], pm[-1]), flush=True)
z = np.array(z)
print("runs %d: mean logZ %.3f +/- %.3f (s.e.m.), scatter %.3f, mean reported error %.3f" % (n, z.mean(), z.std(ddof=1) / np.sqrt(n), z.std(ddof=1), np.mean(e)))
