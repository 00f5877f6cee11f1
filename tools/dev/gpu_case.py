"""dev: one parity case against the oracle; usage: gpu_case.py kind D nDer nlive nr B [general] [clustering]"""
import ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np
from polychordlite_amd import _ctypes_api as api
from tests import oracle_api as orc
kind, D, nDer, nlive, nr, B = sys.argv[1], *[int(x) for x in sys.argv[2:7]]
general = int(sys.argv[7]) if len(sys.argv) > 7 else 0
clus = int(sys.argv[8]) if len(sys.argv) > 8 else 0
BOX = {"gaussian": (None, None), "rastrigin": (-5.12, 5.12), "twin_gaussian": (-1.0, 1.0)}
lo, hi = BOX[kind]
lib = api.load()
s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, nDer)
s.nlive, s.num_repeats, s.seed, s.batch, s.force_general, s.do_clustering = nlive, nr, 5, B, general, clus
L, P, keep = api.make_problem(kind, D, nDer, lo, hi)
g = api.run(s, L, P)
so = orc.settings(D, nDer, nlive=nlive, num_repeats=nr, seed=5, batch=B, do_clustering=clus)
Lo, Po, k2 = orc.make_problem(kind, D, lo, hi)
o = orc.run(so, Lo, Po)
print({k: (g[k], o[k]) for k in ("ndead", "nlike", "niter", "nbatches", "logZ")})
n = min(g["ndead"], o["ndead"])
d = np.abs(g["dead"][:n] - o["dead"][:n]).max(axis=1)
bad = np.nonzero(d > 1e-7)[0]
print("first differing dead row", bad[:1], "of", n)
