"""one configuration of tools/dev/fuzz_parity.py (developer script): kind D nlive nr B clustering cf max_ndead seed"""
import ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np
from polychordlite_amd import _ctypes_api as api
from tests import oracle_api as orc
kind = sys.argv[1]; D, nlive, nr, B, clu = (int(x) for x in sys.argv[2:7]); cf = float(sys.argv[7]); maxnd = int(sys.argv[8]); seed = int(sys.argv[9])
lib = api.load(); olib = orc.load()
nDer = 2 if kind == "gaussian" else (1 if kind == "twin_gaussian" else 0)
kw = dict(nlive=nlive, num_repeats=nr, seed=seed, do_clustering=clu, compression_factor=cf, max_ndead=maxnd)
s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, nDer)
for k, v in kw.items(): setattr(s, k, v)
s.batch = B
lo, hi = (-5.12, 5.12) if kind == "rastrigin" else ((-1.0, 1.0) if kind == "twin_gaussian" else (None, None))
L, P, keep = api.make_problem(kind, D, nDer, lo, hi)
g = api.run(s, L, P)
so = orc.settings(D, nDer, batch=g["batch"], **kw)
Lo, Po, keep2 = orc.make_problem(kind, D, *(() if lo is None else (lo, hi)))
o = orc.run(so, Lo, Po)
print("engine", g["ndead"], g["nlike"], g["niter"], g["ncluster_dead"], g["logZ"], "oracle", o["ndead"], o["nlike"], o["niter"], o["ncluster_dead"], o["logZ"])
n = min(len(g["dead"]), len(o["dead"]))
d = np.abs(g["dead"][:n] - o["dead"][:n]).max(axis=1)
bad = np.nonzero(d > 1e-7)[0]
print("first differing dead row:", bad[:3], "of", n)
