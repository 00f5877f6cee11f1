# the keep rule (settings.epoch_discard = 0) at a shape with NaN chains and dozens of tiny clusters: one-wave kernel twice, general kernel, oracle
import ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np
from polychordlite_amd import _ctypes_api as api
from tests import oracle_api as orc
lib = api.load()
L, P, keep = api.make_problem("rastrigin", 10, 0, -5.12, 5.12)
Lo, Po, k2 = orc.make_problem("rastrigin", 10, -5.12, 5.12)
for disc in (0, 1):
    out = []
    for ab in (0, 0, 32):
        s = api.Settings(); lib.pchip_settings_default(C.byref(s), 10, 0)
        s.nlive, s.num_repeats, s.seed, s.batch, s.do_clustering, s.compression_factor, s.epoch_discard = 200, 2, 8222, 100, 1, 0.9, disc
        s.ablate = ab
        g = api.run(s, L, P); out.append(g)
        print("disc", disc, "ablate", ab, {k: g[k] for k in ("ndead", "nlike", "niter", "ncluster", "ncluster_dead", "nupdates", "logZ")}, flush=True)
    so = orc.settings(10, 0, nlive=200, num_repeats=2, seed=8222, batch=100, do_clustering=1, compression_factor=0.9, epoch_discard=disc)
    o = orc.run(so, Lo, Po)
    print("disc", disc, "oracle   ", {k: o[k] for k in ("ndead", "nlike", "niter", "ncluster", "ncluster_dead", "logZ")}, flush=True)
    for nm, g in zip(("cl", "cl2", "gen"), out):
        n = min(g["ndead"], o["ndead"])
        bad = np.nonzero(~np.all(np.isclose(g["dead"][:n, :10], o["dead"][:n, :10], rtol=1e-7, atol=1e-9, equal_nan=True), axis=1))[0]
        print("   ", nm, "first dead row that differs from the oracle's:", (int(bad[0]) if bad.size else None), "of", n)
