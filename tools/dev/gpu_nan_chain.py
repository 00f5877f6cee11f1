"""dev: the shape of test_clustered_contraction_with_a_chain_that_has_no_number under the three contraction kernels, twice each"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api
lib = api.load()
L, P, keep = api.make_problem("rastrigin", 10, 0, -5.12, 5.12)
res = {}
for ab in (0, 0, 1024, 1024, 32):
    s = api.Settings(); lib.pchip_settings_default(C.byref(s), 10, 0)
    s.nlive, s.num_repeats, s.seed, s.batch, s.do_clustering, s.compression_factor, s.epoch_discard = 200, 2, 8222, 100, 1, 0.9, 1
    if len(sys.argv) > 1: s.max_ndead = int(sys.argv[1])
    s.ablate = ab
    g = api.run(s, L, P)
    print(ab, g["ndead"], g["nlike"], g["niter"], g["ncluster"], g["ncluster_dead"], g["nupdates"], repr(g["logZ"]), {k: v for k, v in g["path"].items() if v})
    res.setdefault(ab, []).append(g)
a, c, b = res[0][0], res[1024][0], res[32][0]
for name, o in (("clp again", res[0][1]), ("serial", c), ("general", b)):
    n = min(a["ndead"], o["ndead"])
    d = np.nonzero(~np.all((a["dead"][:n] == o["dead"][:n]) | (np.isnan(a["dead"][:n]) & np.isnan(o["dead"][:n])), axis=1))[0]
    lw = np.nonzero(a["logweights"][:n] != o["logweights"][:n])[0]
    print(name, "first differing dead row", d[:3], "first differing logweight", lw[:3])
    if len(d):
        i = d[0]; print(" a:", a["dead"][i, -2:], a["logweights"][i], " o:", o["dead"][i, -2:], o["logweights"][i])
i0 = 4216
for name, o in (("clp", a), ("serial", c)):
    print(name)
    for i in range(i0, i0 + 14):
        print("  %d birth %.8f logL %.8f logw %.6g x0 %.6f" % (i, o["dead"][i, -2], o["dead"][i, -1], o["logweights"][i], o["dead"][i, 0]))
