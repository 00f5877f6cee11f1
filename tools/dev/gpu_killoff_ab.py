"""dev: the clustered kill-off by the one-wave kernel (pc_clus.hip k_killoff_cl) against the general kernel (settings.ablate bit 9):
BASELINE configs[2] / [3], t_final and t_total of each, and whether the two runs are the same run"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api
lib = api.load()
for kind, D, nDer, nlive, nr, box in (("rastrigin", 10, 0, 1000, 30, (-5.12, 5.12)), ("twin_gaussian", 30, 1, 500, 40, (-1.0, 1.0))):
    L, P, keep = api.make_problem(kind, D, nDer, *box)
    out = {}
    for ab in (0, 512, 0, 512):
        s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, nDer)
        s.nlive, s.num_repeats, s.seed, s.do_clustering, s.ablate = nlive, nr, 11, 1, ab
        r = api.run(s, L, P)
        out[ab] = r
        print(kind, "ablate", ab, "t_total %.2f ms t_final %.3f ms ndead %d ncluster_dead %d logZ %.12f" % (r["t_total"] * 1e3, r["t_final"] * 1e3, r["ndead"], r["ncluster_dead"], r["logZ"]))
    a, b = out[0], out[512]
    print("  same:", a["logZ"] == b["logZ"], np.array_equal(a["dead"], b["dead"]), np.array_equal(a["logweights"], b["logweights"]),
          "max |dlogw|", float(np.abs(a["logweights"] - b["logweights"]).max()), "dlogZ", a["logZ"] - b["logZ"])
