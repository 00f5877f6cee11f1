# evidence statistics of the clustered BASELINE configurations over many seeds against the analytic values
# (Rastrigin 10-D on [-5.12, 5.12]^10: -10 ln 10.24; twin Gaussian 30-D on [-1, 1]^30: -30 ln 2).  usage: gpu_cluster_stats.py [nseeds]
import ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np
from polychordlite_amd import _ctypes_api as api
lib = api.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for name, kind, D, nDer, nlive, nr, box, truth in (("C3 Rastrigin 10-D", "rastrigin", 10, 0, 1000, 30, (-5.12, 5.12), -10 * np.log(10.24)),
                                                   ("C4 twin Gaussian 30-D", "twin_gaussian", 30, 1, 500, 40, (-1.0, 1.0), -30 * np.log(2.0))):
    s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, nDer)
    s.nlive, s.num_repeats, s.feedback, s.do_clustering = nlive, nr, 0, 1
    L, P, keep = api.make_problem(kind, D, nDer, *box)
    z, e, nc = [], [], []
    for i in range(n):
        s.seed = 7000 + i
        g = api.run(s, L, P)
        z.append(g["logZ"]); e.append(g["logZerr"]); nc.append(g["ncluster"] + g["ncluster_dead"])
    z, e = np.array(z), np.array(e)
    print("%s, %d runs: mean logZ %.3f +/- %.3f (truth %.3f), scatter %.3f, mean reported error %.3f, chi2/n %.2f, clusters found %d..%d" %
          (name, n, z.mean(), z.std(ddof=1) / np.sqrt(n), truth, z.std(ddof=1), e.mean(), np.mean(((z - truth) / e) ** 2), min(nc), max(nc)))
