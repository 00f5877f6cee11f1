# evidence statistics of the metric configuration over many seeds: mean and scatter of logZ against the analytic value
# (0) and against the run's own error estimate; posterior mean / sd per dimension.  usage: gpu_c2_stats.py [nseeds]
import ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np
from polychordlite_amd import _ctypes_api as api
lib = api.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
s = api.Settings(); lib.pchip_settings_default(C.byref(s), 20, 2)
s.nlive, s.num_repeats, s.feedback = 2000, 40, 0
L, P, keep = api.make_problem("gaussian", 20, 2)
z, e, pm, pv = [], [], [], []
for i in range(n):
    s.seed = 5000 + i
    g = api.run(s, L, P)
    z.append(g["logZ"]); e.append(g["logZerr"]); pm.append(g["post_mean"][:20].mean()); pv.append(g["post_var"][:20].mean())
z, e = np.array(z), np.array(e)
print("runs %d: mean logZ %.4f +/- %.4f (standard error), scatter %.4f, mean reported error %.4f, chi2/n of (logZ/err) %.3f" %
      (n, z.mean(), z.std(ddof=1) / np.sqrt(n), z.std(ddof=1), e.mean(), np.mean((z / e) ** 2)))
print("posterior mean %.5f (0.5), posterior sd %.5f (0.1)" % (np.mean(pm), np.sqrt(np.mean(pv))))
