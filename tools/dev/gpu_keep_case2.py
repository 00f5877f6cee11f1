# where the one-wave and the general contraction part ways under the keep rule: first differing dead row, then runs cut short around it
import ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np
from polychordlite_amd import _ctypes_api as api
lib = api.load()
L, P, keep = api.make_problem("rastrigin", 10, 0, -5.12, 5.12)
def run(ab, mx=-1):
    s = api.Settings(); lib.pchip_settings_default(C.byref(s), 10, 0)
    s.nlive, s.num_repeats, s.seed, s.batch, s.do_clustering, s.compression_factor = 200, 2, 8222, 100, 1, 0.9
    s.ablate = ab; s.max_ndead = mx
    return api.run(s, L, P)
a, b = run(0), run(32)
n = min(a["ndead"], b["ndead"])
eq = np.all((a["dead"][:n] == b["dead"][:n]) | (np.isnan(a["dead"][:n]) & np.isnan(b["dead"][:n])), axis=1) & (a["logweights"][:n] == b["logweights"][:n])
i0 = int(np.nonzero(~eq)[0][0]); print("first differing dead row", i0, "of", n)
for i in range(max(0, i0 - 3), i0 + 3):
    print(i, "cl ", a["dead"][i, -1], a["dead"][i, -2], a["logweights"][i], "| gen", b["dead"][i, -1], b["dead"][i, -2], b["logweights"][i])
keys = ("ndead", "nlike", "niter", "ncluster", "ncluster_dead", "nupdates", "nbatches", "nrounds")
for m in range(max(1, i0 - 120), i0 + 40, 8):
    x, y = run(0, m), run(32, m)
    print(m, [x[k] for k in keys], [y[k] for k in keys], "" if all(x[k] == y[k] for k in keys[:6]) else "   <-- differ")
