"""dev: one configuration alone (twice) and in step (twice), log Z to the last digit -- is a difference between the two a property of
the kernels (the same every time) or of timing?  usage: gpu_step_vs_solo.py [ablate]"""
import ctypes as C, sys
sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api
from polychordlite_amd.repeats import run_repeats
lib = api.load()
abl = int(sys.argv[1]) if len(sys.argv) > 1 else 0
kind, D, nDer, nlive, nr, clus, box = "twin_gaussian", 30, 1, 120, 40, 1, (-1.0, 1.0)
L, P, keep = api.make_problem(kind, D, nDer, *box)
def settings(seed):
    s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, nDer)
    s.nlive, s.num_repeats, s.seed, s.do_clustering, s.ablate = nlive, nr, seed, clus, abl
    return s
seeds = [31, 32, 33, 34, 35, 36, 37, 38, 39]
for rep in range(2):
    print("solo   ", ["%.15f" % api.run(settings(sd), L, P)["logZ"] for sd in seeds[:4]])
for rep in range(2):
    m, runs = run_repeats(settings(0), L, P, seeds, max_in_flight=len(seeds))
    print("in step", ["%.15f" % r["logZ"] for r in runs[:4]], [int(r["nrounds"]) for r in runs[:4]])
import numpy as np
one = api.run(settings(31), L, P)
m, runs = run_repeats(settings(0), L, P, seeds, max_in_flight=len(seeds))
r = runs[0]
print("counters", [(k, one[k], r[k]) for k in ("ndead", "nlike", "niter", "nupdates", "nbatches", "ncluster_dead")])
d = one["dead"] != r["dead"]
d &= ~(np.isnan(one["dead"]) & np.isnan(r["dead"]))
rows = np.where(d.any(1))[0]
print("rows that differ:", rows.size, "first", rows[:5], "columns", np.where(d.any(0))[0][:10], "of", one["dead"].shape)
lw = np.where(one["logweights"] != r["logweights"])[0]
print("logweights differ at", lw.size, "first", lw[:5], "max abs", np.abs(one["logweights"] - r["logweights"])[one["logweights"] > -1e29].max() if lw.size else 0)
if rows.size:
    i = rows[0]; c = np.where(d[i])[0]
    print("row", i, "cols", c[:6], one["dead"][i, c[:3]], r["dead"][i, c[:3]])
