"""dev (checker): runs in step against the same runs alone at sizes where the phantom arrays are compacted and grown, the dead
arrays grow, and the runs end rounds apart: the metric configuration and two others, bit for bit."""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api
from polychordlite_amd.repeats import run_repeats
lib = api.load()
bad = 0
for (D, nDer, nlive, nr, R, box) in [(20, 2, 2000, 40, 6, None), (10, 0, 1000, 30, 9, (-0.2, 1.1)), (24, 1, 1500, 48, 5, None), (6, 3, 3000, 12, 12, None)]:
    L, P, keep = api.make_problem("gaussian", D, nDer, *box) if box else api.make_problem("gaussian", D, nDer)
    def settings(seed):
        s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, nDer)
        s.nlive, s.num_repeats, s.seed = nlive, nr, seed
        return s
    seeds = [7000 + D * 10 + j for j in range(R)]
    singles = [api.run(settings(sd), L, P) for sd in seeds]
    merged, runs = run_repeats(settings(0), L, P, seeds, max_in_flight=R)
    ok = True
    for one, r in zip(singles, runs):
        ok = ok and all(one[k] == r[k] for k in ("ndead", "nlike", "niter", "nupdates", "nbatches"))
        ok = ok and one["logZ"] == r["logZ"] and np.array_equal(one["dead"], r["dead"], equal_nan=True) and np.array_equal(one["logweights"], r["logweights"]) \
            and np.array_equal(one["live"], r["live"], equal_nan=True) and np.array_equal(one["post_mean"], r["post_mean"], equal_nan=True)
    print(f"nDims {D} nDerived {nDer} nlive {nlive} nr {nr}, {R} runs: {'same' if ok else 'DIFFERENT'} ({singles[0]['ndead']} dead points, rounds {sorted(r['nrounds'] for r in singles)})", flush=True)
    bad += 0 if ok else 1
print("all the same" if bad == 0 else f"{bad} shapes differ")
sys.exit(1 if bad else 0)
