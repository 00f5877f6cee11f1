"""dev (checker, like tests/): runs in step against the same runs alone, bit for bit, over every compiled width of the lane-per-chain
kernels (nDims 1 .. 24), derived parameters 0 .. 3, unit and other prior boxes, nurseries that do not fill their last wavefront.
usage: fuzz_in_step.py [first_D last_D [seed_offset]]"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api
from polychordlite_amd.repeats import run_repeats
lib = api.load()
d0, d1 = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1, 24)
off = int(sys.argv[3]) if len(sys.argv) > 3 else 0
bad = 0
for D in range(d0, d1 + 1):
    nDer = D % 4
    box = (None if D % 2 else (-0.5 + 0.01 * D, 1.25))
    nlive = 100 + 37 * ((D + off) % 5) + 16 * (off % 7)
    nr = max(2, 2 * D if D < 8 else D + (D % 3))
    L, P, keep = api.make_problem("gaussian", D, nDer, *box) if box else api.make_problem("gaussian", D, nDer)
    def settings(seed):
        s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, nDer)
        s.nlive, s.num_repeats, s.seed = nlive, nr, seed
        return s
    seeds = [900 + 1000 * off + D * 10 + j for j in range(5 + off % 4)]
    singles = [api.run(settings(sd), L, P) for sd in seeds]
    merged, runs = run_repeats(settings(0), L, P, seeds, max_in_flight=len(seeds))
    ok = True
    for one, r in zip(singles, runs):
        ok = ok and all(one[k] == r[k] for k in ("ndead", "nlike", "niter", "nupdates", "nbatches"))
        ok = ok and one["logZ"] == r["logZ"] and np.array_equal(one["dead"], r["dead"], equal_nan=True) and np.array_equal(one["logweights"], r["logweights"]) \
            and np.array_equal(one["live"], r["live"], equal_nan=True) and np.array_equal(one["post_mean"], r["post_mean"], equal_nan=True)
    print(f"nDims {D:2d} nDerived {nDer} nlive {nlive} nr {nr} box {box}: {'same' if ok else 'DIFFERENT'} ({singles[0]['ndead']} dead points, logZ {singles[0]['logZ']:.4f})", flush=True)
    bad += 0 if ok else 1
print("all the same" if bad == 0 else f"{bad} shapes differ")
sys.exit(1 if bad else 0)
