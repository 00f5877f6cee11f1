"""dev (checker): one run with the lane-per-chain sampling kernel (ablate 64), the packed bases kernels (128), both (192) against the
one-run kernels (0), bit for bit, over the shapes of fuzz_in_step.py.  usage: fuzz_ablate_t.py first_D last_D"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api
lib = api.load()
d0, d1 = int(sys.argv[1]), int(sys.argv[2])
for D in range(d0, d1 + 1):
    nDer = D % 4
    box = (None if D % 2 else (-0.5 + 0.01 * D, 1.25))
    nlive = 100 + 37 * (D % 5)
    nr = max(2, 2 * D if D < 8 else D + (D % 3))
    L, P, keep = api.make_problem("gaussian", D, nDer, *box) if box else api.make_problem("gaussian", D, nDer)
    out = []
    for ab in (0, 64, 128, 192):
        s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, nDer)
        s.nlive, s.num_repeats, s.seed, s.ablate = nlive, nr, 900 + D * 10, ab
        out.append(api.run(s, L, P))
    a = out[0]
    res = []
    for ab, b in zip((64, 128, 192), out[1:]):
        same = a["nlike"] == b["nlike"] and a["logZ"] == b["logZ"] and np.array_equal(a["dead"], b["dead"], equal_nan=True)
        first = -1
        if not same and a["dead"].shape == b["dead"].shape:
            diff = np.argwhere(~((a["dead"] == b["dead"]) | (np.isnan(a["dead"]) & np.isnan(b["dead"]))))
            first = tuple(diff[0]) if len(diff) else -2
        res.append(f"{ab}: {'same' if same else 'DIFFERENT first at ' + str(first)}")
    print(f"nDims {D} nDerived {nDer} nr {nr} nT {2*D+nDer+2}: " + "; ".join(res), flush=True)
