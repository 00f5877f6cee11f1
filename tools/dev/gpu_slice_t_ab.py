"""dev: the lane = chain sampling kernel (settings.ablate bit 6) against k_slice on the same runs: every number, and the kernel's time"""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api
lib = api.load()
def run(kind, D, nDer, nlive, nr, ab, seed=11, box=(0.0, 1.0), mnd=-1, batch=0):
    s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, nDer)
    s.nlive, s.num_repeats, s.seed, s.ablate, s.max_ndead, s.batch = nlive, nr, seed, ab, mnd, batch
    s.profile = 1 << 2
    L, P, keep = api.make_problem(kind, D, nDer, *box)
    t0 = time.perf_counter(); g = api.run(s, L, P); t1 = time.perf_counter()
    g["wall"] = t1 - t0
    return g
cases = [("gaussian", 20, 2, 2000, 40, {}), ("gaussian", 8, 0, 300, 16, {}), ("gaussian", 16, 1, 500, 32, {}), ("gaussian", 24, 2, 400, 48, {}),
         ("gaussian", 5, 2, 200, 25, dict(box=(-0.5, 1.5))), ("gaussian", 20, 2, 2000, 40, dict(batch=700))]
for kind, D, nDer, nlive, nr, kw in cases:
    run(kind, D, nDer, nlive, nr, 0, **kw)
    a = run(kind, D, nDer, nlive, nr, 0, **kw); b = run(kind, D, nDer, nlive, nr, int(sys.argv[1]) if len(sys.argv) > 1 else 64, **kw)
    same = all(a[k] == b[k] for k in ("ndead", "nlike", "niter", "nupdates")) and a["logZ"] == b["logZ"] and np.array_equal(a["dead"], b["dead"]) and np.array_equal(a["live"], b["live"]) and np.array_equal(a["logweights"], b["logweights"])
    ka, kb = a["kernel_time"]["k_slice"], b["kernel_time"]["k_slice"]
    print(kind, D, nDer, nlive, nr, kw, "IDENTICAL" if same else "DIFFERENT", "ndead", a["ndead"], b["ndead"], "nlike", a["nlike"], b["nlike"], "logZ", a["logZ"], b["logZ"],
          "k_slice us %.1f vs %.1f" % (ka["total_s"] / ka["launches"] * 1e6, kb["total_s"] / kb["launches"] * 1e6), "wall ms %.2f vs %.2f" % (a["wall"] * 1e3, b["wall"] * 1e3), flush=True)
    if not same:
        nd = min(a["ndead"], b["ndead"])
        bad = np.nonzero(np.any(a["dead"][:nd] != b["dead"][:nd], axis=1))[0]
        if len(bad):
            i = bad[0]; print("  first differing dead row", i, "cols", np.nonzero(a["dead"][i] != b["dead"][i])[0][:8], a["dead"][i][-4:], b["dead"][i][-4:])
