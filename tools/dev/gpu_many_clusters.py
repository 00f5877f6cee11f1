"""dev: which Rastrigin configurations keep more than 128 clusters alive at once (the engine's initial capacity)?"""
import ctypes as C, sys, time
sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api
lib = api.load()
for D, nlive, nr in ((3, 8000, 9), (4, 8000, 12), (2, 8000, 6), (3, 4000, 9), (5, 8000, 15)):
    s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, 0)
    s.nlive, s.num_repeats, s.seed, s.do_clustering = nlive, nr, 3, 1
    L, P, keep = api.make_problem("rastrigin", D, 0, -5.12, 5.12)
    t0 = time.time(); g = api.run(s, L, P)
    print(D, nlive, nr, "peak", g["ncluster_peak"], "dead clusters", g["ncluster_dead"], "logZ %.3f +- %.3f (truth %.3f)" % (g["logZ"], g["logZerr"], -2.326314 * D),
          "ndead", g["ndead"], "nlike", g["nlike"], "failed", g["nlike_failed"], "%.1fs" % (time.time() - t0), flush=True)
