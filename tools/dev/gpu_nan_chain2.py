"""dev: where k_consume_clp and k_consume_cl part ways on the shape of test_clustered_contraction_with_a_chain_that_has_no_number: counters at max_ndead"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api
lib = api.load()
L, P, keep = api.make_problem("rastrigin", 10, 0, -5.12, 5.12)
lo, hi, st = [int(x) for x in sys.argv[1:4]]
for nd in range(lo, hi, st):
    out = []; gs = []
    for ab in (0, 1024):
        s = api.Settings(); lib.pchip_settings_default(C.byref(s), 10, 0)
        s.nlive, s.num_repeats, s.seed, s.batch, s.do_clustering, s.compression_factor, s.epoch_discard = 200, 2, 8222, 100, 1, 0.9, 1
        s.max_ndead = nd; s.ablate = ab
        g = api.run(s, L, P); gs.append(g)
        out.append((g["ndead"], g["nlike"], g["niter"], g["nlike_failed"], g["ncluster"], g["ncluster_dead"], g["nupdates"], g["nbatches"], g["nrounds"]))
    same_live = np.array_equal(gs[0]["live"], gs[1]["live"], equal_nan=True)
    same_dead = np.array_equal(gs[0]["dead"], gs[1]["dead"], equal_nan=True)
    print(nd, "clp", out[0], "serial", out[1], "SAME" if out[0] == out[1] else "DIFF", "live", same_live, "dead", same_dead)
