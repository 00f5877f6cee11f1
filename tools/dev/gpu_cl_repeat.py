"""dev: the same clustered run several times in one process (block cache reuse) -- any difference between the repetitions is a read
of memory the run did not write.  usage: [PC_POISON=1] gpu_cl_repeat.py [ablate]"""
import ctypes as C, sys
sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api
lib = api.load()
ab = int(sys.argv[1]) if len(sys.argv) > 1 else 0
L, P, keep = api.make_problem("rastrigin", 10, 0, -5.12, 5.12)
def run(mnd, seed=8222):
    s = api.Settings(); lib.pchip_settings_default(C.byref(s), 10, 0)
    s.nlive, s.num_repeats, s.seed, s.do_clustering, s.compression_factor, s.max_ndead, s.batch, s.ablate = 200, 2, seed, 1, 0.9, mnd, 100, ab
    g = api.run(s, L, P)
    return g["ndead"], g["nlike"], g["niter"], round(g["logZ"], 9), g["ncluster_dead"], g["nupdates"]
for mnd in (-1, -1, 2116, -1, 4500, 2116, -1):
    print(mnd, run(mnd), flush=True)
