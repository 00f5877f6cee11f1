import os
os.environ.setdefault("PC_DEBUG", "3")   # developer counters of the engine
import ctypes as C, sys
sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api
lib = api.load()
for (kind, D, nDer, nlive, nr, lo, hi) in (("rastrigin", 10, 0, 1000, 30, -5.12, 5.12), ("twin_gaussian", 30, 1, 500, 40, -1.0, 1.0)):
    s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, nDer)
    s.nlive, s.num_repeats, s.seed, s.profile, s.do_clustering, s.feedback = nlive, nr, 1, 1, 1, 3
    L, P, keep = api.make_problem(kind, D, nDer, lo, hi)
    g = api.run(s, L, P)
    print(f"{kind}: logZ {g['logZ']:.3f}+-{g['logZerr']:.3f} t_total {g['t_total']:.2f}s (loop {g['t_loop']:.2f}) niter {g['niter']} ndead {g['ndead']} nlike {g['nlike']} batches {g['nbatches']} rounds {g['nrounds']} updates {g['nupdates']} nclusters {g['ncluster_dead']} " + " ".join(f"{n}={v['total_s']:.2f}s/{v['launches']}" for n, v in g['kernel_time'].items()), flush=True)
