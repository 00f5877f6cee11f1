"""BASELINE configs[4]: 100-D correlated Gaussian, nlive 5000, num_repeats 200 (developer script, not a test)"""
import ctypes as C, sys, time
sys.path.insert(0, ".")
import numpy as np
from polychordlite_amd import _ctypes_api as api
from bench import random_correlated_gaussian
lib = api.load()
D = 100
nlive = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
nr = int(sys.argv[2]) if len(sys.argv) > 2 else 200
B = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ic, mean, logdet = random_correlated_gaussian(D)          # the bench's matrix (bench.py --workload c5)
s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, 0)
s.nlive, s.num_repeats, s.seed, s.batch, s.profile = nlive, nr, 3, B, 1
s.feedback = int(__import__("os").environ.get("PC_FB", "0"))
L, P, keep = api.make_problem("corr_gaussian", D, 0, invcov=ic, mean=mean, logdet=logdet)
t0 = time.time(); g = api.run(s, L, P); dt = time.time() - t0
print(f"C5 D={D} nlive={nlive} nr={nr} B={g['batch']}: logZ {g['logZ']:.4f} +- {g['logZerr']:.4f} (truth ~0 up to prior truncation) ndead {g['ndead']} nlike {g['nlike']} "
      f"t {g['t_total']:.3f}s (wall {dt:.2f}) -> {g['nlike']/g['t_total']:.3e} evals/s rounds {g['nrounds']} updates {g['nupdates']}",
      {k: (round(v['total_s'] * 1e3, 1), v['launches']) for k, v in g['kernel_time'].items()}, flush=True)
