"""Dev script (not pytest): time the metric config (20-D Gaussian, nlive 2000, nr 40) and print the
engine's kernel stopwatch + the contraction kernel's phase counters.  usage: gpu_c2_prof.py [feedback] [nruns]"""
import ctypes as C
import sys
import time

sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api

fb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nruns = int(sys.argv[2]) if len(sys.argv) > 2 else 3
D, nDer = 20, 1
lib = api.load()
for it in range(nruns):
    s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, nDer)
    s.nlive = 2000; s.num_repeats = 40; s.seed = 7 + it; s.batch = 1000; s.profile = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    s.feedback = fb if it == nruns - 1 else 0
    L, P, keep = api.make_problem("gaussian", D, nDer)
    t0 = time.perf_counter()
    r = api.run(s, L, P)
    dt = time.perf_counter() - t0
    print(f"run {it}: wall {dt*1e3:.1f} ms  t_total {r['t_total']*1e3:.1f}  loop {r['t_loop']*1e3:.1f}  setup {r['t_setup']*1e3:.1f} results {r['t_results']*1e3:.1f} teardown {r['t_teardown']*1e3:.1f}"
          f"  nlike {r['nlike']} ndead {r['ndead']} logZ {r['logZ']:.4f}  evals/s {r['nlike']/dt/1e6:.1f}M", flush=True)
    print("   kernels:", {k: (round(v['total_s']*1e3, 2), v['launches']) for k, v in r['kernel_time'].items()}, flush=True)
