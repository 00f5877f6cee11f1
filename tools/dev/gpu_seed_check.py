"""dev: does the wall time of sixteen clustered runs in step depend on the seeds?  usage: gpu_seed_check.py [c3|c4]"""
import ctypes as C, sys, time
sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api
from polychordlite_amd.repeats import run_repeats
lib = api.load()
cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
kind, D, nDer, nlive, nr, box = {"c3": ("rastrigin", 10, 0, 1000, 30, (-5.12, 5.12)), "c4": ("twin_gaussian", 30, 1, 500, 40, (-1.0, 1.0))}[cfg]
s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, nDer)
s.nlive, s.num_repeats, s.do_clustering = nlive, nr, 1
L, P, keep = api.make_problem(kind, D, nDer, *box)
for base in (400000, 500000, 310000, 311000, 312000, 501000):
    m, held = run_repeats(s, L, P, [base + j for j in range(16)], max_in_flight=16)
    print(base, "%.1f ms" % (m["t_runs_s"] * 1e3), "%.0f M evals/s" % (m["nlike"] / m["t_runs_s"] / 1e6), "peaks", sorted(int(h["ncluster_peak"]) for h in held), "rounds", max(int(h["nrounds"]) for h in held), flush=True)
    held = None
