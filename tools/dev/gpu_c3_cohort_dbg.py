"""dev: where the wall time of R clustered runs in step goes (PC_DEBUG=5: the cohort driver's own clocks, per scheduler group) and the
rounds / launches it needed.  usage: PC_DEBUG=5 [PC_REPEATS_SCHED=n] gpu_c3_cohort_dbg.py [c3|c4] [R] [ablate]"""
import ctypes as C, sys, time
sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api
from polychordlite_amd.repeats import run_repeats
lib = api.load()
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"; R = int(sys.argv[2]) if len(sys.argv) > 2 else 16; abl = int(sys.argv[3]) if len(sys.argv) > 3 else 0
kind, D, nDer, nlive, nr, box = {"c3": ("rastrigin", 10, 0, 1000, 30, (-5.12, 5.12)), "c4": ("twin_gaussian", 30, 1, 500, 40, (-1.0, 1.0))}[cfg]
s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, nDer)
s.nlive, s.num_repeats, s.do_clustering, s.ablate = nlive, nr, 1, abl
L, P, keep = api.make_problem(kind, D, nDer, *box)
for w in range(2):
    m, held = run_repeats(s, L, P, [400000 + 1000 * w + j for j in range(R)], max_in_flight=R); held = None
print("==== timed call", flush=True); sys.stderr.write("==== timed call\n"); sys.stderr.flush()
t0 = time.perf_counter()
m, held = run_repeats(s, L, P, [500000 + j for j in range(R)], max_in_flight=R)
print("wall %.1f ms, runs %.1f ms, rounds per run: %s" % ((time.perf_counter() - t0) * 1e3, m["t_runs_s"] * 1e3, [int(h["nrounds"]) for h in held]))
print("nbatches", [int(h["nbatches"]) for h in held], "nupdates", [int(h["nupdates"]) for h in held], "ncluster_dead", [int(h["ncluster_dead"]) for h in held])
