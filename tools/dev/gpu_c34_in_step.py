# R runs of a clustered BASELINE configuration in step on one device (pchip_run_repeats) against one run on its own: wall, evaluations
# per second, evaluations per lived dead point.  usage: gpu_c34_in_step.py [c3|c4|c3,c4] [R list] [B list] [samples]
import ctypes as C, json, os, sys, time
sys.path.insert(0, ".")
import numpy as np
from polychordlite_amd import _ctypes_api as api
from polychordlite_amd.repeats import run_repeats
lib = api.load()
which = (sys.argv[1] if len(sys.argv) > 1 else "c3,c4").split(",")
Rs = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "4,8,16,32").split(",")]
Bs = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0").split(",")]
nsamp = int(sys.argv[4]) if len(sys.argv) > 4 else 2
CFG = {"c3": ("rastrigin", 10, 0, 1000, 30, (-5.12, 5.12)), "c4": ("twin_gaussian", 30, 1, 500, 40, (-1.0, 1.0))}
out = {}
for name in which:
    kind, D, nDer, nlive, nr, box = CFG[name]
    L, P, keep = api.make_problem(kind, D, nDer, *box)
    rows = []
    for B in Bs:
        s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, nDer)
        s.nlive, s.num_repeats, s.feedback, s.do_clustering, s.batch = nlive, nr, 0, 1, B
        s.seed = 6999; api.run(s, L, P)
        t0 = time.perf_counter(); solo = []
        for i in range(3):
            s.seed = 7000 + i; g = api.run(s, L, P); solo.append((g["nlike"], int((g["logweights"] > g["logzero"]).sum())))
        ts = (time.perf_counter() - t0) / 3
        r = dict(B=B, runs=1, wall_ms=ts * 1e3, evals_per_s=float(np.mean([x[0] for x in solo]) / ts), lived_dead_per_s=float(np.mean([x[1] for x in solo]) / ts),
                 evals_per_lived_dead=float(np.sum([x[0] for x in solo]) / np.sum([x[1] for x in solo])))
        rows.append(r); print(name, json.dumps(r), flush=True)
        for R in Rs:
            for w in range(2):
                m, held = run_repeats(s, L, P, [400000 + 1000 * w + j for j in range(R)], max_in_flight=R); held = None
            best = None
            for k in range(nsamp):
                m, held = run_repeats(s, L, P, [500000 + 1000 * k + j for j in range(R)], max_in_flight=R)
                lived = sum(int((h["logweights"] > h["logzero"]).sum()) for h in held); held = None
                cur = dict(B=B, runs=R, wall_ms=m["t_runs_s"] * 1e3, evals_per_s=m["nlike"] / m["t_runs_s"], lived_dead_per_s=lived / m["t_runs_s"],
                           evals_per_lived_dead=m["nlike"] / lived, merged_logZ=m["logZ"], merged_logZerr=m["logZerr"])
                if best is None or cur["evals_per_s"] > best["evals_per_s"]: best = cur
            best["speedup_vs_solo"] = best["evals_per_s"] / rows[0]["evals_per_s"] if rows else None
            rows.append(best); print(name, json.dumps(best), flush=True)
    out[name] = rows
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/c34_in_step.json", "w"), indent=1)
