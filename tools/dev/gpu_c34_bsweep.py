# chains per nursery (B) at the clustered BASELINE configurations: wall, evaluations per lived dead point (the reference's
# linear mode: C3 155.5, C4 ~ 200), clusters found, reported error and scatter.  usage: gpu_c34_bsweep.py [nseeds] [c3|c4|c3,c4]
import ctypes as C, json, sys, time
sys.path.insert(0, ".")
import numpy as np
from polychordlite_amd import _ctypes_api as api
lib = api.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
which = (sys.argv[2] if len(sys.argv) > 2 else "c3,c4").split(",")
CFG = {"c3": ("rastrigin", 10, 0, 1000, 30, (-5.12, 5.12), -10 * np.log(10.24), (16, 32, 64, 125, 250, 500)),
       "c4": ("twin_gaussian", 30, 1, 500, 40, (-1.0, 1.0), -30 * np.log(2.0), (16, 32, 64, 125, 250))}
out = {}
for name in which:
    kind, D, nDer, nlive, nr, box, truth, Bs = CFG[name]
    s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, nDer)
    s.nlive, s.num_repeats, s.feedback, s.do_clustering = nlive, nr, 0, 1
    L, P, keep = api.make_problem(kind, D, nDer, *box)
    rows = []
    for B in Bs:
        s.batch = B
        s.seed = 6999; api.run(s, L, P)          # (untimed: sizes the block cache)
        z, e, nc, wall, epd, nl, lived, rounds = [], [], [], [], [], [], [], []
        for i in range(n):
            s.seed = 7000 + i
            t0 = time.perf_counter(); g = api.run(s, L, P); dt = time.perf_counter() - t0
            nlv = int((g["logweights"] > g["logzero"]).sum())
            z.append(g["logZ"]); e.append(g["logZerr"]); nc.append(g["ncluster"] + g["ncluster_dead"]); wall.append(dt)
            nl.append(g["nlike"]); lived.append(nlv); epd.append(g["nlike"] / nlv); rounds.append(g["nrounds"])
        z = np.array(z)
        r = dict(B=B, wall_ms=float(np.mean(wall) * 1e3), evals_per_lived_dead=float(np.mean(epd)), nlike=float(np.mean(nl)), lived_dead=float(np.mean(lived)),
                 evals_per_s=float(np.sum(nl) / np.sum(wall)), lived_dead_per_s=float(np.sum(lived) / np.sum(wall)), clusters=[int(min(nc)), int(max(nc))],
                 logZ_mean=float(z.mean()), logZ_scatter=float(z.std(ddof=1)) if n > 1 else None, logZerr_mean=float(np.mean(e)), truth=float(truth), rounds=float(np.mean(rounds)))
        rows.append(r)
        print(name, json.dumps(r), flush=True)
    out[name] = rows
json.dump(out, open("gpurun_out/c34_bsweep.json", "w"), indent=1)
