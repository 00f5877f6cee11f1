"""dev: the three evidence estimates of repeated clustered runs -- each run's own (the clusters' volume bookkeeping, run_time_info.f90:211-296),
the replay of each run alone (merge of one run: rank order and live counts only, no clusters), the replay of the union of all runs --
next to the analytic value.  usage: gpu_merge_bias.py [c3|c4] [runs]"""
import ctypes as C, json, sys
sys.path.insert(0, ".")
import numpy as np
from polychordlite_amd import _ctypes_api as api
from polychordlite_amd import merge as mg
from polychordlite_amd.repeats import run_repeats
lib = api.load()
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"; R = int(sys.argv[2]) if len(sys.argv) > 2 else 32
kind, D, nDer, nlive, nr, box, truth = {"c3": ("rastrigin", 10, 0, 1000, 30, (-5.12, 5.12), -10 * np.log(10.24)),
                                        "c4": ("twin_gaussian", 30, 1, 500, 40, (-1.0, 1.0), -30 * np.log(2.0))}[cfg]
s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, nDer)
s.nlive, s.num_repeats, s.do_clustering = nlive, nr, 1
L, P, keep = api.make_problem(kind, D, nDer, *box)
m, runs = run_repeats(s, L, P, [8000 + j for j in range(R)], max_in_flight=min(R, 32))
own = np.array([r["logZ"] for r in runs]); err = np.array([r["logZerr"] for r in runs])
alone = np.array([mg.merge_runs(r, None, D, nDer)["logZ_replay"] for r in runs])
sem = lambda x: x.std(ddof=1) / np.sqrt(x.size)
lme = lambda x: float(np.log(np.mean(np.exp(x - x.max()))) + x.max())
print(json.dumps({"config": cfg, "runs": R, "truth": float(truth),
                  "own_mean": float(own.mean()), "own_sem": float(sem(own)), "own_scatter": float(own.std(ddof=1)), "own_reported_err": float(err.mean()),
                  "own_log_mean_Z": lme(own),
                  "replay_alone_mean": float(alone.mean()), "replay_alone_sem": float(sem(alone)), "replay_minus_own_mean": float((alone - own).mean()), "replay_minus_own_sd": float((alone - own).std(ddof=1)),
                  "replay_alone_log_mean_Z": lme(alone),
                  "union": float(m["logZ"]), "union_err": float(m["logZerr"]), "union_evidence_rule": m["evidence_rule"], "union_nclustered": m["nclustered"],
                  "union_replay": float(m["logZ_replay"]), "union_replay_err": float(m["logZerr_replay"]),
                  "union_post_mean_absmax": float(np.abs(m["post_mean"]).max())}))
