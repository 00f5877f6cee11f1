"""Developer fuzz (not pytest; uses the oracle as the checker, like tests/): random small configurations of the built-in
likelihoods through the engine's default (production) path against the oracle -- counters exact, logZ 1e-8, dead rows 1e-7.
usage: fuzz_parity.py [ncases] [seed]"""
import ctypes as C, os, sys, time
sys.path.insert(0, ".")
import numpy as np
from polychordlite_amd import _ctypes_api as api
from tests import oracle_api as orc

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
lib = api.load(); olib = orc.load()
bad = 0
for case in range(ncases):
    kind = rng.choice(["gaussian", "gaussian", "corr_gaussian", "rastrigin", "twin_gaussian"] if not os.environ.get("FUZZ_CLUSTERED") else ["rastrigin", "rastrigin", "twin_gaussian"])
    D = int(rng.choice([2, 3, 5, 8, 13, 20, 24, 25, 31, 32, 33, 40, 64, 65, 70, 100])) if kind != "twin_gaussian" else int(rng.choice([2, 4, 10, 30]))
    if kind == "rastrigin": D = min(D, 10)
    nlive = int(rng.choice([25, 50, 100, 200, 400])); nr = int(rng.choice([1, 2, 5, D, 2 * D])) if D <= 20 else int(rng.choice([2, 5, 10]))
    nr = max(1, min(nr, 60))
    clustering = int(kind in ("rastrigin", "twin_gaussian") and rng.random() < 0.7)
    if os.environ.get("FUZZ_NOCLUSTER"): clustering = 0
    if clustering and nlive < 8 * D: nlive = 8 * D          # (clusters of fewer points than dimensions have singular covariances: whether
                                                               #  calc_cholesky falls back to the identity is then decided by pivots of 1e-19, in the reference too)
    B = int(rng.choice([0, 0, 1, 7, nlive // 2, nlive]))
    cf = float(rng.choice([np.exp(-1.0), 0.5, 0.1, 0.9]))
    nDer = 2 if kind == "gaussian" else (1 if kind == "twin_gaussian" else 0)
    kw = dict(nlive=nlive, num_repeats=nr, seed=int(rng.integers(1, 10000)), do_clustering=clustering, compression_factor=cf,
              max_ndead=int(rng.choice([-1, 3 * nlive, 10 * nlive])) if D <= 20 else 6 * nlive)
    s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, nDer)
    for k, v in kw.items(): setattr(s, k, v)
    s.batch = B
    s.ablate = int(os.environ.get("FUZZ_ABLATE", "0")); s.epoch_discard = int(os.environ.get("FUZZ_DISCARD", "0"))
    lo, hi = (-5.12, 5.12) if kind == "rastrigin" else ((-1.0, 1.0) if kind == "twin_gaussian" else (None, None))
    extra = {}
    if kind == "corr_gaussian":
        ic = np.zeros((D, D)); ld = C.c_double()
        olib.pc_random_invcov(4321 + case, D, C.c_double(0.1), orc.dptr(ic), C.byref(ld))
        extra = dict(invcov=ic, mean=np.full(D, 0.5), logdet=ld.value)
    if os.environ.get("FUZZ_ONLY") and int(os.environ["FUZZ_ONLY"]) != case: continue     # (the draws above keep the sequence)
    L, P, keep = api.make_problem(kind, D, nDer, lo, hi, **extra)
    t0 = time.time(); g = api.run(s, L, P); tg = time.time() - t0
    so = orc.settings(D, nDer, batch=g["batch"], epoch_discard=int(os.environ.get("FUZZ_DISCARD", "0")), **kw)
    Lo, Po, keep2 = orc.make_problem(kind, D, *(() if lo is None else (lo, hi)), **extra)
    t0 = time.time(); o = orc.run(so, Lo, Po); to = time.time() - t0
    ok = all(g[k] == o[k] for k in ("ndead", "nlike", "niter", "ncluster_dead"))
    if ok:
        ok = abs(g["logZ"] - o["logZ"]) < 1e-8 * max(1.0, abs(o["logZ"]))
        rel = np.abs(g["dead"] - o["dead"]) / np.maximum(1.0, np.abs(o["dead"]))
        ok = ok and rel.max() < 1e-7
    print(("ok  " if ok else "FAIL"), case, kind, "D", D, "nlive", nlive, "nr", nr, "B", g["batch"], "clu", clustering, "cf %.2f" % cf, "maxnd", kw["max_ndead"],
          "ndead", g["ndead"], o["ndead"], "nlike", g["nlike"], o["nlike"], "logZ %.6f %.6f" % (g["logZ"], o["logZ"]), "t %.2f/%.1f" % (tg, to), flush=True)
    bad += 0 if ok else 1
print("cases", ncases, "failures", bad)
