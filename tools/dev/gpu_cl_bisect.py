"""dev: first death at which the one-wave clustered contraction and the general kernel part ways, for one fuzz configuration
(bisection over max_ndead; both kernels are deterministic).  usage: gpu_cl_bisect.py"""
import ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np
from polychordlite_amd import _ctypes_api as api
lib = api.load()
kind, D, nDer, nlive, nr, B, cf, seed = "rastrigin", 10, 0, 200, 2, 100, 0.9, None
# reproduce the fuzz generator's seed for case 212 of `fuzz_parity.py 250 3`
rng = np.random.default_rng(3)
for case in range(213):
    k = rng.choice(["gaussian", "gaussian", "corr_gaussian", "rastrigin", "twin_gaussian"])
    Dd = int(rng.choice([2, 3, 5, 8, 13, 20, 24, 25, 31, 32, 33, 40, 64, 65, 70, 100])) if k != "twin_gaussian" else int(rng.choice([2, 4, 10, 30]))
    if k == "rastrigin": Dd = min(Dd, 10)
    nl = int(rng.choice([25, 50, 100, 200, 400])); nrr = int(rng.choice([1, 2, 5, Dd, 2 * Dd])) if Dd <= 20 else int(rng.choice([2, 5, 10]))
    cl = int(k in ("rastrigin", "twin_gaussian") and rng.random() < 0.7)
    if cl and nl < 8 * Dd: nl = 8 * Dd
    Bb = int(rng.choice([0, 0, 1, 7, nl // 2, nl])); cff = float(rng.choice([np.exp(-1.0), 0.5, 0.1, 0.9]))
    sd = int(rng.integers(1, 10000)); mnd = int(rng.choice([-1, 3 * nl, 10 * nl])) if Dd <= 20 else 6 * nl
    if case == 212: seed = sd; print("case", k, Dd, nl, nrr, Bb, cff, sd, mnd)
L, P, keep = api.make_problem(kind, D, nDer, -5.12, 5.12)
def run(ab, mnd):
    s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, nDer)
    s.nlive, s.num_repeats, s.seed, s.do_clustering, s.compression_factor, s.max_ndead, s.batch, s.ablate = nlive, nr, seed, 1, cf, mnd, B, ab
    g = api.run(s, L, P)
    return g["ndead"], g["nlike"], g["niter"], g["logZ"], g["ncluster_dead"], g["nupdates"]
lo, hi = 0, 9000
full = (run(0, -1), run(32, -1)); print("full", full)
while hi - lo > 1:
    mid = (lo + hi) // 2
    a, b = run(0, mid), run(32, mid)
    same = a[:3] == b[:3] and abs(a[3] - b[3]) < 1e-9
    print(mid, "same" if same else "DIFF", a, b, flush=True)
    if same: lo = mid
    else: hi = mid
print("first difference at max_ndead", hi)
