# one-off sweep: a likelihood fused into the kernel and the same function called back on the host (scalar and vectorised
# Python callables) must give the same run -- same Philox streams, same decisions.  usage: fuzz_callback.py [n] [seed]
import sys, os, tempfile
sys.path.insert(0, ".")
import numpy as np
from polychordlite_amd import pypolychord
from polychordlite_amd.pypolychord import device_likelihoods as dl

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
tmp = tempfile.mkdtemp()
from polychordlite_amd import _ctypes_api as api
lib = api.load()
bad = 0
for k in range(n):
    kind = ["gaussian", "rastrigin", "twin"][int(rng.integers(0, 3))]
    D = int(rng.integers(2, 30))
    nDer = int(rng.integers(0, 3)) if kind == "gaussian" else (int(rng.integers(0, 2)) if kind == "twin" else 0)
    nlive = int(rng.integers(20, 90)) + D
    nr = int(rng.integers(1, 2 * D + 2))
    clustering = bool(rng.integers(0, 2)) and D <= 6
    # a run is reproducible for a (seed, chains per nursery) pair: pin the nursery size, which callback mode would
    # otherwise choose from the measured cost of a call
    lib.polychord_hip_set_option(b"batch", float([1, 8, 32, 64][int(rng.integers(0, 4))]))
    fun = {"gaussian": lambda: dl.Gaussian(0.5, 0.1, nDerived=nDer), "rastrigin": lambda: dl.Rastrigin(), "twin": lambda: dl.TwinGaussian(0.1, nDerived=nDer)}[kind]()
    box = {"gaussian": (0.0, 1.0), "rastrigin": (-5.12, 5.12), "twin": (-1.0, 1.0)}[kind]
    prior = dl.UniformPrior(*box)
    mk = lambda root: pypolychord.PolyChordSettings(D, nDer, nlive=nlive, num_repeats=nr, seed=100 + k, do_clustering=clustering, read_resume=False,
                                                    write_resume=False, base_dir=tmp, file_root=root, feedback=-1, max_ndead=6 * nlive)
    dev = pypolychord.run_polychord(fun, D, nDer, mk("d%d" % k), prior)
    def scalar(theta): return fun(theta)
    def host_prior(cube): return box[0] + (box[1] - box[0]) * np.asarray(cube)
    cb = pypolychord.run_polychord(scalar, D, nDer, mk("c%d" % k), host_prior)
    def vec(theta):
        out = [fun(t) for t in theta]
        if nDer: return np.array([o[0] for o in out]), np.array([o[1] for o in out])
        return np.array(out)
    vec.vectorised = True
    vb = pypolychord.run_polychord(vec, D, nDer, mk("v%d" % k), host_prior)
    same = all((dev.ndead, dev.nlike) == (x.ndead, x.nlike) and abs(dev.logZ - x.logZ) < 1e-9 * max(1.0, abs(dev.logZ)) for x in (cb, vb))
    if not same:
        bad += 1
        print("MISMATCH", k, kind, D, nDer, nlive, nr, clustering, (dev.ndead, dev.nlike, dev.logZ), (cb.ndead, cb.nlike, cb.logZ), (vb.ndead, vb.nlike, vb.logZ))
print("fuzz_callback: %d configurations, %d mismatches" % (n, bad))
