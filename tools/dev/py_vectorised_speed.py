"""Python likelihoods through pypolychord.run: scalar callable vs a callable marked `vectorised` (developer timing script)"""
import sys, time, tempfile
sys.path.insert(0, ".")
import numpy as np
from polychordlite_amd import pypolychord
D = 20
def like(theta):
    return float(-D * (np.log(0.1) + 0.5 * np.log(2 * np.pi)) - 0.5 * np.sum(((theta - 0.5) / 0.1) ** 2))
def vlike(theta):
    return -D * (np.log(0.1) + 0.5 * np.log(2 * np.pi)) - 0.5 * np.sum(((theta - 0.5) / 0.1) ** 2, axis=1)
vlike.vectorised = True
def vprior(cube):
    return cube
vprior.vectorised = True
kw = dict(nlive=500, num_repeats=40, seed=1, do_clustering=False, read_resume=False, write_resume=False, write_dead=False,
          write_live=False, write_stats=True, posteriors=False, equals=False, write_prior=False, feedback=0)
for name, f, p in (("vectorised", vlike, vprior), ("scalar", like, lambda c: c.copy())):
    with tempfile.TemporaryDirectory() as d:
        t0 = time.time()
        pypolychord.run(f, D, prior=p, base_dir=d, file_root="t", **kw)
        dt = time.time() - t0
        st = open(d + "/t.stats").read().splitlines()
        nlike = int([l for l in st if l.startswith(" nlike:")][0].split(":")[1].split()[0])
        print(f"{name}: {nlike} evaluations in {dt:.2f} s = {nlike / dt / 1e6:.3f} M evals/s, {st[8].strip()}", flush=True)
