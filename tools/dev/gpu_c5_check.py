import ctypes as C, sys, time
sys.path.insert(0, ".")
import numpy as np
from polychordlite_amd import _ctypes_api as api
from tests import oracle_api as orc
lib = api.load(); olib = orc.load()
def run(D, nlive, nr, B, seed=3):
    ic = np.zeros((D, D)); ld = C.c_double()
    olib.pc_random_invcov(12345, D, 0.1, orc.dptr(ic), C.byref(ld))
    mean = np.full(D, 0.5)
    s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, 0)
    s.nlive, s.num_repeats, s.seed, s.batch, s.profile = nlive, nr, seed, B, 1
    L, P, keep = api.make_problem("corr_gaussian", D, 0, invcov=ic, mean=mean, logdet=ld.value)
    g = api.run(s, L, P)
    print(f"GPU corr_gaussian D={D} nlive={nlive} nr={nr} B={B}: logZ {g['logZ']:.4f} +- {g['logZerr']:.4f} ndead {g['ndead']} nlike {g['nlike']} t {g['t_total']:.3f}s -> {g['nlike']/g['t_total']:.3e} evals/s", {k: round(v['total_s']*1e3,1) for k,v in g['kernel_time'].items()}, flush=True)
    return g, (ic, mean, ld.value)
g, (ic, mean, ld) = run(100, 200, 40, 32)
so = orc.settings(100, 0, nlive=200, num_repeats=40, seed=3, batch=32)
Lo, Po, k2 = orc.make_problem("corr_gaussian", 100, invcov=ic, mean=mean, logdet=ld)
t0=time.time(); o = orc.run(so, Lo, Po); print("oracle", o['logZ'], o['ndead'], o['nlike'], f"{time.time()-t0:.1f}s")
print("match:", g['ndead']==o['ndead'], g['nlike']==o['nlike'], abs(g['logZ']-o['logZ']))
run(100, 1000, 200, 256)
