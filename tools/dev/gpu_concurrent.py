"""Dev script (not pytest): R independent runs of the metric config driven concurrently from R host threads
(one engine + one HIP stream each) on ONE GPU.  usage: gpu_concurrent.py R [rounds]"""
import ctypes as C
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
D, nDer = 20, 2
lib = api.load()


def one(seed):
    s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, nDer)
    s.nlive = 2000; s.num_repeats = 40; s.seed = seed; s.batch = 1000
    L, P, keep = api.make_problem("gaussian", D, nDer)
    r = api.run(s, L, P)
    return r["nlike"], r["logZ"]


one(1)
with ThreadPoolExecutor(R) as ex:
    list(ex.map(one, range(100, 100 + R)))      # warm the block cache for R engines
    for it in range(rounds):
        t0 = time.perf_counter()
        res = list(ex.map(one, range(1000 * (it + 1), 1000 * (it + 1) + R)))
        dt = time.perf_counter() - t0
        print(f"R={R}: {dt*1e3:.1f} ms for {R} runs, {sum(n for n, _ in res)/dt/1e6:.1f}M evals/s, logZ {[round(z, 3) for _, z in res][:4]}", flush=True)
