"""dev: the 100-D configuration after a sweep of runs in step: does its side stream still run next to its main stream?"""
import ctypes as C, sys, time
sys.path.insert(0, ".")
import bench
from polychordlite_amd import _ctypes_api as api
from polychordlite_amd.repeats import run_repeats
lib = api.load()
s = api.Settings(); lib.pchip_settings_default(C.byref(s), 20, 2)
s.nlive, s.num_repeats = 2000, 40
L, P, keep = api.make_problem("gaussian", 20, 2)
for k in range(10):
    s.seed = k; api.run(s, L, P)
for R in (4, 8, 16, 32, 64):
    for k in range(4):
        m, held = run_repeats(s, L, P, [1000 * k + j for j in range(R)], max_in_flight=R); held = None
lib.polychord_hip_set_option(b"trim_cache", 0.0)
w = bench.WORKLOADS["c5"]
s2 = api.Settings(); lib.pchip_settings_default(C.byref(s2), w["D"], w["nDer"])
s2.nlive, s2.num_repeats = w["nlive"], w["nr"]
ic, mean, logdet = bench.random_correlated_gaussian(w["D"])
L2, P2, keep2 = api.make_problem("corr_gaussian", w["D"], w["nDer"], invcov=ic, mean=mean, logdet=logdet)
s2.profile = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for k in range(6):
    s2.seed = 2000 + k
    t0 = time.perf_counter(); r = api.run(s2, L2, P2); print("c5 run %d: %.0f ms" % (k, (time.perf_counter() - t0) * 1e3), flush=True); r = None
