"""dev: does a sweep of many runs in step (dozens of pooled streams) slow the single runs after it?"""
import ctypes as C, sys, time
sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api
from polychordlite_amd.repeats import run_repeats
lib = api.load()
s = api.Settings(); lib.pchip_settings_default(C.byref(s), 20, 2)
s.nlive, s.num_repeats = 2000, 40
L, P, keep = api.make_problem("gaussian", 20, 2)
def solo(tag):
    ts = []
    for k in range(4):
        s.seed = 900 + k
        t0 = time.perf_counter(); api.run(s, L, P); ts.append((time.perf_counter() - t0) * 1e3)
    print(tag, ["%.2f" % t for t in ts], flush=True)
solo("before the sweep")
for R in (4, 8, 16, 32, 64):
    for k in range(4):
        m, held = run_repeats(s, L, P, [1000 * k + j for j in range(R)], max_in_flight=R); held = None
    solo("after R=%d" % R)
