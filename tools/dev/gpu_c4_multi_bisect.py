"""dev: why sixteen C4 runs in step take 590 ms inside `bench.py --workload c4` and 370 ms elsewhere.  usage: gpu_c4_multi_bisect.py <variant>"""
import ctypes as C, sys, time
sys.path.insert(0, ".")
v = sys.argv[1]
if "torch" in v:
    import torch
from polychordlite_amd import _ctypes_api as api
from polychordlite_amd.repeats import run_repeats
lib = api.load()
kind, D, nDer, nlive, nr, box = ("twin_gaussian", 30, 1, 500, 40, (-1.0, 1.0))
s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, nDer)
s.nlive, s.num_repeats, s.do_clustering = nlive, nr, 1
if "dev0" in v: s.device = 0
L, P, keep = api.make_problem(kind, D, nDer, *box)
if "solo" in v:
    s.profile = 0x3e if "prof" in v else 0
    for i in range(3):
        s.seed = 1000 + i; r = api.run(s, L, P)
        if "merge" in v:
            from polychordlite_amd.merge import merge_runs
            merge_runs(r, None, D, nDer)
        r = None
    s.profile = 0
if "sync" in v: torch.cuda.synchronize()
for k in range(4):
    if "sync" in v: torch.cuda.synchronize()
    t0 = time.perf_counter()
    m, held = run_repeats(s, L, P, [310000 + 1000 * k + j for j in range(16)], max_in_flight=16); held = None
    print(v, k, "%.1f ms" % (m["t_runs_s"] * 1e3), flush=True)
