"""dev: as gpu_repeats_trace.py, with what bench.py has around it switched on one by one.  usage: gpu_repeats_ctx.py R [torch] [solo] [noclus]"""
import ctypes as C, sys, time
sys.path.insert(0, ".")
flags = set(sys.argv[2:])
if "import" in flags:
    import torch
if "init" in flags:
    import torch
    torch.cuda.init()
if "torch" in flags:
    import torch
    torch.zeros(1, device="cuda"); torch.cuda.synchronize()
if "sleep" in flags:
    import time as _t; _t.sleep(5)
from polychordlite_amd import _ctypes_api as api
from polychordlite_amd.repeats import run_repeats
R = int(sys.argv[1])
lib = api.load()
s = api.Settings(); lib.pchip_settings_default(C.byref(s), 20, 2)
s.nlive, s.num_repeats = 2000, 40
print("do_clustering default", s.do_clustering)
if "noclus" in flags: s.do_clustering = 0
L, P, keep = api.make_problem("gaussian", 20, 2)
if "solo" in flags:
    for k in range(12):
        s.seed = 50 + k; api.run(s, L, P)
run_repeats(s, L, P, [100 + j for j in range(R)], max_in_flight=R)
for it in range(4):
    m, held = run_repeats(s, L, P, [1000 * (it + 1) + j for j in range(R)], max_in_flight=R)
    held = None
    print(f"{sorted(flags)} R={R}: runs {m['t_runs_s']*1e3:.1f} ms, {m['nlike']/m['t_runs_s']/1e9:.2f} G evals/s", flush=True)
