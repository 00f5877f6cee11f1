"""dev: R runs in flight through pchip_run_repeats (one host thread going round the engines); with rocprofv3 --kernel-trace around
it, tools/dev/trace_overlap.py tells how much of the kernels' time overlaps.  usage: gpu_repeats_trace.py R"""
import ctypes as C, sys, time
sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api
from polychordlite_amd.repeats import run_repeats
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4
lib = api.load()
s = api.Settings(); lib.pchip_settings_default(C.byref(s), 20, 2)
s.nlive, s.num_repeats = 2000, 40
L, P, keep = api.make_problem("gaussian", 20, 2)
run_repeats(s, L, P, [100 + j for j in range(R)], max_in_flight=R)
for it in range(3):
    t0 = time.perf_counter()
    m, held = run_repeats(s, L, P, [1000 * (it + 1) + j for j in range(R)], max_in_flight=R)
    held = None
    dt = time.perf_counter() - t0
    print(f"R={R}: runs {m['t_runs_s']*1e3:.1f} ms, call {dt*1e3:.1f} ms, {m['nlike']/m['t_runs_s']/1e9:.2f} G evals/s", flush=True)
