"""dev: does a call's first wait for the device depend on how long the device was idle before it?  (PC_DEBUG=5 prints the first
run's set-up + live points time of every call)  usage: PC_DEBUG=5 gpu_first_wait.py"""
import ctypes as C, sys, time
sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api
from polychordlite_amd.repeats import run_repeats
lib = api.load()
s = api.Settings(); lib.pchip_settings_default(C.byref(s), 20, 2)
s.nlive, s.num_repeats = 2000, 40
L, P, keep = api.make_problem("gaussian", 20, 2)
run_repeats(s, L, P, [100 + j for j in range(8)], max_in_flight=8)
for it, nap in enumerate([0, 0, 0, 0.5, 0.5, 0.5, 0, 0, 0, 2.0, 2.0]):
    time.sleep(nap)
    print(f"after a nap of {nap} s:", file=sys.stderr, flush=True)
    m, held = run_repeats(s, L, P, [1000 * (it + 1) + j for j in range(8)], max_in_flight=8)
    held = None
