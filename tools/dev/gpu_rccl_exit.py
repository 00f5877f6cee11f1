"""does a process that used the library's RCCL path exit cleanly?  variants by argv: torch (import torch first), noclose, nomerge"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
flags = set(sys.argv[1:])
if "torch" in flags:
    import torch
from polychordlite_amd import _ctypes_api as api
from polychordlite_amd import merge as mg
lib = api.load()
s = api.Settings(); lib.pchip_settings_default(C.byref(s), 6, 1)
s.nlive, s.num_repeats, s.batch, s.seed = 120, 12, 50, 31
L, P, keep = api.make_problem("gaussian", 6, 1)
run = api.run(s, L, P)
comm = mg.Comm(0, 1, 0)
print("library", comm.library, flush=True)
if "nomerge" not in flags:
    a = mg.comm_merge(run, comm, 6, 1)
    print("merged", a["logZ"], run["logZ"], flush=True)
if "noclose" not in flags:
    comm.close()
print("done", flags, flush=True)
