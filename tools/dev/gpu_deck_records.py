"""dev (checker): the deck records k_nhats leaves behind the bases (pc_deck_record) against stale memory -- one process, one seed, runs whose
num_repeats / nlive / nDims change from call to call so that the engine's cached blocks hold older records of the same keys: every run alone
(decks from the records) against the same run in step with a second one (lane-per-chain kernels: decks made in the kernel), bit for bit."""
import ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np
from polychordlite_amd import _ctypes_api as api
from polychordlite_amd.repeats import run_repeats
lib = api.load()
bad = 0
for D, nDer, nlive, nr in ((8, 0, 200, 16), (8, 0, 200, 8), (8, 0, 200, 16), (8, 0, 100, 16), (8, 0, 200, 32), (12, 1, 200, 16), (8, 0, 200, 16), (8, 0, 200, 9)):
    L, P, keep = api.make_problem("gaussian", D, nDer)
    def settings(seed):
        s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, nDer)
        s.nlive, s.num_repeats, s.seed = nlive, nr, seed
        return s
    one = api.run(settings(5), L, P)
    merged, runs = run_repeats(settings(0), L, P, [5, 6], max_in_flight=2)
    same = one["ndead"] == runs[0]["ndead"] and one["nlike"] == runs[0]["nlike"] and np.array_equal(one["dead"], runs[0]["dead"], equal_nan=True) and one["logZ"] == runs[0]["logZ"]
    print("D %d nlive %d nr %d: %s (ndead %d, nlike %d)" % (D, nlive, nr, "same" if same else "DIFFERENT", one["ndead"], one["nlike"]))
    bad += 0 if same else 1
print("failures", bad)
