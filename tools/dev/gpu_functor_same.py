"""dev: settings.ablate bit 0 (the built-in Gaussian evaluated as a device functor) against the closed form along the chord: the same run?"""
import ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np
from polychordlite_amd import _ctypes_api as api
lib = api.load()
for D, nDer, nlive, nr, box in ((20, 2, 2000, 40, None), (6, 1, 300, 12, (-0.25, 1.5)), (3, 0, 200, 9, None)):
    L, P, keep = api.make_problem("gaussian", D, nDer, *box) if box else api.make_problem("gaussian", D, nDer)
    out = []
    for ab in (0, 1):
        s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, nDer)
        s.nlive, s.num_repeats, s.ablate, s.seed = nlive, nr, ab, 5
        out.append(api.run(s, L, P))
    a, b = out
    n = min(a["ndead"], b["ndead"])
    rel = np.abs(a["dead"][:n] - b["dead"][:n]) / np.maximum(1.0, np.abs(a["dead"][:n]))
    bad = np.nonzero(rel.max(axis=1) > 1e-9)[0]
    print(D, "ndead", a["ndead"], b["ndead"], "nlike", a["nlike"], b["nlike"], "logZ", a["logZ"], b["logZ"], "first differing dead row", bad[:1], "max rel", rel.max())
