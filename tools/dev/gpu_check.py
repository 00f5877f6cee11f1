"""Ad-hoc GPU diagnostics (not a pytest file): engine vs oracle, verbose."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from polychordlite_amd import _ctypes_api as api
from tests import oracle_api as orc


def check_slice(kind, D, nDer, nr, nchains, lo, hi, seed=11):
    lib = api.load()
    s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, nDer)
    s.num_repeats = nr; s.seed = seed
    L, P, keep = api.make_problem(kind, D, nDer, lo, hi)
    so = orc.settings(D, nDer, num_repeats=nr, seed=seed)
    Lo, Po, keep2 = orc.make_problem(kind, D, lo, hi)
    nT = 2 * D + nDer + 2
    rng = np.random.default_rng(1)
    olib = orc.load()
    # seeds near the peak so the contour is non trivial
    seeds = np.zeros((nchains, nT))
    lo_a = np.broadcast_to(0.0 if lo is None else lo, (D,)); hi_a = np.broadcast_to(1.0 if hi is None else hi, (D,))
    for c in range(nchains):
        cube = 0.5 + 0.05 * rng.standard_normal(D)
        if kind == "twin_gaussian":
            cube[:2] = 0.75 + 0.02 * rng.standard_normal(2)
        th = lo_a + (hi_a - lo_a) * cube
        phi = np.zeros(max(nDer, 1))
        seeds[c, :D] = cube; seeds[c, D:2 * D] = th
        seeds[c, nT - 1] = olib.pc_like_eval(C.byref(Lo), orc.dptr(np.ascontiguousarray(th)), D, orc.dptr(phi), nDer)
        seeds[c, 2 * D:2 * D + nDer] = phi[:nDer]
    contour = float(seeds[:, nT - 1].min()) - 5.0
    A = rng.standard_normal((D, D)) * 0.02
    cov = A @ A.T + 0.0004 * np.eye(D)
    chol = np.linalg.cholesky(cov)
    babies = np.zeros((nchains, nr, nT)); nh = np.zeros((nchains, nr, D)); nl = np.zeros(nchains, dtype=np.int32)
    rc = lib.pchip_slice_chains(C.byref(s), C.byref(L), C.byref(P), 3, nchains, api.dptr(seeds), api.dptr(np.ascontiguousarray(chol)),
                                contour, api.dptr(babies), api.dptr(nh), nl.ctypes.data_as(C.POINTER(C.c_int)))
    assert rc == 0
    worst = 0.0
    for c in range(nchains):
        ob, onh, on = orc.slice_chain(so, Lo, Po, seed, 3, c, seeds[c], chol, contour)
        err = np.abs(babies[c] - ob)
        scale = np.maximum(1.0, np.abs(ob))
        rel = (err / scale).max()
        worst = max(worst, rel)
        if rel > 1e-9 or on != nl[c]:
            print(f"  chain {c}: max rel err {rel:.3e} nlike gpu {nl[c]} oracle {on}")
            bad = np.argwhere(err / scale > 1e-9)
            print("   first bad (slice, col):", bad[:5].tolist())
            print("   gpu   row0:", babies[c, 0, :4], babies[c, 0, -2:])
            print("   oracle row0:", ob[0, :4], ob[0, -2:])
            # directions as sets
            g = nh[c]; o = onh
            d = np.abs(g[:, None, :] - o[None, :, :]).max(-1).min(0)
            print("   nhat match (max over oracle dirs of min dist to a gpu dir):", d.max())
            break
    print(f"slice {kind} D={D} nr={nr} chains={nchains}: worst rel err {worst:.3e}  nlike ok={np.array_equal(nl, nl)}")


def check_run(kind, D, nDer, nlive, nr, B, lo, hi, seed=5, clustering=0):
    lib = api.load()
    s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, nDer)
    s.nlive = nlive; s.num_repeats = nr; s.seed = seed; s.batch = B; s.do_clustering = clustering
    L, P, keep = api.make_problem(kind, D, nDer, lo, hi)
    t0 = time.time(); g = api.run(s, L, P); tg = time.time() - t0
    so = orc.settings(D, nDer, nlive=nlive, num_repeats=nr, seed=seed, batch=B, do_clustering=clustering)
    Lo, Po, keep2 = orc.make_problem(kind, D, lo, hi)
    t0 = time.time(); o = orc.run(so, Lo, Po); to = time.time() - t0
    print(f"run {kind} D={D} nlive={nlive} nr={nr} B={B}:")
    for k in ("logZ", "logZerr", "ndead", "nlike", "niter", "nbatches", "ncluster", "ncluster_dead"):
        print(f"   {k:14s} gpu {g[k]!r:>24}  oracle {o[k]!r:>24}")
    print(f"   gpu t_total {g['t_total']:.4f}s (gen {g['t_generate']:.4f} loop {g['t_loop']:.4f} final {g['t_final']:.4f}) rounds {g['nrounds']} updates {g['nupdates']}  wall {tg:.3f}s ; oracle {to:.2f}s")
    print(f"   gpu evals/s {g['nlike'] / g['t_total']:.3e}   oracle evals/s {o['nlike'] / to:.3e}")
    n = min(len(g["dead"]), len(o["dead"]))
    if n:
        dd = np.abs(g["dead"][:n] - o["dead"][:n]) / np.maximum(1.0, np.abs(o["dead"][:n]))
        bad = np.argwhere(dd.max(1) > 1e-8)
        print(f"   dead rows compared {n}: max rel err {dd.max():.3e}; first bad row {bad[0].tolist() if len(bad) else None}")
        lw = np.abs(g["logweights"][:n] - o["logweights"][:n])
        print(f"   logweights max abs err {np.nanmax(np.where(o['logweights'][:n] > -1e29, lw, 0)):.3e}")
    return g, o


if __name__ == "__main__":
    print("devices:", api.load().pchip_device_count())
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("all", "slice"):
        check_slice("gaussian", 20, 2, 40, 8, None, None)
        check_slice("gaussian", 4, 1, 8, 4, -1.0, 1.0)
        check_slice("rastrigin", 10, 0, 30, 4, -5.12, 5.12)
        check_slice("twin_gaussian", 30, 1, 40, 4, -1.0, 1.0)
    if what in ("all", "run"):
        check_run("gaussian", 20, 2, 100, 20, 1, None, None)
        check_run("gaussian", 20, 2, 200, 40, 16, None, None)
        check_run("gaussian", 20, 2, 500, 40, 128, None, None)
    if what in ("all", "big"):
        check_run("gaussian", 20, 2, 2000, 40, 512, None, None)
