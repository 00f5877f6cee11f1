"""GPU (-m gpu): the HIP engine, called through its C ABI, against the oracle on identical seeded
inputs (the engine and the oracle draw the same Philox numbers, so trajectories coincide):
  * kernel level: K0 directions + K1 slice chains == oracle pc_slice_chain, rel. error <= 1e-9
  * run level: identical ndead / nlike / niter / nbatches, logZ to 1e-8, every dead row to 1e-7
  * golden level: logZ within 3 sigma of the analytic truth and of the reference's own numbers
  * properties at BASELINE size: evidence replay of the engine's own dead points == its logZ,
    monotone death sequence, reproducibility for a fixed seed.
Floating-point tolerance (north_star): fp64 everywhere; 1e-9 relative on kernel outputs, 1e-8 absolute
on logZ; integers exact."""
import ctypes as C
import os

import numpy as np
import pytest

from tests import oracle_api as orc

pytestmark = pytest.mark.gpu

BOX = {"gaussian": (None, None), "rastrigin": (-5.12, 5.12), "twin_gaussian": (-1.0, 1.0)}


def _settings(api, D, nDer, **kw):
    lib = api.load()
    s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, nDer)
    for k, v in kw.items():
        setattr(s, k, v)
    return s


@pytest.mark.parametrize("kind,D,nDer,nr,nchains", [("gaussian", 20, 2, 40, 6), ("gaussian", 4, 1, 8, 3),
                                                   ("gaussian", 2, 0, 5, 3), ("rastrigin", 10, 0, 30, 4),
                                                   ("twin_gaussian", 30, 1, 40, 3), ("gaussian", 20, 2, 100, 2),
                                                   ("gaussian", 33, 0, 40, 2)])
def test_slice_chains_match_oracle(engine, kind, D, nDer, nr, nchains):
    api = engine; lib = api.load(); olib = orc.load()
    lo, hi = BOX[kind]
    seed = 11
    s = _settings(api, D, nDer, num_repeats=nr, seed=seed)
    L, P, keep = api.make_problem(kind, D, nDer, lo, hi)
    so = orc.settings(D, nDer, num_repeats=nr, seed=seed)
    Lo, Po, keep2 = orc.make_problem(kind, D, lo, hi)
    nT = 2 * D + nDer + 2
    rng = np.random.default_rng(D * 1000 + nr)
    lo_a = np.broadcast_to(0.0 if lo is None else lo, (D,)); hi_a = np.broadcast_to(1.0 if hi is None else hi, (D,))
    seeds = np.zeros((nchains, nT))
    for c in range(nchains):
        cube = 0.5 + 0.04 * rng.standard_normal(D)
        if kind == "twin_gaussian":
            cube[:2] = 0.75 + 0.02 * rng.standard_normal(2)
        th = np.ascontiguousarray(lo_a + (hi_a - lo_a) * cube)
        phi = np.zeros(max(nDer, 1))
        seeds[c, :D] = cube; seeds[c, D:2 * D] = th
        seeds[c, nT - 1] = olib.pc_like_eval(C.byref(Lo), orc.dptr(th), D, orc.dptr(phi), nDer)
        seeds[c, 2 * D:2 * D + nDer] = phi[:nDer]
    contour = float(seeds[:, nT - 1].min()) - 4.0
    A = rng.standard_normal((D, D)) * 0.02
    chol = np.ascontiguousarray(np.linalg.cholesky(A @ A.T + 4e-4 * np.eye(D)))
    babies = np.zeros((nchains, nr, nT)); nh = np.zeros((nchains, nr, D)); nl = np.zeros(nchains, dtype=np.int32)
    rc = lib.pchip_slice_chains(C.byref(s), C.byref(L), C.byref(P), 5, nchains, api.dptr(seeds), api.dptr(chol), contour,
                                api.dptr(babies), api.dptr(nh), nl.ctypes.data_as(C.POINTER(C.c_int)))
    assert rc == 0
    for c in range(nchains):
        ob, onh, on = orc.slice_chain(so, Lo, Po, seed, 5, c, seeds[c], chol, contour)
        assert on == nl[c]
        rel = np.abs(babies[c] - ob) / np.maximum(1.0, np.abs(ob))
        assert rel.max() < 1e-9
        assert np.all(ob[:, -1] >= contour)            # every baby is inside the contour it was born under
        assert np.all(ob[:, -2] == contour)
        # the shuffled oracle directions are exactly the generated set (unit vectors, first one fixed)
        d = np.abs(nh[c][:, None, :] - onh[None, :, :]).max(-1).min(0)
        assert d.max() < 1e-10
        assert np.allclose(np.linalg.norm(nh[c], axis=1), 1.0, atol=1e-12)


RUNS = [  # kind D nDer nlive nr B general clustering
    ("gaussian", 20, 2, 100, 20, 1, 0, 0), ("gaussian", 20, 2, 200, 40, 16, 0, 0), ("gaussian", 20, 2, 200, 40, 16, 1, 0),
    ("gaussian", 4, 1, 100, 20, 32, 0, 0), ("gaussian", 20, 2, 500, 40, 128, 0, 0), ("rastrigin", 4, 0, 200, 12, 50, 0, 0),
    ("twin_gaussian", 6, 1, 150, 12, 40, 1, 0),
    # kNN clustering on the device: cluster splits, deaths, phantom re-homing, evidence splitting
    ("rastrigin", 2, 0, 300, 6, 1, 0, 1), ("rastrigin", 2, 0, 300, 6, 40, 0, 1), ("twin_gaussian", 6, 1, 150, 12, 30, 0, 1),
    ("rastrigin", 4, 0, 200, 12, 50, 0, 1), ("gaussian", 20, 2, 200, 40, 16, 0, 1),
    # other kernel variants: nDims in (32, 64], num_repeats > 64 (deck in LDS), no derived parameters, serial fast kernel (general=2)
    ("gaussian", 40, 0, 120, 40, 32, 0, 0), ("gaussian", 12, 2, 150, 70, 32, 0, 0), ("gaussian", 6, 1, 300, 12, 150, 2, 0),
    ("rastrigin", 3, 0, 120, 9, 60, 2, 0),
    # BASELINE configs[4] live-set size: the parallel contraction at nlive = 5000 with 1024 chains per nursery
    ("gaussian", 3, 0, 5000, 3, 1024, 0, 0),
    # num_repeats beyond 512 (5 nDims at nDims > 102): sixteen phantom mask words per chain
    ("gaussian", 6, 1, 60, 600, 16, 0, 0), ("gaussian", 5, 0, 50, 1000, 8, 1, 0),
    # live sets beyond the LDS-resident kernels (20 B of LDS per slot): per-slot arrays in HBM, serial contraction
    ("gaussian", 2, 0, 9000, 2, 512, 0, 0),
    # degenerate sizes: two live points, one repeat, more chains per nursery than live points (one live point has no
    # covariance: the reference's directions are 0/0 there, and so are the oracle's and the engine's)
    # (fewer live points than nDims + 1 is left out: the covariance is singular in exact arithmetic, so whether the
    #  Cholesky pivot comes out <= 0 -- the scaled-identity fallback of utils.F90:633-638 -- is decided by round-off)
    ("gaussian", 1, 0, 2, 1, 1, 0, 0), ("gaussian", 4, 0, 7, 5, 64, 0, 1),
]


def _run_against_oracle(api, kind, D, nDer, nlive, nr, B, general, clustering, **extra):
    lo, hi = BOX[kind]
    s = _settings(api, D, nDer, nlive=nlive, num_repeats=nr, seed=5, batch=B, force_general=general,
                  do_clustering=clustering, **extra)
    L, P, keep = api.make_problem(kind, D, nDer, lo, hi)
    g = api.run(s, L, P)
    so = orc.settings(D, nDer, nlive=nlive, num_repeats=nr, seed=5, batch=B, do_clustering=clustering, **extra)
    Lo, Po, keep2 = orc.make_problem(kind, D, lo, hi)
    o = orc.run(so, Lo, Po)
    _same_run(g, o, clustering)
    return g, o


@pytest.mark.parametrize("kind,D,nDer,nlive,nr,B,general,clustering", RUNS)
def test_full_run_matches_oracle(engine, kind, D, nDer, nlive, nr, B, general, clustering):
    _run_against_oracle(engine, kind, D, nDer, nlive, nr, B, general, clustering)


@pytest.mark.parametrize("general", [0, 1])
@pytest.mark.parametrize("kind,D,nDer,nlive,nr,B", [("rastrigin", 2, 0, 300, 6, 40), ("twin_gaussian", 6, 1, 150, 12, 30),
                                                     ("rastrigin", 4, 0, 200, 12, 50), ("rastrigin", 3, 0, 400, 9, 200)])
def test_chains_in_flight_when_the_cluster_list_changes(engine, kind, D, nDer, nlive, nr, B, general):
    """A cluster dies or is split while chains seeded before are still in the nursery.  settings.epoch_discard = 1 is
    nested_sampling.F90:313 + :331-341 as written (all of them are lost), 0 -- the default -- keeps the chains of the clusters
    the change left alone (oracle: remap_chains).  Both rules, by the one-wave and by the general contraction kernel, are the
    oracle's run (what the default saves is asserted at BASELINE configs[2]'s size, test_baseline_configs.py)."""
    gd, od = _run_against_oracle(engine, kind, D, nDer, nlive, nr, B, general, 1, epoch_discard=1)
    gk, ok = _run_against_oracle(engine, kind, D, nDer, nlive, nr, B, general, 1, epoch_discard=0)
    assert od["ncluster"] + od["ncluster_dead"] > 1            # (the cluster list did change)
    if kind == "rastrigin":
        assert gk["nlike"] != gd["nlike"] or gk["ndead"] != gd["ndead"]      # (and chains were in flight when it did)


def _same_run(g, o, clustering):
    for k in ("ndead", "nlike", "niter", "nbatches", "ncluster", "ncluster_dead"):
        assert g[k] == o[k], (k, g[k], o[k])
    assert abs(g["logZ"] - o["logZ"]) < 1e-8
    if clustering:   # per-cluster evidences of the dead clusters, in order of death
        assert np.allclose(g["logZp"], o["logZp"], atol=1e-8)
    assert abs(g["logZerr"] - o["logZerr"]) < 1e-8
    rel = np.abs(g["dead"] - o["dead"]) / np.maximum(1.0, np.abs(o["dead"]))
    assert rel.max() < 1e-7
    ok = o["logweights"] > -1e29
    assert np.array_equal(ok, g["logweights"] > -1e29)
    assert np.abs(g["logweights"][ok] - o["logweights"][ok]).max() < 1e-9
    nD = o["post_mean"].size                     # the engine also reports the derived parameters' moments
    assert np.allclose(g["post_mean"][:nD], o["post_mean"], atol=1e-8)


def test_analytic_evidence_and_reference_numbers(engine, golden):
    """20-D Gaussian, nlive 500 (ini/gaussian.ini): truth logZ = 0; the reference's 8 seeds give
    mean 0.027 +- 0.186/sqrt(8); posterior mean 0.5, sd 0.1 per dimension."""
    api = engine
    ref = [c for c in golden["ref_native"] if c["like"] == "gaussian" and c["nlive"] == 500]
    L, P, keep = api.make_problem("gaussian", 20, 2)
    zs, nds = [], []
    for seed in range(6):
        s = _settings(api, 20, 2, nlive=500, num_repeats=40, seed=200 + seed, batch=128)
        g = api.run(s, L, P)
        assert abs(g["logZ"]) < 3 * g["logZerr"]
        assert abs(g["logZerr"] - ref[0]["logZerr"]) < 0.02
        assert np.all(np.abs(g["post_mean"][:20] - 0.5) < 0.02) and np.all(np.abs(np.sqrt(g["post_var"][:20]) - 0.1) < 0.02)
        zs.append(g["logZ"]); nds.append(g["ndead"])
    sig = ref[0]["logZerr"]
    assert abs(np.mean(zs) - np.mean([c["logZ"] for c in ref])) < 3 * sig * np.sqrt(1 / 6 + 1 / 8)
    # synchronous batches waste ~B/(2 nlive) of the spawns, so ndead is a little larger than linear mode
    assert 0.98 * np.mean([c["ndead"] for c in ref]) < np.mean(nds) < 1.25 * np.mean([c["ndead"] for c in ref])


def test_baseline_size_properties(engine):
    """BASELINE configs[1] (20-D Gaussian nlive=2000): size-independent properties"""
    api = engine
    from tests.replay_oracle import evidence_replay, lived_records
    s = _settings(api, 20, 2, nlive=2000, num_repeats=40, seed=77, batch=0)
    L, P, keep = api.make_problem("gaussian", 20, 2)
    g = api.run(s, L, P)
    assert abs(g["logZ"]) < 3 * g["logZerr"] and 0.08 < g["logZerr"] < 0.11
    lived = g["logweights"] > -1e29
    d = g["dead"][lived]
    assert np.all(np.diff(d[:, -1]) >= 0)                       # deaths are sorted by logL
    assert np.all(d[:, -1] > d[:, -2])                          # each point lies above its birth contour
    lz, var = evidence_replay(*lived_records(g))                # replay of its own records == its evidence
    assert abs(lz - g["logZ"]) < 1e-6 and abs(var - g["varlogZ"]) < 1e-6
    assert np.all((d[:, :20] >= 0) & (d[:, :20] <= 1))          # cube coordinates
    assert np.allclose(d[:, 20:40], d[:, :20], atol=1e-15)      # uniform prior on [0,1] is the identity
    g2 = api.run(s, L, P)                                        # same seed -> bit-identical run
    assert g2["ndead"] == g["ndead"] and g2["nlike"] == g["nlike"] and g2["logZ"] == g["logZ"]
    assert np.array_equal(g2["dead"], g["dead"])
    s.seed = 78
    g3 = api.run(s, L, P)
    assert g3["logZ"] != g["logZ"]


def test_edge_cases(engine):
    api = engine
    L, P, keep = api.make_problem("gaussian", 3, 0)
    # max_ndead stops the run early (nested_sampling.F90:528)
    s = _settings(api, 3, 0, nlive=50, num_repeats=6, seed=1, batch=8, max_ndead=120)
    g = api.run(s, L, P)
    so = orc.settings(3, 0, nlive=50, num_repeats=6, seed=1, batch=8, max_ndead=120)
    Lo, Po, k2 = orc.make_problem("gaussian", 3)
    o = orc.run(so, Lo, Po)
    assert g["ndead"] == o["ndead"] and g["nlike"] == o["nlike"] and abs(g["logZ"] - o["logZ"]) < 1e-8
    # nprior > nlive: the initial set is trimmed (nested_sampling.F90:201-205)
    s = _settings(api, 3, 0, nlive=40, nprior=90, num_repeats=6, seed=2, batch=4)
    g = api.run(s, L, P)
    so = orc.settings(3, 0, nlive=40, nprior=90, num_repeats=6, seed=2, batch=4)
    o = orc.run(so, Lo, Po)
    assert g["ndead"] == o["ndead"] and g["nlike"] == o["nlike"] and abs(g["logZ"] - o["logZ"]) < 1e-8
    # batch larger than nlive and a 1-D problem
    L1, P1, k1 = api.make_problem("gaussian", 1, 0)
    s = _settings(api, 1, 0, nlive=30, num_repeats=3, seed=3, batch=64)
    g = api.run(s, L1, P1)
    so = orc.settings(1, 0, nlive=30, num_repeats=3, seed=3, batch=64)
    Lo1, Po1, k3 = orc.make_problem("gaussian", 1)
    o = orc.run(so, Lo1, Po1)
    assert g["ndead"] == o["ndead"] and g["nlike"] == o["nlike"] and abs(g["logZ"] - o["logZ"]) < 1e-8


@pytest.mark.parametrize("B", [1, 24])
def test_dynamic_nlive_and_nprior_match_oracle(engine, B):
    """nlives/loglikes (run_time_info.f90:766-779: the number of live points follows the contour) and
    nprior > nlive (nested_sampling.F90:201-205): same trajectory as the oracle."""
    api = engine
    D, nDer = 4, 1
    ll = np.array([-20.0, 0.0]); nl = np.array([60, 150], dtype=np.int32)
    kw = dict(nlive=100, num_repeats=8, seed=9, batch=B, nprior=160, n_nlives=2)
    s = _settings(api, D, nDer, **kw)
    s.loglikes = ll.ctypes.data_as(C.POINTER(C.c_double)); s.nlives = nl.ctypes.data_as(C.POINTER(C.c_int))
    L, P, keep = api.make_problem("gaussian", D, nDer)
    g = api.run(s, L, P)
    so = orc.settings(D, nDer, **kw)
    so.loglikes = ll.ctypes.data_as(C.POINTER(C.c_double)); so.nlives = nl.ctypes.data_as(C.POINTER(C.c_int))
    Lo, Po, keep2 = orc.make_problem("gaussian", D)
    o = orc.run(so, Lo, Po)
    for k in ("ndead", "nlike", "niter"):
        assert g[k] == o[k], (k, g[k], o[k])
    assert abs(g["logZ"] - o["logZ"]) < 1e-8
    assert np.abs(g["dead"] - o["dead"]).max() < 1e-7
    assert abs(g["logZ"]) < 4 * g["logZerr"]            # truth 0


@pytest.mark.parametrize("D,nlive,nr,B", [(100, 60, 10, 16), (40, 80, 12, 24), (128, 50, 8, 16), (113, 40, 6, 8),
                                          # the other tile counts of the panel / matrix-core kernels (k_basis, k_whiten, k_upd_gather)
                                          (50, 60, 10, 16), (70, 40, 6, 8), (90, 40, 6, 8),
                                          # beyond 128 dimensions: bases in HBM (k_nhats_big), Cholesky factor built in HBM
                                          (150, 170, 6, 16), (200, 30, 210, 4), (256, 280, 4, 32), (131, 150, 140, 8),
                                          # negative nDims: the spherical Gaussian of gaussian.f90 in that many dimensions
                                          (-120, 140, 8, 8), (-140, 160, 8, 8)])
def test_correlated_gaussian_high_dim_first_generations_match_oracle(engine, D, nlive, nr, B):
    """random_gaussian.f90 in 100 (and 40) dimensions: the wide-nDims kernel variants, and live sets whose logL
    spans ~1e5 nats inside one nursery (the live log-sum-exp must not lose the old points to underflow).
    NOT a full run: the first six generations of live points (max_ndead = 6 nlive) -- in wide boxes round-off is amplified at
    every covariance update until a comparison flips; full wide runs are pinned draw for draw in sequential-stream mode and
    statistically against the reference binary (tests/test_baseline_configs.py)."""
    api = engine; olib = orc.load()
    spherical = D < 0
    D = abs(D)
    ic = np.zeros((D, D)); ld = C.c_double()
    olib.pc_random_invcov(12345, D, C.c_double(0.1), orc.dptr(ic), C.byref(ld))
    mean = np.full(D, 0.5)
    # (a few live-set generations: in wide boxes round-off is amplified at every covariance update until a comparison
    #  flips -- see test_graded_runs_match_oracle)
    kw = dict(nlive=nlive, num_repeats=nr, seed=3, batch=B, max_ndead=6 * nlive)
    s = _settings(api, D, 0, **kw)
    if spherical: L, P, keep = api.make_problem("gaussian", D, 0)
    else: L, P, keep = api.make_problem("corr_gaussian", D, 0, invcov=ic, mean=mean, logdet=ld.value)
    g = api.run(s, L, P)
    so = orc.settings(D, 0, **kw)
    if spherical: Lo, Po, k2 = orc.make_problem("gaussian", D)
    else: Lo, Po, k2 = orc.make_problem("corr_gaussian", D, invcov=ic, mean=mean, logdet=ld.value)
    o = orc.run(so, Lo, Po)
    for k in ("ndead", "nlike", "niter"):
        assert g[k] == o[k], (k, g[k], o[k])
    assert g["ndead"] > 5 * nlive
    assert abs(g["logZ"] - o["logZ"]) < 1e-6 * max(1.0, abs(o["logZ"]))
    rel = np.abs(g["dead"] - o["dead"]) / np.maximum(1.0, np.abs(o["dead"]))
    assert rel.max() < 1e-7


@pytest.mark.parametrize("kw", [dict(max_ndead=777), dict(nfail=3, precision_criterion=1e-9, max_ndead=5000), dict(precision_criterion=0.2)])
def test_termination_triggers_match_oracle(engine, kw):
    """max_ndead, nfail and the precision criterion (nested_sampling.F90:514-543, :315-319) stop the run at the same
    step as the oracle, inside a nursery."""
    api = engine
    base = dict(nlive=150, num_repeats=10, seed=21, batch=64)
    base.update(kw)
    s = _settings(api, 5, 1, **base)
    L, P, keep = api.make_problem("gaussian", 5, 1)
    g = api.run(s, L, P)
    so = orc.settings(5, 1, **base)
    Lo, Po, keep2 = orc.make_problem("gaussian", 5)
    o = orc.run(so, Lo, Po)
    for k in ("ndead", "nlike", "niter"):
        assert g[k] == o[k], (k, g[k], o[k], kw)
    assert abs(g["logZ"] - o["logZ"]) < 1e-8


def test_engine_reproduces_the_reference_binary(engine, golden):
    """The closing link: with ONE Philox stream consumed in the reference's program order (sequential_rng: batch 1,
    reference list rule) the HIP engine walks the same trajectory as the REFERENCE BINARY whose `random_number` was fed
    that stream (tests/golden/ref_injected.json, made by oracle/gen_golden.py with oracle/ref_rng_shim.c): identical
    ndead and nlike, logZ to round-off -- Gaussian, Rastrigin and twin Gaussian, with and without clustering,
    dynamic nlive, nprior > nlive."""
    api = engine
    done = 0
    for c in golden["ref_injected"]:
        lo, hi = BOX[c["like"]]
        kw = dict(nlive=c["nlive"], num_repeats=c["num_repeats"], seed=c["seed"], do_clustering=c["clustering"],
                  nprior=c.get("nprior", -1), sequential_rng=1)
        s = _settings(api, c["nDims"], c["nDerived"], **kw)
        keep_arrays = None
        if "nlives" in c:
            pairs = [p.split(":") for p in c["nlives"].split(",")]
            ll = np.array([float(a) for a, _ in pairs]); nl = np.array([int(b) for _, b in pairs], dtype=np.int32)
            s.n_nlives = len(pairs)
            s.loglikes = ll.ctypes.data_as(C.POINTER(C.c_double)); s.nlives = nl.ctypes.data_as(C.POINTER(C.c_int))
            keep_arrays = (ll, nl)
        graded = "grade_dims" in c          # fast/slow grades with explicit repeats (chordal_sampling.f90:94-145)
        keep_grades = api.set_grades(s, c["grade_dims"], c["grade_repeats"]) if graded else None
        L, P, keep = api.make_problem(c["like"], c["nDims"], c["nDerived"], lo, hi)
        g = api.run(s, L, P)
        assert g["ndead"] == c["ndead"], (c, g["ndead"])
        if graded:                          # .stats lists RTI%nlike per grade
            assert g["nlike_grade"][:len(c["nlike_grades"])] == c["nlike_grades"], (c, g["nlike_grade"])
            assert g["nlike"] == sum(c["nlike_grades"])
        else:
            assert g["nlike"] == c["nlike"], (c, g["nlike"])
        assert abs(g["logZ"] - c["logZ"]) < 1e-8 and abs(g["logZerr"] - c["logZerr"]) < 1e-8, (c, g["logZ"], g["logZerr"])
        done += 1
    assert done == len(golden["ref_injected"])


GRADED = [  # kind D nDer nlive B clustering grade_dims grade_repeats
    ("gaussian", 4, 1, 100, 16, 0, [2, 2], [8, 4]), ("gaussian", 20, 2, 200, 64, 0, [8, 12], [20, 24]),
    ("gaussian", 6, 0, 80, 1, 0, [2, 2, 2], [6, 4, 5]), ("rastrigin", 4, 0, 200, 50, 1, [1, 3], [4, 9]),
    ("gaussian", 33, 0, 60, 16, 0, [30, 3], [33, 8]), ("gaussian", 40, 0, 80, 24, 0, [10, 30], [12, 70]),
    ("twin_gaussian", 6, 1, 150, 30, 1, [3, 3], [6, 7]),
    ("gaussian", 140, 0, 160, 8, 0, [100, 40], [6, 45]),
    ("gaussian", 120, 0, 140, 8, 0, [90, 30], [6, 35]),        # bases in HBM (nDims > 128), the fast grade's truncated
]


@pytest.mark.parametrize("kind,D,nDer,nlive,B,clustering,dims,reps", GRADED)
def test_graded_runs_match_oracle(engine, kind, D, nDer, nlive, B, clustering, dims, reps):
    """fast/slow parameter grades, production mode (B chains per nursery, keyed RNG): every grade's directions span
    its own and the faster parameters only; same trajectory and same per-grade likelihood counts as the oracle
    (which the reference binary pins in sequential mode, tests/golden/ref_injected.json)."""
    api = engine
    lo, hi = BOX[kind]
    kw = dict(nlive=nlive, num_repeats=sum(reps), seed=13, batch=B, do_clustering=clustering)
    if D > 100:
        # Wide boxes with few more live points than dimensions: round-off (1e-16 in a direction) is amplified at every
        # covariance update -- 1e-10 in the live points after 2000 deaths, 1e-7 after 7000 at nDims = 120, then a
        # comparison flips.  The oracle and the engine agree for as long as that takes; the test stops well before.
        kw["max_ndead"] = 6 * nlive
    s = _settings(api, D, nDer, **kw)
    keep_g = api.set_grades(s, dims, reps)
    L, P, keep = api.make_problem(kind, D, nDer, lo, hi)
    g = api.run(s, L, P)
    so = orc.settings(D, nDer, **kw)
    keep_o = orc.set_grades(so, dims, reps)
    Lo, Po, keep2 = orc.make_problem(kind, D, lo, hi)
    o = orc.run(so, Lo, Po)
    for k in ("ndead", "nlike", "niter", "nbatches", "ncluster", "ncluster_dead"):
        assert g[k] == o[k], (k, g[k], o[k])
    assert g["nlike_grade"] == o["nlike_grade"]
    tol = 1e-8 if D <= 100 else 1e-6
    assert abs(g["logZ"] - o["logZ"]) < tol and abs(g["logZerr"] - o["logZerr"]) < tol
    rel = np.abs(g["dead"] - o["dead"]) / np.maximum(1.0, np.abs(o["dead"]))
    assert rel.max() < 1e-7
    if kind == "gaussian" and D <= 100:
        assert abs(g["logZ"]) < 4 * g["logZerr"]                # truth 0


def _random_cases(n, seed=2024, dhi=13, seq=False, big=False):
    """small random configurations: every front-door knob of the path at once"""
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        kind = ["gaussian", "rastrigin", "twin_gaussian"][int(rng.integers(0, 3))]
        D = int(rng.integers(2 if kind == "twin_gaussian" else 1, 13)) if dhi <= 13 else int(rng.integers(13, dhi))
        nDer = 0 if kind == "rastrigin" else int(rng.integers(0, 3 if kind == "gaussian" else 2))
        nlive = int(rng.integers(25, 220)) + (D if dhi > 13 else 0)
        nr = int(rng.integers(1, 4 * D + 3)) if dhi <= 13 else int(rng.integers(1, 2 * D + 3))
        B = int([1, 2, 7, 16, 33, 64, 128][int(rng.integers(0, 7))])
        clustering = int(rng.integers(0, 2)) if D <= 6 else 0
        general = int(rng.integers(0, 3)) if not clustering else 0
        grades = None
        if D >= 2 and rng.random() < 0.35:
            cut = int(rng.integers(1, D))
            grades = ([cut, D - cut], [int(rng.integers(2, 2 * D + 2)), int(rng.integers(2, 2 * D + 2))])
        extra = {}
        r = rng.random()
        if r < 0.15:
            extra["max_ndead"] = int(rng.integers(nlive, 6 * nlive))
        elif r < 0.3:
            extra["precision_criterion"] = float(10 ** rng.uniform(-4, -0.5))
        elif r < 0.4:
            extra["nprior"] = nlive + int(rng.integers(1, nlive))
        if big:   # large live sets, hundreds of chains per nursery: the order-statistic contraction at full width
            D = min(D, 6); nDer = min(nDer, 1)
            nlive = int(rng.integers(500, 6000)); nr = int(rng.integers(1, 2 * D + 3)); B = int([256, 512, 1000, 1024][int(rng.integers(0, 4))])
            grades = None if D < 2 else grades
            if grades is not None: grades = ([1, D - 1], [int(rng.integers(2, D + 2)), int(rng.integers(2, D + 2))])
            extra = {k2: v for k2, v in extra.items() if k2 == "precision_criterion"}
        if seq:   # the reference's own draw order and list rule, one chain at a time (the mode the reference binary pins)
            B, general = 1, 0
            extra["sequential_rng"] = 1
        if dhi > 13:
            # Round-off (1e-16: the engine sums in another order than the oracle) grows with every generation of live
            # points -- 1e-11 after 600 deaths, 1e-7 after 2600 for a 35-D Rastrigin at nlive 63 -- until a comparison
            # flips; wide configurations are compared over the first generations only.
            extra["max_ndead"] = min(extra.get("max_ndead", 1 << 30), 8 * nlive)
        out.append((k, kind, D, nDer, nlive, nr, B, general, clustering, grades, extra))
    return out


# PC_FUZZ="seed:n" adds n more configurations from another seed (a one-off wider sweep on the GPU box)
_FUZZ = [int(x) for x in os.environ.get("PC_FUZZ", "0:0").split(":")]
_FUZZ_WIDE = [int(x) for x in os.environ.get("PC_FUZZ_WIDE", "0:0").split(":")]     # the same with nDims 13 ... 47
_FUZZ_SEQ = [int(x) for x in os.environ.get("PC_FUZZ_SEQ", "0:0").split(":")]       # the same in sequential-stream mode
_FUZZ_BIG = [int(x) for x in os.environ.get("PC_FUZZ_BIG", "0:0").split(":")]       # nlive 500 ... 6000, 256 ... 1024 chains per nursery


@pytest.mark.parametrize("case", _random_cases(24) + _random_cases(24, seed=77) + _random_cases(16, seed=5, dhi=48) + _random_cases(_FUZZ[1], seed=_FUZZ[0]) + _random_cases(_FUZZ_WIDE[1], seed=_FUZZ_WIDE[0], dhi=48) + _random_cases(8, seed=11, seq=True)
                         + _random_cases(_FUZZ_SEQ[1], seed=_FUZZ_SEQ[0], seq=True) + _random_cases(_FUZZ_BIG[1], seed=_FUZZ_BIG[0], big=True), ids=lambda c: f"rnd{c[0]}-{c[1]}-D{c[2]}-N{c[4]}-nr{c[5]}-B{c[6]}-g{c[7]}c{c[8]}" + ("-first8gen" if c[10].get("max_ndead") == 8 * c[4] else ""))
def test_random_configurations_match_oracle(engine, case):
    """(ids ending in -first8gen: the FIRST EIGHT GENERATIONS of live points next to the oracle, not a whole run -- the wide-nDims
    cases, see _random_cases.)
    48 + 16 (nDims 13 ... 47) + 8 (sequential-stream mode) seeded random configurations (likelihood, nDims, derived parameters, nlive, num_repeats, chains per nursery,
    contraction kernel, clustering, parameter grades, termination knobs, nprior): same trajectory as the oracle.
    A wider one-off sweep (PC_FUZZ=31337:300 and 4242:400 on the GPU box): 698 of 700 further configurations identical;
    the two that part ways are 6-D Rastrigin runs with clustering whose clusters hold fewer points than dimensions -- a
    Cholesky pivot of -1.7e-21 in the oracle (PC_ORACLE_TRACE_CHOL=1), a tiny positive one in the engine: the
    scaled-identity fallback of utils.F90:633-638 is decided by round-off there, in the reference as well.
    PC_FUZZ_WIDE=99:120 and 7:200: 320 of 320 configurations with nDims 13 ... 47 identical over their first eight
    generations of live points (see _random_cases for why not longer).  PC_FUZZ_SEQ=321:150: 150 of 150 in
    sequential-stream mode (the reference's draw order and list rule -- the mode in which the reference binary itself
    pins both sides).  PC_FUZZ_BIG=5:80 (nlive 500 ... 6000, 256 ... 1024 chains per nursery): 77 of 80; the three that
    part ways are clustered Rastrigin runs late in the run (identical for the first 52 115 of 61 481 deaths in the one
    looked at), where the live points sit 1e-5 apart and the similarity matrix r_a + r_b - 2 x_a.x_b of calculate.f90
    cancels to the last digits: a 1e-12 difference in a coordinate re-orders the neighbour lists."""
    k, kind, D, nDer, nlive, nr, B, general, clustering, grades, extra = case
    api = engine
    lo, hi = BOX[kind]
    kw = dict(nlive=nlive, num_repeats=nr if grades is None else sum(grades[1]), seed=300 + k, batch=B, do_clustering=clustering)
    kw.update(extra)
    s = _settings(api, D, nDer, force_general=general, **kw)
    kwo = dict(kw)
    if kw.get("sequential_rng"):   # one gaussian deviate goes to time_speeds unless the repeats per grade are given (generate.F90:285-287)
        kwo["time_speeds_draw"] = 1 if grades is None else 0
    so = orc.settings(D, nDer, **kwo)
    keep = (api.set_grades(s, *grades), orc.set_grades(so, *grades)) if grades else None
    L, P, k1 = api.make_problem(kind, D, nDer, lo, hi)
    Lo, Po, k2 = orc.make_problem(kind, D, lo, hi)
    g = api.run(s, L, P)
    o = orc.run(so, Lo, Po)
    for key in ("ndead", "nlike", "niter", "nbatches", "ncluster", "ncluster_dead"):
        assert g[key] == o[key], (key, g[key], o[key], case)
    assert g["nlike_grade"] == o["nlike_grade"]
    assert abs(g["logZ"] - o["logZ"]) < 1e-8 * max(1.0, abs(o["logZ"])) and abs(g["logZerr"] - o["logZerr"]) < 1e-8
    # (a cluster that is down to one live point has no covariance: its chains' directions are 0/0 in the reference, in
    #  the oracle and in the engine alike, and the babies of such a chain are NaN rows in all three)
    assert np.array_equal(np.isnan(g["dead"]), np.isnan(o["dead"]))
    rel = np.nan_to_num(np.abs(g["dead"] - o["dead"]) / np.maximum(1.0, np.abs(o["dead"])), nan=0.0)
    assert rel.max() < 1e-7


def test_maximiser_reproduces_the_reference(engine, golden, tmp_path):
    """maximise = T: in sequential-RNG mode the engine ends on the reference binary's live set, and the Nelder-Mead
    polish of its best points (pchip_maximise: maximiser.F90 / nelder_mead.f90 restated on the host) must land where the
    reference's <root>.maximum says (tests/golden/ref_maximum.json, written by the reference binary)"""
    api = engine; lib = api.load()
    f = lib.pchip_maximise
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_int,
                  C.POINTER(C.c_double), C.c_char_p]
    for c in golden["ref_maximum"]:
        D, nDer = c["nDims"], c["nDerived"]
        lo, hi = BOX[c["like"]]
        s = _settings(api, D, nDer, nlive=c["nlive"], num_repeats=c["num_repeats"], seed=c["seed"], do_clustering=c["clustering"], sequential_rng=1)
        L, P, keep = api.make_problem(c["like"], D, nDer, lo, hi)
        lib_r = api.Result()
        assert lib.pchip_run(C.byref(s), C.byref(L), C.byref(P), C.byref(lib_r)) == 0
        assert lib_r.ndead == c["ndead"]
        lo_a = np.full(D, 0.0 if lo is None else lo); hi_a = np.full(D, 1.0 if hi is None else hi)
        lib.polychord_hip_set_gaussian(0.5, 0.1)
        lib.polychord_hip_set_uniform_prior(D, api.dptr(lo_a), api.dptr(hi_a))
        like = C.cast(getattr(lib, "polychord_hip_" + c["like"]), C.c_void_p); prior = C.cast(lib.polychord_hip_uniform_prior, C.c_void_p)
        path = tmp_path / (c["like"] + ".maximum")
        assert f(like, prior, D, nDer, -1e30, lib_r.live, lib_r.live_cluster, lib_r.nlive_final, None, str(path).encode()) == 0
        lib.pchip_result_free(C.byref(lib_r))
        lines = path.read_text().splitlines()
        num = lambda k: np.array([float(x) for x in lines[k].split()])
        assert abs(num(1)[0] - c["max_loglike"]) < 1e-9 and np.allclose(num(3), c["max_point"], rtol=0, atol=1e-9)
        assert abs(num(6)[0] - c["max_posterior"]) < 1e-6 and abs(num(8)[0] - c["loglike_at_posterior"]) < 1e-9
        assert np.allclose(num(10), c["posterior_point"], rtol=0, atol=1e-9)


def test_twin_gaussian_evidence_statistics_match_the_reference(engine, golden):
    """BASELINE configs[3] in production mode (keyed streams, nlive/2 chains per nursery) against 32 runs of the
    reference binary with its own RNG (tests/golden/ref_c4_seeds.json): same mean evidence -- including the reference's
    own bias of -0.14 +- 0.06 against the analytic -30 ln 2 at num_repeats = 40 -- and the same run-to-run scatter."""
    api = engine
    ref = golden["ref_c4_seeds"]
    zr = np.array([r["logZ"] for r in ref["runs"]])
    c = ref["config"]
    s = _settings(api, c["nDims"], c["nDerived"], nlive=c["nlive"], num_repeats=c["num_repeats"], do_clustering=c["clustering"])
    s.batch = 0                                         # the engine's default nursery
    lo, hi = BOX["twin_gaussian"]
    L, P, keep = api.make_problem("twin_gaussian", c["nDims"], c["nDerived"], lo, hi)
    z, nd, per, per_all = [], [], [], []
    for i in range(32):
        s.seed = 900 + i
        g = api.run(s, L, P)
        z.append(g["logZ"]); nd.append(int((g["logweights"] > -1e29).sum()))   # dead points proper, without the failed spawns of a nursery
        per.append((g["nlike"] - g["nlike_failed"]) / nd[-1]); per_all.append(g["nlike"] / nd[-1])
    z = np.array(z)
    # likelihood evaluations per dead point: the chains that placed a point cost what the reference's cost (177.3); with the babies born
    # below the contour by the time a nursery of nlive/2 gets to them, 1.23 times that
    ref_per = np.mean([r["nlike"] / r["ndead"] for r in ref["runs"]])
    assert abs(np.mean(per) / ref_per - 1.0) < 0.05, (np.mean(per), ref_per)
    assert np.mean(per_all) / ref_per < 1.3, (np.mean(per_all), ref_per)
    sem = np.sqrt(z.var(ddof=1) / z.size + zr.var(ddof=1) / zr.size)
    assert abs(z.mean() - zr.mean()) < 3.0 * sem, (z.mean(), zr.mean(), sem)
    assert 0.6 < z.std(ddof=1) / zr.std(ddof=1) < 1.6
    assert abs(np.mean(nd) / np.mean([r["ndead"] for r in ref["runs"]]) - 1.0) < 0.03


@pytest.mark.parametrize("D,nDer,nlive,nr,kind", [(20, 2, 400, 20, "gaussian"), (70, 0, 150, 8, "corr_gaussian")])
def test_modes_change_no_number(engine, D, nDer, nlive, nr, kind):
    """pool mode, the deferred update and the fused update are rearrangements of WHERE rows live and WHEN kernels run: a run
    with each of them switched off (settings.ablate bits 1, 2, 3; bit 4 for the two ways of the evidence prefix sums) must reproduce the default run -- every counter exactly,
    every number to round-off (the covariance sums group the rows differently: differences of a few 1e-16 in the directions)"""
    api = engine; olib = orc.load()
    if kind == "gaussian": L, P, keep = api.make_problem("gaussian", D, nDer)
    else:
        ic = np.zeros((D, D)); ld = C.c_double()
        olib.pc_random_invcov(12345, D, C.c_double(0.1), orc.dptr(ic), C.byref(ld))
        L, P, keep = api.make_problem("corr_gaussian", D, nDer, invcov=ic, mean=np.full(D, 0.5), logdet=ld.value)
    runs = []
    for ab in (0, 2, 2 | 4, 2 | 4 | 8, 16):            # (bit 4: the contraction's evidence prefixes as pair scans instead of in linear space)
        s = _settings(api, D, nDer, nlive=nlive, num_repeats=nr, seed=11, batch=0, max_ndead=8 * nlive)
        s.ablate = ab
        runs.append(api.run(s, L, P))
    g0 = runs[0]
    assert g0["nupdates"] >= 3
    for g in runs[1:]:
        for k in ("ndead", "nlike", "niter"):
            assert g[k] == g0[k], (k, g[k], g0[k])
        assert abs(g["logZ"] - g0["logZ"]) < 1e-11 * max(1.0, abs(g0["logZ"])) and abs(g["logZerr"] - g0["logZerr"]) < 1e-11
        assert np.abs(g["dead"] - g0["dead"]).max() < 1e-9 * max(1.0, np.abs(g0["dead"]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("kind,D,nDer,nlive,nr,clus,box", [("gaussian", 20, 2, 2000, 40, 0, None), ("gaussian", 5, 1, 200, 10, 0, (-0.5, 1.5)), ("gaussian", 24, 0, 256, 48, 0, None),
                                                          ("gaussian", 12, 3, 400, 24, 1, None), ("rastrigin", 4, 0, 400, 12, 1, (-5.12, 5.12)),
                                                          ("rastrigin", 10, 0, 400, 30, 1, (-5.12, 5.12)), ("gaussian", 9, 2, 200, 64, 0, None)])
def test_the_helper_wavefronts_change_no_number(engine, kind, D, nDer, nlive, nr, clus, box):
    """the fused sampling kernel with four chains a workgroup and a helper wavefront each (deck shuffle and whitening next to the seed choice:
    pc_slice_body.inc) against one wavefront a workgroup (settings.ablate bit 13): bit for bit the same run -- closed form, general functor
    (bit 0), Rastrigin with clusters (several clusters: the helper leaves the whitening to the chain, whose seed names the factor)."""
    api = engine
    L, P, keep = api.make_problem(kind, D, nDer, *box) if box else api.make_problem(kind, D, nDer)
    for base in ((0, 1) if kind == "gaussian" else (0,)):
        runs = []
        for ab in (base, base | 8192, base | 16384):      # (bit 14: the direction's s.M.s reduced by the chain instead of taken from the helper's table)
            s = _settings(api, D, nDer, nlive=nlive, num_repeats=nr, seed=77, batch=0, do_clustering=clus, max_ndead=12 * nlive)
            s.ablate = ab
            runs.append(api.run(s, L, P))
        a = runs[0]
        assert a["nupdates"] >= 3 and a["nbatches"] > 5
        for b in runs[1:]:
            for k in ("ndead", "nlike", "niter", "nbatches", "nupdates", "ncluster", "ncluster_dead"):
                assert a[k] == b[k], (k, a[k], b[k])
            assert a["logZ"] == b["logZ"] and a["logZerr"] == b["logZerr"]
            assert np.array_equal(a["dead"], b["dead"], equal_nan=True) and np.array_equal(a["live"], b["live"], equal_nan=True) and np.array_equal(a["logweights"], b["logweights"])


@pytest.mark.gpu
@pytest.mark.parametrize("D,nDer,nlive,nr,kind,box", [(20, 2, 2000, 40, "gaussian", None), (6, 1, 300, 12, "gaussian", (-0.25, 1.5)), (3, 0, 200, 9, "gaussian", None),
                                                     (24, 2, 300, 24, "gaussian", None), (40, 0, 200, 10, "gaussian", None), (12, 0, 200, 12, "corr_gaussian", None),
                                                     (70, 0, 150, 8, "corr_gaussian", None)])
def test_the_quadratic_likelihoods_as_general_functors_are_the_same_run(engine, D, nDer, nlive, nr, kind, box):
    """settings.ablate bit 0 takes the closed form along the chord away from the built-in Gaussians: every trial point is then a call of the
    likelihood with a wave reduction, as for any device functor (the path bench.py's value_general_functor times).  The same run: every
    counter exactly -- the evaluations of the initial bracket, the steps out and the shrinkage are the same points -- and every number to
    round-off.  (Until round 6 the bracket's two ends of these kinds were summed by another likelihood's formula: no step out was ever
    made, 3.3 evaluations a slice instead of 4.5 -- and no test looked.)"""
    api = engine; olib = orc.load()
    if kind == "gaussian": L, P, keep = api.make_problem("gaussian", D, nDer, *box) if box else api.make_problem("gaussian", D, nDer)
    else:
        ic = np.zeros((D, D)); ld = C.c_double()
        olib.pc_random_invcov(4321, D, C.c_double(0.1), orc.dptr(ic), C.byref(ld))
        L, P, keep = api.make_problem("corr_gaussian", D, nDer, invcov=ic, mean=np.full(D, 0.5), logdet=ld.value)
    runs = []
    for ab in (0, 1):
        s = _settings(api, D, nDer, nlive=nlive, num_repeats=nr, seed=21, batch=0, max_ndead=10 * nlive)
        s.ablate = ab
        runs.append(api.run(s, L, P))
    a, b = runs
    assert a["nupdates"] >= 3
    for k in ("ndead", "nlike", "niter", "nbatches", "nupdates"):
        assert a[k] == b[k], (k, a[k], b[k])
    assert abs(a["logZ"] - b["logZ"]) < 1e-10 * max(1.0, abs(a["logZ"])) and abs(a["logZerr"] - b["logZerr"]) < 1e-10
    assert np.abs(a["dead"] - b["dead"]).max() < 1e-9 * max(1.0, np.abs(a["dead"]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("kind,D,nDer,nlive,nr,box", [("rastrigin", 4, 0, 400, 12, (-5.12, 5.12)), ("twin_gaussian", 8, 1, 300, 16, (-1.0, 1.0)),
                                                     ("rastrigin", 2, 0, 600, 6, (-5.12, 5.12)),
                                                     ("rastrigin", 2, 0, 400, 70, (-5.12, 5.12)),          # two phantom-mask words per chain
                                                     ("rastrigin", 3, 0, 2400, 9, (-5.12, 5.12)),          # more than 64 clusters alive: two clusters per lane
                                                     ("rastrigin", 2, 0, 1000, 6, (-5.12, 5.12)),          # ... in a live set the LDS-resident kernels take (121 modes, eight points a mode)
                                                     ("twin_gaussian", 20, 1, 400, 10, (-1.0, 1.0))])      # steep: launches that end at their 300-nat window
def test_clustered_contraction_kernels_agree(engine, kind, D, nDer, nlive, nr, box):
    """several clusters, three kernels, one run: the contraction with its decisions made in parallel (k_consume_clp, pc_consume_clp_body.inc: the
    acceptance vector as a fixed point over dominance counts, everything else from prefix sums and the merged order of deaths), the one
    whose single wavefront decides chain after chain (k_consume_cl: settings.ablate bit 10) and the general kernel both replace (bit 5
    sends every launch there) -- every counter, every dead row, the evidences of all clusters, the live set; the first two also the same
    BITS in the evidences (the same statements in the same order on the evidence side)"""
    api = engine
    L, P, keep = api.make_problem(kind, D, nDer, *box)
    runs = []
    for ab in (0, 1024, 32):
        s = _settings(api, D, nDer, nlive=nlive, num_repeats=nr, seed=21, batch=0, do_clustering=1)
        s.ablate = ab
        runs.append(api.run(s, L, P))
    a, c, b = runs
    # (nlive 2400 / 1024 chains: no LDS-resident kernel takes a live set of that size with that nursery -- all three runs go through the general kernel,
    #  which pchip_result.path now shows; the shape stays for the general kernel's own two-clusters-per-lane code)
    if nlive < 2000:
        assert a["path"]["consume_general"] == 0 and a["path"]["consume_cl"] > 0 and a["path"]["consume_cl_serial"] == 0
    if nlive < 2000:
        assert c["path"]["consume_cl_serial"] > 0 and c["path"]["consume_cl"] == 0
    assert b["path"]["consume_general"] > 0 and b["path"]["consume_cl"] == 0
    assert a["ncluster_peak"] >= 2 and a["ncluster_dead"] >= 2          # clusters were found, and clusters died on the way
    # (more than 64 clusters alive at once -- two clusters per lane in the LDS-resident kernels -- needs a live set those kernels' LDS does not take:
    #  2400 points reach it, through the general kernel; the largest live set that fits holds 40 clusters at its peak)
    if nlive == 2400: assert a["ncluster_peak"] > 64, a["ncluster_peak"]
    if nlive == 1000: assert a["ncluster_peak"] > 32, a["ncluster_peak"]
    for b in (c, b):
        for k in ("ndead", "nlike", "niter", "ncluster", "ncluster_dead", "nupdates", "ncluster_peak", "nlike_failed"):
            assert a[k] == b[k], (k, a[k], b[k])
        assert abs(a["logZ"] - b["logZ"]) < 1e-10 and abs(a["logZerr"] - b["logZerr"]) < 1e-10
        assert np.array_equal(a["dead"][:, :-2], b["dead"][:, :-2]) and np.array_equal(a["dead"][:, -1], b["dead"][:, -1])
        assert np.abs(a["logweights"] - b["logweights"]).max() < 1e-9
        assert np.allclose(a["logZp"], b["logZp"], atol=1e-9) and np.allclose(a["varlogZp"], b["varlogZp"], atol=1e-9)
        assert np.array_equal(a["live"], b["live"])
    # (k_consume_clp goes on through a cluster's end where k_consume_cl stages again with new references for its linear-space sums: the same
    #  numbers to rounding, not the same bits; with settings.ablate bit 8 -- a pass ends at a cluster's death -- the two are bit for bit the same run)
    runs8 = []
    for ab in (256, 256 | 1024):
        s = _settings(api, D, nDer, nlive=nlive, num_repeats=nr, seed=21, batch=0, do_clustering=1)
        s.ablate = ab
        runs8.append(api.run(s, L, P))
    a8, c8 = runs8
    if nlive < 2000:
        assert a8["path"]["consume_cl"] > 0 and c8["path"]["consume_cl_serial"] > 0
    assert np.array_equal(a8["dead"], c8["dead"]) and np.array_equal(a8["live"], c8["live"])
    if a8["path"]["nn_fallbacks"] == 0:      # (a baby for the full search ends k_consume_clp's pass -- new references -- where k_consume_cl searches on the spot)
        assert a8["logZ"] == c8["logZ"] and np.array_equal(a8["logweights"], c8["logweights"])
    else:
        assert abs(a8["logZ"] - c8["logZ"]) < 1e-12 and np.abs(a8["logweights"] - c8["logweights"]).max() < 1e-10
    # the evidence as sums over all of a pass's deaths (bit 11: a walk per cluster, pair sums, prefix sums -- an experiment, not the default): <Z> and the
    # log weights come out of the same statements in the same order, <Z^2> and <Z X_q> in another grouping (the reported log Z is 2 log<Z> - log<Z^2> / 2)
    s = _settings(api, D, nDer, nlive=nlive, num_repeats=nr, seed=21, batch=0, do_clustering=1)
    s.ablate = 256 | 2048
    a9 = api.run(s, L, P)
    assert np.array_equal(a9["dead"], c8["dead"]) and np.array_equal(a9["live"], c8["live"]) and a9["ndead"] == c8["ndead"] and a9["nupdates"] == c8["nupdates"]
    assert abs(a9["logZ"] - c8["logZ"]) < 1e-12 and abs(a9["logZerr"] - c8["logZerr"]) < 1e-10 and np.allclose(a9["logZp"], c8["logZp"], atol=1e-10)
    if a9["path"]["nn_fallbacks"] == 0:
        assert np.array_equal(a9["logweights"], c8["logweights"])
    assert a8["ndead"] == a["ndead"] and abs(a8["logZ"] - a["logZ"]) < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("kind,D,nDer,nlive,nr,box,maxnd", [("rastrigin", 4, 0, 400, 12, (-5.12, 5.12), 7000),         # 11 clusters alive at the end
                                                            ("rastrigin", 2, 0, 600, 6, (-5.12, 5.12), 3000),          # 25
                                                            ("twin_gaussian", 8, 1, 300, 16, (-1.0, 1.0), -1), ("twin_gaussian", 30, 1, 500, 40, (-1.0, 1.0), 20000),      # 2
                                                            ("rastrigin", 10, 0, 1000, 30, (-5.12, 5.12), 24000),
                                                            ("rastrigin", 3, 0, 2400, 9, (-5.12, 5.12), 12000),        # 15, a live set of 4096 sort entries
                                                            ("rastrigin", 3, 0, 2400, 9, (-5.12, 5.12), 20000)])       # 69: more than the kernel takes, both runs by the general one
def test_clustered_kill_off_kernels_agree(engine, kind, D, nDer, nlive, nr, box, maxnd):
    """the end of a run with several clusters alive (nested_sampling.F90:381-384: every live point dies, lowest first; delete_cluster
    whenever a cluster has lost its last point): the one-wave kill-off of pc_clus.hip (deaths in the sorted order, cluster = lane)
    against the general kernel it stands in for (settings.ablate bit 9) -- the same statements in the same order: the same bits.
    max_ndead stops some of the runs early, with more clusters alive than at a run's natural end"""
    api = engine
    L, P, keep = api.make_problem(kind, D, nDer, *box)
    runs = []
    for ab in (0, 512):
        s = _settings(api, D, nDer, nlive=nlive, num_repeats=nr, seed=33, batch=0, do_clustering=1, max_ndead=maxnd)
        s.ablate = ab
        runs.append(api.run(s, L, P))
    a, b = runs
    assert a["ncluster"] >= 2                                           # (clusters alive when the run stopped: the kill-off's)
    for k in ("ndead", "nlike", "niter", "ncluster", "ncluster_dead", "nupdates", "ncluster_peak"):
        assert a[k] == b[k], (k, a[k], b[k])
    assert a["logZ"] == b["logZ"] and a["logZerr"] == b["logZerr"], (a["logZ"], b["logZ"])
    assert np.array_equal(a["dead"], b["dead"])
    assert np.array_equal(a["logweights"], b["logweights"])
    assert np.array_equal(a["logZp"], b["logZp"]) and np.array_equal(a["varlogZp"], b["varlogZp"])
    assert np.array_equal(a["post_mean"], b["post_mean"]) if "post_mean" in a else True


@pytest.mark.gpu
@pytest.mark.parametrize("kind,D,nDer,nlive,nr,box,maxnd", [("rastrigin", 4, 0, 300, 8, (-5.12, 5.12), 2400), ("twin_gaussian", 30, 1, 300, 10, (-1.0, 1.0), -1)])
def test_phantoms_find_the_same_clusters_by_both_kernels(engine, kind, D, nDer, nlive, nr, box, maxnd):
    """a split's phantoms go to the cluster of their nearest live point (run_time_info.f90:444-453): the lane-per-phantom kernel
    (nDims <= 32: coordinates in registers, live points by LDS broadcast) against the general one it stands in for
    (PC_PH_REHOME_GENERAL=1, read once per process: a child process) -- the same run to the last bit"""
    import json, subprocess, sys
    api = engine
    L, P, keep = api.make_problem(kind, D, nDer, *box)
    s = _settings(api, D, nDer, nlive=nlive, num_repeats=nr, seed=77, batch=0, do_clustering=1, max_ndead=maxnd)
    a = api.run(s, L, P)
    assert a["ncluster_peak"] >= 2
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, json, ctypes as C; sys.path.insert(0, %r)\n"
            "from polychordlite_amd import _ctypes_api as api\n"
            "lib = api.load(); L, P, keep = api.make_problem(%r, %d, %d, *%r)\n"
            "s = api.Settings(); lib.pchip_settings_default(C.byref(s), %d, %d)\n"
            "s.nlive, s.num_repeats, s.seed, s.batch, s.do_clustering, s.max_ndead = %d, %d, 77, 0, 1, %d\n"
            "g = api.run(s, L, P)\n"
            "print(json.dumps(dict(ndead=int(g['ndead']), nlike=int(g['nlike']), logZ=float(g['logZ']).hex(), peak=int(g['ncluster_peak']), sumdead=float(g['dead'][:, :-2].sum()).hex())))\n"
            % (root, kind, D, nDer, box, D, nDer, nlive, nr, maxnd))
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PC_PH_REHOME_GENERAL="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-1500:]
    b = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert (b["ndead"], b["nlike"], b["peak"]) == (int(a["ndead"]), int(a["nlike"]), int(a["ncluster_peak"]))
    assert b["logZ"] == float(a["logZ"]).hex() and b["sumdead"] == float(a["dead"][:, :-2].sum()).hex()


@pytest.mark.gpu
def test_clustered_contraction_with_a_chain_that_has_no_number(engine):
    """two repeats in ten dimensions, 200 live points and a nursery of 100: clusters of fewer points than dimensions have singular
    covariances, and a chain that starts there comes back with NaN for its last logL.  Such a chain must take the same place among
    the launch's candidates in both contraction kernels (it used to share rank 0 with the lowest real candidate in the one-wave
    kernel: two writers of one record, a run that ended early and not the same way twice); found by tools/dev/fuzz_parity.py"""
    api = engine
    L, P, keep = api.make_problem("rastrigin", 10, 0, -5.12, 5.12)
    runs = []
    for ab in (0, 0, 32):
        # (epoch_discard = 1: the shape was found under the reference's rule for chains in flight.  Under the default rule the two
        #  kernels' logweights differ in the last bit at dead point 373 -- different groupings of the same sums -- and with clusters
        #  of two points in ten dimensions that is enough to part the two trajectories a few hundred deaths later)
        s = _settings(api, 10, 0, nlive=200, num_repeats=2, seed=8222, batch=100, do_clustering=1, compression_factor=0.9, epoch_discard=1)
        s.ablate = ab
        runs.append(api.run(s, L, P))
    a, a2, b = runs
    assert a["ncluster_dead"] >= 2 and a["ndead"] > 4000
    for o in (a2, b):
        for k in ("ndead", "nlike", "niter", "ncluster", "ncluster_dead", "nupdates"):
            assert a[k] == o[k], (k, a[k], o[k])
        assert abs(a["logZ"] - o["logZ"]) < 1e-10
        assert np.array_equal(a["dead"][:, :-2], o["dead"][:, :-2], equal_nan=True)
        assert np.array_equal(a["live"], o["live"], equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("D,nDer,nlive,nr,kw", [(20, 2, 2000, 40, {}), (8, 0, 300, 16, {}), (16, 1, 500, 32, {}), (24, 2, 400, 48, {}),
                                                (5, 2, 200, 25, dict(box=(-0.5, 1.5))), (20, 2, 2000, 40, dict(batch=700)),
                                                (3, 1, 120, 9, dict(batch=70)), (17, 0, 250, 17, dict(box=(0.1, 0.9)))])
def test_lane_per_chain_kernels_change_no_number(engine, D, nDer, nlive, nr, kw):
    """the kernels a run uses next to other runs of its device (pc_slice_t.hip: 64 chains to a wavefront, one per lane; a basis
    per lane) against the ones it uses alone (k_slice: a wavefront per chain; k_nhats: a wavefront per basis) -- the same run
    bit for bit: settings.ablate bit 6 = lane-per-chain sampling, bit 7 = lane-per-basis directions, both, neither.  Covers the
    three compiled widths of the old kernels (nDims <= 8, <= 16, <= 24), a prior box that is not the unit cube, nurseries that
    do not fill their last wavefront, odd nDims (an odd number of deviates per basis: stream calls shared between bases)"""
    api = engine
    box = kw.get("box", (None, None))
    L, P, keep = api.make_problem("gaussian", D, nDer, *box) if box[0] is not None else api.make_problem("gaussian", D, nDer)
    runs = []
    for ab in (0, 64, 128, 192):
        s = _settings(api, D, nDer, nlive=nlive, num_repeats=nr, seed=11, batch=kw.get("batch", 0))
        s.ablate = ab
        runs.append(api.run(s, L, P))
    a = runs[0]
    assert a["ndead"] > 3 * nlive
    for b in runs[1:]:
        for k in ("ndead", "nlike", "niter", "nupdates", "nbatches"):
            assert a[k] == b[k], (k, a[k], b[k])
        assert a["logZ"] == b["logZ"] and a["logZerr"] == b["logZerr"]
        assert np.array_equal(a["dead"], b["dead"]) and np.array_equal(a["logweights"], b["logweights"]) and np.array_equal(a["live"], b["live"])
        assert np.array_equal(a["post_mean"], b["post_mean"])
    # ... and the run with both lane-per-chain kernels is the ORACLE's run of these settings (whole runs up to nDims 7, beyond that
    # the first generations: round-off grows with every covariance update)
    g = runs[3]
    so = orc.settings(D, nDer, nlive=nlive, num_repeats=nr, seed=11, batch=g["batch"])
    Lo, Po, keep2 = orc.make_problem("gaussian", D, *box)
    o = orc.run(so, Lo, Po)
    _next_to_the_oracle(g, o, D, nlive)


def _next_to_the_oracle(g, o, D, nlive):
    if D <= 7:
        for k in ("ndead", "nlike", "niter", "nbatches", "ncluster", "ncluster_dead"):
            assert g[k] == o[k], (k, g[k], o[k])
        assert abs(g["logZ"] - o["logZ"]) < 1e-8 and abs(g["logZerr"] - o["logZerr"]) < 1e-8
        rel = np.abs(g["dead"] - o["dead"]) / np.maximum(1.0, np.abs(o["dead"]))
        assert rel.max() < 1e-7
    else:
        n0 = min(6 * nlive, int(g["ndead"]), int(o["ndead"]))
        rel = np.abs(g["dead"][:n0] - o["dead"][:n0]) / np.maximum(1.0, np.abs(o["dead"][:n0]))
        assert rel.max() < 1e-7
        assert abs(g["logZ"] - o["logZ"]) < 3.0 * (g["logZerr"] + o["logZerr"])
