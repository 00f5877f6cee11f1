"""GPU (-m gpu): every BASELINE.json configuration at its STATED size, through the C ABI.

  C2  20-D Gaussian, nlive 2000, num_repeats 40, default nursery (1000 chains): oracle walked next to the engine
      for 3 nlive deaths at the production shape (k_consume_par at full width), every dead row compared
  C3  10-D Rastrigin, nlive 1000, num_repeats 30, kNN clustering (likelihoods/examples/rastrigin.f90:20-35,
      clustering.f90:15-97, run_time_info.f90:913-949): an oracle prefix long enough to split clusters, full runs
      against runs of the reference binary (tests/golden/ref_c3_seeds.json) and the analytic evidence
  C4  30-D twin Gaussian, nlive 500, num_repeats 40, kNN clustering (twin_gaussian.f90:14-56): an oracle prefix through the
      split of the two modes (the statistics against the reference binary's 32 runs: test_gpu_parity.py)
  C5  100-D correlated Gaussian, nlive 5000, num_repeats 200 (random_gaussian.f90:17-30, run_time_info.f90:601-641):
      an oracle prefix at the real nlive and nDims and one full run checked through size-independent properties

Tolerances as in test_gpu_parity.py: integers exact, dead rows 1e-7 relative, logZ 1e-8 (1e-6 relative where logZ
is O(1e5))."""
import ctypes as C

import numpy as np
import pytest

from tests import oracle_api as orc

pytestmark = pytest.mark.gpu


def _settings(api, D, nDer, **kw):
    lib = api.load()
    s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, nDer)
    for k, v in kw.items():
        setattr(s, k, v)
    return s


def _same_trajectory(g, o, ztol=1e-8):
    for k in ("ndead", "nlike", "niter", "nbatches", "ncluster", "ncluster_dead"):
        assert g[k] == o[k], (k, g[k], o[k])
    assert abs(g["logZ"] - o["logZ"]) < ztol * max(1.0, abs(o["logZ"]))
    assert abs(g["logZerr"] - o["logZerr"]) < ztol * max(1.0, abs(o["logZerr"]))
    rel = np.abs(g["dead"] - o["dead"]) / np.maximum(1.0, np.abs(o["dead"]))
    assert rel.max() < 1e-7
    ok = o["logweights"] > -1e29
    assert np.array_equal(ok, g["logweights"] > -1e29)
    assert np.abs(g["logweights"][ok] - o["logweights"][ok]).max() < 1e-8 * max(1.0, np.abs(o["logweights"][ok]).max())


def _replay_properties(g, D, clustered=False):
    """size-independent properties of a full run: deaths in ascending logL, each point above its birth contour,
    cube coordinates inside the box, and the evidence recursion replayed over the run's own (logL, entry) records"""
    from tests.replay_oracle import evidence_replay, lived_records
    lived = g["logweights"] > -1e29
    d = g["dead"][lived]
    assert np.all(np.diff(d[:, -1]) >= 0)
    assert np.all(d[:, -1] > d[:, -2])
    assert np.all((d[:, :D] >= 0) & (d[:, :D] <= 1))
    lz, var = evidence_replay(*lived_records(g))
    if not clustered:       # one cluster: the replay IS the engine's recursion
        assert abs(lz - g["logZ"]) < 1e-6 * max(1.0, abs(g["logZ"])) and abs(var - g["varlogZ"]) < 1e-6
    else:
        # several clusters: the reference tracks one volume per cluster (split in proportion to the points at a split,
        # run_time_info.f90:458-503), so its evidence is NOT the single-volume replay of the same deaths -- the two are
        # different estimators that agree within a few error bars; what holds exactly is that the evidence is the sum of
        # the clusters' evidences (in <Z>; the reported numbers are log-normal location parameters)
        assert abs(lz - g["logZ"]) < 4.0 * g["logZerr"]
        zp = g["logZp"] + g["varlogZp"] / 2.0
        assert abs(np.logaddexp.reduce(zp) - (g["logZ"] + g["varlogZ"] / 2.0)) < 1e-8
    # posterior weights: logweight + logL normalised by the evidence sums to one
    lw = g["logweights"][lived] + d[:, -1]
    assert abs(np.log(np.exp(lw - lw.max()).sum()) + lw.max() - g["logZ"]) < 3.0 * g["logZerr"] + 1e-6


def _paths(g, **want):
    """pchip_result.path: which kernels the run went through.  A shape that silently drops to the general serial kernel (an LDS layout
    that no longer fits, a guard that no longer holds) produces the same numbers -- only these counters tell (two such fallbacks shipped
    green in round 5).  want: name = exact count, or (lo, hi), or ">0"."""
    p = g["path"]
    for k, v in want.items():
        if v == ">0":
            assert p[k] > 0, (k, p)
        elif isinstance(v, tuple):
            assert v[0] <= p[k] <= v[1], (k, p)
        else:
            assert p[k] == v, (k, p)


def test_c2_production_shape_matches_oracle(engine):
    """BASELINE configs[1] exactly as bench.py runs it (batch = 0 -> 1000 chains per nursery, parallel contraction at
    full width), stopped after 3 nlive deaths: same trajectory as the oracle, row for row"""
    api = engine
    kw = dict(nlive=2000, num_repeats=40, seed=5, max_ndead=6000)
    s = _settings(api, 20, 2, batch=0, **kw)
    L, P, keep = api.make_problem("gaussian", 20, 2)
    g = api.run(s, L, P)
    assert g["batch"] == 1000
    so = orc.settings(20, 2, batch=1000, **kw)
    Lo, Po, keep2 = orc.make_problem("gaussian", 20)
    o = orc.run(so, Lo, Po)
    _same_trajectory(g, o)
    assert g["ndead"] == 8000 and g["nbatches"] >= 6
    # the production path: parallel contraction + its kill-off, fused update, pool mode, deferred update -- and nothing else
    _paths(g, consume_par=">0", consume_general=0, consume_cl=0, consume_fast=0, killoff_par=1, killoff_general=0, killoff_fast=0,
           update_fused=">0", update_steps=0, slice_wave=">0", pool_mode=1, defer_update=1)


def test_c3_prefix_matches_oracle(engine):
    """BASELINE configs[2] at its size, engine defaults (500 chains per nursery, candidate lists for identify_cluster),
    until the kNN clustering has split the live set several times: same trajectory as the oracle"""
    api = engine
    kw = dict(nlive=1000, num_repeats=30, seed=5, do_clustering=1, max_ndead=20000)
    s = _settings(api, 10, 0, batch=0, **kw)
    L, P, keep = api.make_problem("rastrigin", 10, 0, -5.12, 5.12)
    g = api.run(s, L, P)
    assert g["batch"] == 500
    so = orc.settings(10, 0, batch=500, **kw)
    Lo, Po, keep2 = orc.make_problem("rastrigin", 10, -5.12, 5.12)
    o = orc.run(so, Lo, Po)
    _same_trajectory(g, o)
    assert g["ncluster"] >= 4 and g["ndead"] >= 20000
    assert np.allclose(g["logZp"], o["logZp"], atol=1e-8)
    # one cluster: the parallel contraction; several: k_consume_cl and its kill-off; the general kernel never
    _paths(g, consume_par=">0", consume_cl=">0", consume_general=0, consume_fast=0, killoff_cl=1, killoff_general=0, nn_lists=">0",
           update_steps=">0")
    assert g["path"]["nn_fallbacks"] <= 0.002 * g["niter"], g["path"]


def test_c4_prefix_matches_oracle(engine):
    """BASELINE configs[3] at its size (ini/twin_gaussian.ini at 30-D: nlive 500, num_repeats 40, kNN clustering, engine default
    nursery of 250 chains; twin_gaussian.f90:14-56): the oracle walked next to the engine through the split of the two modes
    (before death 17 000 with this seed) and six live sets' worth of deaths beyond it -- k_nhats_q's two halves (nDims 25 ... 64),
    k_consume_cl at nlive 500 / 250 chains, the clustered kill-off: same trajectory, row for row"""
    api = engine
    kw = dict(nlive=500, num_repeats=40, seed=11, do_clustering=1, max_ndead=20000)
    s = _settings(api, 30, 1, batch=0, **kw)
    L, P, keep = api.make_problem("twin_gaussian", 30, 1, -1.0, 1.0)
    g = api.run(s, L, P)
    assert g["batch"] == 250
    so = orc.settings(30, 1, batch=250, **kw)
    Lo, Po, keep2 = orc.make_problem("twin_gaussian", 30, -1.0, 1.0)
    o = orc.run(so, Lo, Po)
    assert o["ncluster"] == 2 and o["ndead"] >= 20000
    _same_trajectory(g, o)
    assert np.allclose(g["logZp"], o["logZp"], atol=1e-8)
    _paths(g, consume_par=">0", consume_cl=">0", consume_general=0, consume_fast=0, killoff_cl=1, killoff_general=0, nn_lists=">0")


def test_c3_full_runs_against_the_reference_binary(engine, golden):
    """BASELINE configs[2] in full (ini/rastrigin.ini scaled to 10-D as BASELINE states it; num_repeats = 3 nDims):
    analytic logZ = -23.263; twelve runs of the reference binary (own RNG) in
    tests/golden/ref_c3_seeds.json.  Which of the ~100 modes a run finds is not in its reported error (for the
    reference either: its runs scatter by 0.67 against a reported 0.21), so the comparison is between means."""
    api = engine
    ref = golden["ref_c3_seeds"]
    c = ref["config"]
    zr = np.array([r["logZ"] for r in ref["runs"]])
    s = _settings(api, c["nDims"], c["nDerived"], nlive=c["nlive"], num_repeats=c["num_repeats"], do_clustering=1, batch=0)
    L, P, keep = api.make_problem("rastrigin", c["nDims"], 0, -5.12, 5.12)
    z, nd, ncl, errs, per, per_all = [], [], [], [], [], []
    for i in range(12):
        s.seed = 700 + i
        g = api.run(s, L, P)
        if i == 0:
            _replay_properties(g, c["nDims"], clustered=True)
        z.append(g["logZ"]); errs.append(g["logZerr"])
        nd.append(int((g["logweights"] > -1e29).sum())); ncl.append(g["ncluster_dead"])
        per.append((g["nlike"] - g["nlike_failed"]) / nd[-1]); per_all.append(g["nlike"] / nd[-1])
    z = np.array(z)
    truth = -23.263
    sem = np.sqrt(z.var(ddof=1) / z.size + zr.var(ddof=1) / zr.size)
    assert abs(z.mean() - zr.mean()) < 3.0 * sem, (z.mean(), zr.mean(), sem)
    assert abs(z.mean() - truth) < 3.0 * max(z.std(ddof=1) / np.sqrt(z.size), np.mean(errs) / np.sqrt(z.size)), (z.mean(), z.std(ddof=1))
    assert abs(np.mean(errs) / np.mean([r["logZerr"] for r in ref["runs"]]) - 1.0) < 0.15
    assert abs(np.mean(nd) / np.mean([r["ndead"] for r in ref["runs"]]) - 1.0) < 0.05
    assert min(ncl) >= 50, ncl                           # every run resolves dozens of the modes as separate clusters
    # likelihood evaluations per dead point: the chains that put a point into the live set cost what the reference's cost (155.5);
    # with the chains a nursery of nlive/2 loses -- babies born below the contour by the time they are looked at, chains seeded
    # in a cluster that died or was split meanwhile -- a dead point costs at most 1.35 times that (the
    # reference's own rule for its farm, settings.epoch_discard = 1, costs 2.2 times: 338)
    ref_per = np.mean([r["nlike"] / r["ndead"] for r in ref["runs"]])
    assert abs(np.mean(per) / ref_per - 1.0) < 0.05, (np.mean(per), ref_per)
    assert np.mean(per_all) / ref_per < 1.35, (np.mean(per_all), ref_per)


def _c5_problem(api, D=100):
    olib = orc.load()
    ic = np.zeros((D, D)); ld = C.c_double()
    olib.pc_random_invcov(12345, D, C.c_double(0.1), orc.dptr(ic), C.byref(ld))
    mean = np.full(D, 0.5)
    L, P, keep = api.make_problem("corr_gaussian", D, 0, invcov=ic, mean=mean, logdet=ld.value)
    Lo, Po, keep2 = orc.make_problem("corr_gaussian", D, invcov=ic, mean=mean, logdet=ld.value)
    return (L, P, keep), (Lo, Po, keep2), ld.value


def test_c5_prefix_matches_oracle(engine):
    """BASELINE configs[4] at the real nlive AND nDims (5000 x 100, num_repeats 200, 1024 chains per nursery): matrix-core
    covariance over 5000 live points + their phantoms, whitening, the parallel contraction at nlive 5000 -- two
    nurseries next to the oracle"""
    api = engine
    (L, P, keep), (Lo, Po, keep2), _ = _c5_problem(api)
    kw = dict(nlive=5000, num_repeats=200, seed=5, max_ndead=2048)
    s = _settings(api, 100, 0, batch=0, **kw)
    g = api.run(s, L, P)
    assert g["batch"] == 1024
    so = orc.settings(100, 0, batch=1024, **kw)
    o = orc.run(so, Lo, Po)
    _same_trajectory(g, o, ztol=1e-6)
    assert g["ndead"] == 2048 + 5000
    _paths(g, consume_par=">0", consume_general=0, consume_fast=0, killoff_par=1, killoff_general=0, update_steps=0)


def _bench_matrix(api, D=100):
    """the matrix the reference binary was given for tests/golden/ref_c5_seeds.json (bench.py random_correlated_gaussian:
    random orthonormal eigenbasis from numpy's generator, eigen-sigma 0.1 ... 0.001)"""
    from bench import random_correlated_gaussian
    ic, mean, logdet = random_correlated_gaussian(D)
    return api.make_problem("corr_gaussian", D, 0, invcov=ic, mean=mean, logdet=logdet)


def test_c5_runs_against_the_reference_binary(engine, golden):
    """BASELINE configs[4]'s likelihood, dimension and repeats (100-D correlated Gaussian, num_repeats = 2 nDims) at the
    live-set size the REFERENCE BINARY finishes in half an hour: eight of its runs (own RNG) in
    tests/golden/ref_c5_seeds.json.  At num_repeats = 2 nDims PolyChord's evidence is biased high in 100 dimensions --
    the reference's eight runs give +3.5 +- 0.25 against the analytic 0 -- and the engine must show THE SAME bias, the
    same reported error, the same number of dead points and the same cost per dead point: the algorithm's behaviour,
    not the engine's."""
    api = engine
    ref = golden["ref_c5_seeds"]
    c = ref["config"]
    zr = np.array([r["logZ"] for r in ref["runs"]])
    assert c["nDims"] == 100 and c["num_repeats"] == 200 and zr.size >= 8
    L, P, keep = _bench_matrix(api, c["nDims"])
    s = _settings(api, c["nDims"], 0, nlive=c["nlive"], num_repeats=c["num_repeats"], batch=0)
    z, errs, nd, per = [], [], [], []
    for i in range(10):
        s.seed = 900 + i
        g = api.run(s, L, P)
        z.append(g["logZ"]); errs.append(g["logZerr"])
        lived = int((g["logweights"] > -1e29).sum())
        nd.append(lived); per.append((g["nlike"] - g["nlike_failed"]) / lived)
    z = np.array(z)
    sem = np.sqrt(z.var(ddof=1) / z.size + zr.var(ddof=1) / zr.size)
    assert abs(z.mean() - zr.mean()) < 3.0 * sem, (z.mean(), zr.mean(), sem)
    assert zr.mean() > 2.0 and z.mean() > 2.0                                     # the bias is there, in both
    assert abs(np.mean(errs) / np.mean([r["logZerr"] for r in ref["runs"]]) - 1.0) < 0.05
    assert abs(np.mean(nd) / np.mean([r["ndead"] for r in ref["runs"]]) - 1.0) < 0.02
    ref_per = np.mean([r["nlike"] / r["ndead"] for r in ref["runs"]])
    assert abs(np.mean(per) / ref_per - 1.0) < 0.05, (np.mean(per), ref_per)      # likelihood evaluations per dead point


def test_c5_full_run_properties(engine, golden):
    """BASELINE configs[4] in full (about 1.5e9 likelihood evaluations, 1.9e6 dead points).  The analytic evidence is
    0 (the Gaussian is normalised and sits well inside the unit box); at num_repeats = 2 nDims the algorithm is biased
    high in 100 dimensions -- by 3.5 at nlive 500 for the reference binary itself (tests/golden/ref_c5_seeds.json,
    test_c5_runs_against_the_reference_binary) -- and the bias falls with the number of live points: with ten times
    as many the run must lie between the truth and a fraction of that; the properties are exact."""
    api = engine
    (L, P, keep), _, _ = _c5_problem(api)
    s = _settings(api, 100, 0, nlive=5000, num_repeats=200, seed=31, batch=0)
    g = api.run(s, L, P)
    assert g["ncluster_dead"] == 1 and g["nlike"] > 1.0e9 and g["ndead"] > 1.5e6
    assert 0.2 < g["logZerr"] < 0.3
    ref_bias = float(np.mean([r["logZ"] for r in golden["ref_c5_seeds"]["runs"]]))        # reference, nlive 500
    assert -4.0 * g["logZerr"] < g["logZ"] < 0.5 * ref_bias + 4.0 * g["logZerr"], (g["logZ"], ref_bias)
    _replay_properties(g, 100)
    # posterior: mean 0.5 in every dimension, total variance = sum of the eigen-variances
    sig = 0.1 * (1e-2) ** (np.arange(100) / 99.0)
    assert np.all(np.abs(g["post_mean"][:100] - 0.5) < 0.01)
    assert abs(g["post_var"][:100].sum() / (sig ** 2).sum() - 1.0) < 0.1
