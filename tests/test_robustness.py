"""GPU (-m gpu): capacities that grow on demand and fatal conditions that come back as error codes.

The reference reallocates its cluster and phantom arrays whenever they fill up (run_time_info.f90:392-418,
:747-757) and answers fatal conditions with a message and `stop 1` (abort.F90:19-29).  The engine's arrays are flat
with capacities: they must grow the same way, with the trajectory unchanged (oracle), and a failure inside the
engine must come back through the C ABI as a return code -- an engine inside a Python interpreter must not take it
down."""
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


def _same(g, o):
    for k in ("ndead", "nlike", "niter", "nbatches", "ncluster", "ncluster_dead"):
        assert g[k] == o[k], (k, g[k], o[k])
    assert abs(g["logZ"] - o["logZ"]) < 1e-8 and abs(g["logZerr"] - o["logZerr"]) < 1e-8
    rel = np.abs(g["dead"] - o["dead"]) / np.maximum(1.0, np.abs(o["dead"]))
    assert rel.max() < 1e-7


@pytest.fixture
def capacities(engine):
    lib = engine.load()
    lib.pchip_set_capacity.argtypes = [C.c_int, C.c_int]
    yield lib
    lib.pchip_set_capacity(128, 0)
    lib.pchip_inject_fault(0)


def test_cluster_arrays_grow_on_demand(engine, capacities):
    """a clustered Rastrigin run that needs a dozen clusters, started with room for two: the per-cluster arrays (lists,
    evidences, the cross-volume matrix with its leading dimension, Cholesky factors) double four times; same trajectory
    as the oracle, which has no capacity at all"""
    api = engine
    capacities.pchip_set_capacity(2, -1)
    kw = dict(nlive=300, num_repeats=6, seed=5, batch=40, do_clustering=1)
    s = _settings(api, 2, 0, **kw)
    L, P, keep = api.make_problem("rastrigin", 2, 0, -5.12, 5.12)
    g = api.run(s, L, P)
    so = orc.settings(2, 0, **kw)
    Lo, Po, keep2 = orc.make_problem("rastrigin", 2, -5.12, 5.12)
    o = orc.run(so, Lo, Po)
    _same(g, o)
    assert g["ncluster_dead"] > 8
    assert np.allclose(g["logZp"], o["logZp"], atol=1e-8)


def test_more_than_128_clusters(engine):
    """3-D Rastrigin (1331 modes in the box) with 8000 live points: 150 modes are alive as separate clusters at the
    same time -- beyond the initial capacity of 128, where the engine used to stop.  Analytic logZ = 3 x (-2.326314).
    (2-D has 121 modes in all and 4-D / 5-D runs at this nlive peak at ~105 clusters: tools/dev/gpu_many_clusters.py.)"""
    api = engine
    s = _settings(api, 3, 0, nlive=8000, num_repeats=9, seed=3, do_clustering=1, batch=0)
    L, P, keep = api.make_problem("rastrigin", 3, 0, -5.12, 5.12)
    g = api.run(s, L, P)
    assert g["ncluster_peak"] > 128, g["ncluster_peak"]
    assert g["ncluster_dead"] >= g["ncluster_peak"]
    assert abs(g["logZ"] - 3 * (-2.326314)) < 3.0 * g["logZerr"], (g["logZ"], g["logZerr"])
    lived = g["logweights"] > -1e29
    assert np.all(np.diff(g["dead"][lived][:, -1]) >= 0)
    # the evidence is the sum of the clusters' evidences (in <Z>, not in the log-normal location parameter)
    zp = g["logZp"] + g["varlogZp"] / 2.0
    assert abs(np.logaddexp.reduce(zp) - (g["logZ"] + g["varlogZ"] / 2.0)) < 1e-8


def test_phantom_array_grows_on_demand(engine, capacities):
    """compression_factor = 0.01: the phantoms are cleaned every 4.6 nlive deaths and pile up to ~5 x num_repeats x nlive
    in between (the initial estimate is 4 x); and a deliberately small initial array.  Same trajectory as the oracle."""
    api = engine
    kw = dict(nlive=200, num_repeats=12, seed=7, batch=32, compression_factor=0.01)
    L, P, keep = api.make_problem("gaussian", 6, 1)
    so = orc.settings(6, 1, **kw)
    Lo, Po, keep2 = orc.make_problem("gaussian", 6)
    o = orc.run(so, Lo, Po)
    for cap in (0, 600):
        capacities.pchip_set_capacity(-1, cap)
        g = api.run(_settings(api, 6, 1, **kw), L, P)
        _same(g, o)


@pytest.mark.parametrize("fault,code", [(1, 7), (2, 8), (3, 7)])
def test_failures_return_codes_and_the_process_goes_on(engine, capacities, fault, code, capfd):
    """an out-of-memory condition at an allocation, a cluster limit at a split, a failed growth of the phantom array:
    pchip_run returns the documented code with a message on stderr, releases everything, and the next run in the same
    process is the run it would have been"""
    api = engine; lib = capacities
    kw = dict(nlive=300, num_repeats=6, seed=5, batch=40, do_clustering=1)
    L, P, keep = api.make_problem("rastrigin", 2, 0, -5.12, 5.12)
    good = api.run(_settings(api, 2, 0, **kw), L, P)
    if fault == 3:
        lib.pchip_set_capacity(-1, 300)
    lib.pchip_inject_fault(fault)
    r = api.Result()
    s = _settings(api, 2, 0, **kw)
    rc = lib.pchip_run(C.byref(s), C.byref(L), C.byref(P), C.byref(r))
    assert rc == code
    assert not r.dead and not r.logweights and not r.live          # nothing left to free
    err = capfd.readouterr().err
    assert "polychord_hip:" in err and ("memory" in err or "clusters" in err)
    lib.pchip_set_capacity(128, 0)
    again = api.run(_settings(api, 2, 0, **kw), L, P)
    assert again["ndead"] == good["ndead"] and again["logZ"] == good["logZ"]
    assert np.array_equal(again["dead"], good["dead"])


def test_python_front_door_raises_instead_of_exiting(engine, capacities, tmp_path):
    """through pypolychord.run_polychord: the reference's `stop 1` would take the interpreter with it; the binding asks
    the library to return (halt_returns) and raises RuntimeError with the engine's message"""
    from polychordlite_amd import pypolychord
    from polychordlite_amd.pypolychord.settings import PolyChordSettings
    from polychordlite_amd.pypolychord import device_likelihoods as dl
    st = PolyChordSettings(3, 0, nlive=60, num_repeats=6, base_dir=str(tmp_path), file_root="f", feedback=0,
                           write_resume=False, read_resume=False, maximise=False, write_prior=False, seed=3)
    capacities.pchip_inject_fault(1)
    with pytest.raises(RuntimeError, match="engine failure"):
        pypolychord.run_polychord(dl.Gaussian(), 3, 0, st, dl.UniformPrior(0.0, 1.0))
    out = pypolychord.run_polychord(dl.Gaussian(), 3, 0, st, dl.UniformPrior(0.0, 1.0))      # and the next run is fine
    assert abs(out.logZ) < 5 * out.logZerr + 1.0


@pytest.mark.gpu
def test_a_nursery_of_more_than_65535_chains_is_refused(engine):
    """the contraction kernels stamp the nursery position into 16 bits of the host mirror: a larger batch must be refused (code 8, a
    message), not truncated"""
    api = engine
    lib = api.load()
    s = api.Settings(); lib.pchip_settings_default(C.byref(s), 4, 0)
    s.nlive, s.num_repeats, s.batch = 200, 8, 70000
    L, P, keep = api.make_problem("gaussian", 4, 0)
    r = api.Result()
    assert lib.pchip_run(C.byref(s), C.byref(L), C.byref(P), C.byref(r)) == 8
    s.batch = 0
    assert api.run(s, L, P)["ndead"] > 0          # the process goes on
