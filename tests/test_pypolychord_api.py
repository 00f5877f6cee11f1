"""The Python drop-in (polychordlite_amd.pypolychord) mirrors the reference's own test file
tests/test_run_pypolychord.py: same call patterns (run_polychord / run), seed reproducibility,
with/without derived parameters, grade_dims ValueError, unknown-kwarg TypeError."""
import ctypes as C

import numpy as np
import pytest

from polychordlite_amd import pypolychord
from polychordlite_amd.pypolychord import device_likelihoods as dl

nDims, nlive = 4, 60


def gaussian_likelihood(theta):
    """reference tests/test_run_pypolychord.py:10-20 (4-D Gaussian, sigma 0.1, one derived parameter)"""
    sigma = 0.1
    n = theta.size
    r2 = float(np.sum(theta ** 2))
    logL = -np.log(2 * np.pi * sigma * sigma) * n / 2.0 - r2 / 2 / sigma / sigma
    return float(logL), [r2]


def uniform_prior(cube):
    return -1.0 + 2.0 * cube


# ---------------------------------------------------------------- CPU: surface / validation
def test_settings_defaults_and_errors():
    s = pypolychord.PolyChordSettings(nDims, 1)
    assert (s.nlive, s.num_repeats, s.do_clustering, s.equals, s.base_dir, s.file_root) == (100, 20, True, True, "chains", "test")
    assert abs(s.compression_factor - np.exp(-1)) < 1e-15 and s.cluster_dir.endswith("chains/clusters")
    with pytest.raises(TypeError):
        pypolychord.PolyChordSettings(nDims, 1, not_a_setting=3)
    with pytest.raises(ValueError):
        pypolychord.PolyChordSettings(nDims, 1, grade_dims=[1, 2])


def test_run_rejects_bad_arguments(tmp_path):
    with pytest.raises(ValueError):                      # reference test_grade_dims (tests/...:122-130)
        pypolychord.run(gaussian_likelihood, nDims, nDerived=1, grade_dims=[1, 2], base_dir=str(tmp_path))
    with pytest.raises(TypeError):                       # polychord.py:560-562
        pypolychord.run(gaussian_likelihood, nDims, bogus_keyword=1)
    with pytest.raises(TypeError):                       # _pypolychord.cpp:178-186
        from polychordlite_amd.pypolychord import _pypolychord
        _pypolychord.run(3, uniform_prior, lambda *a: None, *([0] * 34))
    assert (tmp_path / "clusters").is_dir()              # polychord.py:566-568


def test_compiled_extension_surface():
    """the package runs on the compiled CPython module `_pypolychord` (reference pypolychord/_pypolychord.cpp:119-228):
    one function `run`, 34 positional arguments in the reference's parse format, the reference's argument checks --
    all of them made before the engine is entered, so they hold without a GPU"""
    from polychordlite_amd.pypolychord import polychord as pc
    m = pc._pypolychord
    assert m.__file__.endswith(".so") and [n for n in dir(m) if not n.startswith("_")] == ["backend", "run"]
    base = [lambda t, p: 0.0, lambda c, t: None, lambda *a: None, 4, 0, 10, 4, -1, -1, 0, 0, 1e-3, -1e30, -1, 0.0,
            0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.36, 1, "chains", "t"]
    with pytest.raises(ValueError, match="sum to nDims"):            # _pypolychord.cpp:201-204
        m.run(*base, [1.0], [3], {}, 1)
    with pytest.raises(ValueError, match="same size"):               # :196-200
        m.run(*base, [1.0, 2.0], [4], {}, 1)
    with pytest.raises(TypeError):                                   # "O!" wants a list
        m.run(*base, (1.0,), [4], {}, 1)
    with pytest.raises(TypeError, match="list of integers"):         # :190-194
        m.run(*base, [1.0], [4.5], {}, 1)
    with pytest.raises(TypeError, match="dict mapping"):             # :205-209
        m.run(*base, [1.0], [4], {"a": 3}, 1)
    with pytest.raises(TypeError, match="callable"):
        m.run(*([3] + base[1:]), [1.0], [4], {}, 1)
    with pytest.raises(TypeError):                                   # wrong number of positional arguments
        m.run(*base)


def test_under_mpirun_only_rank_0_runs_the_engine(tmp_path, monkeypatch):
    """a script launched with mpirun (the reference's MPI farm, polychord.py:513-518): the GPU is the engine's
    parallelism, so rank 0 runs it and the other ranks wait at a barrier -- they must not start duplicate runs into the
    same base_dir.  mpi4py is simulated."""
    import sys
    import types
    from polychordlite_amd.pypolychord import polychord as pc
    calls, barriers = [], []

    def fake_comm(rank):
        comm = types.SimpleNamespace(Get_rank=lambda: rank, Get_size=lambda: 4, Barrier=lambda: barriers.append(rank),
                                     bcast=lambda obj, root=0: obj if rank == 0 else "done", scatter=None, gather=None)
        return types.SimpleNamespace(MPI=types.SimpleNamespace(COMM_WORLD=comm))
    monkeypatch.setattr(pc, "_pypolychord", types.SimpleNamespace(run=lambda *a: calls.append(a)))
    for rank in (1, 0):
        monkeypatch.setitem(sys.modules, "mpi4py", fake_comm(rank))
        base = tmp_path / f"r{rank}"
        pc.run(gaussian_likelihood, nDims, nDerived=1, base_dir=str(base), nlive=20, num_repeats=4, feedback=0)
        assert (base / "clusters").is_dir() == (rank == 0)              # polychord.py:566-568: rank 0 makes the directories
    assert len(calls) == 1 and barriers == [1, 0]
    # a failure on rank 0 is raised after the barrier
    def boom(*a):
        raise RuntimeError("engine")
    monkeypatch.setattr(pc, "_pypolychord", types.SimpleNamespace(run=boom))
    monkeypatch.setitem(sys.modules, "mpi4py", fake_comm(0))
    with pytest.raises(RuntimeError):
        pc.run(gaussian_likelihood, nDims, nDerived=1, base_dir=str(tmp_path / "x"), nlive=20, num_repeats=4, feedback=0)
    assert barriers[-1] == 0


def test_builtin_functors_are_plain_callables():
    g = dl.Gaussian(mu=0.5, sigma=0.1, nDerived=2)
    logL, phi = g(np.full(20, 0.5))
    assert abs(logL - 20 * (-np.log(0.1) - 0.5 * np.log(2 * np.pi))) < 1e-12 and phi[0] == 0.0
    p = dl.UniformPrior(-1.0, 1.0)
    assert np.allclose(p(np.array([0.0, 0.5, 1.0])), [-1.0, 0.0, 1.0])


# ---------------------------------------------------------------- GPU: runs
@pytest.mark.gpu
def test_run_python_callbacks_seed_reproducibility(engine, tmp_path):
    """reference test_seed: equal seeds -> identical runs, other seed -> different"""
    outs, calls = [], []
    for seed in (1, 1, 2):
        mine = []
        def dumper(live, dead, logweights, logZ, logZerr):
            mine.append((dead.copy(), logweights.copy(), logZ, logZerr, live.shape[0]))
        pypolychord.run(gaussian_likelihood, nDims, nDerived=1, prior=uniform_prior, dumper=dumper, nlive=nlive,
                        num_repeats=8, seed=seed, do_clustering=False, read_resume=False, write_resume=False,
                        base_dir=str(tmp_path), file_root=f"s{seed}", feedback=0)
        outs.append(mine[-1][:4]); calls.append(mine)
    # the dumper is called at every update and at the end (nested_sampling.F90:335,392)
    assert len(calls[0]) >= 5
    nds = [c[0].shape[0] for c in calls[0]]
    assert nds == sorted(nds) and nds[0] < nds[-1]
    assert all(c[4] == nlive for c in calls[0][:-1]) and calls[0][-1][4] == 0     # all live points die at the end
    assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][2] == outs[1][2]
    assert outs[0][2] != outs[2][2]
    dead, logw, logZ, logZerr = outs[0]
    assert dead.shape[1] == nDims + 1 + 2 and abs(np.logaddexp.reduce(logw)) < 1e-9
    # analytic evidence: unit-normalised Gaussian inside U(-1,1)^4 -> log Z = -4 log 2
    assert abs(logZ - (-4 * np.log(2))) < 4 * logZerr
    txt = (tmp_path / "s1.stats").read_text().splitlines()
    assert txt[8].startswith("log(Z)") and "+/-" in txt[8]           # output.py:57-99 parses by line number
    rows = np.loadtxt(tmp_path / "s1_dead-birth.txt")
    assert rows.shape == (dead.shape[0], nDims + 1 + 2) and np.allclose(rows[:, -2], dead[:, -1], rtol=1e-13)
    # posterior files (read_write.F90:479-617): weight, -2 logL, theta, phi
    post = np.loadtxt(tmp_path / "s1.txt"); eq = np.loadtxt(tmp_path / "s1_equal_weights.txt")
    assert post.shape[1] == 2 + nDims + 1 and eq.shape[1] == post.shape[1]
    assert post[:, 0].max() == 1.0 and np.all(eq[:, 0] == 1.0) and 10 < eq.shape[0] < post.shape[0]
    wmean = np.average(post[:, 2:2 + nDims], axis=0, weights=post[:, 0])
    assert np.all(np.abs(wmean) < 0.05)                                   # posterior is N(0, 0.1^2)
    assert np.all(np.abs(eq[:, 2:2 + nDims].mean(axis=0)) < 0.1)


@pytest.mark.gpu
def test_no_derived_gives_identical_samples(engine, tmp_path):
    """reference test_no_derived (tests/...:93-119)"""
    res = []
    for nder, like in ((1, gaussian_likelihood), (0, lambda th: gaussian_likelihood(th)[0])):
        got = []
        pypolychord.run(like, nDims, nDerived=nder, prior=uniform_prior, dumper=lambda l, d, w, z, e: got.append((d.copy(), z)),
                        nlive=nlive, num_repeats=8, seed=3, do_clustering=False, read_resume=False, write_resume=False,
                        write_dead=False, write_live=False, write_stats=False, base_dir=str(tmp_path), feedback=0)
        res.append(got[-1])
    assert res[0][1] == res[1][1]
    assert np.array_equal(res[0][0][:, :nDims], res[1][0][:, :nDims])


@pytest.mark.gpu
def test_run_polychord_legacy_and_device_functor(engine, tmp_path):
    """legacy interface + a likelihood fused on the device; a host callback computing the same
    Gaussian must reproduce the device run (same Philox streams, same decisions)"""
    D = 6
    s = pypolychord.PolyChordSettings(D, 0, nlive=80, num_repeats=12, seed=5, do_clustering=False, read_resume=False,
                                      write_resume=False, base_dir=str(tmp_path), file_root="dev", feedback=0)
    out_dev = pypolychord.run_polychord(dl.Gaussian(0.5, 0.1), D, 0, s, dl.UniformPrior(0.0, 1.0))
    s.file_root = "cb"
    lib = engine.load(); lib.polychord_hip_set_option(b"batch", 40.0)
    def host_gauss(theta):
        return float(-D * (np.log(0.1) + 0.5 * np.log(2 * np.pi)) - 0.5 * np.sum(((theta - 0.5) / 0.1) ** 2))
    out_cb = pypolychord.run_polychord(host_gauss, D, 0, s, lambda c: c.copy())
    lib.polychord_hip_set_option(b"batch", 0.0)
    assert out_dev.ndead == out_cb.ndead and out_dev.nlike == out_cb.nlike
    assert abs(out_dev.logZ - out_cb.logZ) < 1e-9
    assert abs(out_dev.logZ) < 4 * out_dev.logZerr


@pytest.mark.gpu
def test_python_exception_propagates(engine, tmp_path):
    def bad(theta):
        raise RuntimeError("boom")
    with pytest.raises(RuntimeError):
        pypolychord.run(bad, 3, nlive=20, num_repeats=3, base_dir=str(tmp_path), feedback=0, read_resume=False)


@pytest.mark.gpu
def test_ini_front_end_and_cli(engine, tmp_path):
    """configs/*.ini in the reference's ini grammar through polychord_c_interface_ini (CLI driver)"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "tools", "polychord_hip_cli")
    out = subprocess.run([cli, os.path.join(root, "configs", "gaussian.ini"), "gaussian"], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    stats = (tmp_path / "chains" / "gaussian.stats").read_text().splitlines()
    logZ, err = [float(x) for x in stats[8].split("=")[1].split("+/-")]
    assert abs(logZ) < 3.5 * err and 0.15 < err < 0.22                 # truth 0; reference sigma 0.186
    rows = np.loadtxt(tmp_path / "chains" / "gaussian_dead-birth.txt")
    assert rows.shape[1] == 20 + 2 + 2
    # 2-D Rastrigin with clustering (ini/rastrigin.ini of the reference): truth -2 ln 10.24 = -4.6526
    out = subprocess.run([cli, os.path.join(root, "configs", "rastrigin_2d.ini"), "rastrigin"], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    stats = (tmp_path / "chains" / "rastrigin_2d.stats").read_text().splitlines()
    logZ, err = [float(x) for x in stats[8].split("=")[1].split("+/-")]
    assert abs(logZ + 4.6526) < 4 * err + 0.05
    nclusters = sum(1 for l in stats if l.startswith("log(Z_"))
    assert nclusters >= 10                                            # the reference finds ~40 modes
    # parameter speeds in the ini file (priors.f90:708-737): two grades with explicit repeats; .stats has one likelihood
    # count per grade.  In the second file the fast parameters are listed first: the hypercube is ordered by speed, the
    # prior then runs on the host (the device prior maps cube coordinate i to parameter i)
    for name in ("gaussian_grades", "gaussian_grades_permuted"):
        out = subprocess.run([cli, os.path.join(root, "configs", name + ".ini"), "gaussian"], cwd=tmp_path, capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        stats = (tmp_path / "chains" / (name + ".stats")).read_text().splitlines()
        logZ, err = [float(x) for x in stats[8].split("=")[1].split("+/-")]
        assert abs(logZ) < 4 * err
        counts = [int(x) for x in [l for l in stats if l.startswith(" nlike:")][0].split(":")[1].split()]
        assert len(counts) == 2 and min(counts) > 1000
        rows = np.loadtxt(tmp_path / "chains" / (name + "_dead-birth.txt"))
        assert rows.shape[1] == 6 + 2 and abs(np.average(rows[-300:, :6]) - 0.5) < 0.05
    # a missing mandatory key is a fatal error with exit status 1 (abort.F90:19-29)
    bad = tmp_path / "bad.ini"; bad.write_text("num_repeats = 4\nP : a | a | 1 | uniform | 1 | 0 1\n")
    out = subprocess.run([cli, str(bad), "gaussian"], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode == 1 and "nlive" in out.stderr


@pytest.mark.gpu
def test_files_written_at_every_update(engine, tmp_path):
    """read_write.F90: dead files grow as points die, live/stats files are refreshed at every update;
    the final files agree with the run's own record (row counts, evidence, posterior table)."""
    import ctypes as C
    import os
    api = engine
    lib = api.load()
    base = str(tmp_path / "chains"); os.makedirs(base)
    seen = []

    DUMPER = C.CFUNCTYPE(None, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                         C.POINTER(C.c_double), C.c_double, C.c_double)

    def dumper(ndead, nlive, npars, live, dead, logw, logZ, logZerr):
        # what is on disk when the dumper runs is the previous update's state or newer
        p = os.path.join(base, "t_dead-birth.txt")
        rows = sum(1 for _ in open(p)) if os.path.exists(p) else 0
        stats = open(os.path.join(base, "t.stats")).read() if os.path.exists(os.path.join(base, "t.stats")) else ""
        seen.append((ndead, nlive, rows, "(Still Active)" in stats))

    cb = DUMPER(dumper)
    lib.polychord_c_interface.restype = None
    nD, nDer = 4, 1
    gauss = C.cast(lib.polychord_hip_gaussian, C.c_void_p)
    prior = C.cast(lib.polychord_hip_uniform_prior, C.c_void_p)
    lib.polychord_hip_set_gaussian(C.c_double(0.5), C.c_double(0.1))
    comm = C.c_int(0)
    args = [gauss, prior, C.cast(cb, C.c_void_p), C.c_int(200), C.c_int(8), C.c_int(-1), C.c_int(-1), C.c_bool(False), C.c_int(0),
            C.c_double(1e-3), C.c_double(-1e30), C.c_int(-1), C.c_double(0.0), C.c_bool(True), C.c_bool(True), C.c_bool(False),
            C.c_bool(False), C.c_bool(False), C.c_bool(False), C.c_bool(True), C.c_bool(True), C.c_bool(True), C.c_bool(False),
            C.c_bool(False), C.c_double(np.exp(-1)), C.c_bool(True), C.c_int(nD), C.c_int(nDer), C.c_char_p(base.encode()),
            C.c_char_p(b"t"), C.c_int(1), (C.c_double * 1)(1.0), (C.c_int * 1)(nD), C.c_int(0), None, None, C.c_int(3), C.byref(comm)]
    lib.polychord_c_interface(*args)
    assert len(seen) >= 3
    # dead rows on disk never exceed the dumper's count and grow monotonically; active cluster while running
    assert all(r <= nd for nd, _, r, _ in seen) and [r for *_, r, _ in seen] == sorted(r for *_, r, _ in seen)
    assert seen[1][2] > 0 and seen[1][3]
    ndead = seen[-1][0]
    db = np.loadtxt(os.path.join(base, "t_dead-birth.txt")); d = np.loadtxt(os.path.join(base, "t_dead.txt"))
    assert db.shape == (ndead, nD + nDer + 2) and d.shape == (ndead, nD + nDer + 1)
    assert np.allclose(db[:, nD + nDer], d[:, 0])                      # logL column in both layouts
    assert np.all(db[:, -1] <= db[:, -2] + 1e-12)                      # birth contour below logL
    stats = open(os.path.join(base, "t.stats")).read().splitlines()
    logZ, err = [float(x) for x in stats[8].split("=")[1].split("+/-")]
    assert abs(logZ) < 4 * err
    assert not any("(Still Active)" in l for l in stats)
    assert any(l.startswith(" ndead:") and int(l.split()[1]) == ndead for l in stats)
    k = stats.index("Dim No.       Mean        Sigma")
    mean1, sig1 = [float(x) for x in stats[k + 1][3:].split("+/-")]
    assert abs(mean1 - 0.5) < 0.03 and abs(sig1 - 0.1) < 0.03
    post = np.loadtxt(os.path.join(base, "t.txt"))
    assert post.shape[1] == 2 + nD + nDer and abs(post[:, 0].max() - 1.0) < 1e-12
    assert os.path.getsize(os.path.join(base, "t_phys_live.txt")) == 0   # every live point has been killed


@pytest.mark.gpu
def test_resume_and_cube_samples(engine, tmp_path):
    """write_resume / read_resume (read_write.F90:219-288, 384-476): a run restarted from the .resume file of an
    earlier update keeps the dead points of the first run and finishes with a consistent evidence; a finished
    run's file gives its result back; cube_samples start a run from user supplied live points."""
    import shutil
    from polychordlite_amd import pypolychord as pc
    from polychordlite_amd.pypolychord.device_likelihoods import Gaussian
    nD = 4
    base1, base2 = tmp_path / "a", tmp_path / "b"
    snap = {}

    def dumper(live, dead, logw, logZ, logZerr):
        f = base1 / "r.resume"
        if "n" not in snap and f.exists() and len(dead) > 400:
            (base2 / "clusters").mkdir(parents=True, exist_ok=True)
            shutil.copy(f, base2 / "r.resume")
            snap["n"] = int(open(f).read().splitlines()[5])        # dead points in that file
    kw = dict(nDerived=1, nlive=100, num_repeats=8, do_clustering=False, feedback=0, seed=5, file_root="r",
              write_resume=True, read_resume=True, posteriors=False, equals=False, write_live=False, write_prior=False)
    pc.run(Gaussian(mu=0.5, sigma=0.1), nD, base_dir=str(base1), dumper=dumper, **kw)
    assert "n" in snap and snap["n"] > 0
    full = np.loadtxt(base1 / "r_dead-birth.txt")
    # restart from the copied mid-run file
    pc.run(Gaussian(mu=0.5, sigma=0.1), nD, base_dir=str(base2), **kw)
    cont = np.loadtxt(base2 / "r_dead-birth.txt")
    assert cont.shape[0] > snap["n"] + 100
    assert np.array_equal(cont[:snap["n"]], full[:snap["n"]])       # the first run's dead points, as written
    # every row written after the restart is a real point: logL column = likelihood of its theta columns (the points
    # that were alive in the resume file die now; their rows must come from the restored live set)
    gl = -nD * (np.log(0.1) + 0.5 * np.log(2 * np.pi)) - 0.5 * np.sum(((cont[:, :nD] - 0.5) / 0.1) ** 2, axis=1)
    assert np.allclose(cont[:, nD + 1], gl, rtol=0, atol=1e-9)
    st = open(base2 / "r.stats").read().splitlines()
    logZ, err = [float(x) for x in st[8].split("=")[1].split("+/-")]
    assert abs(logZ) < 4 * err and 0.1 < err < 0.6                   # truth: logZ = 0
    # the file a finished run leaves behind (no live point): the same result comes back without sampling
    st1 = open(base1 / "r.stats").read().splitlines()
    nlike1 = [l for l in st1 if l.startswith(" nlike:")][0]
    pc.run(Gaussian(mu=0.5, sigma=0.1), nD, base_dir=str(base1), **kw)
    st1b = open(base1 / "r.stats").read().splitlines()
    va, vb = ([float(x) for x in l[8].split("=")[1].split("+/-")] for l in (st1, st1b))
    assert np.allclose(va, vb, rtol=0, atol=1e-12)                   # the file stores 15 significant digits
    assert [l for l in st1b if l.startswith(" nlike:")][0] == nlike1
    # cube_samples: user supplied live points (python callbacks, host evaluation)
    base3 = tmp_path / "c"
    cubes = np.random.default_rng(1).random((60, 2))
    pc.run(lambda th: -0.5 * float(np.sum(((th - 0.5) / 0.1) ** 2)), 2, cube_samples=cubes, base_dir=str(base3), file_root="cs",
           nlive=60, num_repeats=6, do_clustering=False, feedback=0, seed=2, write_resume=False, posteriors=False, equals=False)
    st3 = open(base3 / "cs.stats").read().splitlines()
    logZ3, err3 = [float(x) for x in st3[8].split("=")[1].split("+/-")]
    assert abs(logZ3 - np.log(2 * np.pi * 0.01)) < 4 * err3          # Z = integral of exp(-r^2/(2 s^2)) over the unit square


@pytest.mark.gpu
def test_resume_with_clusters_and_large_nlive(engine, tmp_path):
    """a restart from a .resume file written while several clusters were alive (per-cluster live points, phantoms,
    evidences, covariances); and a run whose live set does not fit the LDS-resident parallel contraction"""
    import shutil
    from polychordlite_amd import pypolychord as pc
    from polychordlite_amd.pypolychord.device_likelihoods import Rastrigin, Gaussian, UniformPrior
    base1, base2 = tmp_path / "a", tmp_path / "b"
    snap = {}

    def dumper(live, dead, logw, logZ, logZerr):
        f = base1 / "r.resume"
        if "nc" not in snap and f.exists():
            lines = open(f).read().splitlines()
            if int(lines[7]) >= 3:                                     # "=== Number of clusters ===" value
                (base2 / "clusters").mkdir(parents=True, exist_ok=True)
                shutil.copy(f, base2 / "r.resume")
                snap["nc"], snap["nd"] = int(lines[7]), int(lines[5])
    kw = dict(nDerived=0, nlive=300, num_repeats=6, do_clustering=True, feedback=0, seed=2, file_root="r",
              prior=UniformPrior(-5.12, 5.12), write_resume=True, read_resume=True, posteriors=False, equals=False,
              write_live=False, write_prior=False)
    pc.run(Rastrigin(), 2, base_dir=str(base1), dumper=dumper, **kw)
    assert snap.get("nc", 0) >= 3
    pc.run(Rastrigin(), 2, base_dir=str(base2), **kw)
    st = open(base2 / "r.stats").read().splitlines()
    logZ, err = [float(x) for x in st[8].split("=")[1].split("+/-")]
    assert abs(logZ + 4.6526) < 4 * err + 0.05                          # truth -2 ln 10.24
    assert sum(1 for l in st if l.startswith("log(Z_")) >= snap["nc"]
    rows2 = np.loadtxt(base2 / "r_dead-birth.txt")
    assert rows2.shape[0] > snap["nd"] + 200
    # every dead row is a real point: its logL column is the likelihood of its theta columns (the points that were alive
    # in the resume file die after the restart; their rows must come from the restored live set)
    th = rows2[:, :2]
    assert np.allclose(rows2[:, 2], -np.sum(np.log(4991.21750) + th ** 2 - 10.0 * np.cos(2 * np.pi * th), axis=1), rtol=0, atol=1e-9)
    assert np.all(rows2[:, 3] < rows2[:, 2])                            # born below where they died
    # nlive = 6000: serial single-wave contraction and kill-off (the parallel kernel's LDS budget is exceeded)
    base3 = tmp_path / "c"
    pc.run(Gaussian(mu=0.5, sigma=0.1), 3, base_dir=str(base3), file_root="big", nlive=6000, num_repeats=6, do_clustering=False,
           feedback=0, seed=4, write_resume=False, read_resume=False, posteriors=False, equals=False, write_live=False,
           write_prior=False, write_dead=False)
    st3 = open(base3 / "big.stats").read().splitlines()
    logZ3, err3 = [float(x) for x in st3[8].split("=")[1].split("+/-")]
    assert abs(logZ3) < 4 * err3 and err3 < 0.06


@pytest.mark.gpu
def test_fortran_binding(engine, tmp_path):
    """bindings/fortran: a Fortran program (ISO_C_BINDING) drives the engine through polychord_c_interface, with the
    device likelihood and with a likelihood written in Fortran (the reference's Fortran callers, interfaces.F90:10-12)"""
    import os
    import shutil
    import subprocess
    if shutil.which("amdflang") is None:
        pytest.skip("no Fortran compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "bindings", "fortran")
    lib = os.path.join(root, "polychordlite_amd")
    subprocess.check_call(["amdflang", "-c", os.path.join(src, "polychord_hip.f90"), "-o", "ph.o"], cwd=tmp_path)
    subprocess.check_call(["amdflang", os.path.join(src, "example_gaussian.f90"), "ph.o", "-L" + lib, "-lpolychord_hip",
                           "-Wl,-rpath," + lib, "-o", "ex"], cwd=tmp_path)
    (tmp_path / "chains").mkdir()
    out = subprocess.run(["./ex"], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    for rootname in ("f_device", "f_host"):
        st = (tmp_path / "chains" / (rootname + ".stats")).read_text().splitlines()
        logZ, err = [float(x) for x in st[8].split("=")[1].split("+/-")]
        assert abs(logZ) < 4 * err and err < 0.5, (rootname, logZ, err)       # truth 0
        rows = np.loadtxt(tmp_path / "chains" / (rootname + "_dead-birth.txt"))
        assert rows.shape[1] == 4 + 1 + 2


@pytest.mark.gpu
def test_cluster_posterior_files(engine, tmp_path):
    """clusters/<root>_<rank>.txt (read_write.F90:521-592): one file per cluster, ranked by evidence; the largest
    weight of file k is Z_k / Z; every cluster sits on one Rastrigin mode; the mixture of the cluster posteriors
    with weights Z_k / Z is the global posterior (points inherited through splits carry the evidence fractions)."""
    from polychordlite_amd import pypolychord as pc
    from polychordlite_amd.pypolychord.device_likelihoods import Rastrigin, UniformPrior
    base = tmp_path / "ch"
    pc.run(Rastrigin(), 2, base_dir=str(base), file_root="r", nlive=400, num_repeats=6, do_clustering=True, feedback=0, seed=7,
           prior=UniformPrior(-5.12, 5.12), posteriors=True, equals=True, cluster_posteriors=True, write_resume=False,
           read_resume=False, write_live=False, write_prior=False)
    st = open(base / "r.stats").read().splitlines()
    logZ = float(st[8].split("=")[1].split("+/-")[0])
    zk = sorted((float(l.split("=")[1].split("+/-")[0]) for l in st if l.startswith("log(Z_")), reverse=True)
    files = sorted((base / "clusters").glob("r_[0-9]*.txt"), key=lambda p: int(p.stem.split("_")[1]))
    files = [f for f in files if "equal" not in f.name]
    assert len(files) == len(zk) >= 10
    glob_post = np.loadtxt(base / "r.txt")
    gmean = (glob_post[:, 0:1] * glob_post[:, 2:4]).sum(0) / glob_post[:, 0].sum()
    mix = np.zeros(2); ftot = 0.0
    for k, f in enumerate(files):
        a = np.atleast_2d(np.loadtxt(f))
        if a.size == 0:
            continue
        frac = np.exp(zk[k] - logZ)
        assert abs(a[:, 0].max() - frac) < 1e-9 * max(1.0, frac)              # normalisation and ranking
        m = (a[:, 0:1] * a[:, 2:4]).sum(0) / a[:, 0].sum()
        if k < 8:                                                              # the heavy clusters: one mode each
            sd = np.sqrt((a[:, 0:1] * (a[:, 2:4] - m) ** 2).sum(0) / a[:, 0].sum())
            assert np.all(np.abs(m - np.round(m)) < 0.2) and np.all(sd < 0.45), (k, m, sd)
        mix += frac * m; ftot += frac
        eq = base / "clusters" / (f.stem + "_equal_weights.txt")
        assert eq.exists()
    assert abs(ftot - 1.0) < 0.1
    assert np.all(np.abs(mix / ftot - gmean) < 0.15)


@pytest.mark.gpu
def test_boost_posterior(engine, tmp_path):
    """boost_posterior (run_time_info.f90:845-870): phantoms that fall below the contour are kept as posterior samples
    with probability boost/num_repeats and the weight of the death that overtook them: more samples, same posterior."""
    from polychordlite_amd import pypolychord as pc
    from polychordlite_amd.pypolychord.device_likelihoods import Gaussian
    out = {}
    for boost in (0.0, 4.0):
        base = tmp_path / ("b%d" % int(boost))
        pc.run(Gaussian(mu=0.5, sigma=0.1), 4, base_dir=str(base), file_root="g", nDerived=1, nlive=200, num_repeats=8,
               do_clustering=False, feedback=0, seed=11, boost_posterior=boost, posteriors=True, equals=True,
               write_resume=False, read_resume=False, write_live=False, write_prior=False)
        post = np.loadtxt(base / "g.txt")
        w = post[:, 0]
        mean = (w[:, None] * post[:, 2:6]).sum(0) / w.sum()
        sd = np.sqrt((w[:, None] * (post[:, 2:6] - mean) ** 2).sum(0) / w.sum())
        ess = w.sum() ** 2 / (w ** 2).sum()
        out[boost] = (post.shape[0], mean, sd, ess)
        assert abs(w.max() - 1.0) < 1e-12
        assert np.all(np.abs(mean - 0.5) < 0.03) and np.all(np.abs(sd - 0.1) < 0.03), (boost, mean, sd)
    n0, n4 = out[0.0][0], out[4.0][0]
    assert n4 > 2.0 * n0                      # ~ (1 + boost) times the dead points
    assert out[4.0][3] > 1.5 * out[0.0][3]    # and the effective sample size grows with it


@pytest.mark.gpu
def test_grades_through_the_python_surface(engine, tmp_path):
    """grade_dims / grade_frac as pypolychord passes them (polychord.py:593-634): explicit repeats per grade
    (every grade_frac > 1) are deterministic -- a host callback reproduces the device run -- and .stats lists the
    likelihood calls per grade (read_write.F90:880-889); fractions <= 1 go through the speed timing
    (generate.F90:303-309, time_speeds)."""
    D = 6
    kw = dict(nlive=80, num_repeats=12, seed=5, do_clustering=False, read_resume=False, write_resume=False,
              base_dir=str(tmp_path), feedback=0, grade_dims=[2, 4], grade_frac=[6.0, 9.0])
    s = pypolychord.PolyChordSettings(D, 0, file_root="gdev", **kw)
    out_dev = pypolychord.run_polychord(dl.Gaussian(0.5, 0.1), D, 0, s, dl.UniformPrior(0.0, 1.0))
    s.file_root = "gcb"
    lib = engine.load(); lib.polychord_hip_set_option(b"batch", 40.0)
    def host_gauss(theta):
        return float(-D * (np.log(0.1) + 0.5 * np.log(2 * np.pi)) - 0.5 * np.sum(((theta - 0.5) / 0.1) ** 2))
    out_cb = pypolychord.run_polychord(host_gauss, D, 0, s, lambda c: c.copy())
    lib.polychord_hip_set_option(b"batch", 0.0)
    assert out_dev.ndead == out_cb.ndead and abs(out_dev.logZ - out_cb.logZ) < 1e-9
    assert abs(out_dev.logZ) < 4 * out_dev.logZerr
    for root in ("gdev", "gcb"):
        line = [l for l in (tmp_path / (root + ".stats")).read_text().splitlines() if l.startswith(" nlike:")][0]
        counts = [int(x) for x in line.split(":")[1].split()]
        assert len(counts) == 2 and all(c > 0 for c in counts)
    a = [l for l in (tmp_path / "gdev.stats").read_text().splitlines() if l.startswith(" nlike:")][0]
    b = [l for l in (tmp_path / "gcb.stats").read_text().splitlines() if l.startswith(" nlike:")][0]
    assert a == b
    # fractions: device likelihoods have equal speeds, so grade 2 gets nint(frac_2 / frac_1 * num_repeats) repeats
    s2 = pypolychord.PolyChordSettings(D, 0, file_root="gfrac", **dict(kw, grade_frac=[1.0, 0.5]))
    out = pypolychord.run_polychord(dl.Gaussian(0.5, 0.1), D, 0, s2, dl.UniformPrior(0.0, 1.0))
    assert abs(out.logZ) < 4 * out.logZerr
    line = [l for l in (tmp_path / "gfrac.stats").read_text().splitlines() if l.startswith(" nlike:")][0]
    c = [int(x) for x in line.split(":")[1].split()]
    assert len(c) == 2 and 0.3 < c[1] / c[0] < 0.8        # 12 repeats in 6-D vs 6 repeats in the 4-D fast subspace
    # a Python likelihood with fractions: the speeds are timed on the host, the run completes
    s3 = pypolychord.PolyChordSettings(D, 0, file_root="gtime", **dict(kw, grade_frac=[1.0, 1.0]))
    out = pypolychord.run_polychord(host_gauss, D, 0, s3, lambda c: c.copy())
    assert abs(out.logZ) < 4 * out.logZerr


@pytest.mark.gpu
def test_compiled_and_ctypes_bindings_agree(engine, tmp_path):
    """the compiled `_pypolychord` and the ctypes binding drive the same C entry point: identical dumps, files and
    exceptions -- Python likelihood with a derived parameter, Python prior, dynamic nlive dict, device functors"""
    from polychordlite_amd.pypolychord import _pypolychord as comp, _pypolychord_ctypes as ct
    def like(theta, phi):
        logL, p = gaussian_likelihood(theta)
        phi[0] = p[0]
        return logL
    def prior(cube, theta):
        theta[:] = -1.0 + 2.0 * cube
    res = {}
    for name, mod in (("comp", comp), ("ct", ct)):
        got = []
        mod.run(like, prior, lambda live, dead, w, z, e: got.append((dead.copy(), w.copy(), z, e)), nDims, 1, nlive, 8, -1, -1,
                False, 0, 1e-3, -1e30, -1, 0.0, True, True, False, False, False, False, True, False, True, False, False,
                0.36787944117144233, True, str(tmp_path), name, [1.0], [nDims], {-5.0: 40}, 11)
        res[name] = got
    assert len(res["comp"]) == len(res["ct"]) >= 3
    for a, b in zip(res["comp"], res["ct"]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2] and a[3] == b[3]
    assert (tmp_path / "comp.stats").read_text() == (tmp_path / "ct.stats").read_text()
    assert (tmp_path / "comp_dead-birth.txt").read_text() == (tmp_path / "ct_dead-birth.txt").read_text()
    # read-only views, float check and exception transport (_pypolychord.cpp:35,47-51,219-224)
    def nosy(theta, phi):
        assert not theta.flags.writeable and phi.flags.writeable
        return 1                                                     # not a float
    with pytest.raises(TypeError, match="must be a float"):
        comp.run(nosy, prior, lambda *a: None, 3, 0, 20, 3, -1, -1, False, 0, 1e-3, -1e30, -1, 0.0, False, False, False, False,
                 False, False, False, False, False, False, False, 0.36787944117144233, True, str(tmp_path), "bad", [1.0], [3], {}, 1)
    class Boom(Exception):
        pass
    def bad(theta, phi):
        raise Boom("boom")
    with pytest.raises(Boom):
        comp.run(bad, prior, lambda *a: None, 3, 0, 20, 3, -1, -1, False, 0, 1e-3, -1e30, -1, 0.0, False, False, False, False,
                 False, False, False, False, False, False, False, 0.36787944117144233, True, str(tmp_path), "bad", [1.0], [3], {}, 1)
    # device functors resolve to the library's own symbols in both bindings: same run as above tests use
    outs = []
    for mod, root in ((comp, "dcomp"), (ct, "dct")):
        got = []
        mod.run(dl.Gaussian(0.5, 0.1), dl.UniformPrior(0.0, 1.0), lambda live, dead, w, z, e: got.append(z), 6, 0, 80, 12, -1, -1,
                False, 0, 1e-3, -1e30, -1, 0.0, False, False, False, False, False, False, True, False, False, False, False,
                0.36787944117144233, True, str(tmp_path), root, [1.0], [6], {}, 5)
        outs.append(got[-1])
    assert outs[0] == outs[1] and abs(outs[0]) < 1.5


@pytest.mark.gpu
def test_maximise_writes_the_maximum_file(engine, tmp_path):
    """maximise=True through pypolychord.run (polychord.py:397-400): <root>.maximum in the reference's layout
    (read_write.F90:754-807), Python likelihood with a derived parameter and a Python prior on U(-1,1)^4"""
    pypolychord.run(gaussian_likelihood, nDims, nDerived=1, prior=uniform_prior, nlive=nlive, num_repeats=8, seed=3,
                    do_clustering=False, read_resume=False, write_resume=False, maximise=True, posteriors=True,
                    base_dir=str(tmp_path), file_root="mx", feedback=0)
    lines = (tmp_path / "mx.maximum").read_text().splitlines()
    assert lines[0] == "Maximum LogLikelihood:" and lines[5] == "Maximum Posterior:" and lines[12] == "LogLikelihood(mean):"
    peak = -np.log(2 * np.pi * 0.01) * nDims / 2.0
    assert abs(float(lines[1]) - peak) < 1e-3
    pt = np.array([float(x) for x in lines[3].split()])
    assert pt.size == nDims + 1 and np.all(np.abs(pt[:nDims]) < 5e-3)
    # U(-1,1)^4: dX/dtheta = 2^-4, the posterior density is the likelihood times it
    assert abs(float(lines[6]) - (float(lines[8]) - nDims * np.log(2.0))) < 1e-6
    assert abs(float(lines[13]) - peak) < 0.5                      # likelihood at the posterior mean


def test_priors_and_output_modules(tmp_path):
    """pypolychord.priors and pypolychord.output with the reference's names (priors.py:5-47, output.py:20-235)"""
    from polychordlite_amd.pypolychord import priors, PolyChordOutput
    from polychordlite_amd.pypolychord.output import PolyChordOutput as PO2
    assert PolyChordOutput is PO2
    x = np.array([0.2, 0.7, 0.45])
    assert np.allclose(priors.UniformPrior(-1, 2)(x), -1 + 3 * x)
    assert np.allclose(priors.LogUniformPrior(1e-3, 1.0)(x), 1e-3 * 1e3 ** x)
    g = priors.GaussianPrior(2.0, 0.5)(x)
    assert np.allclose(g, [2 + 0.5 * -0.8416212335729143, 2 + 0.5 * 0.5244005127080407, 2 + 0.5 * -0.12566134685507402])
    t = priors.forced_indentifiability_transform(x)
    assert np.all(np.diff(t) > 0) and np.allclose(t, [0.12822809400794053, 0.6411404700397025, 0.7663094323935531])   # = reference's sort_hypercube
    assert np.allclose(priors.SortedUniformPrior(0, 10)(x), 10 * t) and np.allclose(priors.LogSortedUniformPrior(1, 100)(x), 100 ** t)
    # output object: the reference's .stats layout (read_write.F90:809-910), two clusters, two grades, posterior table
    (tmp_path / "clusters").mkdir()
    (tmp_path / "t.stats").write_text("""Evidence estimates:
===================
  - The evidence Z is a log-normally distributed, with location and scale parameters mu and sigma.
  - We denote this as log(Z) = mu +/- sigma.

Global evidence:
----------------

log(Z)       =  -0.334324000000000E+001 +/-   0.257810000000000E+000


Local evidences:
----------------

log(Z_1)     =  -0.400000000000000E+001 +/-   0.300000000000000E+000 (Still Active)
log(Z_2)     =  -0.410000000000000E+001 +/-   0.310000000000000E+000


Run-time information:
---------------------

 ncluster:          1 /       2
 nposterior:      321
 nequals:          45
 ndead:          1358
 nlive:             0
 nlike:        45080   22300
 <nlike>:       12.00    3.00   (    1.50    0.75 per slice )


Dim No.       Mean        Sigma
  1  0.500000000000000E+000 +/-   0.100000000000000E+000
  2  0.400000000000000E+000 +/-   0.200000000000000E+000
-------------------------------
  3  0.100000000000000E+001 +/-   0.300000000000000E+000
""")
    np.savetxt(tmp_path / "t_equal_weights.txt", np.array([[1.0, 2.0, 0.5, 0.4, 1.0], [1.0, 4.0, 0.6, 0.3, 1.1]]))
    o = PolyChordOutput(str(tmp_path), "t")
    assert (o.logZ, o.logZerr) == (-3.34324, 0.25781) and o.logZs == [-4.0, -4.1] and o.logZerrs == [0.3, 0.31]
    assert (o.ncluster, o.nposterior, o.nequals, o.ndead, o.nlive, o.nlike) == (2, 321, 45, 1358, 0, 45080)
    assert o.avnlike == [12.0, 3.0] and o.avnlikeslice == [1.5, 0.75]
    assert o.means == [0.5, 0.4, 1.0] and o.sigmas == [0.1, 0.2, 0.3]
    assert o.root == str(tmp_path / "t") and o.cluster_root(2).endswith("clusters/t_2") and o.paramnames_file.endswith("t.paramnames")
    o.make_paramnames_files([("a", "a"), ("b", "b"), ("r*", "r")])
    assert (tmp_path / "t.paramnames").read_text().splitlines() == ["a   a", "b   b", "r*   r"]
    assert (tmp_path / "clusters" / "t_1.paramnames").exists()
    if o.pandas:
        assert list(o.samples.columns) == ["weight", "loglike", "a", "b", "r*"] and np.allclose(o.loglikes, [-1.0, -2.0])
        assert "Global evidence" in str(o)


@pytest.mark.gpu
def test_cpp_facade(engine, tmp_path):
    """include/polychord_hip.hpp: the reference's C++ `Settings` / `run_polychord` surface (interfaces.hpp:8-87, defaults of
    c_interface.cpp:6-39) -- a driver in the style of src/drivers/polychord_CC.cpp, likelihood fused on the device and
    likelihood written in C++; maximise defaults to true there, so <root>.maximum appears"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "polychordlite_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(root, "include"), os.path.join(root, "bindings", "cpp", "example_gaussian.cpp"),
                           "-L" + lib, "-lpolychord_hip", "-Wl,-rpath," + lib, "-o", "ex"], cwd=tmp_path)
    (tmp_path / "chains" / "clusters").mkdir(parents=True)
    out = subprocess.run(["./ex", "chains"], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "final dump" in out.stdout
    vals = {}
    for rootname in ("cpp_device", "cpp_host"):
        st = (tmp_path / "chains" / (rootname + ".stats")).read_text().splitlines()
        logZ, err = [float(x) for x in st[8].split("=")[1].split("+/-")]
        assert abs(logZ) < 4 * err and err < 0.5, (rootname, logZ, err)       # truth 0
        mx = (tmp_path / "chains" / (rootname + ".maximum")).read_text().splitlines()
        assert abs(float(mx[1]) - (-6 * (np.log(0.1) + 0.5 * np.log(2 * np.pi)))) < 1e-3
        vals[rootname] = logZ
    # a compiled callback this cheap gets the device's chains per nursery (nlive / 2): same draws, same decisions
    assert abs(vals["cpp_device"] - vals["cpp_host"]) < 1e-9


@pytest.mark.gpu
def test_output_files_equal_the_reference_binary(engine, golden, tmp_path, monkeypatch):
    """The file writers (read_write.F90:479-910 restated in pc_abi.hip) against files the REFERENCE BINARY wrote
    (tests/golden/ref_files/, oracle/gen_golden.py): in sequential-stream mode the engine walks the reference's run, so
    <root>_dead-birth.txt must be the same rows in the same order and <root>.stats the same text, line for line
    (evidences to the last printed digit or one unit of it)."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.setenv("PC_SEQUENTIAL_RNG", "1")
    likes = {"gaussian": (dl.Gaussian(0.5, 0.1, nDerived=1), dl.UniformPrior(0.0, 1.0)), "rastrigin": (dl.Rastrigin(), dl.UniformPrior(-5.12, 5.12))}
    for c in golden["ref_files"]:
        like, prior = likes[c["like"]]
        s = pypolychord.PolyChordSettings(c["nDims"], c["nDerived"], nlive=c["nlive"], num_repeats=c["num_repeats"], seed=c["seed"],
                                          do_clustering=bool(c["clustering"]), read_resume=False, write_resume=False, write_dead=True,
                                          write_stats=True, posteriors=False, equals=False, write_prior=False, write_live=False,
                                          base_dir=str(tmp_path), file_root=c["name"], feedback=0)
        pypolychord.run_polychord(like, c["nDims"], c["nDerived"], s, prior)
        ref = np.loadtxt(os.path.join(root, "tests", "golden", "ref_files", c["name"] + "_dead-birth.txt"))
        got = np.loadtxt(tmp_path / (c["name"] + "_dead-birth.txt"))
        assert got.shape == ref.shape == (c["ndead"], c["nDims"] + c["nDerived"] + 2)
        assert np.abs(got - ref).max() < 1e-9 * max(1.0, np.abs(ref[:, :-1]).max()), np.abs(got - ref).max()   # 15 printed digits; fp64 re-association
        ref_lines = open(os.path.join(root, "tests", "golden", "ref_files", c["name"] + ".stats")).read().splitlines()
        got_lines = (tmp_path / (c["name"] + ".stats")).read_text().splitlines()
        assert len(got_lines) == len(ref_lines)
        for a, b in zip(got_lines, ref_lines):
            if a == b:
                continue
            fa, fb = a.replace("+/-", " ").replace("=", " ").split(), b.replace("+/-", " ").replace("=", " ").split()
            assert len(fa) == len(fb), (a, b)
            for x, y in zip(fa, fb):                     # same labels, numbers equal to round-off of the last digit
                try:
                    assert abs(float(x) - float(y)) <= 1e-9 * max(1.0, abs(float(y))), (a, b)
                except ValueError:
                    assert x == y, (a, b)


@pytest.mark.gpu
def test_vectorised_and_farmed_likelihoods(engine, tmp_path, monkeypatch):
    """polychord_hip_set_batch_callback through the Python layer: a likelihood marked `vectorised` gets all the parked
    proposals of a round as one (n, nDims) array, and under mpirun the rows are farmed out to the ranks (mpi4py simulated:
    the 'other ranks' evaluate their shares in this process).  Same draws, same decisions: the runs equal the scalar one."""
    import sys
    import types
    from polychordlite_amd.pypolychord import polychord as pc
    lib = engine.load(); lib.polychord_hip_set_option(b"batch", 32.0)
    kw = dict(nDerived=1, nlive=80, num_repeats=8, seed=7, do_clustering=False, read_resume=False, write_resume=False,
              write_dead=False, write_live=False, write_stats=True, posteriors=False, equals=False, write_prior=False, feedback=0)
    shapes = []

    def vec_like(theta):                                   # theta: (n, nDims)
        shapes.append(theta.shape)
        r2 = np.sum(theta ** 2, axis=1)
        return -np.log(2 * np.pi * 0.01) * theta.shape[1] / 2.0 - r2 / 0.02, r2[:, None]
    vec_like.vectorised = True

    def vec_prior(cube):
        return -1.0 + 2.0 * cube
    vec_prior.vectorised = True
    try:
        out = {}
        dumps = {}
        for name, like, prior in (("scalar", gaussian_likelihood, uniform_prior), ("vec", vec_like, vec_prior), ("mixed", vec_like, uniform_prior)):
            got = []
            pypolychord.run(like, nDims, prior=prior, dumper=lambda l, d, w, z, e: got.append((d.copy(), z)), base_dir=str(tmp_path), file_root=name, **kw)
            dumps[name] = got[-1]
        assert max(s[0] for s in shapes) > 8 and all(s[1] == nDims for s in shapes)          # real blocks of rows arrived
        for name in ("vec", "mixed"):
            # (NumPy sums a row of a 2-D array and a 1-D array in different orders: last-bit differences in logL)
            assert abs(dumps[name][1] - dumps["scalar"][1]) < 1e-10 and np.allclose(dumps[name][0], dumps["scalar"][0], rtol=1e-12, atol=1e-12)
        # simulated MPI farm: 3 ranks; scatter hands rank 0 its share and evaluates the others' shares on the spot
        calls = {"n": 0}

        class Comm:
            def Get_rank(self): return 0
            def Get_size(self): return 3
            def Barrier(self): pass
            def bcast(self, obj, root=0): return obj
            def scatter(self, parts, root=0):
                self.parts = parts
                return parts[0]
            def gather(self, mine, root=0):
                calls["n"] += 1
                return [mine] + [self.ev.local(p) for p in self.parts[1:]]
        comm = Comm()
        monkeypatch.setitem(sys.modules, "mpi4py", types.SimpleNamespace(MPI=types.SimpleNamespace(COMM_WORLD=comm)))
        orig = pc._BatchEvaluator

        class Spy(orig):
            def __init__(self, *a):
                super().__init__(*a)
                comm.ev = self
        monkeypatch.setattr(pc, "_BatchEvaluator", Spy)
        got = []
        pypolychord.run(gaussian_likelihood, nDims, prior=uniform_prior, dumper=lambda l, d, w, z, e: got.append((d.copy(), z)),
                        base_dir=str(tmp_path), file_root="farm", **kw)
        assert calls["n"] > 10
        assert got[-1][1] == dumps["scalar"][1] and np.array_equal(got[-1][0], dumps["scalar"][0])
        # an exception inside a vectorised likelihood comes back out of run()
        def bad(theta):
            raise KeyError("vectorised boom")
        bad.vectorised = True
        monkeypatch.delitem(sys.modules, "mpi4py")
        monkeypatch.setattr(pc, "_BatchEvaluator", orig)
        with pytest.raises(KeyError):
            pypolychord.run(bad, nDims, base_dir=str(tmp_path), file_root="bad", **dict(kw, nDerived=0))
    finally:
        lib.polychord_hip_set_option(b"batch", 0.0)


@pytest.mark.gpu
def test_feedback_levels_print_like_the_reference(engine, tmp_path, capfd):
    """feedback.f90: level 0 prints the final box, level 1 adds a progress block at every update (lives per
    cluster, counters, evidence) -- and changes nothing about the run"""
    D = 4
    def run(fb, root):
        s = pypolychord.PolyChordSettings(D, 0, nlive=60, num_repeats=8, seed=9, do_clustering=True, read_resume=False,
                                          write_resume=False, base_dir=str(tmp_path), file_root=root, feedback=fb)
        return pypolychord.run_polychord(dl.TwinGaussian(0.05), D, 0, s, dl.UniformPrior(-1.0, 1.0))
    capfd.readouterr()
    quiet = run(-1, "fbq")
    assert capfd.readouterr().out.strip() == ""
    box = run(0, "fb0")
    out0 = capfd.readouterr().out
    assert "| ndead  = %12d" % box.ndead in out0 and "| log(Z) =" in out0 and "lives      |" not in out0
    loud = run(1, "fb1")
    out1 = capfd.readouterr().out
    assert "started sampling" in out1 and out1.count("lives      |") >= 3 and out1.count("log(Z)     =") >= 3
    assert "ncluster   =" in out1 and "<nlike>    =" in out1 and "per slice )" in out1
    # the last progress block reports the counters of the last update: below the final ones, above zero
    nd = [int(l.split("=")[1]) for l in out1.splitlines() if l.startswith("ndead      =")]
    assert nd == sorted(nd) and 0 < nd[-1] <= loud.ndead
    assert (quiet.ndead, quiet.nlike, quiet.logZ) == (box.ndead, box.nlike, box.logZ) == (loud.ndead, loud.nlike, loud.logZ)


@pytest.mark.gpu
def test_weighted_posterior_file_equals_the_reference_binary(engine, golden, tmp_path, monkeypatch):
    """R13, engine side: with posteriors = T the reference draws one uniform for every phantom its clean_phantoms removes
    (run_time_info.f90:857-859) -- the engine's sequential-stream mode accounts for them, so it still walks the reference
    binary's run, and its <root>.txt (weight, -2 logL, theta, phi of every dead point: update_posteriors :1050-1054,
    write_posterior_file read_write.F90:560-566) must be the file the reference wrote (tests/golden/ref_files/pgp.txt)."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.setenv("PC_SEQUENTIAL_RNG", "1")
    c = [x for x in golden["ref_posteriors"] if x["name"] == "pgp"][0]
    s = pypolychord.PolyChordSettings(c["nDims"], c["nDerived"], nlive=c["nlive"], num_repeats=c["num_repeats"], seed=c["seed"],
                                      do_clustering=False, read_resume=False, write_resume=False, write_dead=True, write_stats=True,
                                      posteriors=True, equals=False, write_prior=False, write_live=False, base_dir=str(tmp_path),
                                      file_root="pgp", feedback=0)
    out = pypolychord.run_polychord(dl.Gaussian(0.5, 0.1, nDerived=1), c["nDims"], c["nDerived"], s, dl.UniformPrior(0.0, 1.0))
    assert out.ndead == c["ndead"] and out.nposterior == c["nposterior"] and out.nlike == c["nlike"]
    assert abs(out.logZ - c["logZ"]) < 1e-9 and abs(out.logZerr - c["logZerr"]) < 1e-9
    ref = np.loadtxt(os.path.join(root, "tests", "golden", "ref_files", "pgp.txt"))
    got = np.loadtxt(tmp_path / "pgp.txt")
    assert got.shape == ref.shape
    assert (np.abs(got - ref) / np.maximum(1e-300, np.abs(ref))).max() < 1e-8


@pytest.mark.gpu
@pytest.mark.parametrize("kind,D,nDer,nlive,nr,B,boost,clus", [("gaussian", 4, 1, 150, 8, 32, 0.0, 0), ("gaussian", 4, 1, 150, 8, 32, 3.0, 0),
                                                              ("rastrigin", 2, 0, 200, 6, 25, 2.0, 1), ("rastrigin", 2, 0, 200, 6, 25, 0.0, 1),
                                                              # BASELINE configs[0] (ini/gaussian.ini) at its size, 250 chains per nursery
                                                              ("gaussian", 20, 2, 500, 40, 250, 0.0, 0)],
                         ids=["gauss4", "gauss4-boost", "rastrigin2-clusters-boost", "rastrigin2-clusters", "C1-gaussian20-nlive500"])
def test_posterior_samples_match_the_oracle(engine, tmp_path, kind, D, nDer, nlive, nr, B, boost, clus):
    """R13 in production mode (keyed streams, B chains per nursery) against the oracle -- whose posterior machinery the
    reference binary pins (tests/test_oracle_pinned.py): the weighted posterior <root>.txt holds the same points with the
    same weights, the phantoms kept by boost_posterior included (their Bernoulli trials are keyed by the phantom's id in
    both); the equally weighted file is thinned incrementally like the reference's (run_time_info.f90:975-1026: at every update, and
    whenever a cluster has lost its last point -- delete_cluster calls update_posteriors, :533 --, survivors are re-drawn against the
    ratio of successive maxima and move up, newcomers are drawn against the current one).  The trial of posterior row r in round k is
    the draw keyed by (k, r) in the engine and in the oracle's keyed mode (oracle/pc_oracle.c bernoulli_post), so the SAME rows survive
    whatever order the lists are walked in: <root>_equal_weights.txt holds the oracle's rows, row for row once both are sorted -- one
    cluster, with phantoms in the stack (boost_posterior), several clusters, and BASELINE configs[0] at its size.  Several clusters AND
    boost_posterior: the reference's clean_phantoms also runs in the rounds at a cluster's end, the engine's at updates only -- there the
    file's size is compared with its expectation."""
    from tests import oracle_api as orc
    lib = engine.load(); lib.polychord_hip_set_option(b"batch", float(B))
    try:
        lo, hi = (-5.12, 5.12) if kind == "rastrigin" else (0.0, 1.0)
        like = dl.Rastrigin() if kind == "rastrigin" else dl.Gaussian(0.5, 0.1, nDerived=nDer)
        s = pypolychord.PolyChordSettings(D, nDer, nlive=nlive, num_repeats=nr, seed=17, do_clustering=bool(clus), read_resume=False,
                                          write_resume=False, write_dead=False, write_stats=True, posteriors=True, equals=True,
                                          boost_posterior=boost, write_prior=False, write_live=False, base_dir=str(tmp_path),
                                          file_root="k", feedback=0, cluster_posteriors=False)
        out = pypolychord.run_polychord(like, D, nDer, s, dl.UniformPrior(lo, hi))
    finally:
        lib.polychord_hip_set_option(b"batch", 0.0)
    so = orc.settings(D, nDer, nlive=nlive, num_repeats=nr, seed=17, batch=B, do_clustering=clus, posteriors=1, equals=1, boost_posterior=boost)
    Lo, Po, keep = orc.make_problem(kind, D, None if kind == "gaussian" else lo, None if kind == "gaussian" else hi)
    o = orc.run(so, Lo, Po)
    assert out.ndead == o["ndead"] and abs(out.logZ - o["logZ"]) < 1e-8
    w = np.exp(o["post_rows"][:, 0] - o["maxlogweight"])
    ref = np.column_stack([w, -2 * o["post_rows"][:, 1], o["post_rows"][:, 2:]])[w > 0]
    got = np.loadtxt(tmp_path / "k.txt")
    assert got.shape == ref.shape, (got.shape, ref.shape)
    key = lambda a: a[np.lexsort(a.T[::-1])]                     # the engine lists the kept phantoms after the dead points
    assert (np.abs(key(got) - key(ref)) / np.maximum(1e-300, np.abs(key(ref)))).max() < 1e-7
    assert out.nposterior == ref.shape[0]
    eq = np.loadtxt(tmp_path / "k_equal_weights.txt")
    expect, var = got[:, 0].sum(), (got[:, 0] * (1 - got[:, 0])).sum()
    assert abs(eq.shape[0] - expect) < 5 * np.sqrt(var) + 1, (eq.shape[0], expect)
    assert abs(o["nequals"] - expect) < 5 * np.sqrt(var) + 1                       # and so is the oracle's (the reference's)
    assert np.all(eq[:, 0] == 1.0)
    if not (boost != 0.0 and clus):
        ref_eq = o["equal_rows"]                                   # [-2 logL, theta, phi] in the order of RTI%equals_global
        assert eq.shape[0] == o["nequals"] == ref_eq.shape[0], (eq.shape[0], o["nequals"])
        assert (np.abs(key(eq[:, 1:]) - key(ref_eq)) / np.maximum(1e-300, np.abs(key(ref_eq)))).max() < 1e-7
    if clus:
        assert o["ncluster_dead"] > 2                              # clusters ended on the way: rounds between the updates


@pytest.mark.gpu
@pytest.mark.parametrize("discard", [0, 1])
def test_epoch_rule_through_the_front_door(engine, tmp_path, discard, capfd):
    """polychord_hip_set_option("epoch_discard", 1) gives polychord_c_interface callers the reference farm's rule for chains in flight
    when the list of clusters changes (nested_sampling.F90:313); the default keeps the chains of the clusters the change left alone.
    Either way the run pypolychord.run_polychord makes is the oracle's run under that rule."""
    from tests import oracle_api as orc
    lib = engine.load(); lib.polychord_hip_set_option(b"batch", 120.0); lib.polychord_hip_set_option(b"epoch_discard", float(discard))
    D, nlive, nr = 3, 300, 9
    try:
        s = pypolychord.PolyChordSettings(D, 0, nlive=nlive, num_repeats=nr, seed=23, do_clustering=True, read_resume=False,
                                          write_resume=False, write_dead=False, write_stats=True, posteriors=False, equals=False,
                                          write_prior=False, write_live=False, base_dir=str(tmp_path), file_root="e", feedback=0)
        capfd.readouterr()
        s.feedback = 0
        out = pypolychord.run_polychord(dl.Rastrigin(), D, 0, s, dl.UniformPrior(-5.12, 5.12))
        box = capfd.readouterr().out
    finally:
        lib.polychord_hip_set_option(b"batch", 0.0); lib.polychord_hip_set_option(b"epoch_discard", 0.0)
    # a run that held several clusters says, behind the reference's final box, what its error does not contain and which rule it followed
    # for chains in flight (the one engine-specific sampling rule at the drop-in boundary: pchip_settings.epoch_discard)
    assert "| log(Z) =" in box and "clusters were alive at once" in box and "nested_sampling.F90:313" in box
    assert ("THIS ENGINE's rule" in box) == (discard == 0) and ("the reference farm's rule" in box) == (discard == 1)
    so = orc.settings(D, 0, nlive=nlive, num_repeats=nr, seed=23, batch=120, do_clustering=1, epoch_discard=discard)
    Lo, Po, keep = orc.make_problem("rastrigin", D, -5.12, 5.12)
    o = orc.run(so, Lo, Po)
    assert o["ncluster"] + o["ncluster_dead"] > 3
    assert out.ndead == o["ndead"] and abs(out.logZ - o["logZ"]) < 1e-8 and abs(out.logZerr - o["logZerr"]) < 1e-8
    # a caller of the reference's interface can see which rule the run followed: the last line of <root>.stats (behind everything the
    # reference's readers parse)
    assert ("epoch_discard = %d" % discard) in open(tmp_path / "e.stats").read().splitlines()[-1]


@pytest.mark.gpu
def test_resumed_run_keeps_its_posterior_bookkeeping(engine, tmp_path):
    """a clustered run restarted from a .resume file written while it was under way (a copy taken from the dumper): the
    cluster every earlier dead point died in, the genealogy of the splits and the phantoms kept by boost_posterior travel
    in the sidecar <root>.resume.hip, so the per-cluster posterior files of the finished run hold the points from before
    the interruption too (every weighted dead point is in the file of its cluster and of all that cluster's
    descendants), and the resume file carries the reference's thin factor boost / num_repeats (generate.F90:311-316)."""
    import shutil
    from polychordlite_amd import pypolychord as pc
    from polychordlite_amd.pypolychord.device_likelihoods import Rastrigin, UniformPrior
    base1, base = tmp_path / "a", tmp_path / "ch"
    snap = {}

    def dumper(live, dead, logw, logZ, logZerr):
        f = base1 / "r.resume"
        if "nd" not in snap and f.exists() and (base1 / "r.resume.hip").exists():
            lines = open(f).read().splitlines()
            if int(lines[7]) >= 3 and int(lines[5]) >= 1200:           # clusters alive, dead points so far
                (base / "clusters").mkdir(parents=True, exist_ok=True)
                shutil.copy(f, base / "r.resume"); shutil.copy(base1 / "r.resume.hip", base / "r.resume.hip")
                snap["nd"] = int(lines[5])
    kw = dict(file_root="r", nlive=300, num_repeats=6, do_clustering=True, feedback=0, seed=9,
              prior=UniformPrior(-5.12, 5.12), posteriors=True, equals=True, cluster_posteriors=True, boost_posterior=2.0,
              write_resume=True, read_resume=True, write_live=False, write_prior=False)
    pc.run(Rastrigin(), 2, base_dir=str(base1), dumper=dumper, **kw)
    assert snap.get("nd", 0) >= 1200
    lines = (base / "r.resume").read_text().splitlines()
    k = lines.index("=== posterior thin factor ===")
    assert abs(float(lines[k + 1]) - 2.0 / 6.0) < 1e-12
    pc.run(Rastrigin(), 2, base_dir=str(base), **kw)
    st = open(base / "r.stats").read().splitlines()
    logZ = float(st[8].split("=")[1].split("+/-")[0])
    zk = sorted((float(l.split("=")[1].split("+/-")[0]) for l in st if l.startswith("log(Z_")), reverse=True)
    files = sorted((f for f in (base / "clusters").glob("r_[0-9]*.txt") if "equal" not in f.name), key=lambda p: int(p.stem.split("_")[1]))
    assert len(files) == len(zk) >= 8
    glob_post = np.loadtxt(base / "r.txt")
    ndead_file = np.loadtxt(base / "r_dead-birth.txt").shape[0]
    assert glob_post.shape[0] > ndead_file                           # dead points of both legs + the phantoms boost_posterior kept
    rows = 0; mix = np.zeros(2); ftot = 0.0
    for kf, f in enumerate(files):
        a = np.atleast_2d(np.loadtxt(f))
        if a.size == 0:
            continue
        rows += a.shape[0]
        frac = np.exp(zk[kf] - logZ)
        assert abs(a[:, 0].max() - frac) < 1e-9 * max(1.0, frac)
        mix += frac * (a[:, 0:1] * a[:, 2:4]).sum(0) / a[:, 0].sum(); ftot += frac
    assert rows >= glob_post.shape[0]                                # nobody is missing (inherited points are counted more than once)
    gmean = (glob_post[:, 0:1] * glob_post[:, 2:4]).sum(0) / glob_post[:, 0].sum()
    assert abs(ftot - 1.0) < 0.1 and np.all(np.abs(mix / ftot - gmean) < 0.15)
