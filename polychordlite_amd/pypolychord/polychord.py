"""`run` / `run_polychord` -- same signatures, defaults and errors as reference
pypolychord/polychord.py (run_polychord :16-215, run :221-646)."""
import warnings
from pathlib import Path

import numpy as np

try:                                    # compiled CPython extension (pypolychord/_pypolychord_module.cpp)
    from . import _pypolychord
except ImportError:                     # not built: the ctypes binding of the same C entry point
    from . import _pypolychord_ctypes as _pypolychord
from .settings import PolyChordSettings  # noqa: F401


def _mpi_comm():
    """COMM_WORLD when the script runs under mpirun with mpi4py (the reference's MPI farm: polychord.py:513-518,
    nested_sampling.F90:262-301), else None.  The engine's parallelism is the GPU: rank 0 runs it, the other ranks wait
    for it and return the same files' results -- they must not start duplicate runs into the same base_dir."""
    try:
        from mpi4py import MPI
    except ImportError:
        return None
    comm = MPI.COMM_WORLD
    return comm if comm.Get_size() > 1 else None


class _BatchEvaluator:
    """Evaluation of all parked proposals of a round in one go (polychord_hip_set_batch_callback).

    * `loglikelihood.vectorised = True` (and optionally `prior.vectorised = True`): the callable takes theta of shape
      (n, nDims) and returns logL of shape (n,), or (logL, phi) with phi of shape (n, nDerived) -- NumPy-vectorised
      likelihoods, or likelihoods that run on an accelerator themselves.
    * under mpirun with mpi4py: the rows are scattered over the ranks, every rank evaluates its share with its own copy of
      the callables, rank 0 gathers -- the parallel likelihood evaluation the reference gets from its MPI workers
      (nested_sampling.F90:426-498).
    Otherwise no batch callback is registered and the engine calls the scalar callbacks."""

    def __init__(self, loglikelihood, prior, nDims, nDerived, comm, logzero):
        self.like, self.prior, self.nDims, self.nDerived, self.comm, self.logzero = loglikelihood, prior, nDims, nDerived, comm, logzero
        self.vec_like = bool(getattr(loglikelihood, "vectorised", False))
        self.vec_prior = bool(getattr(prior, "vectorised", False))
        builtin = getattr(loglikelihood, "symbol", None) is not None       # runs inside the kernel: nothing to batch
        self.active = not builtin and (self.vec_like or comm is not None)
        self.error = None
        self._cb = None

    # ---- evaluation of a block of rows on this rank
    def local(self, cubes):
        n = cubes.shape[0]
        theta = np.asarray(self.prior(cubes), dtype=float).reshape(n, self.nDims) if self.vec_prior else \
            np.array([np.asarray(self.prior(c), dtype=float) for c in cubes], dtype=float).reshape(n, self.nDims)
        phi = np.zeros((n, self.nDerived))
        if self.vec_like:
            out = self.like(theta)
            if isinstance(out, tuple):
                logL, p = out
                phi[:] = np.asarray(p, dtype=float).reshape(n, self.nDerived)
            else:
                logL = out
            logL = np.asarray(logL, dtype=float).reshape(n)
        else:
            logL = np.empty(n)
            for i in range(n):
                out = self.like(theta[i])
                try:
                    logL[i], phi[i, :] = out
                except TypeError:
                    logL[i] = out
        return theta, phi, logL

    # ---- all ranks together (rank 0 holds the rows)
    def evaluate(self, cubes):
        comm = self.comm
        if comm is None:
            return self.local(cubes)
        size = comm.Get_size()
        comm.bcast("eval", root=0)
        mine = comm.scatter(np.array_split(cubes, size), root=0)
        parts = comm.gather(self.local(mine) if len(mine) else (np.zeros((0, self.nDims)), np.zeros((0, self.nDerived)), np.zeros(0)), root=0)
        return tuple(np.concatenate([q[k] for q in parts]) for k in range(3))

    def serve(self):
        """worker ranks: evaluate shares until rank 0 says the run is over"""
        comm = self.comm
        while comm.bcast(None, root=0) == "eval":
            mine = comm.scatter(None, root=0)
            comm.gather(self.local(mine) if len(mine) else (np.zeros((0, self.nDims)), np.zeros((0, self.nDerived)), np.zeros(0)), root=0)

    # ---- the C callback
    def register(self):
        import ctypes as C
        from .. import _ctypes_api as api
        lib = api.load()
        proto = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                            C.POINTER(C.c_double), C.POINTER(C.c_double))

        def cb(_user, n, nd, nder, cube_p, theta_p, phi_p, logL_p):
            logL = np.ctypeslib.as_array(logL_p, shape=(n,))
            try:
                if self.error is not None:
                    raise self.error
                cubes = np.ctypeslib.as_array(cube_p, shape=(n, nd))
                theta, phi, ll = self.evaluate(cubes)
                np.ctypeslib.as_array(theta_p, shape=(n, nd))[:] = theta
                if nder > 0:
                    np.ctypeslib.as_array(phi_p, shape=(n, nder))[:] = phi
                logL[:] = ll
            except BaseException as e:   # noqa: BLE001 - re-raised by run() once the engine has wound down
                if self.error is None:
                    self.error = e
                logL[:] = self.logzero
                lib.polychord_hip_request_stop()
        self._cb = proto(cb)
        lib.polychord_hip_set_batch_callback.argtypes = [C.c_void_p, C.c_void_p]
        lib.polychord_hip_set_batch_callback(C.cast(self._cb, C.c_void_p), None)
        return lib

    def unregister(self, lib):
        lib.polychord_hip_set_batch_callback(None, None)
        self._cb = None


def _engine_run(loglikelihood, prior, *args):
    """_pypolychord.run(wrapped loglikelihood, wrapped prior, *args) on rank 0; every rank leaves together (a failure on
    rank 0 is re-raised there after the others have been released).  args[1], args[2] = nDims, nDerived; args[10] = logzero."""
    wl, wp = _wrap(loglikelihood, prior)
    comm = _mpi_comm()
    batch = _BatchEvaluator(loglikelihood, prior, args[1], args[2], comm, args[10])
    if comm is None:
        if not batch.active:
            return _pypolychord.run(wl, wp, *args)
        lib = batch.register()
        try:
            _pypolychord.run(wl, wp, *args)
        finally:
            batch.unregister(lib)
        if batch.error is not None:
            raise batch.error
        return None
    err = None
    if comm.Get_rank() == 0:
        lib = batch.register() if batch.active else None
        try:
            _pypolychord.run(wl, wp, *args)
        except BaseException as e:      # noqa: BLE001 - re-raised below
            err = e
        finally:
            if lib is not None:
                batch.unregister(lib)
                comm.bcast("done", root=0)
        err = err or batch.error
    elif batch.active:
        batch.serve()
    comm.Barrier()
    if err is not None:
        raise err
    return None


def default_prior(cube):
    """identity prior on the unit hypercube (polychord.py:9-10)"""
    return cube.copy()


def default_dumper(live, dead, logweights, logZ, logZerr):
    pass


def make_paramnames_file(paramnames, filename):
    """output.py:173-177"""
    with open(filename, "w") as f:
        for name, latex in paramnames:
            f.write("%s   %s\n" % (name, latex))


def _call_prior(prior, cube):
    """theta of one hypercube point, also from a prior written for blocks of rows (`prior.vectorised`)"""
    if getattr(prior, "vectorised", False):
        return np.asarray(prior(np.asarray(cube, dtype=float)[None, :]), dtype=float)[0]
    return prior(cube)


def _call_like(loglikelihood, theta):
    """logL or (logL, phi) of one point, also from a likelihood written for blocks of rows (`loglikelihood.vectorised`)"""
    if getattr(loglikelihood, "vectorised", False):
        out = loglikelihood(np.asarray(theta, dtype=float)[None, :])
        if isinstance(out, tuple):
            return float(np.asarray(out[0]).reshape(-1)[0]), np.asarray(out[1], dtype=float).reshape(1, -1)[0]
        return float(np.asarray(out).reshape(-1)[0])
    return loglikelihood(theta)


def _wrap(loglikelihood, prior):
    builtin_like = getattr(loglikelihood, "symbol", None) is not None
    builtin_prior = getattr(prior, "symbol", None) is not None

    def wrap_loglikelihood(theta, phi):          # polychord.py:581-587
        logL = _call_like(loglikelihood, theta)
        try:
            logL, phi[:] = logL
        except TypeError:
            pass
        return logL

    def wrap_prior(cube, theta):                 # polychord.py:589-590
        theta[:] = _call_prior(prior, cube)

    if builtin_like:
        wrap_loglikelihood.__wrapped_builtin__ = loglikelihood
    if builtin_prior:
        wrap_prior.__wrapped_builtin__ = prior
    return wrap_loglikelihood, wrap_prior


from .output import PolyChordOutput as _Output  # noqa: E402  (the class run_polychord returns)


def _e24(v):
    """Fortran E24.15E3 (src/polychord/utils.F90:19-21)"""
    v = float(v)
    if v == 0.0:
        return "   0.000000000000000E+000"
    m, e = ("%.14E" % abs(v)).split("E")
    ex = int(e) + 1
    return ("%s0.%sE%s%03d" % ("-" if v < 0 else "", m.replace(".", ""), "-" if ex < 0 else "+", abs(ex))).rjust(24)


def _make_resume_file(loglikelihood, **kwargs):
    """A .resume file whose live points are the user's `cube_samples` (polychord.py:650-789 of the reference):
    one cluster, nothing dead yet, identity covariance; the run then starts with read_resume."""
    cubes = np.asarray(kwargs["cube_samples"], dtype=float)
    lives = []
    for cube in cubes:
        theta = np.asarray(_call_prior(kwargs["prior"], cube), dtype=float)
        logL = _call_like(loglikelihood, theta)
        try:
            logL, derived = logL
        except TypeError:
            derived = []
        lives.append(np.concatenate([cube, theta, np.asarray(derived, dtype=float), [kwargs["logzero"], float(logL)]]))
    lives = np.array(lives)
    nDims, nDerived = cubes.shape[1], lives.shape[1] - 2 * cubes.shape[1] - 2
    lz, ident = kwargs["logzero"], np.identity(nDims)
    sep = "---------------------------------------"

    def ints(v):
        return "".join("%12d" % int(x) for x in np.atleast_1d(v))

    def reals(v):
        return "".join(_e24(x) for x in np.atleast_1d(v))

    body = [
        ("Number of dimensions", [ints(nDims)]), ("Number of derived parameters", [ints(nDerived)]),
        ("Number of dead points/iterations", [ints(0)]), ("Number of clusters", [ints(1)]), ("Number of dead clusters", [ints(0)]),
        ("Number of global weighted posterior points", [ints(0)]), ("Number of global equally weighted posterior points", [ints(0)]),
        ("Number of grades", [ints(len(kwargs["grade_dims"]))]), ("positions of grades", [ints(kwargs["grade_dims"])]),
        ("Number of repeats", [ints(kwargs["num_repeats"])]), ("Number of likelihood calls", [ints(len(lives))]),
        ("Number of live points in each cluster", [ints(len(lives))]), ("Number of phantom points in each cluster", [ints(0)]),
        ("Number of weighted posterior points in each cluster", [ints(0)]),
        ("Number of equally weighted posterior points in each cluster", [ints(0)]),
        ("Minimum loglikelihood positions", [ints(np.argmin(lives[:, -1]) + 1)]),
        ("Number of weighted posterior points in each dead cluster", []),
        ("Number of equally weighted posterior points in each dead cluster", []),
        ("global evidence -- log(<Z>)", [reals(lz)]), ("global evidence^2 -- log(<Z^2>)", [reals(lz)]),
        ("posterior thin factor", [reals(kwargs["boost_posterior"])]), ("local loglikelihood bounds", [reals(lives[:, -1].min())]),
        ("local volume -- log(<X_p>)", [reals(0.0)]), ("last update volume", [reals(0.0)]),
        ("global evidence volume cross correlation -- log(<ZX_p>)", [reals(lz)]), ("local evidence -- log(<Z_p>)", [reals(lz)]),
        ("local evidence^2 -- log(<Z_p^2>)", [reals(lz)]), ("local evidence volume cross correlation -- log(<Z_pX_p>)", [reals(lz)]),
        ("local volume cross correlation -- log(<X_pX_q>)", [reals(0.0)]), ("maximum log weights -- log(w_p)", [reals(lz)]),
        ("local dead evidence -- log(<Z_p>)", []), ("local dead evidence^2 -- log(<Z_p^2>)", []),
        ("maximum dead log weights -- log(w_p)", []),
        ("covariance matrices", [sep] + [reals(x) for x in ident]), ("cholesky decompositions", [sep] + [reals(x) for x in ident]),
        ("live points", [sep] + [reals(x) for x in lives]), ("dead points", []), ("logweights of dead points", []),
        ("phantom points", [sep]), ("weighted posterior points", [sep]), ("dead weighted posterior points", []),
        ("global weighted posterior points", []), ("equally weighted posterior points", [sep]),
        ("dead equally weighted posterior points", []), ("global equally weighted posterior points", []),
    ]
    path = Path(kwargs["base_dir"]) / (kwargs["file_root"] + ".resume")
    with open(path, "w") as f:
        for name, lines in body:
            f.write("=== %s ===\n" % name)
            for l in lines:
                f.write(l + "\n")


def _legacy_make_resume_file(settings, loglikelihood, prior):
    kwargs = dict(settings.__dict__)
    kwargs["prior"] = prior
    _make_resume_file(loglikelihood, **kwargs)


def run_polychord(loglikelihood, nDims, nDerived, settings, prior=default_prior, dumper=default_dumper):
    """legacy interface (polychord.py:16-215)"""
    comm = _mpi_comm()
    root = comm is None or comm.Get_rank() == 0
    if root:
        Path(settings.cluster_dir).mkdir(parents=True, exist_ok=True)
    if settings.cube_samples is not None:                      # polychord.py:170-173 of the reference
        if root:
            _legacy_make_resume_file(settings, loglikelihood, prior)
        settings.read_resume = True
    settings.grade_dims = [int(d) for d in settings.grade_dims]
    settings.nlives = {float(logL): int(nlive) for logL, nlive in settings.nlives.items()}
    _engine_run(loglikelihood, prior, dumper, nDims, nDerived, settings.nlive, settings.num_repeats, settings.nprior, settings.nfail,
                     settings.do_clustering, settings.feedback, settings.precision_criterion, settings.logzero,
                     settings.max_ndead, settings.boost_posterior, settings.posteriors, settings.equals,
                     settings.cluster_posteriors, settings.write_resume, settings.write_paramnames, settings.read_resume,
                     settings.write_stats, settings.write_live, settings.write_dead, settings.write_prior, settings.maximise,
                     settings.compression_factor, settings.synchronous, settings.base_dir, settings.file_root,
                     settings.grade_frac, settings.grade_dims, settings.nlives, settings.seed)
    return _Output(settings.base_dir, settings.file_root)


def run(loglikelihood, nDims, **kwargs):
    """keyword interface (polychord.py:221-646)"""
    paramnames = kwargs.pop("paramnames", None)
    default_kwargs = {
        "nDerived": 0, "prior": default_prior, "dumper": default_dumper, "nlive": nDims * 25, "num_repeats": nDims * 5,
        "nprior": -1, "nfail": -1, "do_clustering": True, "feedback": 1, "precision_criterion": 0.001, "logzero": -1e30,
        "max_ndead": -1, "boost_posterior": 0.0, "posteriors": True, "equals": True, "cluster_posteriors": True,
        "write_resume": True, "write_paramnames": False, "read_resume": True, "write_stats": True, "write_live": True,
        "write_dead": True, "write_prior": True, "maximise": False, "compression_factor": np.exp(-1), "synchronous": True,
        "base_dir": "chains", "file_root": "test", "cluster_dir": "clusters", "grade_dims": [nDims], "nlives": {},
        "seed": -1, "cube_samples": None,
    }
    default_kwargs["grade_frac"] = ([1.0] * len(default_kwargs["grade_dims"]) if "grade_dims" not in kwargs
                                    else [1.0] * len(kwargs["grade_dims"]))
    if not kwargs.keys() <= default_kwargs.keys():
        raise TypeError(f"{__name__} got unknown keyword arguments: {kwargs.keys() - default_kwargs.keys()}")
    default_kwargs.update(kwargs)
    kwargs = default_kwargs
    comm = _mpi_comm()
    root = comm is None or comm.Get_rank() == 0                # polychord.py:566-572 of the reference: rank 0 makes the directories
    if root:
        (Path(kwargs["base_dir"]) / kwargs["cluster_dir"]).mkdir(parents=True, exist_ok=True)
        if paramnames is not None:
            make_paramnames_file(paramnames, Path(kwargs["base_dir"]) / (kwargs["file_root"] + ".paramnames"))
    kwargs["grade_dims"] = [int(d) for d in list(kwargs["grade_dims"])]
    if sum(kwargs["grade_dims"]) != nDims:
        raise ValueError(f"grade_dims ({sum(kwargs['grade_dims'])}) must sum to nDims ({nDims})")
    kwargs["nlives"] = {float(logL): int(nlive) for logL, nlive in kwargs["nlives"].items()}
    if kwargs["cube_samples"] is not None:                     # polychord.py:596-598 of the reference
        if root:
            _make_resume_file(loglikelihood, **kwargs)
        kwargs["read_resume"] = True
    _engine_run(loglikelihood, kwargs["prior"], kwargs["dumper"], nDims, kwargs["nDerived"], kwargs["nlive"], kwargs["num_repeats"],
                     kwargs["nprior"], kwargs["nfail"], kwargs["do_clustering"], kwargs["feedback"],
                     kwargs["precision_criterion"], kwargs["logzero"], kwargs["max_ndead"], kwargs["boost_posterior"],
                     kwargs["posteriors"], kwargs["equals"], kwargs["cluster_posteriors"], kwargs["write_resume"],
                     kwargs["write_paramnames"], kwargs["read_resume"], kwargs["write_stats"], kwargs["write_live"],
                     kwargs["write_dead"], kwargs["write_prior"], kwargs["maximise"], kwargs["compression_factor"],
                     kwargs["synchronous"], kwargs["base_dir"], kwargs["file_root"], kwargs["grade_frac"],
                     kwargs["grade_dims"], kwargs["nlives"], kwargs["seed"])
    try:
        import anesthetic
    except ImportError:
        warnings.warn("anesthetic not installed. Cannot return NestedSamples object.")
        return None
    return anesthetic.read_chains(str(Path(kwargs["base_dir"]) / kwargs["file_root"]))
