"""ctypes view of libpolychord_hip.so (include/polychord_hip.h) -- plumbing only.

The product path is the HIP library; there is no Python/NumPy fallback.  Importing this module
loads the shared object and fails loudly if it has not been built (`python -m polychordlite_amd.build`).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PCHIP_LIB") or os.path.join(_HERE, "libpolychord_hip.so")     # (PCHIP_LIB: A/B runs of two builds on one GPU box)

LOGLIKE_FN = C.CFUNCTYPE(C.c_double, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_int)
PRIOR_FN = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int)
DUMPER_FN = C.CFUNCTYPE(None, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                        C.POINTER(C.c_double), C.c_double, C.c_double)

LIKE_CALLBACK, LIKE_GAUSSIAN, LIKE_RASTRIGIN, LIKE_TWIN_GAUSSIAN, LIKE_CORR_GAUSSIAN = range(5)
KERNEL_CLASSES = ("k_nhats", "k_slice", "k_consume", "k_apply", "k_clean", "k_covmats", "k_bases_side")
LIKE_KINDS = {"gaussian": LIKE_GAUSSIAN, "rastrigin": LIKE_RASTRIGIN, "twin_gaussian": LIKE_TWIN_GAUSSIAN,
              "corr_gaussian": LIKE_CORR_GAUSSIAN}


class Settings(C.Structure):
    _fields_ = [("nDims", C.c_int), ("nDerived", C.c_int), ("nlive", C.c_int), ("num_repeats", C.c_int),
                ("nprior", C.c_int), ("nfail", C.c_int), ("do_clustering", C.c_int),
                ("precision_criterion", C.c_double), ("logzero", C.c_double), ("max_ndead", C.c_int),
                ("boost_posterior", C.c_double), ("posteriors", C.c_int), ("equals", C.c_int),
                ("cluster_posteriors", C.c_int), ("compression_factor", C.c_double), ("n_nlives", C.c_int),
                ("loglikes", C.POINTER(C.c_double)), ("nlives", C.POINTER(C.c_int)), ("seed", C.c_int),
                ("batch", C.c_int), ("device", C.c_int), ("feedback", C.c_int), ("profile", C.c_int),
                ("force_general", C.c_int), ("ablate", C.c_int),
                ("resume_write", C.c_char_p), ("sequential_rng", C.c_int), ("resume_read", C.c_char_p),
                ("nGrade", C.c_int), ("grade_dims", C.POINTER(C.c_int)), ("grade_repeats", C.POINTER(C.c_int)),
                ("epoch_discard", C.c_int), ("device_records", C.c_int)]


class Like(C.Structure):
    _fields_ = [("kind", C.c_int), ("mu", C.c_double), ("sigma", C.c_double), ("invcov", C.POINTER(C.c_double)),
                ("mean", C.POINTER(C.c_double)), ("logdetcov", C.c_double), ("fn", C.c_void_p)]


class Prior(C.Structure):
    _fields_ = [("kind", C.c_int), ("lo", C.POINTER(C.c_double)), ("hi", C.POINTER(C.c_double)), ("fn", C.c_void_p)]


class Result(C.Structure):
    _fields_ = [("logZ", C.c_double), ("varlogZ", C.c_double), ("ndead", C.c_long), ("nlike", C.c_long),
                ("niter", C.c_long), ("nbatches", C.c_long), ("nrounds", C.c_long), ("nupdates", C.c_long),
                ("ncluster", C.c_int), ("ncluster_dead", C.c_int), ("nTotal", C.c_int), ("batch", C.c_int),
                ("t_generate", C.c_double), ("t_loop", C.c_double), ("t_final", C.c_double), ("t_total", C.c_double),
                ("t_setup", C.c_double), ("t_results", C.c_double), ("t_teardown", C.c_double),
                ("k_time_s", C.c_double * 8), ("k_launches", C.c_long * 8),
                ("dead", C.POINTER(C.c_double)), ("logweights", C.POINTER(C.c_double)), ("entry", C.POINTER(C.c_double)),
                ("live", C.POINTER(C.c_double)), ("nlive_final", C.c_int),
                ("logZp", C.POINTER(C.c_double)), ("varlogZp", C.POINTER(C.c_double)), ("nZp", C.c_int),
                ("post_mean", C.POINTER(C.c_double)), ("post_var", C.POINTER(C.c_double)),
                ("nlike_grade", C.c_long * 8), ("live_cluster", C.POINTER(C.c_int)),
                ("nlike_failed", C.c_long), ("ncluster_peak", C.c_int), ("epoch_discard", C.c_int),
                ("d_records", C.c_void_p), ("n_records", C.c_long), ("records_cap", C.c_long), ("records_device", C.c_int),
                ("path", C.c_long * 24)]


# pchip_result.path[]: launches per kernel variant (include/polychord_hip.h PCHIP_PATH_*)
PATH_NAMES = ("consume_par", "consume_cl", "consume_general", "consume_fast", "killoff_par", "killoff_cl", "killoff_general",
              "killoff_fast", "update_fused", "update_steps", "slice_wave", "slice_lane", "nn_lists", "nn_fallbacks", "pool_mode",
              "defer_update", "consume_cl_serial")


_lib = None


def load():
    """dlopen libpolychord_hip.so (RTLD_GLOBAL so that the HIP runtime is shared with torch)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -m polychordlite_amd.build` "
                          "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    lib.pchip_settings_default.argtypes = [C.POINTER(Settings), C.c_int, C.c_int]
    lib.pchip_settings_default.restype = None
    lib.pchip_device_count.restype = C.c_int
    lib.pchip_run.argtypes = [C.POINTER(Settings), C.POINTER(Like), C.POINTER(Prior), C.POINTER(Result)]
    lib.pchip_run.restype = C.c_int
    lib.pchip_result_free.argtypes = [C.POINTER(Result)]
    lib.pchip_result_free.restype = None
    lib.pchip_slice_chains.argtypes = [C.POINTER(Settings), C.POINTER(Like), C.POINTER(Prior), C.c_uint, C.c_int,
                                       C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double,
                                       C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    lib.pchip_slice_chains.restype = C.c_int
    lib.polychord_hip_set_gaussian.argtypes = [C.c_double, C.c_double]
    lib.polychord_hip_set_uniform_prior.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.polychord_hip_set_corr_gaussian.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double]
    lib.polychord_hip_set_option.argtypes = [C.c_char_p, C.c_double]
    # this mirror against the library that was loaded (the structs grow at their end: include/polychord_hip.h PCHIP_ABI_VERSION)
    lib.pchip_sizeof.argtypes = [C.c_char_p]
    lib.pchip_sizeof.restype = C.c_ulong
    for name, cls in (("settings", Settings), ("result", Result), ("like", Like), ("prior", Prior)):
        if lib.pchip_sizeof(name.encode()) != C.sizeof(cls):
            raise ImportError(f"{LIB_PATH}: pchip_{name} is {lib.pchip_sizeof(name.encode())} bytes, this binding's mirror {C.sizeof(cls)} "
                              f"(library ABI version {lib.pchip_abi_version()}): rebuild with `python -m polychordlite_amd.build`")
    _lib = lib
    return lib


def dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def make_problem(kind, nDims, nDerived=0, lo=None, hi=None, mu=0.5, sigma=0.1, invcov=None, mean=None, logdet=0.0):
    """(Like, Prior, keepalive) for a built-in device likelihood and a uniform box prior."""
    keep = []
    L = Like()
    L.kind = LIKE_KINDS[kind]
    L.mu, L.sigma, L.logdetcov = mu, sigma, logdet
    if invcov is not None:
        ic = np.ascontiguousarray(invcov, dtype=np.float64)
        mn = np.ascontiguousarray(mean, dtype=np.float64)
        keep += [ic, mn]
        L.invcov, L.mean = dptr(ic), dptr(mn)
    P = Prior()
    P.kind = 1
    if lo is not None:
        lo_a = np.ascontiguousarray(np.broadcast_to(lo, (nDims,)), dtype=np.float64)
        hi_a = np.ascontiguousarray(np.broadcast_to(hi, (nDims,)), dtype=np.float64)
        keep += [lo_a, hi_a]
        P.lo, P.hi = dptr(lo_a), dptr(hi_a)
    return L, P, keep


def set_grades(settings, dims, repeats):
    """fast/slow parameter grades with explicit repeats per grade; returns the arrays to keep alive"""
    gd = np.array(dims, dtype=np.int32)
    gr = np.array(repeats, dtype=np.int32)
    settings.nGrade = len(dims)
    settings.grade_dims = gd.ctypes.data_as(C.POINTER(C.c_int))
    settings.grade_repeats = gr.ctypes.data_as(C.POINTER(C.c_int))
    return gd, gr


class _Owner:
    """Keeps a pchip_result alive for the numpy views handed out by run(); frees it when the last view dies."""

    def __init__(self, lib, res):
        self.lib, self.res = lib, res

    def __del__(self):
        try:
            self.lib.pchip_result_free(C.byref(self.res))
        except Exception:
            pass


class _View:
    """array-interface shim: numpy keeps this object (and through it the owner) as the array's base"""

    def __init__(self, owner, ptr, shape):
        self.owner = owner
        addr = C.cast(ptr, C.c_void_p).value or 0
        self.__array_interface__ = {"data": (addr, False), "shape": tuple(shape), "typestr": "<f8", "version": 3}


def _view(owner, ptr, shape):
    if int(np.prod(shape)) == 0:
        return np.zeros(shape)
    return np.asarray(_View(owner, ptr, shape))


def run(settings, like, prior):
    """pchip_run -> dict; the big arrays (dead points, weights, live points) are zero-copy views of the
    engine's pinned result buffers, released when the last view is garbage collected."""
    lib = load()
    r = Result()
    rc = lib.pchip_run(C.byref(settings), C.byref(like), C.byref(prior), C.byref(r))
    if rc != 0:
        raise RuntimeError(f"pchip_run failed with code {rc}")
    return result_dict(r, settings)


def result_dict(r, settings):
    """dict view of a filled pchip_result; takes ownership (the block is freed when the last array view dies)"""
    lib = load()
    own = _Owner(lib, r)
    nT, nd, D = r.nTotal, r.ndead, settings.nDims
    out = dict(logZ=r.logZ, logZerr=float(np.sqrt(abs(r.varlogZ))), varlogZ=r.varlogZ, ndead=nd, nlike=r.nlike,
               niter=r.niter, nbatches=r.nbatches, nrounds=r.nrounds, nupdates=r.nupdates, ncluster=r.ncluster,
               ncluster_dead=r.ncluster_dead, nTotal=nT, batch=r.batch, t_generate=r.t_generate, t_loop=r.t_loop,
               t_final=r.t_final, t_total=r.t_total, t_setup=r.t_setup, t_results=r.t_results, t_teardown=r.t_teardown,
               kernel_time={n: {"total_s": r.k_time_s[i], "launches": r.k_launches[i]}
                            for i, n in enumerate(KERNEL_CLASSES) if r.k_launches[i] > 0},
               dead=_view(own, r.dead, (nd, nT)),
               logweights=_view(own, r.logweights, (nd,)),
               entry=_view(own, r.entry, (nd,)),
               live=_view(own, r.live, (r.nlive_final, nT)),
               logZp=np.ctypeslib.as_array(r.logZp, shape=(max(r.nZp, 1),))[:r.nZp].copy(),
               post_mean=np.ctypeslib.as_array(r.post_mean, shape=(D + settings.nDerived,)).copy(),
               post_var=np.ctypeslib.as_array(r.post_var, shape=(D + settings.nDerived,)).copy(),
               nlike_grade=[int(v) for v in r.nlike_grade], nlike_failed=r.nlike_failed, ncluster_peak=r.ncluster_peak, epoch_discard=r.epoch_discard, n_records=int(r.n_records) if r.d_records else None,
               path={n: int(r.path[i]) for i, n in enumerate(PATH_NAMES)},
               varlogZp=np.ctypeslib.as_array(r.varlogZp, shape=(max(r.nZp, 1),))[:r.nZp].copy(),
               logzero=settings.logzero,
               _owner=own)          # the pchip_result itself (merge.comm_merge hands it back to the library)
    return out
