"""ctypes view of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY (the checker, never the product)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")

LOGL_FN = C.CFUNCTYPE(C.c_double, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_int, C.c_void_p)
PRIOR_FN = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.c_void_p)


class Rng(C.Structure):
    _fields_ = [("key", C.c_uint32 * 2), ("sequential", C.c_int), ("seq", C.c_uint64), ("post", C.c_uint64)]


class Like(C.Structure):
    _fields_ = [("kind", C.c_int), ("mu", C.c_double), ("sigma", C.c_double), ("invcov", C.POINTER(C.c_double)),
                ("logdetcov", C.c_double), ("mean", C.POINTER(C.c_double)), ("fn", C.c_void_p), ("ctx", C.c_void_p)]


class Prior(C.Structure):
    _fields_ = [("kind", C.c_int), ("lo", C.POINTER(C.c_double)), ("hi", C.POINTER(C.c_double)),
                ("fn", C.c_void_p), ("ctx", C.c_void_p)]


class Settings(C.Structure):
    _fields_ = [("nDims", C.c_int), ("nDerived", C.c_int), ("nlive", C.c_int), ("num_repeats", C.c_int),
                ("nprior", C.c_int), ("nfail", C.c_int), ("do_clustering", C.c_int),
                ("precision_criterion", C.c_double), ("logzero", C.c_double), ("max_ndead", C.c_int),
                ("boost_posterior", C.c_double), ("posteriors", C.c_int), ("equals", C.c_int),
                ("cluster_posteriors", C.c_int), ("compression_factor", C.c_double), ("n_nlives", C.c_int),
                ("loglikes", C.POINTER(C.c_double)), ("nlives", C.POINTER(C.c_int)), ("seed", C.c_int),
                ("batch", C.c_int), ("sequential_rng", C.c_int), ("time_speeds_draw", C.c_int),
                ("nGrade", C.c_int), ("grade_dims", C.POINTER(C.c_int)), ("grade_frac", C.POINTER(C.c_double)), ("farm", C.c_int), ("epoch_discard", C.c_int)]


class Result(C.Structure):
    _fields_ = [("logZ", C.c_double), ("varlogZ", C.c_double), ("ndead", C.c_long), ("nlike", C.c_long),
                ("ncluster", C.c_int), ("ncluster_dead", C.c_int), ("niter", C.c_long), ("nbatches", C.c_long),
                ("nTotal", C.c_int), ("dead", C.POINTER(C.c_double)), ("logweights", C.POINTER(C.c_double)),
                ("live", C.POINTER(C.c_double)), ("nlive_final", C.c_int), ("logZp", C.POINTER(C.c_double)),
                ("varlogZp", C.POINTER(C.c_double)), ("nZp", C.c_int), ("post_mean", C.POINTER(C.c_double)),
                ("post_var", C.POINTER(C.c_double)), ("nposterior_global", C.c_long), ("nequals_global", C.c_long),
                ("nlike_grade", C.c_long * 8), ("post_rows", C.POINTER(C.c_double)), ("equal_rows", C.POINTER(C.c_double)),
                ("maxlogweight", C.c_double)]


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"])
    lib = C.CDLL(LIB)
    d = C.c_double
    pd = C.POINTER(C.c_double)
    lib.pc_inv_normal_cdf.argtypes = [d]; lib.pc_inv_normal_cdf.restype = d
    lib.pc_logsumexp.argtypes = [pd, C.c_int]; lib.pc_logsumexp.restype = d
    lib.pc_logaddexp.argtypes = [d, d]; lib.pc_logaddexp.restype = d
    lib.pc_cholesky.argtypes = [pd, C.c_int, pd]
    lib.pc_covmat.argtypes = [pd, C.c_int, pd, C.c_int, C.c_int, C.c_int, pd]
    lib.pc_similarity.argtypes = [pd, C.c_int, C.c_int, C.c_int, pd]
    lib.pc_nn_clustering.argtypes = [pd, C.c_int, C.POINTER(C.c_int)]; lib.pc_nn_clustering.restype = C.c_int
    lib.pc_compute_knn.argtypes = [pd, C.c_int, C.c_int, C.POINTER(C.c_int)]
    lib.pc_evidence_replay.argtypes = [pd, pd, C.c_long, pd, pd]
    lib.pc_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.pc_uniform_keyed.argtypes = [C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    lib.pc_uniform_keyed.restype = d
    lib.pc_settings_default.argtypes = [C.POINTER(Settings), C.c_int, C.c_int]
    lib.pc_oracle_run.argtypes = [C.POINTER(Settings), C.POINTER(Like), C.POINTER(Prior), C.POINTER(Result)]
    lib.pc_oracle_run.restype = C.c_int
    lib.pc_result_free.argtypes = [C.POINTER(Result)]
    lib.pc_slice_chain.argtypes = [C.POINTER(Settings), C.POINTER(Like), C.POINTER(Prior), C.POINTER(Rng), C.c_uint32,
                                   C.c_uint32, pd, pd, d, pd, pd]
    lib.pc_slice_chain.restype = C.c_long
    lib.pc_random_invcov.argtypes = [C.c_uint32, C.c_int, d, pd, pd]
    lib.pc_like_eval.argtypes = [C.POINTER(Like), pd, C.c_int, pd, C.c_int]; lib.pc_like_eval.restype = d
    _lib = lib
    return lib


def dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


KINDS = {"gaussian": 1, "rastrigin": 2, "twin_gaussian": 3, "corr_gaussian": 4}


def make_problem(kind, nDims, lo=None, hi=None, mu=0.5, sigma=0.1, invcov=None, mean=None, logdet=0.0):
    keep = []
    L = Like(); L.kind = KINDS[kind]; L.mu = mu; L.sigma = sigma; L.logdetcov = logdet
    if invcov is not None:
        ic = np.ascontiguousarray(invcov, dtype=np.float64); mn = np.ascontiguousarray(mean, dtype=np.float64)
        keep += [ic, mn]; L.invcov = dptr(ic); L.mean = dptr(mn)
    P = Prior(); P.kind = 1
    if lo is not None:
        lo_a = np.ascontiguousarray(np.broadcast_to(lo, (nDims,)), dtype=np.float64)
        hi_a = np.ascontiguousarray(np.broadcast_to(hi, (nDims,)), dtype=np.float64)
        keep += [lo_a, hi_a]; P.lo = dptr(lo_a); P.hi = dptr(hi_a)
    return L, P, keep


def settings(nDims, nDerived=0, **kw):
    lib = load()
    s = Settings()
    lib.pc_settings_default(C.byref(s), nDims, nDerived)
    for k, v in kw.items():
        setattr(s, k, v)
    return s


def set_grades(s, dims, repeats):
    """fast/slow grades with explicit repeats (chordal_sampling.f90:94-145); returns the arrays to keep alive"""
    gd = np.array(dims, dtype=np.int32); gf = np.array(repeats, dtype=np.float64)
    s.nGrade = len(dims); s.grade_dims = gd.ctypes.data_as(C.POINTER(C.c_int)); s.grade_frac = dptr(gf)
    return gd, gf


def run(s, like, prior):
    lib = load()
    r = Result()
    rc = lib.pc_oracle_run(C.byref(s), C.byref(like), C.byref(prior), C.byref(r))
    assert rc == 0
    nT, nd, D = r.nTotal, r.ndead, s.nDims
    out = dict(logZ=r.logZ, logZerr=float(np.sqrt(abs(r.varlogZ))), varlogZ=r.varlogZ, ndead=nd, nlike=r.nlike,
               niter=r.niter, nbatches=r.nbatches, ncluster=r.ncluster, ncluster_dead=r.ncluster_dead, nTotal=nT,
               dead=np.ctypeslib.as_array(r.dead, shape=(nd, nT)).copy(),
               logweights=np.ctypeslib.as_array(r.logweights, shape=(nd,)).copy(),
               live=np.ctypeslib.as_array(r.live, shape=(max(r.nlive_final, 1), nT))[:r.nlive_final].copy(),
               logZp=np.ctypeslib.as_array(r.logZp, shape=(max(r.nZp, 1),))[:r.nZp].copy(),
               post_mean=np.ctypeslib.as_array(r.post_mean, shape=(D,)).copy(),
               post_var=np.ctypeslib.as_array(r.post_var, shape=(D,)).copy(),
               nlike_grade=[int(v) for v in r.nlike_grade], nposterior=int(r.nposterior_global), nequals=int(r.nequals_global),
               maxlogweight=r.maxlogweight,
               post_rows=np.ctypeslib.as_array(r.post_rows, shape=(max(r.nposterior_global, 1), 2 + D + s.nDerived))[:r.nposterior_global].copy(),
               equal_rows=np.ctypeslib.as_array(r.equal_rows, shape=(max(r.nequals_global, 1), 1 + D + s.nDerived))[:r.nequals_global].copy())
    lib.pc_result_free(C.byref(r))
    return out


def slice_chain(s, like, prior, key_seed, batch, chain, seed_point, chol, contour):
    """one oracle chain in keyed mode -> (babies[nr][nT], nhats[nr][D], nlike)"""
    lib = load()
    D, nT, nr = s.nDims, 2 * s.nDims + s.nDerived + 2, s.num_repeats
    rng = Rng(); rng.key[0] = key_seed & 0xFFFFFFFF; rng.key[1] = 0x504F4C59; rng.sequential = 0
    babies = np.zeros((nr, nT)); nh = np.zeros((nr, D))
    sp = np.ascontiguousarray(seed_point, dtype=np.float64); ch = np.ascontiguousarray(chol, dtype=np.float64)
    n = lib.pc_slice_chain(C.byref(s), C.byref(like), C.byref(prior), C.byref(rng), batch, chain, dptr(sp), dptr(ch),
                           contour, dptr(babies), dptr(nh))
    return babies, nh, n
