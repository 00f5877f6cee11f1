"""`_pypolychord.run` -- the 37-positional-argument entry of the reference's CPython extension
(pypolychord/_pypolychord.cpp:119-228, parse format "OOOiiiiiiiiddidiiiiiiiiiiidissO!O!O!i"),
implemented over the C ABI of libpolychord_hip.so with ctypes.  The package prefers the compiled module
`_pypolychord` (pypolychord/_pypolychord_module.cpp, same surface); this one is the binding of choice when no C++
compiler / Python headers are around, and the cross-check of the compiled one in tests.

Callbacks follow _pypolychord.cpp:29-115: `loglikelihood(theta, phi) -> float` gets numpy views of
engine-owned buffers (theta read-only, phi written in place), must return a Python float;
`prior(cube, theta)` writes theta in place; `dumper(live, dead, logweights, logZ, logZerr)`.
A Python exception raised inside a callback is re-raised by `run` (the reference throws it through
the Fortran frames, :219-224; here the run is finished with logzero for that point first).
"""
import ctypes as C

import numpy as np

from .. import _ctypes_api as api


def run(loglikelihood, prior, dumper, nDims, nDerived, nlive, num_repeats, nprior, nfail, do_clustering, feedback,
        precision_criterion, logzero, max_ndead, boost_posterior, posteriors, equals, cluster_posteriors,
        write_resume, write_paramnames, read_resume, write_stats, write_live, write_dead, write_prior, maximise,
        compression_factor, synchronous, base_dir, file_root, grade_frac, grade_dims, nlives, seed):
    for name, fn in (("loglikelihood", loglikelihood), ("prior", prior), ("dumper", dumper)):
        if not callable(fn):
            raise TypeError(f"{name} must be callable")                        # _pypolychord.cpp:178-186
    if not isinstance(grade_frac, list) or not isinstance(grade_dims, list) or not isinstance(nlives, dict):
        raise TypeError("grade_frac and grade_dims must be lists, nlives a dict")   # format "O!O!O!"
    if len(grade_frac) != len(grade_dims):
        raise ValueError("grade_dims and grade_frac must have the same length")    # _pypolychord.cpp:196-200
    lib = api.load()
    f = lib.polychord_c_interface
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_bool, C.c_int, C.c_double,
                  C.c_double, C.c_int, C.c_double] + [C.c_bool] * 11 + [C.c_double, C.c_bool, C.c_int, C.c_int, C.c_char_p,
                  C.c_char_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_double),
                  C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]
    errors = []
    npars = nDims + nDerived + 2

    # built-in device functors pass straight through as the library's own function pointers
    def builtin_ptr(obj):
        sym = getattr(obj, "symbol", None)
        if sym is None:
            return None
        obj.configure(nDims)
        return C.cast(getattr(lib, sym), C.c_void_p)

    like_ptr = builtin_ptr(getattr(loglikelihood, "__wrapped_builtin__", loglikelihood))
    prior_ptr = builtin_ptr(getattr(prior, "__wrapped_builtin__", prior))
    keep = []
    if like_ptr is None:
        def c_like(theta_p, nd, phi_p, nder):
            try:
                theta = np.ctypeslib.as_array(theta_p, shape=(nd,))
                theta.flags.writeable = False                                 # _pypolychord.cpp:35
                phi = np.ctypeslib.as_array(phi_p, shape=(nder,)) if nder > 0 else np.zeros(0)
                out = loglikelihood(theta, phi)
                if not isinstance(out, float):
                    raise ValueError("Return from loglikelihood must be a float")   # _pypolychord.cpp:47-51
                return out
            except BaseException as e:   # noqa: BLE001 - re-raised after the run
                errors.append(e)
                lib.polychord_hip_request_stop()
                return logzero
        cb = api.LOGLIKE_FN(c_like); keep.append(cb)
        like_ptr = C.cast(cb, C.c_void_p)
    if prior_ptr is None:
        def c_prior(cube_p, theta_p, nd):
            try:
                cube = np.ctypeslib.as_array(cube_p, shape=(nd,))
                theta = np.ctypeslib.as_array(theta_p, shape=(nd,))
                prior(cube, theta)
            except BaseException as e:   # noqa: BLE001
                errors.append(e)
                lib.polychord_hip_request_stop()
        cb = api.PRIOR_FN(c_prior); keep.append(cb)
        prior_ptr = C.cast(cb, C.c_void_p)

    def c_dumper(ndead, nlive_, npars_, live_p, dead_p, logw_p, logZ, logZerr):
        try:
            live = np.ctypeslib.as_array(live_p, shape=(max(nlive_, 0), npars_)) if nlive_ > 0 else np.zeros((0, npars_))
            dead = np.ctypeslib.as_array(dead_p, shape=(ndead, npars_)) if ndead > 0 else np.zeros((0, npars_))
            logw = np.ctypeslib.as_array(logw_p, shape=(ndead,)) if ndead > 0 else np.zeros(0)
            dumper(live, dead, logw, logZ, logZerr)
        except BaseException as e:   # noqa: BLE001
            errors.append(e)
    cbd = api.DUMPER_FN(c_dumper); keep.append(cbd)

    gf = (C.c_double * len(grade_frac))(*[float(x) for x in grade_frac])
    gd = (C.c_int * len(grade_dims))(*[int(x) for x in grade_dims])
    keys = sorted(nlives.keys())
    ll = (C.c_double * max(len(keys), 1))(*[float(k) for k in keys])
    nl = (C.c_int * max(len(keys), 1))(*[int(nlives[k]) for k in keys])
    comm = C.c_int(0)
    lib.polychord_hip_set_option(b"halt_returns", 1.0)       # engine failures come back as exceptions, not as `stop 1`
    lib.polychord_hip_last_error.restype = C.c_char_p
    f(like_ptr, prior_ptr, C.cast(cbd, C.c_void_p), int(nlive), int(num_repeats), int(nprior), int(nfail), bool(do_clustering),
      int(feedback), float(precision_criterion), float(logzero), int(max_ndead), float(boost_posterior), bool(posteriors),
      bool(equals), bool(cluster_posteriors), bool(write_resume), bool(write_paramnames), bool(read_resume),
      bool(write_stats), bool(write_live), bool(write_dead), bool(write_prior), bool(maximise), float(compression_factor),
      bool(synchronous), int(nDims), int(nDerived), str(base_dir).encode(), str(file_root).encode(), len(grade_frac), gf, gd,
      len(keys), ll, nl, int(seed), C.byref(comm))
    if errors:
        raise errors[0]
    msg = lib.polychord_hip_last_error()
    if msg:
        raise RuntimeError(msg.decode())
    return None
