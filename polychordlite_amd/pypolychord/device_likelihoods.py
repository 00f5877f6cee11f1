"""Handles for the likelihoods / prior that run fused inside the slice-sampling kernel.

Passing one of these objects as `loglikelihood` (or `prior`) to `run` / `run_polychord` makes the
engine evaluate it on the GPU instead of calling back into Python for every proposal; they are also
ordinary callables (the formula is evaluated by the library's host function), so the same script runs
unchanged against the reference `pypolychord`.
"""
import ctypes as C

import numpy as np

from .. import _ctypes_api as api


class _Builtin:
    symbol = None

    def _fn(self):
        lib = api.load()
        f = getattr(lib, self.symbol)
        f.restype = C.c_double
        f.argtypes = [C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_int]
        return f

    def configure(self, nDims):
        pass

    def __call__(self, theta):
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        self.configure(theta.size)
        phi = np.zeros(max(self.nDerived, 1))
        logL = self._fn()(api.dptr(theta), theta.size, api.dptr(phi), self.nDerived)
        return (logL, phi[:self.nDerived]) if self.nDerived else logL


class Gaussian(_Builtin):
    """likelihoods/examples/gaussian.f90: N(mu, sigma^2 I); phi = (radius, log volume)"""
    symbol = "polychord_hip_gaussian"

    def __init__(self, mu=0.5, sigma=0.1, nDerived=0):
        self.mu, self.sigma, self.nDerived = mu, sigma, nDerived

    def configure(self, nDims):
        api.load().polychord_hip_set_gaussian(self.mu, self.sigma)


class Rastrigin(_Builtin):
    """likelihoods/examples/rastrigin.f90"""
    symbol = "polychord_hip_rastrigin"
    nDerived = 0


class TwinGaussian(_Builtin):
    """likelihoods/examples/twin_gaussian.f90; phi = (+-1 by theta_1 > 0.5)"""
    symbol = "polychord_hip_twin_gaussian"

    def __init__(self, sigma=0.1, nDerived=0):
        self.sigma, self.nDerived = sigma, nDerived

    def configure(self, nDims):
        api.load().polychord_hip_set_gaussian(0.0, self.sigma)


class CorrelatedGaussian(_Builtin):
    """likelihoods/examples/random_gaussian.f90 with an explicit inverse covariance"""
    symbol = "polychord_hip_corr_gaussian"
    nDerived = 0

    def __init__(self, invcov, mean, logdetcov):
        self.invcov = np.ascontiguousarray(invcov, dtype=np.float64)
        self.mean = np.ascontiguousarray(mean, dtype=np.float64)
        self.logdetcov = float(logdetcov)

    def configure(self, nDims):
        api.load().polychord_hip_set_corr_gaussian(nDims, api.dptr(self.invcov), api.dptr(self.mean), self.logdetcov)


class UniformPrior:
    """priors.f90:40-55 uniform box; evaluated on the device when the likelihood is a built-in"""
    symbol = "polychord_hip_uniform_prior"

    def __init__(self, lo, hi):
        self.lo, self.hi = lo, hi

    def configure(self, nDims):
        lo = np.ascontiguousarray(np.broadcast_to(self.lo, (nDims,)), dtype=np.float64)
        hi = np.ascontiguousarray(np.broadcast_to(self.hi, (nDims,)), dtype=np.float64)
        api.load().polychord_hip_set_uniform_prior(nDims, api.dptr(lo), api.dptr(hi))

    def __call__(self, cube):
        cube = np.asarray(cube, dtype=np.float64)
        return np.broadcast_to(self.lo, cube.shape) + (np.broadcast_to(self.hi, cube.shape) - np.broadcast_to(self.lo, cube.shape)) * cube
