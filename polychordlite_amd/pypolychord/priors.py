"""Hypercube -> physical transforms with the names and call behaviour of the reference's `pypolychord.priors`
(reference pypolychord/priors.py:5-47): every class is constructed with its parameters and called with an array of
unit-hypercube coordinates.  `forced_indentifiability_transform` (the reference's spelling) is the order-statistics map
of priors.f90:245-262."""
import numpy as np

try:                                    # scipy is what the reference imports; the engine's own AS241 serves without it
    from scipy.special import ndtri as _inv_normal_cdf
except ImportError:                     # pragma: no cover
    def _inv_normal_cdf(p):
        import ctypes as C
        from .. import _ctypes_api as api
        f = api.load().polychord_hip_inv_normal_cdf
        f.restype, f.argtypes = C.c_double, [C.c_double]
        return np.vectorize(lambda q: f(float(q)))(np.asarray(p, dtype=float))


class UniformPrior:
    """theta = a + (b - a) x"""

    def __init__(self, a, b):
        self.a, self.b = a, b

    def __call__(self, x):
        return self.a + (self.b - self.a) * x


class GaussianPrior:
    """theta = mu + sigma Phi^-1(x)"""

    def __init__(self, mu, sigma):
        self.mu, self.sigma = mu, sigma

    def __call__(self, x):
        return self.mu + self.sigma * _inv_normal_cdf(x)


class LogUniformPrior(UniformPrior):
    """theta = a (b / a)^x"""

    def __call__(self, x):
        return self.a * (self.b / self.a) ** x


def forced_indentifiability_transform(x):
    """x uniform in the cube -> t sorted ascending, uniform on the simplex 0 < t_1 < ... < t_N < 1"""
    x = np.asarray(x, dtype=float)
    n = len(x)
    t = np.empty(n)
    t[n - 1] = x[n - 1] ** (1.0 / n)
    for k in range(n - 2, -1, -1):
        t[k] = x[k] ** (1.0 / (k + 1)) * t[k + 1]
    return t


class SortedUniformPrior(UniformPrior):
    def __call__(self, x):
        return UniformPrior.__call__(self, forced_indentifiability_transform(x))


class LogSortedUniformPrior(LogUniformPrior):
    def __call__(self, x):
        return LogUniformPrior.__call__(self, forced_indentifiability_transform(x))
