"""PolyChordSettings -- mirrors reference pypolychord/settings.py:176-222 (names and defaults)."""
import os

import numpy


class PolyChordSettings:
    """Container of the run settings of the legacy `run_polychord` interface.

    Same attributes and defaults as the reference class (settings.py:176-210); unknown keyword
    arguments raise TypeError (:212-214) and grade_dims must sum to nDims (ValueError, :216-218).
    """

    _DEFAULTS = (("nprior", -1), ("nfail", -1), ("do_clustering", True), ("feedback", 1),
                 ("precision_criterion", 0.001), ("logzero", -1e30), ("max_ndead", -1), ("boost_posterior", 0.0),
                 ("posteriors", True), ("equals", True), ("cluster_posteriors", True), ("write_resume", True),
                 ("write_paramnames", False), ("read_resume", True), ("write_stats", True), ("write_live", True),
                 ("write_dead", True), ("write_prior", True), ("maximise", False),
                 ("compression_factor", numpy.exp(-1)), ("synchronous", True), ("base_dir", "chains"),
                 ("file_root", "test"), ("seed", -1))

    def __init__(self, nDims, nDerived, **kwargs):
        self.nlive = kwargs.pop("nlive", nDims * 25)
        self.num_repeats = kwargs.pop("num_repeats", nDims * 5)
        for name, default in self._DEFAULTS:
            setattr(self, name, kwargs.pop(name, default))
        self.grade_dims = list(kwargs.pop("grade_dims", [nDims]))
        self.grade_frac = list(kwargs.pop("grade_frac", [1.0] * len(self.grade_dims)))
        self.nlives = kwargs.pop("nlives", {})
        self.cube_samples = kwargs.pop("cube_samples", None)
        if kwargs:
            raise TypeError("Unexpected **kwargs in Contours constructor: %r" % kwargs)
        if sum(self.grade_dims) != nDims:
            raise ValueError("grade_dims must sum to the total dimensionality: sum(%s) /= %i" % (self.grade_dims, nDims))

    @property
    def cluster_dir(self):
        return os.path.join(self.base_dir, "clusters")
