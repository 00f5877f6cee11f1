"""Drop-in for the reference's `pypolychord` package on top of the MI355X engine.

    from polychordlite_amd import pypolychord
    pypolychord.run(loglikelihood, nDims, nlive=..., ...)          # reference pypolychord/polychord.py:221
    pypolychord.run_polychord(loglikelihood, nDims, nDerived, settings, prior, dumper)   # :16

Same call surface, defaults and error behaviour as reference pypolychord 1.22.2; the work is done by
libpolychord_hip.so through the reference's C entry point `polychord_c_interface`.
"""
__version__ = "1.22.2+hip0.1"
from .settings import PolyChordSettings
from .polychord import run_polychord, run
from .output import PolyChordOutput
from . import priors
from . import device_likelihoods
