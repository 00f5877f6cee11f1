"""polychordlite_amd -- MI355X-native nested-sampling engine behind PolyChordLite's own API.

`polychordlite_amd.pypolychord` mirrors the reference's `pypolychord` package (run / run_polychord /
PolyChordSettings); the compute path is libpolychord_hip.so (hand-written HIP for gfx950).
"""
__version__ = "0.1.0"
