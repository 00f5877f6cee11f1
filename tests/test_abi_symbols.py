"""CPU: libpolychord_hip.so loads without a GPU and exports every symbol include/*.h declares."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if not fn.endswith(".h"):
            continue
        src = open(os.path.join(ROOT, "include", fn)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        for m in re.finditer(r"^\s*(?:void|int|double)\s+\*?\s*((?:polychord|pchip)_\w+)\s*\(", src, flags=re.M):
            names.add(m.group(1))
    return names


def test_library_exports_declared_symbols():
    from polychordlite_amd import _ctypes_api as api
    lib = api.load()
    names = _declared()
    assert {"polychord_c_interface", "polychord_c_interface_ini", "pchip_run", "pchip_slice_chains"} <= names
    for n in sorted(names):
        assert hasattr(lib, n), f"{n} declared in include/ but not exported"


def test_reference_boundary_signature_has_38_args():
    """interfaces.h:2-45: 3 callbacks + 34 scalars/pointers + comm by reference = 38 parameters"""
    src = open(os.path.join(ROOT, "include", "polychord_hip.h")).read()
    m = re.search(r"void polychord_c_interface\((.*?)\);", src, flags=re.S)
    args = [a for a in re.sub(r"\s+", " ", m.group(1)).split(",")]
    assert len(args) == 38


def test_builtin_host_functions_evaluate():
    """the built-in likelihood symbols are real host functions with the reference callback signature"""
    import numpy as np
    from polychordlite_amd import _ctypes_api as api
    from tests import oracle_api as orc
    lib = api.load(); olib = orc.load()
    lib.polychord_hip_gaussian.restype = C.c_double
    lib.polychord_hip_gaussian.argtypes = [C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_int]
    th = np.linspace(0.3, 0.7, 20); phi = np.zeros(2); phi2 = np.zeros(2)
    L, P, keep = orc.make_problem("gaussian", 20)
    a = lib.polychord_hip_gaussian(api.dptr(th), 20, api.dptr(phi), 2)
    b = olib.pc_like_eval(C.byref(L), orc.dptr(th), 20, orc.dptr(phi2), 2)
    assert abs(a - b) < 1e-12 and np.allclose(phi, phi2, rtol=1e-13)
