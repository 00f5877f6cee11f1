import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# the engine recycles device blocks between runs; in the tests every block it hands out is first filled with a
# garbage pattern (pc_engine.hip dalloc), so that a buffer a code path forgets to initialise cannot pass by luck
os.environ.setdefault("PC_POISON", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        from polychordlite_amd import _ctypes_api as api
        return api.load().pchip_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def engine():
    """the HIP engine through its C ABI; GPU tests FAIL (not skip) if the .so is missing on a GPU box"""
    from polychordlite_amd import _ctypes_api as api
    lib = api.load()
    if lib.pchip_device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests need the MI355X")
    return api


@pytest.fixture(scope="session")
def golden():
    import json
    d = os.path.join(ROOT, "tests", "golden")
    return {n[:-5]: json.load(open(os.path.join(d, n))) for n in os.listdir(d) if n.endswith(".json")}
