"""The .resume grammar (src/polychord/read_write.F90:219-288 writer, :384-476 reader): the engine's reader
and writer against a file written by the reference itself in the middle of a clustered run
(tests/golden/ref_rastrigin2d_mid.resume, made by oracle/gen_golden.py) and against the file pypolychord
builds for `cube_samples` (pypolychord/polychord.py:650-789).  Host code only: runs without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from polychordlite_amd import _ctypes_api as api

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_rastrigin2d_mid.resume")


def sections(path):
    out, cur = [], None
    for l in open(path):
        l = l.rstrip("\n")
        if l.startswith("==="):
            cur = [l, []]; out.append(cur)
        elif not l.startswith("---") and cur is not None and l.strip():
            cur[1].extend(float(x) for x in l.split())
    return out


def test_reference_resume_file_round_trip(tmp_path):
    lib = api.load()
    lib.polychord_hip_resume_copy.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int)]
    counts = (C.c_int * 6)()
    out = str(tmp_path / "copy.resume")
    assert lib.polychord_hip_resume_copy(GOLD.encode(), out.encode(), counts) == 0
    assert list(counts) == [2, 0, 349, 4, 4, 40]
    a, b = sections(GOLD), sections(out)
    assert [s[0] for s in a] == [s[0] for s in b]            # same sections, same order
    stacks = re.compile(r"posterior points|maximum .*log weights")
    for (name, va), (_, vb) in zip(a, b):
        if stacks.search(name):
            continue                                         # posterior stacks are rebuilt from the dead points
        assert len(va) == len(vb), name
        assert np.array_equal(np.array(va), np.array(vb)), name   # E24.15E3 survives a read + write unchanged
    # text layout of a line: integers I12, reals E24.15E3
    lines = open(out).read().splitlines()
    assert lines[1] == "%12d" % 2 and len(lines[lines.index("=== global evidence -- log(<Z>) ===") + 1]) == 24
    # a second pass over our own output is a fixed point
    out2 = str(tmp_path / "copy2.resume")
    assert lib.polychord_hip_resume_copy(out.encode(), out2.encode(), counts) == 0
    assert open(out).read() == open(out2).read()


def test_cube_samples_resume_file(tmp_path):
    """the file pypolychord writes for cube_samples: one cluster, no dead points, identity covariance"""
    from polychordlite_amd.pypolychord.polychord import _make_resume_file
    nD = 3
    cubes = np.random.default_rng(0).random((25, nD))
    kw = dict(base_dir=str(tmp_path), file_root="cs", cube_samples=cubes, prior=lambda c: 2 * np.asarray(c) - 1,
              logzero=-1e30, grade_dims=[nD], num_repeats=15, boost_posterior=0.0)
    _make_resume_file(lambda th: (-0.5 * float(np.sum(th ** 2)), [float(th[0])]), **kw)
    lib = api.load()
    lib.polychord_hip_resume_copy.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int)]
    counts = (C.c_int * 6)()
    assert lib.polychord_hip_resume_copy(str(tmp_path / "cs.resume").encode(), None, counts) == 0
    assert list(counts) == [nD, 1, 0, 1, 0, 25]


def test_malformed_resume_file_is_rejected(tmp_path):
    lib = api.load()
    lib.polychord_hip_resume_copy.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int)]
    bad = tmp_path / "bad.resume"
    bad.write_text("\n".join(open(GOLD).read().splitlines()[:60]) + "\n")
    assert lib.polychord_hip_resume_copy(str(bad).encode(), None, None) == 1
