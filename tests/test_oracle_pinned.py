"""CPU: the oracle in sequential-RNG mode against FULL RUNS of the reference binary whose
`random_number` was fed the same Philox stream (tests/golden/ref_injected.json).  Every integer must
coincide and logZ must agree to round-off: the restatement is pinned line by line, including kNN
clustering, cluster splitting / death and the evidence bookkeeping."""
import numpy as np
import pytest

from tests import oracle_api as orc

BOX = {"gaussian": (None, None), "rastrigin": (-5.12, 5.12), "twin_gaussian": (-1.0, 1.0)}


def _cases(golden):
    return [c for c in golden["ref_injected"] if c["nlike"] < 1300000]   # keep the CPU suite short


def test_oracle_reproduces_injected_reference(golden):
    cases = _cases(golden)
    assert len(cases) >= 6
    for c in cases:
        lo, hi = BOX[c["like"]]
        graded = "grade_dims" in c     # explicit repeats per grade: time_speeds is not called (generate.F90:285-287)
        s = orc.settings(c["nDims"], c["nDerived"], nlive=c["nlive"], num_repeats=c["num_repeats"], seed=c["seed"],
                         batch=1, sequential_rng=1, time_speeds_draw=0 if graded else 1, do_clustering=c["clustering"],
                         nprior=c.get("nprior", -1))
        keep_g = orc.set_grades(s, c["grade_dims"], c["grade_repeats"]) if graded else None
        if "nlives" in c:                                # dynamic nlive: "logL:n,logL:n" (run_time_info.f90:766-779)
            import ctypes as C
            import numpy as np
            pairs = [p.split(":") for p in c["nlives"].split(",")]
            ll = np.array([float(a) for a, _ in pairs]); nl = np.array([int(b) for _, b in pairs], dtype=np.int32)
            s.n_nlives = len(pairs)
            s.loglikes = ll.ctypes.data_as(C.POINTER(C.c_double)); s.nlives = nl.ctypes.data_as(C.POINTER(C.c_int))
        L, P, keep = orc.make_problem(c["like"], c["nDims"], lo, hi)
        o = orc.run(s, L, P)
        assert o["ndead"] == c["ndead"], c
        if graded:                  # .stats prints RTI%nlike per grade; the golden "nlike" is the first of them
            assert o["nlike_grade"][:len(c["nlike_grades"])] == c["nlike_grades"], c
            assert o["nlike"] == sum(c["nlike_grades"]), c
        else:
            assert o["nlike"] == c["nlike"], c
        assert abs(o["logZ"] - c["logZ"]) < 1e-10, c
        assert abs(o["logZerr"] - c["logZerr"]) < 1e-10, c


def test_oracle_farm_mode_reproduces_the_reference_farm(golden):
    """B > 1 pinned to the reference itself.  tests/golden/ref_farm.json: full runs of the reference binary built with -DMPI
    (oracle/Makefile ref_mpi, the container's mpich) under mpiexec with workers + 1 ranks in its synchronous mode
    (nested_sampling.F90:262-286: seeds for all workers from one snapshot, the babies stored at the worker's index, consumed from the
    last worker down, the epoch guard of :313, the updates in the middle of a nursery), every rank's generator fed from its own
    Philox stream.  The oracle's farm mode -- the SAME nursery loop its keyed mode runs for the engine, with the streams arranged as
    the farm's ranks consume them -- must give the same run: integers exact, logZ to round-off, the dead file's columns."""
    cases = golden["ref_farm"]
    assert len(cases) >= 6 and {c["workers"] for c in cases} >= {2, 4, 8} and any(c["clustering"] for c in cases)
    for c in cases:
        lo, hi = BOX[c["like"]]
        s = orc.settings(c["nDims"], c["nDerived"], nlive=c["nlive"], num_repeats=c["num_repeats"], seed=c["seed"], batch=c["workers"],
                         sequential_rng=1, time_speeds_draw=1, do_clustering=c["clustering"], farm=1)
        L, P, keep = orc.make_problem(c["like"], c["nDims"], lo, hi)
        o = orc.run(s, L, P)
        assert o["ndead"] == c["ndead"] and o["nlike"] == c["nlike"], (c["like"], c["workers"], o["ndead"], c["ndead"], o["nlike"], c["nlike"])
        assert abs(o["logZ"] - c["logZ"]) < 1e-10 and abs(o["logZerr"] - c["logZerr"]) < 1e-10, c
        D, nDer = c["nDims"], c["nDerived"]
        logL = o["dead"][:, -1]; birth = o["dead"][:, -2]
        assert abs(logL.sum() - c["dead_logL_sum"]) < 1e-9 * max(1.0, abs(c["dead_logL_sum"]))
        assert abs(birth[birth > -1e29].sum() - c["dead_birth_sum"]) < 1e-9 * max(1.0, abs(c["dead_birth_sum"]))
        for row, ref in ((o["dead"][0], c["dead_first_row"]), (o["dead"][-1], c["dead_last_row"])):    # theta, phi, logL, birth
            mine = np.concatenate([row[D:2 * D + nDer], [row[-1], row[-2]]])
            assert np.allclose(mine, np.array(ref), rtol=1e-12, atol=1e-12), (mine, ref)


def test_oracle_keyed_mode_statistics_against_native_reference(golden):
    """keyed RNG (the engine's layout): logZ is statistically compatible with the untouched
    reference over its 8 seeds (mean within 3 standard errors) and with the analytic truth 0"""
    import numpy as np
    ref = [c for c in golden["ref_native"] if c["like"] == "gaussian" and c["nlive"] == 500]
    assert len(ref) == 8
    zs = []
    for seed in range(4):
        s = orc.settings(20, 2, nlive=500, num_repeats=40, seed=100 + seed, batch=64)
        L, P, keep = orc.make_problem("gaussian", 20)
        o = orc.run(s, L, P)
        zs.append(o["logZ"])
        assert abs(o["logZ"]) < 3 * o["logZerr"]
        assert abs(o["post_mean"][0] - 0.5) < 0.02 and abs(np.sqrt(o["post_var"][0]) - 0.1) < 0.02
    ref_mean = np.mean([c["logZ"] for c in ref]); sig = ref[0]["logZerr"]
    assert abs(np.mean(zs) - ref_mean) < 3 * sig * np.sqrt(1 / 4 + 1 / 8)


def test_posterior_machinery_reproduces_the_reference_binary(golden):
    """R13: clean_phantoms / update_posteriors / write_posterior_file (run_time_info.f90:820-877, :955-1066,
    read_write.F90:479-617).  Every Bernoulli trial of the thinning is a draw of the reference's generator; with the injected
    stream (oracle/ref_rng_shim.c) the oracle in sequential mode reproduces the runs with posteriors / equals switched on
    -- with and without boost_posterior, clustered and not, cluster_posteriors -- exactly: nposterior and nequals as
    integers, <root>.txt and <root>_equal_weights.txt row for row (files written by the reference binary:
    tests/golden/ref_files/p*.txt, oracle/gen_golden.py posteriors)."""
    import os
    box = {"gaussian": (None, None), "rastrigin": (-5.12, 5.12)}
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_files")
    assert len(golden["ref_posteriors"]) >= 5
    for c in golden["ref_posteriors"]:
        lo, hi = box[c["like"]]
        so = orc.settings(c["nDims"], c["nDerived"], nlive=c["nlive"], num_repeats=c["num_repeats"], seed=c["seed"],
                          do_clustering=c["clustering"], sequential_rng=1, time_speeds_draw=1, posteriors=c["posteriors"],
                          equals=c["equals"], boost_posterior=c["boost_posterior"], cluster_posteriors=c["cluster_posteriors"])
        L, P, keep = orc.make_problem(c["like"], c["nDims"], lo, hi)
        o = orc.run(so, L, P)
        assert (o["ndead"], o["nlike"], o["nposterior"], o["nequals"]) == (c["ndead"], c["nlike"], c["nposterior"], c["nequals"]), c["name"]
        assert abs(o["logZ"] - c["logZ"]) < 1e-10
        if c["posteriors"]:
            ref = np.loadtxt(os.path.join(gold, c["name"] + ".txt"))
            w = np.exp(o["post_rows"][:, 0] - o["maxlogweight"])
            mine = np.column_stack([w, -2 * o["post_rows"][:, 1], o["post_rows"][:, 2:]])[w > 0]
            assert mine.shape == ref.shape
            assert (np.abs(mine - ref) / np.maximum(1e-300, np.abs(ref))).max() < 1e-12       # fifteen printed digits
        if c["equals"]:
            ref = np.loadtxt(os.path.join(gold, c["name"] + "_equal_weights.txt"))
            mine = np.column_stack([np.ones(o["nequals"]), o["equal_rows"]])
            assert mine.shape == ref.shape and np.abs(mine - ref).max() < 1e-12 * max(1.0, np.abs(ref).max())
