"""Repeat-sharded merge.

CPU: the numpy checker (tests/replay_oracle.py) against the oracle's evidence; the all-gather plumbing of
polychordlite_amd.merge.gather_records between two processes (gloo): counts, padding, full rows + entry contours.
GPU (-m gpu): the device merge pchip_merge_records against the checker -- evidence, posterior weights, live counts and
posterior moments of the union -- on engine-format records, one run (= the engine's own evidence) and several;
pchip_run_repeats over the visible devices; the merged files."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import oracle_api as orc
from tests.replay_oracle import replay, evidence_replay, lived_records, combined_evidence

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(seed, nlive=60, batch=1):
    s = orc.settings(6, 0, nlive=nlive, num_repeats=12, seed=seed, batch=batch)
    L, P, keep = orc.make_problem("gaussian", 6)
    return orc.run(s, L, P)


def test_replay_equals_engine_evidence_linear_mode():
    o = _run(3)
    lz, var = evidence_replay(*lived_records(o))
    assert abs(lz - o["logZ"]) < 1e-9 and abs(var - o["varlogZ"]) < 1e-9
    r = replay(*lived_records(o), rows=o["dead"], p0=6, nP=6)          # weights and posterior mean of the checker == the oracle's
    assert np.abs(r["logweights"] - o["logweights"]).max() < 1e-9
    assert np.allclose(r["post_mean"], o["post_mean"], atol=1e-9)


def test_merged_runs_shrink_the_error():
    runs = [_run(s) for s in range(4)]
    L = np.concatenate([lived_records(r)[0] for r in runs]); B = np.concatenate([lived_records(r)[1] for r in runs])
    lz, var = evidence_replay(L, B)
    single = np.mean([r["varlogZ"] for r in runs])
    assert var < 0.4 * single                      # ~ 1/4
    assert abs(lz) < 4 * np.sqrt(var) + 0.2        # truth ~ 0 (6-D Gaussian inside the unit box)


def test_combined_evidence_of_runs():
    """the checker of evidence_rule 1: one run gives its own evidence back; equal runs shrink the error like 1/sqrt(R); runs that scatter
    more than they say get the scatter as their error and the LINEAR mean as their evidence"""
    lz, var = combined_evidence([-3.2], [0.04])
    assert abs(lz + 3.2) < 1e-12 and abs(var - 0.04) < 1e-12
    lz, var = combined_evidence([-3.2] * 9, [0.04] * 9)
    assert abs(var - np.log1p(np.expm1(0.04) / 9)) < 1e-12 and abs(lz + 0.5 * var - (-3.2 + 0.02)) < 1e-12
    zs = np.array([-23.9, -23.1, -22.6, -23.4, -22.9, -24.2])
    lz, var = combined_evidence(zs, [0.03] * 6)
    m = np.exp(zs + 0.015)
    assert abs(lz + 0.5 * var - np.log(m.mean())) < 1e-12
    assert abs(var - np.log1p(m.var(ddof=1) / 6 / m.mean() ** 2)) < 1e-12 and var > 0.03 / 6
    assert lz > zs.mean()                            # the mean of the Z, not of the log Z


WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from tests import oracle_api as orc
from tests.replay_oracle import replay
from polychordlite_amd.merge import gather_records, lived_records
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
D = 6
def one(r):
    s = orc.settings(D, 0, nlive=60 + 7 * r, num_repeats=12, seed=10 + r, batch=4)     # ranks hold different numbers of records
    L, P, keep = orc.make_problem("gaussian", D)
    o = orc.run(s, L, P)
    o["entry"] = o["dead"][:, -2].copy()            # engine-format record: rows [cube|theta|phi|birth|logL] + entry contour
    return o
mine = one(rank)
g, ks = gather_records(mine, dist, torch, torch.device("cpu"))
g = g.numpy()
runs = [one(r) for r in range(world)]
rows = np.concatenate([lived_records(x)[0] for x in runs]); entry = np.concatenate([lived_records(x)[1] for x in runs])
assert ks == [lived_records(x)[0].shape[0] for x in runs], ks
assert g.shape == (rows.shape[0], rows.shape[1] + 1)
assert np.array_equal(g[:, :-1], rows) and np.array_equal(g[:, -1], entry)          # every rank holds the identical union
m = replay(g[:, -2], g[:, -1], rows=g[:, :-1], p0=D, nP=D)
ref = replay(rows[:, -1], entry, rows=rows, p0=D, nP=D)
assert m["logZ"] == ref["logZ"] and np.array_equal(m["post_mean"], ref["post_mean"])
singles = [replay(*[a for a in (lived_records(x)[0][:, -1], lived_records(x)[1])]) for x in runs]
assert m["varlogZ"] < 0.75 * np.mean([s["varlogZ"] for s in singles])
assert np.all(np.abs(m["post_mean"] - 0.5) < 0.05)                                   # posterior of the union: theta ~ N(0.5, 0.1)
# with the runs' own weights and evidences behind the records (what pchip_merge_records_ex wants for clustered runs)
g2, ks2, ev = gather_records(mine, dist, torch, torch.device("cpu"), with_own=True)
g2 = g2.numpy()
assert ks2 == ks and np.array_equal(g2[:, :-1], g) and len(ev) == world
own = np.concatenate([x["logweights"][x["logweights"] > -1e29] for x in runs])
assert np.array_equal(g2[:, -1], own)
from polychordlite_amd.merge import clustered
assert ev == [(float(x["logZ"]), float(x["varlogZ"]), clustered(x)) for x in runs] and not any(e[2] for e in ev), ev
if rank == 0:
    print("GATHER_OK", ks, m["logZ"], m["post_mean"][:2])
dist.barrier(); dist.destroy_process_group()
"""


def test_all_gather_of_engine_records_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29611", str(script), ROOT],
                         capture_output=True, text=True, env=env, timeout=600)
    assert "GATHER_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_merge_has_no_cpu_path():
    """without a HIP device the merge fails loudly (return code, no numpy fallback behind it)"""
    from polychordlite_amd import _ctypes_api as api
    from polychordlite_amd import merge as mg
    if api.load().pchip_device_count() > 0:
        pytest.skip("a GPU is visible")
    o = _run(3)
    rows, entry = mg.lived_records(o)
    with pytest.raises(RuntimeError):
        mg.merge_records(6, 0, [rows.shape[0]], rows, entry)


def test_a_rank_learns_that_rank_0_has_no_communicator_id():
    """rank 0 could not make the RCCL id (library not loadable): what it broadcasts is empty, and every other rank raises instead of
    waiting in ncclCommInitRank for a peer that will never come"""
    from polychordlite_amd import merge as mg
    with pytest.raises(RuntimeError, match="rank 0"):
        mg.Comm(1, 2, 0, exchange=lambda b: b"")


# ------------------------------------------------------------------------------------------------ GPU
def _engine_runs(api, seeds, D=6, nDer=1, nlive=150, nr=12, batch=50, kind="gaussian", clustering=0, box=(None, None)):
    lib = api.load()
    s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, nDer)
    s.nlive, s.num_repeats, s.batch, s.do_clustering = nlive, nr, batch, clustering
    L, P, keep = api.make_problem(kind, D, nDer, *box)
    runs = []
    for sd in seeds:
        s.seed = sd
        runs.append(api.run(s, L, P))
    return s, L, P, keep, runs


@pytest.mark.gpu
@pytest.mark.parametrize("nruns", [1, 2, 5])
def test_device_merge_matches_the_checker(engine, nruns):
    from polychordlite_amd import merge as mg
    D, nDer = 6, 1
    s, L, P, keep, runs = _engine_runs(engine, [40 + k for k in range(nruns)])
    recs = [mg.lived_records(r) for r in runs]
    rows = np.concatenate([a for a, _ in recs]); entry = np.concatenate([b for _, b in recs])
    m = mg.merge_records(D, nDer, [a.shape[0] for a, _ in recs], rows, entry, want_rows=True)
    ref = replay(rows[:, -1], entry, rows=rows, p0=D, nP=D + nDer)
    assert m["records"] == rows.shape[0] and m["n_runs"] == nruns
    assert abs(m["logZ"] - ref["logZ"]) < 1e-9 and abs(m["varlogZ"] - ref["varlogZ"]) < 1e-9
    assert np.array_equal(m["nlive"], ref["nlive"])
    assert np.abs(m["logweights"] - ref["logweights"]).max() < 1e-9
    assert np.allclose(m["post_mean"], ref["post_mean"], atol=1e-11) and np.allclose(m["post_var"], ref["post_var"], atol=1e-11)
    mr = m["rows"]
    assert np.all(np.diff(mr[:, -1]) >= 0)                                  # merged death order
    assert np.array_equal(np.sort(mr[:, -1]), np.sort(rows[:, -1]))
    srt = rows[ref["order"]]
    assert np.array_equal(mr[:, :-2], srt[:, :-2])                          # the same points, row for row
    assert np.array_equal(mr[:, -2], entry[ref["order"]])                   # birth column = entry contour
    if nruns == 1:                                                          # one run: the union IS the run
        g = runs[0]
        assert abs(m["logZ"] - g["logZ"]) < 1e-7 and abs(m["varlogZ"] - g["varlogZ"]) < 1e-7
        lw = g["logweights"][g["logweights"] > -1e29]
        assert np.abs(m["logweights"] - lw).max() < 1e-8
        assert np.allclose(m["post_mean"], g["post_mean"], atol=1e-9)
    else:
        assert m["varlogZ"] < 1.3 * np.mean([r["varlogZ"] for r in runs]) / nruns


@pytest.mark.gpu
def test_device_merge_of_clustered_and_ragged_runs(engine):
    """runs of different lengths, one of them clustered with dynamic live counts inside (the union does not care how a
    run arrived at its death sequence), one with a single record, one empty"""
    from polychordlite_amd import merge as mg
    api = engine
    _, _, _, _, a = _engine_runs(api, [3], D=2, nDer=0, nlive=300, nr=6, batch=40, kind="rastrigin", clustering=1, box=(-5.12, 5.12))
    _, _, _, _, b = _engine_runs(api, [4, 5], D=2, nDer=0, nlive=80, nr=6, batch=1, kind="rastrigin", box=(-5.12, 5.12))
    recs = [mg.lived_records(r) for r in (a[0], b[0], b[1])]
    one_row = (recs[1][0][-1:].copy(), np.array([-1e30]))                   # a "run" of one point that lived from the start
    empty = (np.zeros((0, 6)), np.zeros(0))
    parts = [recs[0], empty, recs[1], one_row, recs[2]]
    rows = np.concatenate([p[0] for p in parts]); entry = np.concatenate([p[1] for p in parts])
    m = mg.merge_records(2, 0, [p[0].shape[0] for p in parts], rows, entry)
    ref = replay(rows[:, -1], entry, rows=rows, p0=2, nP=2)
    assert abs(m["logZ"] - ref["logZ"]) < 1e-9 and np.array_equal(m["nlive"], ref["nlive"])
    assert np.allclose(m["post_mean"], ref["post_mean"], atol=1e-11)
    assert abs(m["logZ"] - 2 * (-2.326314)) < 4 * m["logZerr"] + 0.05


@pytest.mark.gpu
def test_union_of_clustered_runs_quotes_the_runs_own_evidence(engine, tmp_path):
    """sixteen 10-D Rastrigin runs (nlive 300, kNN clustering: dozens of clusters each).  A run with clusters weighs a dead point by its
    CLUSTER's volume (run_time_info.f90:211-296, :458-503); the replay of the union from ranks and live counts does not know them and sits
    far below the runs' own evidences (round 4: 0.46 at BASELINE configs[2], twenty of its own error bars).  The union must quote the runs'
    own evidences combined in linear space (evidence_rule 1), weigh every record by its own weight over the number of runs, say so in its
    .stats file, and keep the replay beside it."""
    from polychordlite_amd import merge as mg
    from polychordlite_amd.repeats import run_repeats
    from polychordlite_amd.pypolychord.output import PolyChordOutput
    api = engine
    lib = api.load()
    D, R = 10, 16
    s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, 0)
    s.nlive, s.num_repeats, s.do_clustering = 300, 30, 1
    L, P, keep = api.make_problem("rastrigin", D, 0, -5.12, 5.12)
    merged, runs = run_repeats(s, L, P, [600 + k for k in range(R)], max_in_flight=R, want_rows=True, write=(str(tmp_path), "u"))
    assert merged["n_runs"] == R and merged["evidence_rule"] == 1 and merged["nclustered"] == sum(mg.clustered(r) for r in runs) >= R // 2
    lz, var = combined_evidence([r["logZ"] for r in runs], [r["varlogZ"] for r in runs])
    assert abs(merged["logZ"] - lz) < 1e-10 and abs(merged["varlogZ"] - var) < 1e-10
    zs = np.array([r["logZ"] for r in runs])
    assert abs(merged["runs_logZ_mean"] - zs.mean()) < 1e-12
    # within 3 sigma of the runs' own mean -- and the replay is not (that is what this rule is for)
    assert abs(merged["logZ"] - merged["runs_logZ_mean"]) < 3.0 * np.hypot(merged["logZerr"], merged["runs_logZ_sem"])
    assert merged["logZerr"] >= np.sqrt(np.mean([r["varlogZ"] for r in runs]) / R) * 0.99          # never below the propagated error
    recs = [mg.lived_records(r) for r in runs]
    rows = np.concatenate([a for a, _ in recs]); entry = np.concatenate([b for _, b in recs])
    ref = replay(rows[:, -1], entry, rows=rows, p0=D, nP=D)
    assert abs(merged["logZ_replay"] - ref["logZ"]) < 1e-9 and abs(merged["varlogZ_replay"] - ref["varlogZ"]) < 1e-9
    assert np.array_equal(merged["nlive"], ref["nlive"])
    assert merged["logZ_replay"] < merged["runs_logZ_mean"] - 3.0 * merged["logZerr_replay"]
    # weights: the runs' own, over the number of runs, in merged death order
    own = np.concatenate([r["logweights"][r["logweights"] > -1e29] for r in runs]) - np.log(R)
    order = ref["order"]
    assert np.array_equal(merged["rows"][:, -1], rows[order, -1])
    assert np.abs(merged["logweights"] - own[order]).max() < 1e-12
    lp = merged["logweights"] + merged["rows"][:, -1]
    # sum of the posterior weights = mean over the runs of <Z_r>
    assert abs(np.logaddexp.reduce(lp) - np.log(np.mean(np.exp(zs + 0.5 * np.array([r["varlogZ"] for r in runs]))))) < 1e-6
    w = np.exp(lp - lp.max())
    x = merged["rows"][:, D:2 * D]
    mean = (w[:, None] * x).sum(0) / w.sum()
    assert np.allclose(merged["post_mean"], mean, atol=1e-11) and np.allclose(merged["post_var"], (w[:, None] * x * x).sum(0) / w.sum() - mean ** 2, atol=1e-11)
    assert np.all(np.abs(merged["post_mean"]) < 1.0)              # Rastrigin: symmetric about 0, modes one unit apart
    # the files: PolyChordOutput reads the quoted evidence, the file says which it is and keeps the replay
    out = PolyChordOutput(str(tmp_path), "u")
    assert abs(out.logZ - merged["logZ"]) < 1e-12 and abs(out.logZerr - merged["logZerr"]) < 1e-12
    txt = open(tmp_path / "u.stats").read()
    assert "evidence rule 1" in txt and "replay of the union" in txt and ("%d of the runs held" % merged["nclustered"]) in txt
    post = np.loadtxt(tmp_path / "u.txt")
    assert np.allclose((post[:, 0][:, None] * post[:, 2:2 + D]).sum(0) / post[:, 0].sum(), merged["post_mean"], atol=1e-9)
    # the same union through the records interface (what a gloo / MPI transport feeds): identical
    m2 = mg.merge_records(D, 0, [a.shape[0] for a, _ in recs], rows, entry, ownw=own + np.log(R), run_logZ=zs,
                          run_varlogZ=[r["varlogZ"] for r in runs], run_clustered=[mg.clustered(r) for r in runs])
    assert m2["logZ"] == merged["logZ"] and m2["evidence_rule"] == 1 and np.array_equal(m2["logweights"], merged["logweights"])
    # and without what the runs know about themselves it is the replay, as before
    m3 = mg.merge_records(D, 0, [a.shape[0] for a, _ in recs], rows, entry)
    assert m3["evidence_rule"] == 0 and m3["logZ"] == merged["logZ_replay"]
    # one clustered run merged alone gives its own evidence back
    one = mg.comm_merge(runs[0], None, D, 0)
    if mg.clustered(runs[0]):
        assert one["evidence_rule"] == 1 and abs(one["logZ"] - runs[0]["logZ"]) < 1e-12 and abs(one["varlogZ"] - runs[0]["varlogZ"]) < 1e-12
        lw = runs[0]["logweights"]
        assert np.array_equal(one["logweights"], lw[lw > -1e29])


@pytest.mark.gpu
def test_run_repeats_front_door_and_files(engine, tmp_path):
    """polychordlite_amd.repeats.run_repeats(devices=[...]): the repeats are the runs they would be one after the other, the
    union comes from the device merge, and its files read like a single run's (PolyChordOutput parses <root>.stats)"""
    from polychordlite_amd.repeats import run_repeats
    from polychordlite_amd.pypolychord.output import PolyChordOutput
    api = engine
    ndev = api.load().pchip_device_count()
    s, L, P, keep, singles = _engine_runs(api, [11, 12, 13, 14, 15, 16])
    merged, runs = run_repeats(s, L, P, [11, 12, 13, 14, 15, 16], max_in_flight=3, devices=list(range(ndev)),
                               want_rows=True, write=(str(tmp_path), "u"))
    for one, r in zip(singles, runs):               # several runs in flight on one host thread: each is bit for bit its solo run
        assert one["ndead"] == r["ndead"] and one["nlike"] == r["nlike"] and one["logZ"] == r["logZ"]
        assert np.array_equal(one["dead"], r["dead"]) and np.array_equal(one["logweights"], r["logweights"]) and np.array_equal(one["live"], r["live"])
    assert merged["n_runs"] == 6 and merged["nlike"] == sum(r["nlike"] for r in runs)
    assert merged["evidence_rule"] == 0 and merged["logZ_replay"] == merged["logZ"]     # one-cluster runs: the replay of the union, as before
    rows = np.concatenate([r["dead"][r["logweights"] > -1e29] for r in runs]); entry = np.concatenate([r["entry"][r["logweights"] > -1e29] for r in runs])
    ref = replay(rows[:, -1], entry, rows=rows, p0=6, nP=7)
    assert abs(merged["logZ"] - ref["logZ"]) < 1e-9 and np.allclose(merged["post_mean"], ref["post_mean"], atol=1e-11)
    errs = [r["logZerr"] for r in runs]
    assert 0.3 * np.mean(errs) < merged["logZerr"] < 0.55 * np.mean(errs)          # ~ 1 / sqrt(6)
    assert abs(merged["logZ"]) < 4 * merged["logZerr"]                             # truth 0
    assert np.all(np.abs(merged["post_mean"][:6] - 0.5) < 0.02)
    out = PolyChordOutput(str(tmp_path), "u")
    assert abs(out.logZ - merged["logZ"]) < 1e-12 and abs(out.logZerr - merged["logZerr"]) < 1e-12
    db = np.loadtxt(tmp_path / "u_dead-birth.txt")
    assert db.shape == (merged["records"], 6 + 1 + 2)
    assert np.allclose(db[:, :7], merged["rows"][:, 6:13], rtol=1e-14) and np.allclose(db[:, 7], merged["rows"][:, -1], rtol=1e-14)
    lz, var = evidence_replay(db[:, 7], db[:, 8])                                  # the file alone reproduces the evidence
    assert abs(lz - merged["logZ"]) < 1e-9
    post = np.loadtxt(tmp_path / "u.txt")
    w = post[:, 0]
    assert np.allclose((w[:, None] * post[:, 2:8]).sum(0) / w.sum(), merged["post_mean"][:6], atol=1e-9)


WORKER_GPU = r"""
import os, sys, hashlib
sys.path.insert(0, sys.argv[1])
out_dir = sys.argv[2]
import ctypes as C
import numpy as np, torch, torch.distributed as dist
from tests.replay_oracle import replay
from polychordlite_amd import _ctypes_api as api
from polychordlite_amd import merge as mg
from polychordlite_amd.pypolychord.output import PolyChordOutput
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
lib = api.load()
assert lib.pchip_device_count() >= 1
D, nDer = 6, 1
s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, nDer)
s.nlive, s.num_repeats, s.batch, s.seed, s.device = 150 + 20 * rank, 12, 50, 70 + rank, 0      # both ranks on device 0, ragged record counts
L, P, keep = api.make_problem("gaussian", D, nDer)
mine = api.run(s, L, P)                                             # an ENGINE run on this rank
g, ks = mg.gather_records(mine, dist, torch, torch.device("cpu"))   # counts + one padded all-gather (gloo)
g = g.numpy()
rows, entry = np.ascontiguousarray(g[:, :-1]), np.ascontiguousarray(g[:, -1])
assert ks[rank] == int((mine["logweights"] > s.logzero).sum()) and sum(ks) == g.shape[0] and ks[0] != ks[1]
m = mg.merge_records(D, nDer, ks, rows, entry, want_rows=True, write=(out_dir, "u%d" % rank))      # device merge, on every rank
ref = replay(rows[:, -1], entry, rows=rows, p0=D, nP=D + nDer)
assert abs(m["logZ"] - ref["logZ"]) < 1e-9 and abs(m["varlogZ"] - ref["varlogZ"]) < 1e-9
assert np.array_equal(m["nlive"], ref["nlive"]) and np.allclose(m["post_mean"], ref["post_mean"], atol=1e-11)
assert m["n_runs"] == world and m["records"] == g.shape[0]
assert m["logZerr"] < 0.85 * mine["logZerr"] and abs(m["logZ"]) < 4 * m["logZerr"] + 0.05
# the library's own single-process path (records picked on the device, no collective) gives this rank's run back
own = mg.comm_merge(mine, None, D, nDer, want_rows=True)
r1, e1 = mg.lived_records(mine)
chk = mg.merge_records(D, nDer, [r1.shape[0]], r1, e1, want_rows=True)
assert own["records"] == ks[rank] and own["logZ"] == chk["logZ"] and np.array_equal(own["rows"], chk["rows"]) and np.array_equal(own["logweights"], chk["logweights"])
assert abs(own["logZ"] - mine["logZ"]) < 1e-7
# every rank holds the identical union and the identical merged result
dig = (hashlib.sha256(g.tobytes()).hexdigest(), m["logZ"], m["varlogZ"], m["post_mean"].tolist(), hashlib.sha256(m["rows"].tobytes()).hexdigest())
box = [None] * world
dist.all_gather_object(box, dig)
assert all(b == box[0] for b in box), box
dist.barrier()
# clustered runs on both ranks: the records travel with their own weights and the runs' evidences, the union follows evidence_rule 1
from tests.replay_oracle import combined_evidence
s2 = api.Settings(); lib.pchip_settings_default(C.byref(s2), 3, 0)
s2.nlive, s2.num_repeats, s2.do_clustering, s2.seed, s2.device = 200, 9, 1, 90 + rank, 0
L2, P2, keep2 = api.make_problem("rastrigin", 3, 0, -5.12, 5.12)
mine2 = api.run(s2, L2, P2)
assert mg.clustered(mine2)
g2, ks2, ev = mg.gather_records(mine2, dist, torch, torch.device("cpu"), with_own=True)
g2 = g2.numpy()
m2 = mg.merge_records(3, 0, ks2, np.ascontiguousarray(g2[:, :-2]), np.ascontiguousarray(g2[:, -2]), ownw=np.ascontiguousarray(g2[:, -1]),
                      run_logZ=[e[0] for e in ev], run_varlogZ=[e[1] for e in ev], run_clustered=[e[2] for e in ev])
lz, var = combined_evidence([e[0] for e in ev], [e[1] for e in ev])
assert m2["evidence_rule"] == 1 and m2["nclustered"] == 2 and abs(m2["logZ"] - lz) < 1e-10 and abs(m2["varlogZ"] - var) < 1e-10
assert ev[rank][0] == mine2["logZ"]
box2 = [None] * world
dist.all_gather_object(box2, (m2["logZ"], m2["varlogZ"], m2["logZ_replay"], m2["post_mean"].tolist(), hashlib.sha256(m2["logweights"].tobytes()).hexdigest()))
assert all(b == box2[0] for b in box2), box2
dist.barrier()
if rank == 0:
    a, b = PolyChordOutput(out_dir, "u0"), PolyChordOutput(out_dir, "u1")
    assert a.logZ == b.logZ and abs(a.logZ - m["logZ"]) < 1e-12
    for suffix in (".stats", "_dead-birth.txt", ".txt"):
        assert open(os.path.join(out_dir, "u0" + suffix)).read() == open(os.path.join(out_dir, "u1" + suffix)).read(), suffix
    print("MERGE2_OK", ks, m["logZ"], m["logZerr"])
dist.barrier(); dist.destroy_process_group()
"""


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_merge_engine_records(engine, tmp_path):
    """the N > 1 path with ENGINE runs on a 1-GPU box: two ranks share device 0 (RCCL refuses two ranks on one GPU, so
    the records travel over gloo), each merges the union on the device; both must hold the same union, the merged
    evidence must equal the numpy checker's, and the merged files must be identical"""
    script = tmp_path / "worker_gpu.py"
    script.write_text(WORKER_GPU)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29613", str(script), ROOT, str(tmp_path)],
                         capture_output=True, text=True, env=env, timeout=900)
    assert "MERGE2_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


@pytest.mark.gpu
def test_library_rccl_exchange_one_rank(engine):
    """pchip_comm_*: librccl.so resolved by the library itself, a communicator of one rank, the run's records through
    ncclAllGather (counts, then the padded block) and the un-padding kernel -- the result must be the plain device merge
    of the same records, bit for bit"""
    from polychordlite_amd import merge as mg
    s, L, P, keep, runs = _engine_runs(engine, [31], nlive=120)
    run = runs[0]
    comm = mg.Comm(0, 1, 0)
    try:
        assert comm.library and "rccl" in comm.library
        a = mg.comm_merge(run, comm, 6, 1, want_rows=True)
    finally:
        comm.close()
    r1, e1 = mg.lived_records(run)
    b = mg.merge_records(6, 1, [r1.shape[0]], r1, e1, want_rows=True)
    assert a["records"] == b["records"] == r1.shape[0] and a["n_runs"] == 1
    assert a["logZ"] == b["logZ"] and a["varlogZ"] == b["varlogZ"]
    assert np.array_equal(a["rows"], b["rows"]) and np.array_equal(a["logweights"], b["logweights"]) and np.array_equal(a["nlive"], b["nlive"])
    assert a["nlike"] == run["nlike"] and a["ndead_all"] == run["ndead"]
    assert abs(a["logZ"] - run["logZ"]) < 1e-7
    assert a["runs_logZ_mean"] == run["logZ"] and abs(a["runs_logZ_sem"] - run["logZerr"]) < 1e-12      # (the run's own evidence travelled with the counts)
    # a rank that holds several runs (the runs of a GPU in step): header, six words per run, one padded block with the runs one after the other
    s3, L3, P3, keep3, runs3 = _engine_runs(engine, [32, 33, 34], nlive=120)
    comm = mg.Comm(0, 1, 0)
    try:
        c = mg.comm_merge_many(runs3, comm, 6, 1, want_rows=True)
    finally:
        comm.close()
    recs = [mg.lived_records(r) for r in runs3]
    d = mg.merge_records(6, 1, [x.shape[0] for x, _ in recs], np.concatenate([x for x, _ in recs]), np.concatenate([e for _, e in recs]), want_rows=True)
    assert c["n_runs"] == 3 and c["records"] == d["records"] and c["logZ"] == d["logZ"] and c["varlogZ"] == d["varlogZ"]
    assert np.array_equal(c["rows"], d["rows"]) and np.array_equal(c["logweights"], d["logweights"]) and np.array_equal(c["nlive"], d["nlive"])
    assert c["nlike"] == sum(r["nlike"] for r in runs3) and abs(c["runs_logZ_mean"] - np.mean([r["logZ"] for r in runs3])) < 1e-12
    e = mg.comm_merge_many(runs3, None, 6, 1, want_rows=True)              # no communicator: the same code without the collective
    assert e["logZ"] == c["logZ"] and np.array_equal(e["rows"], c["rows"])


class _ThreadGather:
    """an all-gather between THREADS of this process whose ranks share a GPU: every rank posts its send block, all meet, each copies every
    block into its own receive buffer device-to-device (rank after rank: the layout ncclAllGather delivers), all meet again"""

    def __init__(self, world):
        import threading
        from polychordlite_amd import merge as mg
        self.world, self.bar, self.post, self.hip, self.calls = world, threading.Barrier(world, timeout=120), [None] * world, mg.hip_runtime(), 0

    def rank(self, r):
        def all_gather(send, recv, nbytes):
            self.post[r] = (send, nbytes)
            self.bar.wait()
            for q in range(self.world):
                assert self.post[q][1] == nbytes                       # every rank came with the same block size (padded to the largest)
                if self.hip.hipMemcpy(recv + q * nbytes, self.post[q][0], nbytes, 3) != 0:
                    return 2
            if r == 0:
                self.calls += 1
            self.bar.wait()
            return 0
        return all_gather


def _ranks_in_threads(world, body):
    """body(rank) on `world` threads; returns their results (an exception object where one was raised); a rank that hangs fails the test"""
    import threading
    out = [None] * world

    def run(r):
        try:
            out[r] = body(r)
        except Exception as e:      # noqa: BLE001
            out[r] = e
    th = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in th: t.start()
    for t in th: t.join(300)
    assert not any(t.is_alive() for t in th), "a rank is still waiting in the exchange"
    return out


@pytest.mark.gpu
def test_exchange_path_with_two_ranks_on_one_gpu(engine):
    """pchip_comm_merge_many with nranks = 2 -- the path no test had ever run with more than one rank (RCCL forms no communicator with two
    ranks on one device, and the box has one): header all-gather, six words per run, the status word, ONE padded block per rank
    [nmax][nTotal] | entry | own log weight, k_unpad, the device merge -- every statement of it, over a communicator whose all-gather is
    the caller's (pchip_comm_create_with; here: two threads, device-to-device copies in ncclAllGather's layout).  Ragged: rank 0 holds two
    runs (their records left on the device: settings.device_records), rank 1 one CLUSTERED run with three times the records (host arrays,
    packed by the library) -- every rank must get, bit for bit, pchip_merge_records_ex of the union in rank order."""
    from polychordlite_amd import merge as mg
    api = engine
    lib = api.load()
    D, nDer = 2, 0
    s0 = api.Settings(); lib.pchip_settings_default(C.byref(s0), D, nDer)
    L, P, keep = api.make_problem("rastrigin", D, nDer, -5.12, 5.12)
    def make(nlive, batch, clus, seed, on_device):
        s0.nlive, s0.num_repeats, s0.batch, s0.do_clustering, s0.seed, s0.device_records = nlive, 6, batch, clus, seed, on_device
        return api.run(s0, L, P)
    mine = [[make(80, 1, 0, 4, 1), make(120, 8, 0, 5, 1)], [make(300, 40, 1, 3, 0)]]
    assert mine[0][0]["n_records"] and mine[1][0]["n_records"] is None and mine[1][0]["ncluster_peak"] > 1
    tg = _ThreadGather(2)
    comms = [mg.CallbackComm(r, 2, 0, tg.rank(r)) for r in range(2)]
    try:
        got = _ranks_in_threads(2, lambda r: mg.comm_merge_many(mine[r], comms[r], D, nDer, want_rows=True))
    finally:
        for c in comms: c.close()
    for g in got:
        assert not isinstance(g, Exception), g
    assert tg.calls == 4                                              # header, counts, status, records
    union = mine[0] + mine[1]
    recs = [mg.lived_records(r) for r in union]
    ownw = np.concatenate([r["logweights"][r["logweights"] > r["logzero"]] for r in union])
    ref = mg.merge_records(D, nDer, [x.shape[0] for x, _ in recs], np.concatenate([x for x, _ in recs]), np.concatenate([e for _, e in recs]), want_rows=True,
                           ownw=ownw, run_logZ=[r["logZ"] for r in union], run_varlogZ=[r["varlogZ"] for r in union], run_clustered=[mg.clustered(r) for r in union])
    assert ref["evidence_rule"] == 1 and recs[2][0].shape[0] > recs[0][0].shape[0] + recs[1][0].shape[0]      # (ragged: rank 1's block sets the padding)
    for g in got:
        assert g["n_runs"] == 3 and g["records"] == ref["records"] and g["evidence_rule"] == 1
        assert g["logZ"] == ref["logZ"] and g["varlogZ"] == ref["varlogZ"] and g["logZ_replay"] == ref["logZ_replay"]
        assert np.array_equal(g["rows"], ref["rows"]) and np.array_equal(g["logweights"], ref["logweights"]) and np.array_equal(g["nlive"], ref["nlive"])
        assert np.array_equal(g["post_mean"], ref["post_mean"])
        assert g["nlike"] == sum(r["nlike"] for r in union) and g["ndead_all"] == sum(r["ndead"] for r in union)


@pytest.mark.gpu
def test_a_failed_rank_ends_the_exchange_on_every_rank(engine):
    """a rank that cannot pack its records (here: rows of the wrong width) still takes part in the header all-gather and says so there
    (count -1): EVERY rank returns an error, none is left waiting in the next collective -- with two real ranks; and a rank whose
    all-gather itself fails takes its error code back"""
    from polychordlite_amd import merge as mg
    api = engine
    _, _, _, _, good = _engine_runs(api, [7], D=2, nDer=0, nlive=80, nr=6, batch=8, kind="rastrigin", box=(-5.12, 5.12))
    _, _, _, _, wide = _engine_runs(api, [8], nlive=80)                                   # 6-D + 1 derived: 15 columns, not 6
    mine = [good, wide]
    tg = _ThreadGather(2)
    comms = [mg.CallbackComm(r, 2, 0, tg.rank(r)) for r in range(2)]
    try:
        got = _ranks_in_threads(2, lambda r: mg.comm_merge_many(mine[r], comms[r], 2, 0))
    finally:
        for c in comms: c.close()
    assert all(isinstance(g, RuntimeError) for g in got), got
    assert tg.calls == 1                                              # the header's all-gather, and no collective after it
    # the collective itself fails on one rank (its transport): that rank returns the error; one rank only, so nobody waits
    bad = mg.CallbackComm(0, 1, 0, lambda send, recv, n: 5)
    try:
        with pytest.raises(RuntimeError):
            mg.comm_merge_many(good, bad, 2, 0)
    finally:
        bad.close()


@pytest.mark.gpu
def test_lived_records_follow_the_runs_logzero(engine):
    """failed spawns carry logweight = settings.logzero, whatever it is: with logzero = -1e20 the Python selection and the
    library's must both drop them (a hard-coded -1e29 kept them and handed the merge a sequence that does not ascend)"""
    from polychordlite_amd import merge as mg
    lib = engine.load()
    s = engine.Settings(); lib.pchip_settings_default(C.byref(s), 6, 1)
    s.nlive, s.num_repeats, s.batch, s.seed, s.logzero = 150, 12, 75, 5, -1e20
    L, P, keep = engine.make_problem("gaussian", 6, 1)
    run = engine.run(s, L, P)
    assert run["logzero"] == -1e20 and (run["logweights"] <= -1e20).sum() > 0           # this run has failed spawns
    rows, entry = mg.lived_records(run)
    assert rows.shape[0] == int((run["logweights"] > -1e20).sum()) and np.all(np.diff(rows[:, -1]) >= 0)
    a = mg.comm_merge(run, None, 6, 1)
    assert a["records"] == rows.shape[0] and abs(a["logZ"] - run["logZ"]) < 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize("kind,D,nDer,nlive,nr,clus,box", [("gaussian", 20, 2, 400, 20, 0, None), ("gaussian", 7, 1, 300, 14, 0, (-0.25, 1.5)),
                                                            ("rastrigin", 3, 0, 200, 9, 1, (-5.12, 5.12)), ("twin_gaussian", 6, 1, 250, 12, 0, (-1.0, 1.0)),
                                                            ("gaussian", 24, 0, 256, 24, 0, None), ("gaussian", 3, 4, 200, 6, 0, (0.1, 0.9)),
                                                            # the clustered BASELINE shapes, smaller: 10-D Rastrigin (dozens of clusters), 30-D twin Gaussian
                                                            # (k_nhats_q + the plain k_slice), twin Gaussian with a split, Rastrigin without clustering
                                                            ("rastrigin", 10, 0, 300, 30, 1, (-5.12, 5.12)), ("twin_gaussian", 30, 1, 120, 40, 1, (-1.0, 1.0)),
                                                            ("twin_gaussian", 6, 1, 250, 12, 1, (-1.0, 1.0)), ("rastrigin", 4, 0, 200, 12, 0, (-5.12, 5.12))])
def test_runs_in_step_are_their_solo_runs(engine, kind, D, nDer, nlive, nr, clus, box):
    """pchip_run_repeats with all runs of the device in flight: they go round by round together on one stream, every kernel of a
    round launched once for all of them (Gaussian: the lane-per-chain kernels, fused update, pool compaction for all at once;
    the other likelihoods and clustered runs: k_slice / k_nhats_q / k_nn_lists / k_consume_cl / the row copies with the run in the
    grid, the updates with clustering by each engine in between) -- each run bit for bit the run it is alone, whatever round the
    others update, split, compact or end in.  The first seed is also the ORACLE's run of these settings (tests.oracle_api)."""
    from polychordlite_amd.repeats import run_repeats
    api = engine
    lib = api.load()
    seeds = [31, 32, 33, 34, 35, 36, 37, 38, 39]
    L, P, keep = api.make_problem(kind, D, nDer, *box) if box else api.make_problem(kind, D, nDer)
    def settings(seed):
        s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, nDer)
        s.nlive, s.num_repeats, s.seed, s.do_clustering = nlive, nr, seed, clus
        return s
    singles = [api.run(settings(sd), L, P) for sd in seeds]
    merged, runs = run_repeats(settings(0), L, P, seeds, max_in_flight=len(seeds))
    assert len({r["nrounds"] for r in singles}) > 1 or kind != "gaussian"      # (they do not all end in the same round)
    for one, r in zip(singles, runs):
        for k in ("ndead", "nlike", "niter", "nupdates", "nbatches", "ncluster_dead"):
            assert one[k] == r[k], (k, one[k], r[k])
        assert one["logZ"] == r["logZ"] and one["logZerr"] == r["logZerr"]
        assert np.array_equal(one["dead"], r["dead"], equal_nan=True) and np.array_equal(one["logweights"], r["logweights"]) and np.array_equal(one["live"], r["live"], equal_nan=True)
        assert np.array_equal(one["post_mean"], r["post_mean"], equal_nan=True)
    assert merged["n_runs"] == len(seeds)
    # the runs' own evidences beside the union's (the one to quote where runs have many clusters: DESIGN section 8)
    zs = np.array([r["logZ"] for r in runs])
    assert abs(merged["runs_logZ_mean"] - zs.mean()) < 1e-12 and abs(merged["runs_logZ_sem"] - zs.std(ddof=1) / np.sqrt(zs.size)) < 1e-12
    # one of the runs in step next to the oracle (same Philox keys: the same trajectory; nDims <= 7 whole runs, beyond that the first
    # generations -- round-off grows with every covariance update, and clusters of fewer points than dimensions hang on the sign of
    # a 1e-19 Cholesky pivot: DESIGN section 7)
    from tests import oracle_api as orc
    g = runs[0]
    so = orc.settings(D, nDer, nlive=nlive, num_repeats=nr, seed=seeds[0], batch=g["batch"], do_clustering=clus)
    Lo, Po, keep2 = orc.make_problem(kind, D, *(box if box else (None, None)))
    o = orc.run(so, Lo, Po)
    if D <= 7:
        for k in ("ndead", "nlike", "niter", "nbatches", "ncluster", "ncluster_dead"):
            assert g[k] == o[k], (k, g[k], o[k])
        assert abs(g["logZ"] - o["logZ"]) < 1e-8 and abs(g["logZerr"] - o["logZerr"]) < 1e-8
        rel = np.abs(g["dead"] - o["dead"]) / np.maximum(1.0, np.abs(o["dead"]))
        assert rel.max() < 1e-7
    else:
        n0 = min(6 * nlive, int(g["ndead"]), int(o["ndead"]))
        rel = np.abs(g["dead"][:n0] - o["dead"][:n0]) / np.maximum(1.0, np.abs(o["dead"][:n0]))
        assert rel.max() < 1e-7
        assert abs(g["logZ"] - o["logZ"]) < 3.0 * (g["logZerr"] + o["logZerr"])


@pytest.mark.gpu
@pytest.mark.parametrize("D", list(range(1, 25)))
def test_runs_in_step_at_every_width(engine, D):
    """every compiled width of the lane-per-chain kernels (nDims 1 .. 24) with 0 .. 3 derived parameters, unit and other prior boxes,
    nurseries that do not fill their last wavefront: five runs in step, each bit for bit the run it is alone.  (These shapes caught a
    cohort that compacted its phantom arrays a round early: the update's partial sums are grouped by the array's extent.)"""
    from polychordlite_amd.repeats import run_repeats
    api = engine
    lib = api.load()
    nDer = D % 4
    box = None if D % 2 else (-0.5 + 0.01 * D, 1.25)
    nlive = 100 + 37 * (D % 5)
    nr = max(2, 2 * D if D < 8 else D + (D % 3))
    L, P, keep = api.make_problem("gaussian", D, nDer, *box) if box else api.make_problem("gaussian", D, nDer)
    def settings(seed):
        s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, nDer)
        s.nlive, s.num_repeats, s.seed = nlive, nr, seed
        return s
    seeds = [900 + D * 10 + j for j in range(5)]
    singles = [api.run(settings(sd), L, P) for sd in seeds]
    merged, runs = run_repeats(settings(0), L, P, seeds, max_in_flight=len(seeds))
    for one, r in zip(singles, runs):
        for k in ("ndead", "nlike", "niter", "nupdates", "nbatches"):
            assert one[k] == r[k], (k, one[k], r[k])
        assert one["logZ"] == r["logZ"] and one["logZerr"] == r["logZerr"]
        assert np.array_equal(one["dead"], r["dead"], equal_nan=True) and np.array_equal(one["logweights"], r["logweights"]) and np.array_equal(one["live"], r["live"], equal_nan=True)
        assert np.array_equal(one["post_mean"], r["post_mean"], equal_nan=True)
    # the first of the runs in step next to the ORACLE's run of these settings
    from tests import oracle_api as orc
    from tests.test_gpu_parity import _next_to_the_oracle
    g = runs[0]
    so = orc.settings(D, nDer, nlive=nlive, num_repeats=nr, seed=seeds[0], batch=g["batch"])
    Lo, Po, keep2 = orc.make_problem("gaussian", D, *(box if box else (None, None)))
    _next_to_the_oracle(g, orc.run(so, Lo, Po), D, nlive)


@pytest.mark.gpu
def test_bases_with_their_own_deviates_are_the_same_bases(engine):
    """settings.ablate bit 12: runs in step whose Gram-Schmidt kernel makes its own deviates (k_bases_own: Philox and AS241 in the
    registers the vectors live in, the tail arguments of a wavefront finished together) -- every run bit for bit the run it is alone,
    whose bases come from k_nhats; nDims even and odd (an odd nDims puts every other vector on the second half of a Philox call)."""
    from polychordlite_amd.repeats import run_repeats
    api = engine
    lib = api.load()
    for D, nDer, nlive, nr in ((20, 2, 400, 40), (7, 1, 300, 14), (24, 0, 256, 24), (2, 0, 100, 6), (13, 3, 200, 26)):
        L, P, keep = api.make_problem("gaussian", D, nDer)
        def settings(seed, ablate):
            s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, nDer)
            s.nlive, s.num_repeats, s.seed, s.ablate = nlive, nr, seed, ablate
            return s
        seeds = [700 + D + j for j in range(4)]
        singles = [api.run(settings(sd, 0), L, P) for sd in seeds]
        merged, runs = run_repeats(settings(0, 4096), L, P, seeds, max_in_flight=len(seeds))
        for one, r in zip(singles, runs):
            assert r["path"]["slice_lane"] > 0 and one["path"]["slice_lane"] == 0
            for k in ("ndead", "nlike", "niter", "nupdates", "nbatches"):
                assert one[k] == r[k], (D, k, one[k], r[k])
            assert one["logZ"] == r["logZ"] and np.array_equal(one["dead"], r["dead"], equal_nan=True) and np.array_equal(one["live"], r["live"], equal_nan=True)


@pytest.mark.gpu
def test_deck_records_against_stale_memory(engine):
    """k_nhats leaves every chain's deck behind the bases of a nursery under a tag of what the deck is a function of (keys, nursery, chain,
    num_repeats: pc_deck_record), and k_slice takes a record whose tag it finds.  The engine's blocks come back from a cache: one process,
    one seed, shapes that change from call to call, so that older records of the same keys lie where new ones are looked for -- every run
    alone (decks from the records) against the same run in step with a second one (lane-per-chain kernels: decks made in the kernel)."""
    from polychordlite_amd.repeats import run_repeats
    api = engine
    lib = api.load()
    for D, nDer, nlive, nr in ((8, 0, 200, 16), (8, 0, 200, 8), (8, 0, 200, 16), (8, 0, 100, 16), (8, 0, 200, 32), (12, 1, 200, 16), (8, 0, 200, 16), (8, 0, 200, 9)):
        L, P, keep = api.make_problem("gaussian", D, nDer)
        def settings(seed):
            s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, nDer)
            s.nlive, s.num_repeats, s.seed = nlive, nr, seed
            return s
        one = api.run(settings(5), L, P)
        merged, runs = run_repeats(settings(0), L, P, [5, 6], max_in_flight=2)
        assert one["path"]["slice_wave"] > 0 and runs[0]["path"]["slice_lane"] > 0
        for k in ("ndead", "nlike", "niter", "nbatches"):
            assert one[k] == runs[0][k], (D, nlive, nr, k, one[k], runs[0][k])
        assert one["logZ"] == runs[0]["logZ"] and np.array_equal(one["dead"], runs[0]["dead"], equal_nan=True)


@pytest.mark.gpu
def test_a_failing_run_ends_the_runs_in_step_cleanly(engine):
    """one of several runs in step fails (injected: a device allocation during its setup): pchip_run_repeats reports the failure, gives
    every buffer back, and the next call -- the same seeds -- makes the runs as if nothing had happened"""
    from polychordlite_amd.repeats import run_repeats
    api = engine
    lib = api.load()
    L, P, keep = api.make_problem("gaussian", 6, 1)
    s = api.Settings(); lib.pchip_settings_default(C.byref(s), 6, 1)
    s.nlive, s.num_repeats = 150, 12
    good, _ = run_repeats(s, L, P, [71, 72, 73, 74], max_in_flight=4)
    lib.polychord_hip_set_option(b"inject_fault", 1.0)
    with pytest.raises(RuntimeError):
        run_repeats(s, L, P, [71, 72, 73, 74], max_in_flight=4)
    again, runs = run_repeats(s, L, P, [71, 72, 73, 74], max_in_flight=4)
    assert again["logZ"] == good["logZ"] and again["nlike"] == good["nlike"] and len(runs) == 4


@pytest.mark.gpu
def test_a_failing_update_inside_a_fiber_unwinds_the_others(engine):
    """clustered runs in step make their updates as fibers of the driving thread (shared waits).  One of them fails in the middle of its
    update (injected: cluster capacity at the next split): the fibers still suspended are resumed with the cancel flag and unwind their
    frames (their pinned blocks go back to the cache, what they wrote down for the next flush is dropped), the call reports the failure,
    and the next call -- the same seeds -- makes the runs as if nothing had happened"""
    from polychordlite_amd.repeats import run_repeats
    api = engine
    lib = api.load()
    L, P, keep = api.make_problem("rastrigin", 3, 0, -5.12, 5.12)
    s = api.Settings(); lib.pchip_settings_default(C.byref(s), 3, 0)
    s.nlive, s.num_repeats, s.do_clustering = 200, 9, 1
    seeds = [81, 82, 83, 84, 85, 86]
    good, gruns = run_repeats(s, L, P, seeds, max_in_flight=len(seeds))
    assert good["nclustered"] >= 4
    lib.polychord_hip_set_option(b"inject_fault", 2.0)
    with pytest.raises(RuntimeError):
        run_repeats(s, L, P, seeds, max_in_flight=len(seeds))
    again, runs = run_repeats(s, L, P, seeds, max_in_flight=len(seeds))
    assert again["logZ"] == good["logZ"] and again["nlike"] == good["nlike"] and len(runs) == len(seeds)
    for a, b in zip(gruns, runs):
        assert a["logZ"] == b["logZ"] and np.array_equal(a["dead"], b["dead"], equal_nan=True)


@pytest.mark.gpu
def test_lived_records_left_on_the_device(engine):
    """settings.device_records: a run leaves the records of its points that entered the live set on the device that made them (what the
    exchange step sends), picked by the same kernels the merge uses on uploaded host arrays: the merge that reads them there must be, bit
    for bit, the merge of the host arrays -- for a run with failed spawns (batch > 1) and for a clustered one -- and pchip_run_repeats
    (which asks for them itself) must give the union it gave before"""
    from polychordlite_amd import merge as mg
    from polychordlite_amd.repeats import run_repeats
    api = engine
    lib = api.load()
    for kind, D, nDer, nlive, nr, clus, box in (("gaussian", 6, 1, 150, 12, 0, (None, None)), ("rastrigin", 3, 0, 200, 9, 1, (-5.12, 5.12))):
        s = api.Settings(); lib.pchip_settings_default(C.byref(s), D, nDer)
        s.nlive, s.num_repeats, s.batch, s.seed, s.do_clustering, s.device_records = nlive, nr, nlive // 2, 5, clus, 1
        L, P, keep = api.make_problem(kind, D, nDer, *box)
        run = api.run(s, L, P)
        lived = int((run["logweights"] > run["logzero"]).sum())
        assert run["n_records"] == lived < run["ndead"]
        a = mg.comm_merge(run, None, D, nDer, want_rows=True)                      # reads the device block
        s.device_records = 0
        run0 = api.run(s, L, P)
        assert run0["n_records"] is None and np.array_equal(run0["dead"], run["dead"], equal_nan=True)
        b = mg.comm_merge(run0, None, D, nDer, want_rows=True)                     # uploads the host arrays
        assert a["records"] == b["records"] == lived and a["logZ"] == b["logZ"] and a["evidence_rule"] == b["evidence_rule"] == clus
        assert np.array_equal(a["rows"], b["rows"]) and np.array_equal(a["logweights"], b["logweights"]) and np.array_equal(a["nlive"], b["nlive"])
        seeds = [5, 6, 7]
        m1, r1 = run_repeats(s, L, P, seeds, max_in_flight=3, want_rows=True)       # (device records asked for by the library itself)
        os.environ["PC_DEVICE_RECORDS_OFF"] = "1"
        try:
            m0, r0 = run_repeats(s, L, P, seeds, max_in_flight=3, want_rows=True)
        finally:
            del os.environ["PC_DEVICE_RECORDS_OFF"]
        # (the library asked for the device blocks itself and gave them back behind the union: a caller that did not ask must not find
        #  its results pinning device memory -- ADVICE round 5; a caller that did ask keeps them)
        assert all(x["n_records"] is None for x in r1) and all(x["n_records"] is None for x in r0)
        s.device_records = 1
        m3, r3 = run_repeats(s, L, P, seeds, max_in_flight=3, want_rows=True)
        s.device_records = 0
        assert all(x["n_records"] is not None for x in r3) and m3["logZ"] == m1["logZ"] and np.array_equal(m3["rows"], m1["rows"])
        assert m1["logZ"] == m0["logZ"] and np.array_equal(m1["rows"], m0["rows"]) and np.array_equal(m1["logweights"], m0["logweights"])
        m2, _ = run_repeats(s, L, P, seeds, max_in_flight=3)                        # without the merged rows: the same evidence
        assert m2["logZ"] == m1["logZ"] and "rows" not in m2 and np.array_equal(m2["post_mean"], m1["post_mean"])
