"""bench.py's last stdout line is what the driver parses: it must stay small, valid and complete.

Round 4's line (profiles/r04_bench.json, 18.9 KB: per-step lists, every sweep's detail, paragraph-long notes) was not parsed and
the round's measurement did not count.  `compact_record` makes the line from the full record; these tests feed it that very record
(and a clustered one) and hold its size, its JSON round trip and the fields SURVEY 8(d) asks for.  The second test starts bench.py's
own rank bootstrap with two processes over gloo (no GPU) up to, not including, the library's communicator.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _canned(name):
    return json.load(open(os.path.join(ROOT, "profiles", name)))


@pytest.mark.parametrize("name", ["r05_bench_full.json", "r05_bench_c3_full.json", "r05_bench_c5_full.json", "r04_bench.json", "r04_bench_c3.json", "r03_bench.json"])
def test_compact_record_is_small_valid_and_complete(name):
    full = _canned(name)
    # what round 5 adds to the full record
    if full.get("roofline"):
        full["roofline"]["in_step_multi"] = {"runs_per_gpu": 16, "n_gpus": 1, "runs": 16, "value": 4.4e9, "wall_ms": 52.1, "exchange_ms": 3.3,
                                             "merged_logZ": 0.01, "merged_logZerr": 0.023, "note": "x" * 500}
    full.setdefault("config", {})["workload_short"] = "configs[1]: 20-D Gaussian, nlive=2000, num_repeats=40, one full run per step"
    out = bench.compact_record(full, "gpurun_out/bench_full_c2.json")
    line = json.dumps(out, separators=(",", ":"))
    assert len(line) < bench.COMPACT_LIMIT <= 6000, len(line)
    back = json.loads(line)
    assert back == out
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data", "config"):
        assert k in back, k
    assert back["value"] == pytest.approx(full["value"], rel=1e-5)
    assert back["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    assert set(back["config"]) >= {"workload", "batch_chains", "parallelism", "mode"}
    roof = back["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert roof["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-5)
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], rel=1e-4)
    for k in ("kernel", "achieved", "avg_launch_us", "bytes_per_launch", "bytes_per_eval", "whole_run_frac"):
        assert k in roof, k
    cb = back["cpu_baseline"]
    assert cb["value"] == pytest.approx(full["cpu_baseline"]["value"], rel=1e-5)
    assert cb["kind"] in ("reference", "port") and cb["cores"] == 1 and cb["unit"] and cb["sample"]
    # nothing a parser chokes on: no lists of more than six entries, no string of more than 120 characters, no NaN
    def walk(x):
        if isinstance(x, dict):
            for v in x.values():
                walk(v)
        elif isinstance(x, list):
            assert len(x) <= 6
            for v in x:
                walk(v)
        elif isinstance(x, str):
            assert len(x) <= 120, x
        elif isinstance(x, float):
            assert x == x and abs(x) != float("inf")
    walk(back)
    assert "NaN" not in line and "Infinity" not in line


def test_compact_record_sheds_optional_blocks_rather_than_grow():
    full = _canned("r04_bench.json")
    full["other_configs"] = {("c%d" % k): dict(full["other_configs"]["c3"]) for k in range(40)}       # absurdly many
    out = bench.compact_record(full)
    assert len(json.dumps(out, separators=(",", ":"))) < bench.COMPACT_LIMIT
    assert out["roofline"]["frac"] > 0 and out["cpu_baseline"]["value"] > 0


def test_rank_bootstrap_two_processes_over_gloo():
    """`python bench.py --gpus 2` starts its two ranks itself (torch.distributed.run, 127.0.0.1), forms the process group and reduces
    the ranks' clocks and counters the way the timed region does -- here over gloo, stopping before the library's communicator"""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--bootstrap-only"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j == {"bootstrap": "ok", "world": 2, "backend": "gloo", "max": 2.0, "sum": 30.0}
    # a launcher whose world size disagrees with --gpus is refused
    env2 = dict(env, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--bootstrap-only"],
                        cwd=ROOT, env=env2, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p2.returncode != 0 and "WORLD_SIZE=3" in p2.stderr


@pytest.mark.gpu
def test_two_ranks_through_the_whole_harness_on_one_gpu():
    """`python bench.py --gpus 2 --backend gloo` on the GPU box: two ranks (they share the visible GPU; RCCL forms no communicator with two
    ranks on one device), the process group over gloo, and the library's exchange over that group's all-gather (pchip_comm_create_with) --
    so the N > 1 code of the harness AND of the library runs with world_size 2 before the driver's 8-GPU run: the early record, the timed
    region's barrier + max over ranks, the sums of the ranks' counters, R runs per rank in step with one exchange of all N R runs
    (roofline.in_step_multi), the agreement between the ranks before it.  No scaling claim: the record says the ranks share a GPU."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--nlive", "400",
                        "--runs-per-gpu", "3", "--no-cpu", "--full-out", ""], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    early, j = lines[0], lines[-1]
    assert early.get("partial") and early["n_gpus"] == 2 and early["value"] > 0
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["scaling"] == "weak" and "SHARE" in j["config"]["parallelism"] and "partial" not in j
    assert abs(j["value"] / early["value"] - 1.0) < 1e-5                  # the same timed region
    assert j["merged"]["n_runs"] == 2                                        # the last step's run of BOTH ranks in the union
    m = j["roofline"]["in_step_multi"]
    assert "error" not in m and m["n_gpus"] == 2 and m["runs_per_gpu"] == 3 and m["runs"] == 6 and m["value"] > 0 and m["exchange_ms"] > 0
    assert abs(m["merged_logZ"]) < 5 * m["merged_logZerr"] + 0.2             # six runs of the 20-D Gaussian (analytic log Z = 0)
    assert "leg_errors" not in j, j.get("leg_errors")


@pytest.mark.parametrize("name", ["r05_bench.json", "r05_bench_c3.json", "r05_bench_c4.json", "r05_bench_c5.json",
                                  "r06_bench.json", "r06_bench_c3.json", "r06_bench_c4.json", "r06_bench_c5.json"])
def test_the_lines_bench_py_printed_on_the_gpu_box(name):
    """the committed last lines of `python bench.py [--workload cN]` as they came off the MI355X: one line, < 6 KB, the driver's fields"""
    text = open(os.path.join(ROOT, "profiles", name)).read().strip()
    assert "\n" not in text and len(text) < bench.COMPACT_LIMIT
    j = json.loads(text)
    assert j["roofline"]["frac"] > 0 and j["roofline"]["bound"] == "hbm" and j["cpu_baseline"]["value"] > 0 and j["cpu_baseline"]["kind"] == "reference"
    assert j["unit"] == "likelihood evals/s" and j["dtype"] == "f64" and j["scaling"] == "weak" and j["n_gpus"] == 1 and j["vs_baseline"] is None
    if name == "r06_bench.json":
        # round 6: the figure a user's own device functor gets and the reference-equivalent one stand at the top of the record, the general-functor
        # sampling kernel has its own roofline entry, and the record says which kernels the run took
        assert 0 < j["value_general_functor"] < j["value"] and 0 < j["value_reference_equivalent"] < j["value"]
        assert j["general_functor"]["frac"] > 0 and j["general_functor"]["avg_launch_us"] > 0
        assert j["paths"]["consume_par"] > 0 and "consume_general" not in j["paths"]
        assert all(v["general_kernel_launches"] == 0 for v in j["other_configs"].values())
    if name in ("r05_bench.json", "r06_bench.json"):
        assert j["metric"].startswith("likelihood evals/sec, 20D Gaussian nlive=2000") and j["config"]["batch_chains"] == 1000
        assert len(j["roofline"]["in_step"]) == 6 and j["roofline"]["in_step_multi"]["runs"] == 16


def test_every_committed_profile_bench_py_reads_is_a_record():
    """round 5 committed an EMPTY profiles/r05_slice_cycles.json (a collection that failed half way) and `python bench.py` died in
    json.load behind its timed region -- on the GPU box only.  The files bench.py picks up must parse, and one that does not is skipped"""
    import glob
    for pth in glob.glob(os.path.join(ROOT, "profiles", "*.json")):
        assert os.path.getsize(pth) > 0, pth
        json.load(open(pth))
    rec, name = bench.committed_record(bench.SLICE_CYCLE_FILES, ("cycles_per_slice", "cycles_per_slice_total"))
    assert rec and rec["cycles_per_slice_total"] > 0 and name in bench.SLICE_CYCLE_FILES
    rec, name = bench.committed_record(bench.PMC_FILES, ("kernels",))
    assert rec and any(k.startswith("k_slice") for k in rec["kernels"])
    # an empty or truncated file in front of a good one is passed over
    import shutil
    import tempfile
    d = tempfile.mkdtemp()
    try:
        os.makedirs(os.path.join(d, "profiles"))
        open(os.path.join(d, "profiles", "a.json"), "w").close()
        open(os.path.join(d, "profiles", "b.json"), "w").write('{"cycles_per_slice": {')
        open(os.path.join(d, "profiles", "c.json"), "w").write('{"cycles_per_slice": {"x": 1.0}, "cycles_per_slice_total": 1.0}')
        root = bench.ROOT
        bench.ROOT = d
        try:
            rec, name = bench.committed_record(("missing.json", "a.json", "b.json", "c.json"), ("cycles_per_slice", "cycles_per_slice_total"))
        finally:
            bench.ROOT = root
        assert name == "c.json" and rec["cycles_per_slice_total"] == 1.0
    finally:
        shutil.rmtree(d)


def test_the_latency_block_from_the_committed_cycles():
    runs = [{"nlike": 14_000_000, "niter": 80_000, "nbatches": 79, "nrounds": 79}]
    kern = [{"kernel": "k_slice", "avg_launch_us": 70.0}]
    lat = bench.latency_model(runs, kern)
    assert lat["model_min"] > 0 and 0.3 < lat["frac_of_model"] < 1.0 and lat["measured_source"].startswith("profiles/r0")


def test_a_leg_that_fails_is_written_down_not_raised():
    class B(bench.Bench):
        def __init__(self):
            self.leg_errors = {}
    b = B()
    assert b.leg("ok", lambda: 3, None) == 3 and not b.leg_errors
    assert b.leg("broken", lambda: json.loads(""), "dflt") == "dflt"
    assert "broken" in b.leg_errors and "JSONDecodeError" in b.leg_errors["broken"]
    full = _canned("r05_bench_full.json")
    full["leg_errors"] = b.leg_errors
    full["roofline"] = None                                            # (the roofline leg itself lost: the line still leaves, and says so)
    out = bench.compact_record(full)
    assert out["roofline"] is None and "broken" in out["leg_errors"] and out["value"] > 0 and out["cpu_baseline"]["value"] > 0
