#!/usr/bin/env python
"""bench.py -- likelihood evaluations / second of the MI355X nested-sampling engine.

Contract (DESIGN.md "Measurement"):
  * a STEP is one complete nested-sampling run of the workload (default c2 = BASELINE.json configs[1]: 20-D Gaussian of
    likelihoods/examples/gaussian.f90 with ini/gaussian.ini's priors, nlive = 2000, num_repeats = 40, precision_criterion
    1e-3, no clustering, fp64 throughout); step i uses seed = 1000 + i (+ rank * 100003).
  * metric = likelihood evals/sec = sum of the reference's own counter RTI%nlike (calculate.f90:44) over the timed steps
    / wall time (barrier + device sync on both sides, max over ranks).  The exchange step is inside the timed region.
  * N > 1 GPUs: repeat-sharded: every rank runs independent runs with its own seeds; the dead points of the
    last step's runs are all-gathered over RCCL inside the library and merged on every rank; scaling is "weak".
    N = 1 goes through the same merge code.  `--runs-per-gpu R`: behind the timed steps every rank also runs R runs IN STEP
    (pchip_run_repeats), the N R runs are exchanged the same way, reported as `roofline.in_step_multi`, never in `value`.
  * roofline: the kernel class with the largest HIP-event time; `achieved` = SURVEY 8(d)'s algorithmic bytes per
    evaluation x evaluations of one launch / its measured launch duration.
  * cpu_baseline: the REFERENCE itself (oracle/_ref/ref_driver, built from /root/reference by oracle/Makefile; else the C
    restatement oracle/liboracle.so) on one host core, one full run of the same workload, and `all_cores`: one
    independent run per host core at the same time (both legs after every GPU figure, alone).
  * OUTPUT: the LAST stdout line is ONE compact JSON record (< 6 KB: `compact_record`); the full record (per-step lists,
    every kernel class, the sweeps' details, notes) goes to gpurun_out/bench_full_<workload>.json (`--full-out`), never to
    stdout.  Round 4's 18.9-KB line was not parsed by the driver; tests/test_bench_line.py holds the size.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_EVAL = 258.0
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec
GENERAL_PMC_FILES = ("r06_general_pmc.json",)      # PMC passes of the general-functor run (tools/collect_pmc.sh with PC_ABLATE=1)
COMPACT_LIMIT = 6000        # bytes of the last stdout line (the driver parsed 6.5 KB and 13 KB, not 18.9 KB)
# BASELINE.json configs through this harness: kind, nDims, nDerived, nlive, num_repeats, clustering, box, analytic logZ
WORKLOADS = {
    "c2": dict(kind="gaussian", D=20, nDer=2, nlive=2000, nr=40, clustering=0, box=None, truth=0.0, ref_evals_per_dead=173.9,
               short="configs[1]: 20-D Gaussian, nlive=%d, num_repeats=40",
               name="BASELINE configs[1]: 20-D Gaussian (mu=0.5, sigma=0.1, U(0,1)^20), nlive=%d, num_repeats=40"),
    "c3": dict(kind="rastrigin", D=10, nDer=0, nlive=1000, nr=30, clustering=1, box=(-5.12, 5.12), truth=-23.263, ref_evals_per_dead=155.5,
               short="configs[2]: 10-D Rastrigin, nlive=%d, num_repeats=30, kNN clustering",
               name="BASELINE configs[2]: 10-D Rastrigin, U(-5.12,5.12)^10, nlive=%d, num_repeats=30 (= 3 nDims, the ini's ratio), kNN clustering"),
    "c4": dict(kind="twin_gaussian", D=30, nDer=1, nlive=500, nr=40, clustering=1, box=(-1.0, 1.0), truth=-20.794, ref_evals_per_dead=177.3,
               short="configs[3]: 30-D twin Gaussian, nlive=%d, num_repeats=40, kNN clustering",
               name="BASELINE configs[3]: 30-D twin Gaussian (sigma=0.1), U(-1,1)^30, nlive=%d, num_repeats=40, kNN clustering"),
    "c5": dict(kind="corr_gaussian", D=100, nDer=0, nlive=5000, nr=200, clustering=0, box=None, truth=0.0, ref_evals_per_dead=None,
               short="configs[4]: 100-D correlated Gaussian, nlive=%d, num_repeats=200",
               name="BASELINE configs[4]: 100-D correlated Gaussian (random eigenbasis, eigen-sigma 0.1 .. 0.001), U(0,1)^100, nlive=%d, num_repeats=200"),
}
# ref_evals_per_dead: likelihood evaluations per dead point of the REFERENCE BINARY at this configuration (its linear mode, one chain at a
# time: tests/golden/ref_c3_seeds.json, ref_c4_seeds.json; configs[1]: the cpu_baseline leg of this script on the GPU box, 11.45 M / 65.8 k)
METRIC = {"c2": "likelihood evals/sec, 20D Gaussian nlive=%d", "c3": "likelihood evals/sec, 10D Rastrigin nlive=%d",
          "c4": "likelihood evals/sec, 30D twin Gaussian nlive=%d", "c5": "likelihood evals/sec, 100D correlated Gaussian nlive=%d"}
HOST_PHASES = ("t_setup", "t_generate", "t_loop", "t_final", "t_results", "t_teardown")


def algorithmic_bytes_per_iteration(D, nDer, nr, N):
    """chain I/O + covariance pass + phantom compaction + contour scan, per nested-sampling iteration
    (phi = nr + 1 phantoms per live point in steady state)"""
    nT, phi = 2 * D + nDer + 2, nr + 1
    return 8.0 * nT * (1 + nr) + 8.0 * (1 + phi) * D + 8.0 * phi * nT + 8.0 * N


def own_bytes_per_launch(kernel, D, nDer, nr, N, B):
    """a kernel class's OWN algorithmic HBM bytes per launch"""
    nT = 2 * D + nDer + 2
    return {"k_slice": B * (8 * nT * (1 + nr) + 8 * nr * (D + 1) + 16 * nr),    # seed row in, nr baby rows out, directions + widths in, logL twice
            "k_consume": 8 * (2 * N + 3 * B) + 8 * nr * B + 184 * B,               # sorted keys + slots in and out, candidates, the babies' logL (phantom masks), plan records
            "k_nhats": B * 8 * nr * (D + 1) * 2 + 8 * D * D,                       # raw bases in, whitened directions + widths out, the Cholesky factor
            # the orthonormal bases drawn ahead on the side stream: one record per basis and chain (nDims > 64: 16384 doubles in the
            # matrix cores' operand layout; else nDims x nDims)
            "k_bases_side": B * ((nr + D - 1) // D) * 8 * (16384 if D > 64 else D * D),
            "k_apply": B * 8 * nT * (1 + nr) + 8 * nT * B}.get(kernel)


def random_correlated_gaussian(D, seed=12345, sigma0=0.1):
    """random_gaussian.f90 / random_utils.F90:581-614: random orthonormal eigenbasis, eigen-sigma_j = sigma0 (1e-2)^(j/(D-1))"""
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    sig = sigma0 * (1e-2) ** (np.arange(D) / max(D - 1, 1))
    return Q @ np.diag(sig ** -2) @ Q.T, np.full(D, 0.5), float(2.0 * np.log(sig).sum())


# ---- issue model of the dominant kernel (the path is latency bound, not bandwidth bound: SURVEY 8d).  k_slice runs one wavefront per
# chain, one wavefront per SIMD, one cube coordinate per lane.  A wavefront that has its SIMD to itself issues one vector instruction every
# 5-6 cycles WHETHER OR NOT it depends on the one before (tools/dev/ubench_fp64.hip: dependent fma 5.8, eight independent chains 5.2 per
# operation; rounds 2-5 quoted "32 cycles per dependent fp64 operation" from a loop of ONE operation -- 24 of them the loop's own -- and built
# a dependent-chain model on it: withdrawn).  So a launch lasts as long as its longest wavefront has instructions:
#     floor = instructions per wavefront (rocprofv3 SQ_INSTS_*: profiles/rNN_issue.json) x the single-wave issue interval,
# against the wavefront's measured cycles (SQ_WAVE_CYCLES, in units of four clocks; and s_memtime inside the kernel, section by section).
# What closes the gap is fewer instructions a slice, not shorter dependence chains or more bytes.
ISSUE_INTERVAL_CYCLES, CLOCK_MHZ = 5.8, 2400.0
ISSUE_FILES = ("r06_issue.json",)
SLICE_CYCLE_FILES = ("r06_slice_cycles.json", "r05_slice_cycles.json", "r04_slice_cycles.json", "r03_slice_cycles.json")
PMC_FILES = ("r06_pmc.json", "r05_pmc.json", "r04_pmc.json", "r03_pmc.json", "r02_pmc.json", "r01_pmc.json")


def committed_record(names, need):
    """the first of profiles/<names> that is there, parses and holds the keys `need`: (record, name) or (None, None).  A profile that a
    failed collection left empty or half-written must not take the bench line down with it"""
    for n in names:
        pth = os.path.join(ROOT, "profiles", n)
        try:
            rec = json.load(open(pth))
        except (OSError, ValueError):
            continue
        if isinstance(rec, dict) and all(k in rec for k in need):
            return rec, n
    return None, None


def latency_model(runs, kern):
    """the issue floor of k_slice (instructions per wavefront x the interval at which one wavefront issues) next to its measured cycles"""
    ks = [k for k in kern if k["kernel"] == "k_slice"]
    nr = 40
    evals_per_slice = sum(r["nlike"] for r in runs) / max(sum(r["niter"] for r in runs), 1) / nr
    out = {"unit": "shader cycles per wavefront (= chain) per launch, 2.4 GHz", "issue_interval_cycles": ISSUE_INTERVAL_CYCLES,
           "issue_interval_source": "tools/dev/ubench_fp64.hip: one wavefront, sixteen operations to a loop iteration: dependent fp64 fma 5.8 cycles an operation, "
                                    "eight independent chains 5.2 -- issue bound either way", "evaluations_per_slice": evals_per_slice, "slices_per_launch": nr}
    rec, src = committed_record(ISSUE_FILES, ("kernels",))
    k = next((v for n, v in (rec["kernels"] if rec else {}).items() if n.startswith("k_slice")), None)
    if k and k.get("SQ_WAVES"):
        w = k["SQ_WAVES"]
        ins = {c[9:].lower(): k.get(c, 0.0) / w for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM")}
        out["instructions_per_wavefront"] = ins
        out["instructions_per_slice"] = sum(ins.values()) / nr               # (prologue and epilogue included: ~15 % of a launch's instructions)
        out["model_min"] = sum(ins.values()) * ISSUE_INTERVAL_CYCLES
        out["measured"] = 4.0 * k["SQ_WAVE_CYCLES"] / w                      # (the counter ticks every four clocks)
        out["frac_of_model"] = out["model_min"] / out["measured"]
        out["measured_source"] = ("profiles/%s (rocprofv3 --pmc SQ_INSTS_VALU / _SALU / _LDS / _SMEM / _VMEM / SQ_WAVES / SQ_WAVE_CYCLES, one pass each, tools/collect_issue_profile.sh; "
                                  "taken with settings.ablate bit 13 -- one wavefront a chain -- so that the counters are the chain's whole instruction stream: the product kernel "
                                  "hands the deck's shuffle, the whitening and nine of ten Philox calls, ~2.5 k of these instructions, to a helper wavefront on another SIMD)") % src
    m, src2 = committed_record(SLICE_CYCLE_FILES, ("cycles_per_slice", "cycles_per_slice_total"))
    if m:
        out["slice_sections_cycles"] = m["cycles_per_slice"]; out["slice_cycles"] = m["cycles_per_slice_total"]
        out["slice_sections_source"] = "profiles/%s (s_memtime inside k_slice, SLICE_DBG build, tools/collect_slice_dbg.sh)" % src2
    if ks:
        out["launch_cycles_per_slice_this_run"] = ks[0]["avg_launch_us"] * CLOCK_MHZ / nr      # whole launch / slices: includes seed choice, shuffle, whitening, derived parameters
    return out


def live_pmc(workload):
    """HBM bytes per launch of every kernel, measured now: two rocprofv3 passes over a short run of this script (FETCH_SIZE, then
    WRITE_SIZE; --pmc with --kernel-trace only, one counter per pass, as MI355X_MICROARCH.md prescribes), summarised like
    tools/pmc_summary.py (gfx950: FETCH_SIZE counts 64 B per 128-B request and is doubled).  None if the profiler is not there."""
    import shutil
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import pmc_summary
        tmp = tempfile.mkdtemp(prefix="pc_pmc_", dir="/tmp")
        env = dict(os.environ, TMPDIR="/tmp")
        acc = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            subprocess.run([rocprof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
                            os.path.abspath(__file__), "--workload", workload, "--no-cpu", "--no-extras", "--steps", "2", "--warmup", "1",
                            "--full-out", ""],
                           cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240)
            acc[counter] = pmc_summary.per_kernel(d, counter)
        shutil.rmtree(tmp, ignore_errors=True)
        F, W = acc["FETCH_SIZE"], acc["WRITE_SIZE"]
        if not F or not W:
            return None
        out = {}
        for k in set(F) | set(W):
            nf, sf = F.get(k, [0, 0.0]); nw, sw = W.get(k, [0, 0.0])
            out[k] = {"launches": max(nf, nw, 1), "hbm_bytes_per_launch": (2.0 * sf / max(nf, 1) + sw / max(nw, 1)) * 1024.0}
        return out
    except Exception:
        return None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


class CpuBaseline:
    """the reference (preferred) or the restatement on ONE host core -- one full run of the workload (bounded for c5) -- and
    `all_cores`: one full run per host core at the same time.  Both legs run AFTER every GPU figure has been taken, with nothing else
    going on (tried in round 5: the one-core run in the background of the GPU sweeps measured 6 % low, an all-cores leg bounded to
    the first 16000 deaths 30 % low -- a baseline must not be flattered by the harness)."""
    ALL_CORES_NDEAD = None

    def __init__(self, wl, nlive, all_cores=True):
        self.wl, self.nlive, self.all_cores = wl, nlive, all_cores
        self.ref = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
        ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        self.visible = ncores
        try:    # the cores this container may actually use (cgroup v2 quota), not the threads the host shows
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            if q != "max":
                ncores = max(1, min(ncores, int(float(q) / float(per))))
        except (OSError, ValueError):
            pass
        self.ncores = min(ncores, 32)         # bounded sample: at most 32 runs at a time
        self.env, self.bounded, self.proc = "", None, None
        self.sample = "%s, one full run, seed 7" % (wl["short"] % nlive)
        if wl["kind"] == "corr_gaussian":
            # long configurations get a BOUNDED sample: the reference stops after max_ndead deaths (settings%max_ndead)
            covf = "/tmp/pc_ref_bench_cov%d.bin" % wl["D"]
            ic, mean, logdet = random_correlated_gaussian(wl["D"])
            with open(covf, "wb") as f:
                f.write(np.ascontiguousarray(ic).tobytes()); f.write(np.ascontiguousarray(mean).tobytes()); f.write(np.float64(logdet).tobytes())
            self.bounded = 2500
            self.env = "REF_COV_FILE=%s REF_MAX_NDEAD=%d " % (covf, self.bounded)
            self.sample = "%s, the first %d deaths (a full run takes the reference hours), seed 7" % (wl["short"] % nlive, self.bounded)

    def _ref_run(self, seed, tag, max_ndead=None):
        wl = self.wl
        tmp = "/tmp/pc_ref_bench_%s" % tag
        os.makedirs(tmp, exist_ok=True)
        env = self.env + ("REF_MAX_NDEAD=%d " % max_ndead if (max_ndead and not self.bounded) else "")
        cmd = f"ulimit -s unlimited; {env}{self.ref} {wl['kind']} {wl['D']} {wl['nDer']} {self.nlive} {wl['nr']} {seed} {wl['clustering']} {tmp} ref 0"
        return subprocess.Popen(["bash", "-c", cmd], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, cwd=tmp)

    @staticmethod
    def _parse(p):
        out = p.communicate()[0]
        line = [l for l in out.splitlines() if l.startswith("{")]
        return json.loads(line[-1]) if line else None

    def start(self):
        if os.path.exists(self.ref):
            self.proc = self._ref_run(7, "one")

    def finish(self):
        if os.path.exists(self.ref):
            if self.proc is None:
                self.start()
            j = self._parse(self.proc)
            if j:
                res = {"value": j["nlike"] / j["wall"], "unit": "likelihood evals/s", "cores": 1, "kind": "reference",
                       "sample": self.sample + "; PolyChordLite Fortran, amdflang -O2, no file output", "cpu_model": cpu_model(),
                       "logZ": j["logZ"] if not self.bounded else None, "logZerr": j["logZerr"] if not self.bounded else None,
                       "ndead": j["ndead"], "nlike": j["nlike"], "wall_s": j["wall"], "bounded_max_ndead": self.bounded}
                if self.all_cores and self.ncores > 1:
                    t0 = time.time()
                    js = [self._parse(p) for p in [self._ref_run(7 + c, "c%d" % c, self.ALL_CORES_NDEAD) for c in range(self.ncores)]]
                    wall = time.time() - t0
                    js = [x for x in js if x]
                    res["all_cores"] = {"value": sum(x["nlike"] for x in js) / wall, "unit": "likelihood evals/s", "cores": self.ncores, "runs": len(js),
                                        "wall_s": wall, "mean_run_wall_s": float(np.mean([x["wall"] for x in js])),
                                        "hardware_threads_visible": self.visible,
                                        "sample": "one run per core the cgroup grants (<= 32), started together; sum nlike / wall of the slowest"}
                return res
        from tests import oracle_api as orc
        wl = self.wl
        s = orc.settings(wl["D"], wl["nDer"], nlive=self.nlive, num_repeats=wl["nr"], seed=7, batch=1, do_clustering=wl["clustering"])
        lo, hi = wl["box"] if wl["box"] else (None, None)
        L, P, keep = orc.make_problem(wl["kind"], wl["D"], lo, hi)
        t0 = time.time(); o = orc.run(s, L, P); dt = time.time() - t0
        return {"value": o["nlike"] / dt, "unit": "likelihood evals/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(),
                "sample": self.sample + "; oracle/liboracle.so (C restatement, gcc -O2)", "logZ": o["logZ"],
                "logZerr": o["logZerr"], "ndead": int(o["ndead"]), "nlike": int(o["nlike"]), "wall_s": dt}


# ---- the record the driver parses --------------------------------------------------------------------------------------
def _num(x, digits=6):
    """floats to `digits` significant figures (the line is read by a parser and a judge, not by a solver)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return float("%.*g" % (digits, x)) if np.isfinite(x) else None
    if isinstance(x, (np.floating, np.integer)):
        return _num(x.item(), digits)
    if isinstance(x, dict):
        return {k: _num(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_num(v, digits) for v in x]
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d and d[k] is not None}


def compact_record(full, full_path=None):
    """The ONE line the driver parses: BASELINE's metric, `roofline`, `cpu_baseline` and a handful of scalars -- no per-step lists,
    no paragraphs; every string < 120 characters; < COMPACT_LIMIT bytes (asserted).  Everything else is in the full record."""
    out = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling"))
    out["vs_baseline"] = full.get("vs_baseline")
    out.update(_pick(full, ("dtype", "data")))
    cfg = full.get("config") or {}
    out["config"] = {"workload": str(cfg.get("workload_short", cfg.get("workload", "")))[:110], "batch_chains": cfg.get("batch_chains"),
                     "parallelism": cfg.get("parallelism"), "mode": "one run at a time per GPU; R runs in step: roofline.in_step"}
    roof = full.get("roofline")
    if roof:
        r = _pick(roof, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "bytes_per_launch", "bytes_per_eval",
                         "whole_run_frac", "stream"))
        r["traffic_source"] = "live rocprofv3 --pmc" if str(roof.get("traffic_source", "")).startswith("live") else roof.get("traffic_source")
        lat = roof.get("latency")
        if lat and "frac_of_model" in lat:
            r["latency"] = _pick(lat, ("frac_of_model", "measured", "model_min"))
        # the two heaviest classes with their OWN algorithmic bytes, short
        r["kernels"] = [_pick(k, ("kernel", "avg_launch_us", "own_frac", "traffic")) for k in (roof.get("kernels") or [])[:2]]
        ins = []
        for e in (roof.get("in_step") or [])[:6]:
            ins.append(_pick(e, ("config", "runs", "value", "x_solo", "ms_per_run", "whole_run_frac")))
        if ins:
            r["in_step"] = ins
        if roof.get("in_step_multi"):
            r["in_step_multi"] = _pick(roof["in_step_multi"], ("runs_per_gpu", "n_gpus", "runs", "value", "wall_ms", "merged_logZ", "merged_logZerr", "exchange_ms", "error"))
        out["roofline"] = r
    else:
        out["roofline"] = None
    cb = full.get("cpu_baseline")
    if cb:
        c = _pick(cb, ("value", "unit", "cores", "kind", "cpu_model", "wall_s", "logZ", "logZerr", "ndead", "nlike", "bounded_max_ndead"))
        c["sample"] = str(cb.get("sample", ""))[:118]
        if cb.get("all_cores"):
            c["all_cores"] = _pick(cb["all_cores"], ("value", "cores", "runs", "wall_s"))
        out["cpu_baseline"] = c
    else:
        out["cpu_baseline"] = None
    lz, le = full.get("logZ") or [], full.get("logZerr") or []
    if lz:
        out["logZ_mean"] = float(np.mean(lz)); out["logZ_sem"] = float(np.std(lz, ddof=1) / np.sqrt(len(lz))) if len(lz) > 1 else None
        out["logZerr_mean"] = float(np.mean(le)) if le else None
    out.update(_pick(full, ("logZ_truth", "value_reference_equivalent", "speedup_wall_per_run", "speedup_evals_per_s", "merge_ms", "exchange", "paths")))
    mg = full.get("merged")
    if mg:
        out["merged"] = _pick(mg, ("n_runs", "logZ", "logZerr", "evidence_rule", "records"))
    gf = full.get("general_functor")
    if gf:
        # `value` is the built-in Gaussian by its closed form along the chord, failed spawns counted: the figure a user's OWN device functor gets
        # (one reduction per trial) stands next to it at the top, and so does the reference-equivalent one (value_reference_equivalent)
        out["value_general_functor"] = gf.get("value")
        out["general_functor"] = _pick(gf, ("value", "ms_per_step", "frac", "avg_launch_us", "traffic"))
    oc = full.get("other_configs")
    if oc:
        out["other_configs"] = {n: (_pick(v, ("value", "ms_per_step", "logZ", "logZerr", "logZ_truth", "evals_per_lived_dead", "dominant_kernel", "whole_run_frac", "general_kernel_launches"))
                                    if "error" not in v else {"error": str(v["error"])[:80]}) for n, v in oc.items()}
    if full.get("leg_errors"):
        out["leg_errors"] = {k: str(v)[:80] for k, v in list(full["leg_errors"].items())[:6]}
    if full_path:
        out["full_record"] = full_path
    out = _num(out)
    line = json.dumps(out, separators=(",", ":"))
    if len(line) >= COMPACT_LIMIT:          # never print a line the driver cannot parse: drop the optional blocks, largest first
        for k in ("other_configs", "general_functor", "merged"):
            out.pop(k, None)
        if out.get("roofline"):
            out["roofline"].pop("kernels", None)
        line = json.dumps(out, separators=(",", ":"))
    assert len(line) < COMPACT_LIMIT, "bench.py: the compact record is %d bytes" % len(line)
    return out


# ---- ranks ----------------------------------------------------------------------------------------------------------------
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nlive", type=int, default=0, help="0 = the workload's own")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS),
                    help="c2 = BASELINE configs[1], the metric configuration (default, what the driver runs); c3 / c4 / c5 = "
                         "BASELINE configs[2..4] through the same harness, also with --gpus N (their lines are kept under profiles/)")
    ap.add_argument("--runs-per-gpu", type=int, default=16,
                    help="behind the timed steps: every rank runs this many runs of the workload IN STEP (pchip_run_repeats), the N x R runs are "
                         "exchanged over RCCL and merged; reported as roofline.in_step_multi, never part of `value`; 0 = skip")
    ap.add_argument("--concurrent", default="4,8,16,32,64",
                    help="after the timed steps (N = 1): R independent runs in flight on this GPU for each R of the list "
                         "(polychordlite_amd.repeats.run_repeats; reported separately, never part of `value`); '' or 0 = skip")
    ap.add_argument("--concurrent-configs", default="c3,c4", help="after the timed steps (N = 1): these clustered BASELINE configurations with R runs in step, R from --concurrent-clustered; '' = skip")
    ap.add_argument("--concurrent-clustered", default="16,32")
    ap.add_argument("--other-configs", default="c3,c4,c5",
                    help="after the timed steps of the default workload (N = 1): one step each of these BASELINE configurations, reported "
                         "as `other_configs`; '' = skip")
    ap.add_argument("--no-live-pmc", action="store_true", help="roofline.traffic from the committed profiles/ instead of two profiled runs now")
    ap.add_argument("--no-extras", action="store_true", help="skip the figures after the timed region (general functor, concurrent sweep)")
    ap.add_argument("--full-out", default=None, help="where the full record goes (default gpurun_out/bench_full_<workload>.json; '' = nowhere)")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"), help="process-group backend of the ranks (gloo: the launcher test on CPU)")
    ap.add_argument("--bootstrap-only", action="store_true", help="start the ranks, form the process group, reduce one number over it, print and stop "
                                                                   "(before the library's communicator): tests/test_bench_line.py")
    return ap.parse_args(argv)


def bootstrap(args):
    """One process per GPU.  `python bench.py --gpus N` on its own starts the N ranks here, the way the driver does (torch.distributed.run,
    127.0.0.1); under a launcher it reads RANK / LOCAL_RANK / WORLD_SIZE.  Returns (rank, local_rank, world, dist, torch)."""
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        if args.backend == "nccl":
            from polychordlite_amd import _ctypes_api as api0
            ndev = api0.load().pchip_device_count()
            if ndev < args.gpus:
                raise SystemExit("bench.py: --gpus %d asked for, %d HIP device(s) visible -- one rank per GPU, no oversubscription" % (args.gpus, ndev))
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node equal to --gpus" % (args.gpus, world))
    import torch
    if args.backend == "nccl" and torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d has no GPU of its own (%d visible): one rank per GPU" % (local_rank, torch.cuda.device_count()))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    return rank, local_rank, world, dist, torch


GLOO_NOTE = " -- ranks SHARE the visible GPU(s), the exchange goes over a host all-gather (gloo) handed to the library: a test of the N > 1 code path, not a scaling figure"


def reduce_over_ranks(dist, torch, device, dt, sums):
    """(max over ranks of dt, sums added over ranks) -- what turns the ranks' clocks and counters into the job's"""
    if dist is None:
        return dt, list(sums)
    t = torch.tensor([dt] + list(sums), dtype=torch.float64, device=device)
    tm = t.clone(); dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    ts = t.clone(); dist.all_reduce(ts, op=dist.ReduceOp.SUM)
    return float(tm[0]), [float(x) for x in ts[1:]]


class Bench:
    """one rank of the harness: the problem, the engine's library, the process group"""
    BIG = ("dead", "logweights", "entry", "live", "_owner")      # result arrays that are views of pinned engine buffers: not kept across steps
    TIMED_STRIDE = 8

    def __init__(self, args, rank, local_rank, world, dist, torch):
        from polychordlite_amd import _ctypes_api as api
        from polychordlite_amd.merge import Comm
        self.args, self.rank, self.local_rank, self.world, self.dist, self.torch, self.api = args, rank, local_rank, world, dist, torch, api
        self.leg_errors = {}
        self.lib = api.load()
        if self.lib.pchip_device_count() < 1:
            raise SystemExit("bench.py: no HIP device visible; the engine has no CPU path")
        # the exchange step's communicator: RCCL inside the library (its id travels through the process group the ranks were
        # started with); one rank needs none.  --backend gloo on a GPU box (tests, a one-GPU box): the ranks share the visible devices,
        # the process group is gloo on host tensors and the library's exchange runs over it (merge.CallbackComm) -- every statement of
        # pchip_comm_merge_many and of this harness with world > 1, no claim about xGMI
        if args.backend == "gloo":
            from polychordlite_amd.merge import CallbackComm, host_all_gather
            local_rank = self.local_rank = local_rank % self.lib.pchip_device_count()
            torch.cuda.set_device(local_rank)
            self.dev = "cpu"                                        # (where the harness's own reductions live: gloo)
            self.comm = CallbackComm(rank, world, local_rank, host_all_gather(dist, torch, local_rank)) if world > 1 else None
        else:
            self.dev = f"cuda:{local_rank}"
            self.comm = Comm(rank, world, local_rank) if world > 1 else None
        self.wl = WORKLOADS[args.workload]
        self.nlive = args.nlive if args.nlive > 0 else self.wl["nlive"]
        self.s, self.L, self.P, self.keep = self.problem(self.wl, self.nlive, args.batch)
        self.extras = rank == 0 and world == 1 and not args.no_extras and args.steps > 0

    def cls_bit(self, name):
        return 1 << (self.api.KERNEL_CLASSES.index(name) + 1)

    def problem(self, w, nl, batch=0):
        api = self.api
        s_ = api.Settings(); self.lib.pchip_settings_default(C.byref(s_), w["D"], w["nDer"])
        s_.nlive = nl; s_.num_repeats = w["nr"]; s_.batch = batch; s_.device = self.local_rank
        s_.do_clustering = w["clustering"]
        s_.ablate = int(os.environ.get("PC_ABLATE", "0"))      # developer switches of the engine (A/B timing of a code path), 0 in production
        if w["kind"] == "corr_gaussian":
            ic, mean, logdet = random_correlated_gaussian(w["D"])
            L_, P_, keep_ = api.make_problem("corr_gaussian", w["D"], w["nDer"], invcov=ic, mean=mean, logdet=logdet)
        else:
            lo, hi = w["box"] if w["box"] else (None, None)
            L_, P_, keep_ = api.make_problem(w["kind"], w["D"], w["nDer"], lo, hi)
        return s_, L_, P_, keep_

    def one(self, i):
        self.s.seed = 1000 + i + 100003 * self.rank
        return self.api.run(self.s, self.L, self.P)

    def sync(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def reduce(self, dt, sums):
        return reduce_over_ranks(self.dist, self.torch, self.dev, dt, sums)

    def agree(self, ok):
        """all ranks: did everybody's local part succeed?  (one all-reduce; what run_repeats calls between a rank's runs and their exchange, so
        that a rank-local failure inside a leg that swallows exceptions cannot leave the other ranks waiting in the all-gather)"""
        if self.dist is None:
            return bool(ok)
        t = self.torch.tensor([0.0 if ok else 1.0], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t[0]) == 0.0

    # ---- the timed region: K steps (one full run each) + the exchange of the last step's runs, barrier + sync on both sides
    def timed_steps(self):
        from polychordlite_amd.merge import merge_runs
        args, s, wl = self.args, self.s, self.wl
        nDims, nDer = wl["D"], wl["nDer"]
        # HIP-event stopwatch on the run's own stream.  Warm-up: the kernel classes a round consists of, every launch (picks the two
        # heaviest).  Timed steps: those two, every 8th launch of each -- an event pair costs the stream ~6 us, every launch of two
        # classes would be ~2.5 ms of a 22 ms run, every 8th is ~0.3 ms.
        s.profile = sum(self.cls_bit(n) for n in ("k_nhats", "k_slice", "k_consume", "k_apply", "k_bases_side"))
        top2 = ["k_slice", "k_consume"]
        for i in range(args.warmup):
            w = self.one(-1 - i)
            merge_runs(w, self.comm, nDims, nDer)
            kw = w["kernel_time"]
            if kw:
                top2 = sorted(kw, key=lambda n: -kw[n]["total_s"])[:2]
            w = None
        s.profile = sum(self.cls_bit(n) for n in top2) | (self.TIMED_STRIDE << 8)
        self.sync()
        t0 = time.perf_counter()
        # Every step hands back its dead points in pinned host memory (zero-copy views).  Only the last step's arrays are
        # kept (for the merge); of the earlier ones the numbers: holding all of them made every later run allocate a fresh
        # 45 MB pinned buffer (~3 ms) instead of getting the previous one back from the engine's block cache.
        runs, step_ms, last = [], [], None
        for i in range(args.steps):
            ts0 = time.perf_counter()
            last = None                                 # releases the previous step's result buffers
            last = self.one(i)
            runs.append({k: v for k, v in last.items() if k not in self.BIG})
            step_ms.append((time.perf_counter() - ts0) * 1e3)
        # the exchange step: all-gather of the last step's dead points (rows + entry contours) over RCCL, merged on the device
        tm0 = time.perf_counter()
        merged = merge_runs(last, self.comm, nDims, nDer) if args.steps > 0 else None
        merge_ms = (time.perf_counter() - tm0) * 1e3
        self.sync()
        dt = time.perf_counter() - t0
        tmax, (nlike, nfailed) = self.reduce(dt, [float(sum(r["nlike"] for r in runs)), float(sum(r["nlike_failed"] for r in runs))])
        last = None
        return dict(runs=runs, step_ms=step_ms, merged=merged, merge_ms=merge_ms, dt=dt, tmax=tmax, nlike=nlike, nfailed=nfailed)

    # ---- R runs of every GPU in step + the exchange of all N R runs (its own barrier-bracketed region; never part of `value`)
    def in_step_multi(self):
        args = self.args
        if not (args.runs_per_gpu > 1 and args.steps > 0 and not args.no_extras and args.workload != "c5"):
            return None
        from polychordlite_amd.repeats import run_repeats
        R, api = args.runs_per_gpu, self.api
        s_m = api.Settings(); C.memmove(C.byref(s_m), C.byref(self.s), C.sizeof(self.s)); s_m.profile = 0
        seeds = lambda base, k: [base + 100003 * self.rank + 1000 * k + j for j in range(R)]
        for w in range(2):              # untimed: the block cache for R engines, the kernels' first launches
            _, held = run_repeats(s_m, self.L, self.P, seeds(300000, w), max_in_flight=R, comm=self.comm, agree=self.agree)
            held = None
        samples = []
        for k in range(3):
            self.sync()
            tq0 = time.perf_counter()
            mm, held = run_repeats(s_m, self.L, self.P, seeds(310000, k), max_in_flight=R, comm=self.comm, agree=self.agree)
            held = None
            self.sync()
            tq, (nl_all,) = self.reduce(time.perf_counter() - tq0, [float(mm["nlike_local"])])
            samples.append((nl_all / tq, tq, mm))
        v, tq, mm = sorted(samples, key=lambda t_: t_[0])[1]
        self.lib.polychord_hip_set_option(b"trim_cache", 0.0)
        self.sync()
        return {"runs_per_gpu": R, "n_gpus": self.world, "runs": int(mm["n_runs"]), "value": v, "value_min": min(t_[0] for t_ in samples),
                "value_max": max(t_[0] for t_ in samples), "unit": "likelihood evals/s", "wall_ms": tq * 1e3, "exchange_ms": mm["t_merge_s"] * 1e3,
                "merged_logZ": mm["logZ"], "merged_logZerr": mm["logZerr"], "evidence_rule": mm.get("evidence_rule"),
                "runs_logZ_mean": mm["runs_logZ_mean"], "runs_logZ_sem": mm["runs_logZ_sem"],
                "note": "every rank: R runs in step (pchip_run_repeats, their lived records left on the device), then ONE exchange of all N R runs' records "
                        "(RCCL all-gather inside the library; N = 1: none) and the device merge on every rank; barrier + sync on both sides, max over ranks; median of 3"}

    # ---- figures behind the timed region (N = 1 only; never part of `value`)
    def general_functor(self):
        """the same workload with the closed-form chord evaluation of the built-in quadratic likelihoods switched off: every trial point pays
        a wave reduction like any other device functor would (same trajectory up to round-off)"""
        if not (self.extras and self.wl["kind"] in ("gaussian", "corr_gaussian") and self.args.workload != "c5"):
            return None
        s, wl = self.s, self.wl
        prof_keep = s.profile
        s.ablate = 1; s.profile = 0
        self.one(-100)
        tg0 = time.perf_counter()
        gr = []
        for k in range(3):
            # (only the counters are kept: a result's arrays are views of the engine's pinned buffers, and a run that finds them still held
            #  pins fresh ones -- 2.4 ms a run here until round 6, charged to the functor)
            r = self.one(-101 - k)
            gr.append({key: r[key] for key in ("nlike", "nlike_failed", "logZ")})
            r = None
        self.torch.cuda.synchronize()
        tg = time.perf_counter() - tg0
        # ... and its sampling kernel under the HIP-event stopwatch (one more run, every launch timed; not part of the figure above):
        # k_slice<.., LEAN = 0> with its OWN algorithmic bytes -- the roofline entry of the path any user functor takes
        s.profile = self.cls_bit("k_slice")
        pr = self.one(-110)
        s.ablate = int(os.environ.get("PC_ABLATE", "0")); s.profile = prof_keep
        kt = pr["kernel_time"].get("k_slice")
        out = {"value": sum(r["nlike"] for r in gr) / tg, "unit": "likelihood evals/s", "ms_per_step": tg / 3 * 1e3, "logZ": [r["logZ"] for r in gr],
               "value_reference_equivalent": sum(r["nlike"] - r["nlike_failed"] for r in gr) / tg,
               "note": "built-in Gaussian evaluated like a general device functor (one wave reduction per trial, no closed form along the chord): the same run, counter for counter (tests/test_gpu_parity.py); 3 runs"}
        if kt and kt["launches"] > 0:
            own = own_bytes_per_launch("k_slice", wl["D"], wl["nDer"], wl["nr"], self.nlive, pr["batch"])
            avg = kt["total_s"] / kt["launches"]
            bpe = BYTES_PER_EVAL if self.args.workload == "c2" else algorithmic_bytes_per_iteration(wl["D"], wl["nDer"], wl["nr"], self.nlive) * pr["niter"] / pr["nlike"]
            per_launch = pr["nlike"] / max(1, pr["nbatches"]) * bpe
            rec, name = committed_record(GENERAL_PMC_FILES, ("kernels",))
            hit = sorted([v for k, v in (rec["kernels"] if rec else {}).items() if k.startswith("k_slice")], key=lambda v: -v["launches"])
            out.update({"kernel": "k_slice (general functor)", "avg_launch_us": avg * 1e6, "launches_timed": kt["launches"],
                        "achieved": per_launch / avg / 1e9, "peak": HBM_PEAK_GBS, "frac": per_launch / avg / 1e9 / HBM_PEAK_GBS,
                        "own_bytes_per_launch": own, "own_frac": own / avg / 1e9 / HBM_PEAK_GBS,
                        "whole_run_frac": sum(r["nlike"] for r in gr) * bpe / tg / 1e9 / HBM_PEAK_GBS,
                        "traffic": hit[0]["hbm_bytes_per_launch"] if hit else None, "traffic_source": ("profiles/" + name) if hit else None})
        return out

    def concurrent(self, Rs):
        """R independent runs of the metric configuration in step on this GPU, for each R of the list"""
        if not (self.extras and Rs):
            return None
        from polychordlite_amd.repeats import run_repeats
        api, wl, nlive = self.api, self.wl, self.nlive
        s_c = api.Settings(); C.memmove(C.byref(s_c), C.byref(self.s), C.sizeof(self.s)); s_c.profile = 0
        conc = []
        for R in Rs:
            if R * nlive * (4 * wl["nr"] + 64) * (2 * wl["D"] + wl["nDer"] + 2) * 8 * 5 > 200e9:      # phantom buffers of R engines (pool mode: twice the rows, two buffers)
                continue
            # (untimed: the block cache for R engines, then two more calls -- on the HIP runtime PyTorch brings into this process
            #  the host phases of the calls after the first are slow now and then, 60 -> 72 ms for sixteen runs, with no trip to
            #  the driver in them; value_min / value_max keep what the timed ones saw)
            for w in range(3):
                _, held = run_repeats(s_c, self.L, self.P, [400000 + 1000 * w + j for j in range(R)], max_in_flight=R)
                held = None
            samples = []
            for k in range(3):
                # (the runs' result arrays are views of pinned buffers of the engine: given back before the next call, or every
                #  run of it pins a fresh 45 MB -- 7 ms each, on the one thread that drives them all)
                mc, held = run_repeats(s_c, self.L, self.P, [500000 + 1000 * k + j for j in range(R)], max_in_flight=R)
                held = None
                samples.append(mc)
            vals = sorted(m["nlike"] / m["t_runs_s"] for m in samples)
            mc = sorted(samples, key=lambda m: m["nlike"] / m["t_runs_s"])[1]      # the median sample: wall_ms, per_run_ms and value belong together
            conc.append({"runs": R, "wall_ms": mc["t_runs_s"] * 1e3, "value": vals[1], "value_min": vals[0], "value_max": vals[2], "samples": 3,
                         "unit": "likelihood evals/s", "merge_ms": mc["t_merge_s"] * 1e3, "merged_logZ": mc["logZ"], "merged_logZerr": mc["logZerr"],
                         "per_run_ms": mc["t_runs_s"] * 1e3 / R,
                         "whole_run_frac": mc["nlike"] * 258.0 / mc["t_runs_s"] / 8e12,      # algorithmic bytes (258 B per evaluation) / wall / 8 TB/s
                         "note": "R independent runs of this GPU going round by round together (pchip_run_repeats: one stream, every kernel of a round "
                                 "launched once for all runs, the lane-per-chain sampling kernel); each run bit for bit its solo run; median of 3 samples"})
        self.lib.polychord_hip_set_option(b"trim_cache", 0.0)      # (the blocks of 64 engines: the next configurations size their buffers by what is free)
        self.sync()
        return conc

    def concurrent_clustered(self, names, Rs):
        """the configurations north_star shards over the GPUs (Rastrigin, twin Gaussian: clustered runs) in step on this one"""
        out = {}
        if not (self.extras and Rs and self.args.workload == "c2" and names):
            return out
        from polychordlite_amd.repeats import run_repeats
        api = self.api
        for name in names:
            w3 = WORKLOADS[name]
            s3, L3, P3, keep3 = self.problem(w3, w3["nlive"])
            s3.seed = 6999; api.run(s3, L3, P3)
            ts0 = time.perf_counter(); solo = []
            for i in range(3):
                s3.seed = 7000 + i; g3 = api.run(s3, L3, P3); solo.append((g3["nlike"], int((g3["logweights"] > g3["logzero"]).sum()))); g3 = None
            tsolo = (time.perf_counter() - ts0) / 3
            solo_v = float(np.mean([x[0] for x in solo]) / tsolo); solo_ld = float(np.mean([x[1] for x in solo]) / tsolo)
            rows = []
            for R in Rs:
                for w in range(2):
                    _, held = run_repeats(s3, L3, P3, [400000 + 1000 * w + j for j in range(R)], max_in_flight=R); held = None
                samples = []
                for k in range(3):
                    mc, held = run_repeats(s3, L3, P3, [500000 + 1000 * k + j for j in range(R)], max_in_flight=R)
                    lived = sum(int((h["logweights"] > h["logzero"]).sum()) for h in held); held = None
                    samples.append((mc["nlike"] / mc["t_runs_s"], mc, lived))
                v, mc, lived = sorted(samples, key=lambda t: t[0])[1]
                bpe3 = algorithmic_bytes_per_iteration(w3["D"], w3["nDer"], w3["nr"], w3["nlive"]) * lived / mc["nlike"]
                rows.append({"runs": R, "wall_ms": mc["t_runs_s"] * 1e3, "value": v, "value_min": min(t[0] for t in samples), "value_max": max(t[0] for t in samples),
                             "unit": "likelihood evals/s", "x_solo": v / solo_v, "lived_dead_per_s": lived / mc["t_runs_s"], "x_solo_lived_dead": lived / mc["t_runs_s"] / solo_ld,
                             "evals_per_lived_dead": mc["nlike"] / lived, "evals_per_lived_dead_reference": w3["ref_evals_per_dead"],
                             "value_reference_equivalent": lived / mc["t_runs_s"] * w3["ref_evals_per_dead"],
                             "merged_logZ": mc["logZ"], "merged_logZerr": mc["logZerr"], "evidence_rule": mc.get("evidence_rule"), "merged_logZ_replay": mc.get("logZ_replay"),
                             "runs_logZ_mean": mc["runs_logZ_mean"], "runs_logZ_sem": mc["runs_logZ_sem"], "logZ_truth": w3["truth"],
                             "whole_run_frac": mc["nlike"] * bpe3 / mc["t_runs_s"] / 1e9 / HBM_PEAK_GBS})
            out[name] = {"workload": w3["name"] % w3["nlive"],
                         "solo": {"value": solo_v, "lived_dead_per_s": solo_ld, "ms_per_run": tsolo * 1e3,
                                  "evals_per_lived_dead": float(np.sum([x[0] for x in solo]) / np.sum([x[1] for x in solo]))},
                         "in_step": rows,
                         "note": "R independent runs of this GPU in step (pchip_run_repeats), each bit for bit its solo run; median of 3 calls; merged_logZ = the union's "
                                 "evidence by the rule in evidence_rule (clustered runs: the runs' own evidences combined in linear space; merged_logZ_replay = the replay "
                                 "of the union by ranks and live counts, DESIGN section 8); value_reference_equivalent = dead points that lived per second x the "
                                 "reference binary's evaluations per dead point"}
        self.lib.polychord_hip_set_option(b"trim_cache", 0.0)
        self.sync()
        return out

    def other_configs(self, names):
        """the other BASELINE configurations through the same engine, one timed step each (after two untimed steps that size the block
        cache): reported next to the headline, never part of `value`"""
        if not (self.extras and self.args.workload == "c2" and names):
            return None
        api, torch = self.api, self.torch
        others = {}
        for name in names:
            w2 = WORKLOADS[name]
            s2, L2, P2, keep2 = self.problem(w2, w2["nlive"])
            s2.profile = 0
            s2.seed = 2000
            try:
                for _w in range(2):            # two untimed runs: the first pins and allocates every size a run grows through, the second finds the cache settled
                    api.run(s2, L2, P2)
                    torch.cuda.synchronize()
                    s2.seed += 10
                s2.seed = 2001
                to0 = time.perf_counter()
                r2 = api.run(s2, L2, P2)
                torch.cuda.synchronize()
                to = time.perf_counter() - to0
                # the same run once more under the HIP-event stopwatch (every launch of every kernel class of the main stream: 15 % of a
                # configs[3] run, which is why the figure above is taken without it -- until round 6 it was not): which kernel dominates
                s2.profile = 1
                kt2 = api.run(s2, L2, P2)["kernel_time"]
                torch.cuda.synchronize()
            except RuntimeError as e:           # (e.g. a device without the memory for c5)
                others[name] = {"error": str(e)}
                continue
            dom2 = max(kt2, key=lambda n: kt2[n]["total_s"]) if kt2 else None
            bpe2 = algorithmic_bytes_per_iteration(w2["D"], w2["nDer"], w2["nr"], w2["nlive"]) * r2["niter"] / r2["nlike"]
            lived2 = int((r2["logweights"] > r2["logzero"]).sum())
            others[name] = {"workload": w2["name"] % w2["nlive"], "value": r2["nlike"] / to, "unit": "likelihood evals/s", "ms_per_step": to * 1e3,
                            # evaluations of the chains that put a point into the live set (what the reference's one-chain loop would have
                            # needed for these dead points) / wall; all evaluations per dead point that lived, next to the reference binary's
                            "value_reference_equivalent": (r2["nlike"] - r2["nlike_failed"]) / to, "lived_dead": lived2, "lived_dead_per_s": lived2 / to,
                            "evals_per_lived_dead": r2["nlike"] / lived2, "evals_per_lived_dead_reference": w2["ref_evals_per_dead"], "batch_chains": int(r2["batch"]),
                            "engine_ms": r2["t_total"] * 1e3, "logZ": r2["logZ"], "logZerr": r2["logZerr"], "logZ_truth": w2["truth"],
                            "ndead": int(r2["ndead"]), "nlike": int(r2["nlike"]), "clusters_peak": int(r2["ncluster_peak"]),
                            "dominant_kernel": dom2, "dominant_share_of_kernel_time": (kt2[dom2]["total_s"] / sum(v["total_s"] for v in kt2.values())) if dom2 else None,
                            "kernel_time_s": {n: round(v["total_s"], 6) for n, v in kt2.items()},
                            # which kernels the run went through (pchip_result.path): a silent drop to the general serial kernel shows here
                            "paths": {k: v for k, v in r2["path"].items() if v}, "general_kernel_launches": int(r2["path"]["consume_general"] + r2["path"]["killoff_general"]),
                            "bytes_per_eval": bpe2, "whole_run_frac": r2["nlike"] * bpe2 / to / 1e9 / HBM_PEAK_GBS,
                            "note": "ms_per_step / value: an untimed run; kernel_time_s: the same run again under the HIP-event stopwatch (profile = 1)"}
            r2 = None
        self.sync()
        return others

    # ---- a leg behind the timed region: whatever goes wrong in it is written into the record, and the record still leaves
    def leg(self, name, fn, default):
        try:
            return fn()
        except Exception as e:      # noqa: BLE001 -- the line the driver parses must not depend on an optional figure
            self.leg_errors[name] = ("%s: %s" % (type(e).__name__, e))[:160]
            sys.stderr.write("bench.py: leg %s failed: %s\n" % (name, self.leg_errors[name]))
            return default

    # ---- the roofline block: the kernel class with the largest HIP-event time over the timed steps
    def roofline(self, T):
        args, wl, nlive = self.args, self.wl, self.nlive
        nDims, nDer, nr = wl["D"], wl["nDer"], wl["nr"]
        runs = T["runs"]
        B_ = runs[-1]["batch"]
        evals = float(sum(r["nlike"] for r in runs)); niter = float(sum(r["niter"] for r in runs))
        nurseries = float(sum(r["nbatches"] for r in runs))
        bpe = BYTES_PER_EVAL if args.workload == "c2" else algorithmic_bytes_per_iteration(nDims, nDer, nr, nlive) * niter / evals
        # HBM traffic per launch (PMC counters): measured now when the profiler is at hand (two short profiled runs of this
        # script behind the timed region), else the committed passes of profiles/ for the metric configuration
        pmc, pmc_src = {}, None
        if self.extras and not args.no_live_pmc:
            lp = self.leg("live_pmc", lambda: live_pmc(args.workload), None)
            if lp:
                pmc, pmc_src = lp, "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE --kernel-trace, one pass each, over 3 runs of this workload"
        if not pmc and args.workload == "c2":
            rec, name = committed_record(PMC_FILES, ("kernels",))
            if rec:
                pmc, pmc_src = rec["kernels"], "profiles/" + name
        kern = []
        k_last = runs[-1]["kernel_time"]
        for name in sorted(k_last, key=lambda n: -sum(r["kernel_time"][n]["total_s"] for r in runs)):
            kt = sum(r["kernel_time"][name]["total_s"] for r in runs); kl = sum(r["kernel_time"][name]["launches"] for r in runs)
            if kl == 0:
                continue
            own = own_bytes_per_launch(name, nDims, nDer, nr, nlive, B_)
            avg = kt / kl
            pmc_name = {"k_consume": "k_consume_par" if wl["clustering"] == 0 else "k_consume_cl", "k_bases_side": "k_basis" if nDims > 64 else "k_nhats",
                        "k_nhats": "k_whiten" if nDims > 64 else "k_nhats"}.get(name, name)
            hit = sorted([v for k, v in pmc.items() if k.startswith(pmc_name)], key=lambda v: -v["launches"])
            kern.append({"kernel": name, "avg_launch_us": avg * 1e6, "launches_timed": kl, "timed_every": self.TIMED_STRIDE,
                         "own_bytes_per_launch": own, "own_achieved_GBs": own / avg / 1e9 if own else None,
                         "own_frac": own / avg / 1e9 / HBM_PEAK_GBS if own else None,
                         "traffic": hit[0]["hbm_bytes_per_launch"] if hit else None})
        if not kern:
            return None
        # launches of the dominant class per run: nurseries (k_slice, k_nhats) or rounds (k_consume, k_apply)
        dom = kern[0]
        per_launch_evals = evals / (nurseries if dom["kernel"] in ("k_slice", "k_nhats", "k_bases_side") else float(sum(r["nrounds"] for r in runs)))
        achieved = per_launch_evals * bpe / (dom["avg_launch_us"] * 1e-6) / 1e9
        roof = {"bound": "hbm", "kernel": dom["kernel"], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": dom["traffic"], "avg_launch_us": dom["avg_launch_us"],
                "bytes_per_launch": per_launch_evals * bpe, "bytes_per_eval": bpe, "traffic_source": pmc_src,
                "whole_run_frac": evals * bpe / T["dt"] / 1e9 / HBM_PEAK_GBS,
                "kernels": kern,
                "stream": "side" if dom["kernel"] == "k_bases_side" else "main",
                "note": "latency/parallelism bound path (SURVEY 8d): <= B chains x nDims lanes are live.  achieved = SURVEY 8(d) algorithmic "
                        "bytes per likelihood evaluation (whole path) x evaluations of one launch / that launch's HIP-event time, for the "
                        "class with the largest total time; kernels[] = the two heaviest classes with their OWN algorithmic bytes per launch "
                        "and the PMC traffic of profiles/ (per launch); whole_run_frac = all algorithmic bytes of the timed steps / wall / peak"}
        if args.workload == "c2":
            roof["latency"] = self.leg("latency_model", lambda: latency_model(runs, kern), None)
        return roof


def main():
    args = parse_args()
    rank, local_rank, world, dist, torch = bootstrap(args)
    if args.bootstrap_only:
        dev = "cpu" if args.backend == "gloo" else f"cuda:{local_rank}"
        if dist is not None:
            dist.barrier()
        tmax, (total,) = reduce_over_ranks(dist, torch, dev, 1.0 + rank, [10.0 * (rank + 1)])
        if rank == 0:
            print(json.dumps({"bootstrap": "ok", "world": world, "backend": args.backend, "max": tmax, "sum": total}))
        if dist is not None:
            dist.barrier(); dist.destroy_process_group()
        return
    b = Bench(args, rank, local_rank, world, dist, torch)
    wl, nlive = b.wl, b.nlive
    T = b.timed_steps()                                         # `value` comes from here and from nowhere else
    if world > 1 and rank == 0:
        # N > 1: the record of the timed region leaves BEFORE the optional legs -- R runs per GPU in step and their exchange have never run
        # on more than one GPU, and a line that is already out survives whatever happens there.  The LAST line is the complete record.
        early = {"metric": METRIC[args.workload] % nlive, "value": T["nlike"] / T["tmax"], "unit": "likelihood evals/s", "n_gpus": world, "steps": args.steps,
                 "warmup": args.warmup, "ms_per_step": T["tmax"] / max(args.steps, 1) * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                 "dtype": "f64", "data": "synthetic", "config": {"workload": wl["short"] % nlive + ", one full run per step", "parallelism": "repeat-sharded x%d" % world + GLOO_NOTE * (args.backend == "gloo" and world > 1)},
                 "roofline": None, "cpu_baseline": None, "partial": "timed region only; the complete record follows"}
        print(json.dumps(_num(early), separators=(",", ":")), flush=True)
    # (none of the legs below is ever the reason to lose the timed region's record: Bench.leg)
    multi = b.leg("in_step_multi", b.in_step_multi, None)
    if multi is None and "in_step_multi" in b.leg_errors:
        multi = {"error": b.leg_errors["in_step_multi"][:100], "runs_per_gpu": args.runs_per_gpu, "n_gpus": world}
    ints = lambda txt: [int(x) for x in txt.split(",") if x.strip() and int(x) > 1] if txt else []
    names = lambda txt: [x for x in txt.split(",") if x.strip()] if txt else []
    general = b.leg("general_functor", b.general_functor, None)
    conc = b.leg("concurrent", lambda: b.concurrent(ints(args.concurrent)), [])
    conc_other = b.leg("concurrent_clustered", lambda: b.concurrent_clustered(names(args.concurrent_configs), ints(args.concurrent_clustered) if ints(args.concurrent) else []), {})
    others = b.leg("other_configs", lambda: b.other_configs(names(args.other_configs)), {})
    if rank == 0:
        runs, tmax, nlike, nfailed, merged = T["runs"], T["tmax"], T["nlike"], T["nfailed"], T["merged"]
        value = nlike / tmax
        roof = b.leg("roofline", lambda: b.roofline(T), None)
        if roof and multi:
            roof["in_step_multi"] = multi
        if roof and conc:
            # the GPU's best mode: R runs of the metric configuration in step (never part of `value`; `--gpus N` ranks run ONE run at a time each)
            roof["in_step"] = [{"runs": c_["runs"], "value": c_["value"], "ms_per_run": c_["per_run_ms"], "whole_run_frac": c_["whole_run_frac"]} for c_ in conc if c_["runs"] in (16, 64)]
            for nm, v_ in conc_other.items():
                roof["in_step"] += [{"config": nm, "runs": r_["runs"], "value": r_["value"], "x_solo": r_["x_solo"], "value_reference_equivalent": r_["value_reference_equivalent"],
                                     "whole_run_frac": r_["whole_run_frac"]} for r_ in v_["in_step"]]
        keep_merged = ("n_runs", "logZ", "logZerr", "logZ_replay", "logZerr_replay", "evidence_rule", "runs_logZ_mean", "runs_logZ_sem", "records", "post_mean", "t_merge_s")
        full = {"metric": METRIC[args.workload] % nlive, "value": value,
                "unit": "likelihood evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": tmax / max(args.steps, 1) * 1e3, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": wl["name"] % nlive + ", precision_criterion=1e-3, one full nested-sampling run per step",
                           "workload_short": wl["short"] % nlive + ", one full run per step",
                           "batch_chains": runs[-1]["batch"], "parallelism": "repeat-sharded x%d" % world + GLOO_NOTE * (args.backend == "gloo" and world > 1),
                           "mode": "one run at a time per GPU (a step = one run; every rank of --gpus N runs this mode).  R runs of a GPU in step -- its best "
                                   "mode, `concurrent*`, roofline.in_step and roofline.in_step_multi -- are reported beside it, never in `value`"},
                "logZ": [r["logZ"] for r in runs], "logZerr": [r["logZerr"] for r in runs],
                "logZ_truth": wl["truth"], "ndead": [int(r["ndead"]) for r in runs], "nlike": [int(r["nlike"]) for r in runs],
                # evaluations spent on chains whose spawn failed (a nursery of B chains is seeded from ONE snapshot; the
                # reference's one-chain loop has none): what is left is what the reference would have needed for this evidence
                "evals_reference_equivalent": nlike - nfailed, "value_reference_equivalent": (nlike - nfailed) / tmax,
                "merged": {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in merged.items() if k in keep_merged} if merged else None,
                "step_ms": T["step_ms"], "merge_ms": T["merge_ms"], "general_functor": general, "concurrent": conc,
                "concurrent_c3": conc_other.get("c3"), "concurrent_c4": conc_other.get("c4"), "other_configs": others, "roofline": roof,
                "exchange": ("RCCL all-gather inside the library (%s), %d ranks" % (os.path.basename(b.comm.library or "?"), world)) if b.comm is not None else "one rank: no exchange",
                "kernel_time": {n: v for n, v in runs[-1]["kernel_time"].items()},
                "host_time_s": {k: runs[-1][k] for k in HOST_PHASES},
                "host_time_ms_steps": {k: [round(r[k] * 1e3, 3) for r in runs] for k in HOST_PHASES},
                "rounds": int(runs[-1]["nrounds"]), "batches": int(runs[-1]["nbatches"]),
                "paths": {k: v for k, v in runs[-1]["path"].items() if v},
                "reference_cpu_evals_per_s_survey_container": 357e3}
        cb = b.leg("cpu_baseline", lambda: CpuBaseline(wl, nlive).finish(), None) if world == 1 and not args.no_cpu else None      # (both legs now, alone: every GPU figure has been taken)
        if cb:
            full["cpu_baseline"] = cb
            # wall clock of one run of the reference / of the engine: what a user waits for (the evals/s ratio also counts the
            # engine's failed spawns as work)
            full["speedup_wall_per_run"] = cb["wall_s"] / (tmax / max(args.steps, 1)) if not cb.get("bounded_max_ndead") else None
            full["speedup_evals_per_s"] = value / cb["value"]
        else:
            full["cpu_baseline"] = None
        if b.leg_errors:
            full["leg_errors"] = dict(b.leg_errors)
        full_path = None
        dest = args.full_out if args.full_out is not None else os.path.join("gpurun_out", "bench_full_%s.json" % args.workload)
        if dest:
            try:
                os.makedirs(os.path.dirname(os.path.join(ROOT, dest)) or ".", exist_ok=True)
                with open(os.path.join(ROOT, dest), "w") as f:
                    json.dump(full, f)
                full_path = dest
            except OSError:
                full_path = None
        sys.stdout.flush()
        print(json.dumps(compact_record(full, full_path), separators=(",", ":")))
        sys.stdout.flush()
    if b.comm is not None:
        b.comm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
