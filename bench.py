#!/usr/bin/env python
"""bench.py -- likelihood evaluations / second of the MI355X nested-sampling engine.

Contract (see DESIGN.md "Measurement"):
  * a STEP is one complete nested-sampling run of BASELINE.json configs[1]: 20-D Gaussian
    (likelihoods/examples/gaussian.f90, ini/gaussian.ini priors), nlive = 2000, num_repeats = 40,
    precision_criterion 1e-3, no clustering, fp64 throughout; step i uses seed = 1000 + i (+ rank*100003).
  * metric  = likelihood evals/sec = sum of the reference's own counter RTI%nlike (calculate.f90:44)
    over the timed steps / wall time (barrier + device sync on both sides, max over ranks).
  * N > 1 GPUs: repeat-sharded: every rank runs independent runs with its own seeds; the
    (logL, birth) records of all dead points are all-gathered over RCCL (torch.distributed "nccl") and
    merged into one evidence by the replay recursion; scaling is "weak".
  * roofline: the dominant kernel's algorithmic HBM bytes (258 B / evaluation at this
    config) over its HIP-event time, against 8 TB/s.  This path is latency bound; the fraction says so.
  * cpu_baseline: the REFERENCE itself (oracle/_ref/ref_driver, built from /root/reference by
    oracle/Makefile) when the prebuilt binary is present, else the C restatement (oracle/liboracle.so),
    on one host core (the reference is single threaded; its MPI farm does not speed this likelihood up,
    BASELINE.md), on one full run of the same workload (about 10-25 s).
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
# BASELINE.json configs through this harness: kind, nDims, nDerived, nlive, num_repeats, clustering, box, analytic logZ
WORKLOADS = {
    "c2": dict(kind="gaussian", D=20, nDer=2, nlive=2000, nr=40, clustering=0, box=None, truth=0.0,
               name="BASELINE configs[1]: 20-D Gaussian (mu=0.5, sigma=0.1, U(0,1)^20), nlive=%d, num_repeats=40"),
    "c3": dict(kind="rastrigin", D=10, nDer=0, nlive=1000, nr=30, clustering=1, box=(-5.12, 5.12), truth=-23.26,
               name="BASELINE configs[2]: 10-D Rastrigin, U(-5.12,5.12)^10, nlive=%d, num_repeats=30 (= 3 nDims, the ini's ratio), kNN clustering"),
    "c4": dict(kind="twin_gaussian", D=30, nDer=1, nlive=500, nr=40, clustering=1, box=(-1.0, 1.0), truth=-20.79,
               name="BASELINE configs[3]: 30-D twin Gaussian (sigma=0.1), U(-1,1)^30, nlive=%d, num_repeats=40, kNN clustering"),
    "c5": dict(kind="corr_gaussian", D=100, nDer=0, nlive=5000, nr=200, clustering=0, box=None, truth=None,
               name="BASELINE configs[4]: 100-D correlated Gaussian (random eigenbasis, eigen-sigma 0.1 .. 0.001), U(0,1)^100, nlive=%d, num_repeats=200"),
}


def algorithmic_bytes_per_iteration(D, nDer, nr, N):
    """chain I/O + covariance pass + phantom compaction + contour scan, per nested-sampling iteration
    (phi = nr + 1 phantoms per live point in steady state)"""
    nT, phi = 2 * D + nDer + 2, nr + 1
    return 8.0 * nT * (1 + nr) + 8.0 * (1 + phi) * D + 8.0 * phi * nT + 8.0 * N


def random_correlated_gaussian(D, seed=12345, sigma0=0.1):
    """random_gaussian.f90 / random_utils.F90:581-614: random orthonormal eigenbasis, eigen-sigma_j = sigma0 (1e-2)^(j/(D-1))"""
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    sig = sigma0 * (1e-2) ** (np.arange(D) / max(D - 1, 1))
    return Q @ np.diag(sig ** -2) @ Q.T, np.full(D, 0.5), float(2.0 * np.log(sig).sum())
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(nDims, nDer, nr):
    """reference (preferred) or restatement on ONE host core, bounded sample"""
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    sample = "the workload itself: 20-D Gaussian, nlive=2000, num_repeats=40, one full run, seed 7"
    if os.path.exists(ref):
        tmp = "/tmp/pc_ref_bench"
        os.makedirs(tmp, exist_ok=True)
        cmd = f"ulimit -s unlimited; {ref} gaussian {nDims} {nDer} 2000 {nr} 7 0 {tmp} ref 0"
        out = subprocess.run(["bash", "-c", cmd], capture_output=True, text=True, cwd=tmp)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if line:
            j = json.loads(line[-1])
            return {"value": j["nlike"] / j["wall"], "unit": "likelihood evals/s", "cores": 1, "kind": "reference",
                    "sample": sample + "; PolyChordLite Fortran built with amdflang -O2, file output off",
                    "logZ": j["logZ"], "logZerr": j["logZerr"], "ndead": j["ndead"], "nlike": j["nlike"], "wall_s": j["wall"]}
    from tests import oracle_api as orc
    s = orc.settings(nDims, nDer, nlive=2000, num_repeats=nr, seed=7, batch=1)
    L, P, keep = orc.make_problem("gaussian", nDims)
    t0 = time.time(); o = orc.run(s, L, P); dt = time.time() - t0
    return {"value": o["nlike"] / dt, "unit": "likelihood evals/s", "cores": 1, "kind": "port",
            "sample": sample + "; oracle/liboracle.so (C restatement, gcc -O2)", "logZ": o["logZ"],
            "logZerr": o["logZerr"], "ndead": int(o["ndead"]), "nlike": int(o["nlike"]), "wall_s": dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nlive", type=int, default=2000)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS),
                    help="c2 = BASELINE configs[1], the metric configuration (default, what the driver runs); c3 / c4 / c5 = "
                         "BASELINE configs[2..4] through the same harness (their lines are kept under profiles/)")
    ap.add_argument("--concurrent", type=int, default=4,
                    help="after the timed steps: R independent runs driven concurrently from R host threads on this GPU "
                         "(reported separately, never part of `value`); 0 = skip")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from polychordlite_amd import _ctypes_api as api
    from polychordlite_amd.merge import merge_runs
    lib = api.load()
    if lib.pchip_device_count() < 1:
        raise SystemExit("bench.py: no HIP device visible; the engine has no CPU path")

    wl = WORKLOADS[args.workload]
    if args.workload != "c2" and args.nlive == 2000:
        args.nlive = wl["nlive"]
    nDims, nDer, nr = wl["D"], wl["nDer"], wl["nr"]
    s = api.Settings(); lib.pchip_settings_default(C.byref(s), nDims, nDer)
    s.nlive = args.nlive; s.num_repeats = nr; s.batch = args.batch; s.device = local_rank
    s.do_clustering = wl["clustering"]
    # HIP-event stopwatch: the warm-up steps time the two heaviest kernel classes (slice sampling = the likelihood
    # evaluations, contraction), the timed steps only the one that came out on top -- every timed launch costs two
    # event records on the run's stream (all six classes: ~4 ms per 25 ms run, two classes: ~2.5 ms)
    PROFILE_BOTH = (1 << (1 + 1)) | (1 << (2 + 1))
    s.profile = PROFILE_BOTH
    if wl["kind"] == "corr_gaussian":
        ic, mean, logdet = random_correlated_gaussian(nDims)
        L, P, keep = api.make_problem("corr_gaussian", nDims, nDer, invcov=ic, mean=mean, logdet=logdet)
    else:
        lo, hi = wl["box"] if wl["box"] else (None, None)
        L, P, keep = api.make_problem(wl["kind"], nDims, nDer, lo, hi)

    def one(i):
        s.seed = 1000 + i + 100003 * rank
        return api.run(s, L, P)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        w = one(-1 - i)
        merge_runs(w, dist, torch, local_rank)
        kw = w["kernel_time"]
        if kw:
            # (k_slice unless another class is clearly ahead: at the metric config the two are within a few per cent
            #  of each other under the stopwatch, and the kernel trace in profiles/ has k_slice on top)
            top = max(kw, key=lambda n: kw[n]["total_s"] * (1.2 if n == "k_slice" else 1.0))
            s.profile = 1 << (api.KERNEL_CLASSES.index(top) + 1)
        w = None
    sync()
    t0 = time.perf_counter()
    # Every step hands back its dead points in pinned host memory (zero-copy views).  Only the last step's arrays are
    # kept (for the merge); of the earlier ones the numbers: holding all of them made every later run allocate a fresh
    # 45 MB pinned buffer (~3 ms) instead of getting the previous one back from the engine's block cache.
    BIG = ("dead", "logweights", "entry", "live")
    runs, step_ms, last = [], [], None
    for i in range(args.steps):
        ts0 = time.perf_counter()
        last = None                                 # releases the previous step's result buffers
        last = one(i)
        runs.append({k: v for k, v in last.items() if k not in BIG})
        step_ms.append((time.perf_counter() - ts0) * 1e3)
    # repeat-sharded merge: all-gather (logL, entry contour) of every dead point of the last step's runs
    tm0 = time.perf_counter()
    merged = merge_runs(last, dist, torch, local_rank) if args.steps > 0 else None
    merge_ms = (time.perf_counter() - tm0) * 1e3
    sync()
    dt = time.perf_counter() - t0
    tmax = dt
    nlike = float(sum(r["nlike"] for r in runs))
    if dist is not None:
        t = torch.tensor([dt, nlike], dtype=torch.float64, device=f"cuda:{local_rank}")
        tm = t.clone(); dist.all_reduce(tm, op=dist.ReduceOp.MAX); tmax = float(tm[0])
        ts = t.clone(); dist.all_reduce(ts, op=dist.ReduceOp.SUM); nlike = float(ts[1])
    # one run keeps a small part of the chip busy (B chains = B wavefronts, the contraction one CU): independent
    # runs on separate streams overlap.  Reported next to the headline, not in it.
    conc = None
    if args.concurrent > 1 and args.steps > 0:
        from polychordlite_amd.repeats import run_repeats
        R = args.concurrent
        s_c = api.Settings(); C.memmove(C.byref(s_c), C.byref(s), C.sizeof(s)); s_c.profile = 0
        run_repeats(s_c, L, P, [400000 + j + 100003 * rank for j in range(R)], max_in_flight=R)     # block cache for R engines
        mc, _ = run_repeats(s_c, L, P, [500000 + j + 100003 * rank for j in range(R)], max_in_flight=R)
        tc = mc["t_runs_s"]
        nl = mc["nlike"]
        conc = {"runs": R, "wall_ms": tc * 1e3, "value": nl / tc, "unit": "likelihood evals/s", "merge_ms": mc["t_merge_s"] * 1e3,
                "merged_logZ": mc["logZ"], "merged_logZerr": mc["logZerr"],
                "note": "R independent runs of the same workload in flight on one GPU (one host thread + HIP stream each, "
                        "polychordlite_amd.repeats.run_repeats); wall_ms = the runs, merge_ms = evidence replay of their union on the host"}
        sync()
    if rank == 0:
        value = nlike / tmax
        k = runs[-1]["kernel_time"]
        dom = max(k, key=lambda n: k[n]["total_s"]) if k else None
        roof = None
        if dom:
            kt = sum(r["kernel_time"][dom]["total_s"] for r in runs)
            kl = sum(r["kernel_time"][dom]["launches"] for r in runs)
            evals = float(sum(r["nlike"] for r in runs))
            niter = float(sum(r["niter"] for r in runs))
            bpe = BYTES_PER_EVAL if args.workload == "c2" else algorithmic_bytes_per_iteration(nDims, nDer, nr, args.nlive) * niter / evals
            achieved = evals * bpe / kt / 1e9
            # HBM bytes per launch of that kernel from the PMC passes committed under profiles/ (rocprofv3 --pmc
            # FETCH_SIZE / WRITE_SIZE in separate runs of this command; FETCH_SIZE doubled on gfx950)
            traffic, pmc_src = None, os.path.join(ROOT, "profiles", "r01_pmc.json")
            if os.path.exists(pmc_src) and args.workload == "c2":
                pk = json.load(open(pmc_src))["kernels"]
                hit = [v for k, v in pk.items() if k.startswith(dom if dom != "k_consume" else "k_consume_par")]
                if hit:
                    traffic = hit[0]["hbm_bytes_per_launch"]
            # the kernel's own algorithmic I/O per launch, to read `traffic` against
            nT, B_, N_ = 2 * nDims + nDer + 2, runs[-1]["batch"], args.nlive
            own = {"k_slice": B_ * (8 * nT * (1 + nr) + 8 * nr * (nDims + 1) + 16 * nr),
                   "k_consume": 8 * (2 * N_ + 3 * B_) + 120 * B_,
                   "k_nhats": B_ * 8 * nr * (nDims + 1)}.get(dom)
            roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel_own_bytes_per_launch": own,
                    "avg_launch_us": kt / max(kl, 1) * 1e6, "launches": kl,
                    "bytes_per_launch": evals * bpe / max(kl, 1), "bytes_per_eval": bpe,
                    "note": "latency/parallelism bound path (SURVEY 8d): <=B chains x nDims lanes are live; algorithmic bytes = "
                            "SURVEY 8(d) bytes per likelihood evaluation (258 B at the metric config) x evaluations of one nursery; "
                            "traffic = PMC bytes of this kernel alone"}
        metric_name = {"c2": "likelihood evals/sec, 20D Gaussian nlive=%d", "c3": "likelihood evals/sec, 10D Rastrigin nlive=%d",
                       "c4": "likelihood evals/sec, 30D twin Gaussian nlive=%d", "c5": "likelihood evals/sec, 100D correlated Gaussian nlive=%d"}
        out = {"metric": metric_name[args.workload] % args.nlive, "value": value,
               "unit": "likelihood evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": tmax / max(args.steps, 1) * 1e3, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": wl["name"] % args.nlive + ", precision_criterion=1e-3, one full nested-sampling run per step",
                          "batch_chains": runs[-1]["batch"], "parallelism": "repeat-sharded x%d" % world},
               "logZ": [r["logZ"] for r in runs], "logZerr": [r["logZerr"] for r in runs],
               "logZ_truth": wl["truth"], "ndead": [int(r["ndead"]) for r in runs], "nlike": [int(r["nlike"]) for r in runs],
               "merged": merged, "step_ms": step_ms, "merge_ms": merge_ms, "concurrent": conc, "roofline": roof,
               "kernel_time": {n: v for n, v in runs[-1]["kernel_time"].items()},
               "host_time_s": {k: runs[-1][k] for k in ("t_setup", "t_generate", "t_loop", "t_final", "t_results", "t_teardown")},
               "rounds": int(runs[-1]["nrounds"]), "batches": int(runs[-1]["nbatches"]),
               "reference_cpu_evals_per_s_survey_container": 357e3}
        if not args.no_cpu and world == 1 and args.workload == "c2":
            out["cpu_baseline"] = cpu_baseline(nDims, nDer, nr)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
