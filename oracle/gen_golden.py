#!/usr/bin/env python
"""gen_golden.py -- produce tests/golden/*.json from the REFERENCE itself (run in the build container
only; needs /root/reference and amdflang).  Fixtures are numbers only (inputs + expected outputs):

  ref_units.json     unit vectors computed by the reference's Fortran modules (oracle/ref_units.f90)
  ref_injected.json  full runs of the reference binary whose `random_number` is fed the oracle's
                     sequential Philox stream (oracle/ref_rng_shim.c): logZ, logZerr, ndead, nlike,
                     #likelihood callback invocations, #uniforms consumed -- the oracle in
                     sequential mode must reproduce every integer exactly and logZ to ~1e-12
  ref_native.json    the untouched reference (own compiler RNG), several seeds: distribution-level
                     targets (logZ within sigma) and the CPU baseline timing of this container
  ref_replay.json    (logL, birth) columns of one reference dead-birth file + its .stats evidence:
                     pins the evidence recursion (SURVEY 8c "deterministic evidence replay")
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
TMP = "/tmp/pc_golden"


def sh(cmd):
    return subprocess.run(["bash", "-c", "ulimit -s unlimited; " + cmd], capture_output=True, text=True, cwd=TMP)


def last_json(out):
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not lines:
        raise RuntimeError(out.stdout[-2000:] + out.stderr[-2000:])
    return json.loads(lines[-1])


GRADE_CASES = [  # like nDims nDerived nlive seed clustering grade_dims grade_repeats
    ("gaussian", 4, 1, 100, 2, 0, [2, 2], [8, 4]), ("gaussian", 20, 2, 100, 3, 0, [8, 12], [20, 24]),
    ("gaussian", 6, 0, 80, 5, 0, [2, 2, 2], [6, 4, 5]), ("rastrigin", 4, 0, 200, 6, 1, [1, 3], [4, 9]),
    ("twin_gaussian", 6, 1, 150, 7, 1, [3, 3], [6, 7]), ("gaussian", 33, 0, 60, 8, 0, [30, 3], [33, 8]),
]


def grade_cases(inj):
    """fast/slow parameter grades with explicit repeats per grade (every grade_frac > 1: generate.F90:303-309);
    REF_GRADES of ref_driver.  nlike_grades = RTI%nlike per grade as printed in .stats"""
    out = []
    for like, D, nDer, nlive, seed, clus, dims, reps in GRADE_CASES:
        env = "REF_GRADES='" + ",".join(map(str, dims)) + ";" + ",".join(map(str, reps)) + "' "
        j = last_json(sh(f"{env}{inj} {like} {D} {nDer} {nlive} {sum(reps)} {seed} {clus} {TMP}/chains inj 0"))
        j.update(nDerived=nDer, clustering=clus, grade_dims=dims, grade_repeats=reps,
                 nlike_grades=[int(x) for x in j["nlike_grades"].split()])
        j.pop("wall", None)
        out.append(j)
        print("injected", j)
    return out


def maximum_cases(inj):
    """<root>.maximum of the reference (maximise = T, maximiser.F90 / nelder_mead.f90) for two injected runs"""
    out = []
    for like, D, nDer, nlive, nr, seed, clus in (("gaussian", 4, 1, 100, 20, 2, 0), ("rastrigin", 2, 0, 300, 6, 2, 1)):
        j = last_json(sh(f"REF_MAXIMISE=1 {inj} {like} {D} {nDer} {nlive} {nr} {seed} {clus} {TMP}/chains mx 0"))
        lines = open(f"{TMP}/chains/mx.maximum").read().splitlines()
        num = lambda k: [float(x) for x in lines[k].split()]
        out.append(dict(like=like, nDims=D, nDerived=nDer, nlive=nlive, num_repeats=nr, seed=seed, clustering=clus,
                        ndead=j["ndead"], max_loglike=num(1)[0], max_point=num(3), max_posterior=num(6)[0],
                        loglike_at_posterior=num(8)[0], posterior_point=num(10)))
        print("maximum", out[-1])
    return out


def file_cases(inj):
    """output files of the reference itself (read_write.F90 writers) for two injected runs: data the engine's writers must
    reproduce byte for byte in its sequential-RNG mode.  -> tests/golden/ref_files/"""
    import shutil
    dst = os.path.join(GOLD, "ref_files")
    os.makedirs(dst, exist_ok=True)
    meta = []
    for name, like, D, nDer, nlive, nr, seed, clus in (("g3", "gaussian", 3, 1, 50, 6, 5, 0), ("r2", "rastrigin", 2, 0, 100, 6, 3, 1)):
        sh(f"rm -f {TMP}/chains/{name}*")
        j = last_json(sh(f"{inj} {like} {D} {nDer} {nlive} {nr} {seed} {clus} {TMP}/chains {name} 1"))
        shutil.copy(f"{TMP}/chains/{name}.stats", os.path.join(dst, name + ".stats"))
        shutil.copy(f"{TMP}/chains/{name}_dead-birth.txt", os.path.join(dst, name + "_dead-birth.txt"))
        meta.append(dict(name=name, like=like, nDims=D, nDerived=nDer, nlive=nlive, num_repeats=nr, seed=seed, clustering=clus, ndead=j["ndead"]))
        print("files", meta[-1])
    json.dump(meta, open(os.path.join(GOLD, "ref_files.json"), "w"), indent=1)


def random_correlated_gaussian_file(D, path, seed=12345, sigma0=0.1):
    """the matrix of bench.py / tests (random_utils.F90:581-614: random orthonormal eigenbasis, eigen-sigma_j = sigma0 (1e-2)^(j/(D-1)))
    in the binary layout ref_driver's REF_COV_FILE reads: D x D inverse covariance, D means, log det(covariance)"""
    import numpy as np
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    sig = sigma0 * (1e-2) ** (np.arange(D) / max(D - 1, 1))
    with open(path, "wb") as f:
        f.write(np.ascontiguousarray(Q @ np.diag(sig ** -2) @ Q.T).tobytes()); f.write(np.full(D, 0.5).tobytes())
        f.write(np.float64(2.0 * np.log(sig).sum()).tobytes())


def seed_runs(name, like, D, nDer, nlive, nr, seeds, comment, workdir=None, reuse=False, clustering=1, env="", workers=6):
    """N runs of the untouched reference binary (own RNG) of one BASELINE configuration -> tests/golden/<name>.json:
    logZ, logZerr, ndead, nlike and the number of clusters that died (local evidences listed in .stats), per seed.
    The runs are independent: 6 at a time on the cores of this container (a 10-D Rastrigin run takes 140-270 s)."""
    import concurrent.futures as cf
    nat = os.path.join(HERE, "_ref", "ref_driver")
    work = workdir or (TMP + "/" + name)
    os.makedirs(work, exist_ok=True)

    def one(seed):
        d = f"{work}/d{seed}"
        stats = f"{d}/r{seed}.stats"
        if not (reuse and os.path.exists(stats) and os.path.getsize(f"{work}/out{seed}.json") > 0):
            out = subprocess.run(["bash", "-c", f"ulimit -s unlimited; {env} {nat} {like} {D} {nDer} {nlive} {nr} {seed} {clustering} {d} r{seed} 0"],
                                 capture_output=True, text=True, cwd=work)
            open(f"{work}/out{seed}.json", "w").write(out.stdout)
        line = [l for l in open(f"{work}/out{seed}.json").read().splitlines() if l.startswith("{")][-1]
        j = json.loads(line)
        ncd = sum(1 for l in open(stats) if l.startswith("log(Z_"))
        return dict(seed=seed, logZ=j["logZ"], logZerr=j["logZerr"], ndead=j["ndead"], nlike=j["nlike"], ncluster_dead=ncd, wall_s=j.get("wall"))

    with cf.ThreadPoolExecutor(workers) as ex:
        runs = list(ex.map(one, seeds))
    json.dump({"_comment": comment, "config": dict(like=like, nDims=D, nDerived=nDer, nlive=nlive, num_repeats=nr, clustering=1),
               "runs": runs}, open(os.path.join(GOLD, name + ".json"), "w"), indent=0)   # (config.clustering below)
    for r in runs:
        print(name, r)


POST_CASES = [  # name like nDims nDerived nlive nrepeats seed clustering REF_POSTERIORS boost cluster_posteriors
    ("pg", "gaussian", 3, 1, 50, 6, 5, 0, "pe", 0.0, 0), ("pgb", "gaussian", 3, 1, 50, 6, 5, 0, "pe", 3.0, 0),
    ("pgp", "gaussian", 4, 1, 100, 20, 2, 0, "p", 0.0, 0), ("pr", "rastrigin", 2, 0, 100, 6, 3, 1, "pe", 0.0, 1),
    ("prb", "rastrigin", 2, 0, 100, 6, 4, 1, "pe", 2.0, 0),
]


def posterior_cases(inj):
    """update_posteriors / clean_phantoms / write_posterior_file of the reference (run_time_info.f90:820-877, :955-1066,
    read_write.F90:479-617): injected runs with posteriors and / or equals switched on -- every Bernoulli trial of the
    thinning is a draw of the injected stream, so the oracle in sequential mode must reproduce nposterior and nequals
    exactly and the two files row for row.  -> tests/golden/ref_posteriors.json + ref_files/<name>.txt / _equal_weights.txt"""
    import shutil
    dst = os.path.join(GOLD, "ref_files")
    os.makedirs(dst, exist_ok=True)
    meta = []
    for name, like, D, nDer, nlive, nr, seed, clus, flags, boost, cpost in POST_CASES:
        sh(f"rm -rf {TMP}/post_{name}")
        env = f"REF_POSTERIORS={flags} REF_BOOST={boost} " + ("REF_CLUSTER_POST=1 " if cpost else "")
        j = last_json(sh(f"{env}{inj} {like} {D} {nDer} {nlive} {nr} {seed} {clus} {TMP}/post_{name} {name} 0"))
        if "p" in flags:
            shutil.copy(f"{TMP}/post_{name}/{name}.txt", os.path.join(dst, name + ".txt"))
        if "e" in flags:
            shutil.copy(f"{TMP}/post_{name}/{name}_equal_weights.txt", os.path.join(dst, name + "_equal_weights.txt"))
        meta.append(dict(name=name, like=like, nDims=D, nDerived=nDer, nlive=nlive, num_repeats=nr, seed=seed, clustering=clus,
                         posteriors=int("p" in flags), equals=int("e" in flags), boost_posterior=boost, cluster_posteriors=cpost,
                         ndead=j["ndead"], nlike=j["nlike"], logZ=j["logZ"], logZerr=j["logZerr"], nposterior=j["nposterior"],
                         nequals=j["nequals"], rng_consumed=j["rng_consumed"]))
        print("posteriors", meta[-1])
    json.dump(meta, open(os.path.join(GOLD, "ref_posteriors.json"), "w"), indent=1)


FARM_CASES = [  # like nDims nDerived nlive nrepeats seed clustering workers (= nprocs - 1)
    ("gaussian", 4, 1, 60, 8, 3, 0, 4), ("gaussian", 20, 2, 100, 20, 1, 0, 8), ("rastrigin", 2, 0, 100, 6, 7, 1, 2),
    ("rastrigin", 2, 0, 300, 6, 2, 1, 8), ("rastrigin", 4, 0, 200, 12, 5, 1, 4), ("twin_gaussian", 6, 1, 150, 12, 4, 1, 8),
    ("rastrigin", 3, 0, 200, 9, 8, 1, 16),
]


def farm_cases():
    """the reference's own FARM in its synchronous mode (nested_sampling.F90:262-286; mpi_utils.F90:376-600), built with -DMPI against
    the container's mpich (oracle/Makefile ref_mpi) and run under mpiexec with workers + 1 ranks, every rank's `random_number` fed
    from its own sequential Philox stream (ref_rng_shim.c pc_shim_set_rank).  The farm stores a worker's babies at the worker's index:
    the run does not depend on the order of arrival (each case is run twice here to show it).  Numbers only; the oracle's farm mode
    (pc_settings.farm) must reproduce them draw for draw: tests/test_oracle_pinned.py."""
    exe = os.path.join(HERE, "_ref", "ref_driver_mpi_inject")
    out = []
    for c in FARM_CASES:
        runs = []
        for rep in range(2):
            r = sh(f"rm -rf {TMP}/farm; /opt/conda/bin/mpiexec -n {c[7] + 1} {exe} {c[0]} {c[1]} {c[2]} {c[3]} {c[4]} {c[5]} {c[6]} {TMP}/farm f 1")
            j = last_json(r)
            rows = [[float(x) for x in l.split()] for l in open(f"{TMP}/farm/f_dead-birth.txt")]
            j["dead_logL_sum"] = sum(x[-2] for x in rows); j["dead_birth_sum"] = sum(x[-1] for x in rows if x[-1] > -1e29)
            j["dead_first_row"] = rows[0]; j["dead_last_row"] = rows[-1]
            runs.append(j)
        a, b = runs
        assert all(a[k] == b[k] for k in ("logZ", "logZerr", "ndead", "nlike", "dead_logL_sum")), (a, b)
        a.update(nDerived=c[2], clustering=c[6], workers=c[7])
        for k in ("wall", "nlike_grades", "calls", "nposterior", "nequals"):
            a.pop(k, None)
        out.append(a)
        print("farm", {k: a[k] for k in ("like", "nDims", "workers", "logZ", "ndead", "nlike", "ncluster")})
    return out


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "farm":       # only tests/golden/ref_farm.json
        os.makedirs(TMP, exist_ok=True)
        json.dump(farm_cases(), open(os.path.join(GOLD, "ref_farm.json"), "w"), indent=1)
        return
    os.makedirs(GOLD, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "posteriors":
        os.makedirs(TMP, exist_ok=True)
        subprocess.check_call(["make", "-C", HERE, "ref"])
        posterior_cases(os.path.join(HERE, "_ref", "ref_driver_inject"))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "c3seeds":      # BASELINE configs[2]; `c3seeds <dir> reuse` collects finished runs
        subprocess.check_call(["make", "-C", HERE, "ref"])
        seed_runs("ref_c3_seeds", "rastrigin", 10, 0, 1000, 30, list(range(1, 13)),
                  "BASELINE configs[2] (10-D Rastrigin, U(-5.12,5.12)^10, nlive 1000, num_repeats 30 = 3 nDims, kNN clustering) run by the "
                  "REFERENCE BINARY with its own RNG, seeds 1..12: oracle/_ref/ref_driver rastrigin 10 0 1000 30 $s 1 <dir> r$s "
                  "(oracle/Makefile builds the driver from /root/reference with amdflang; python oracle/gen_golden.py c3seeds); "
                  "analytic logZ = -23.263; ncluster_dead = local evidences listed in <root>.stats",
                  workdir=sys.argv[2] if len(sys.argv) > 2 else None, reuse=len(sys.argv) > 3)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "c5seeds":      # BASELINE configs[4] at a live-set size the reference finishes in an hour
        subprocess.check_call(["make", "-C", HERE, "ref"])
        work = sys.argv[2] if len(sys.argv) > 2 else TMP + "/ref_c5_seeds"
        os.makedirs(work, exist_ok=True)
        random_correlated_gaussian_file(100, work + "/cov100.bin")
        seed_runs("ref_c5_seeds", "corr_gaussian", 100, 0, 500, 200, list(range(1, 9)),
                  "BASELINE configs[4]'s likelihood and repeats (100-D correlated Gaussian of random_gaussian.f90: random eigenbasis, eigen-sigma 0.1 .. 0.001, "
                  "U(0,1)^100, num_repeats 200 = 2 nDims, no clustering) at nlive 500, run by the REFERENCE BINARY with its own RNG, seeds 1..8: "
                  "REF_COV_FILE=<matrix of bench.py random_correlated_gaussian(100, seed 12345)> oracle/_ref/ref_driver corr_gaussian 100 0 500 200 $s 0 <dir> r$s "
                  "(python oracle/gen_golden.py c5seeds); analytic logZ = 0 up to the mass outside the unit box (< 1e-6); about an hour per run",
                  workdir=work, reuse=len(sys.argv) > 3, clustering=0, env="REF_COV_FILE=" + work + "/cov100.bin", workers=8)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "files":
        os.makedirs(TMP + "/chains/clusters", exist_ok=True)
        subprocess.check_call(["make", "-C", HERE, "ref"])
        file_cases(os.path.join(HERE, "_ref", "ref_driver_inject"))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "maximum":
        os.makedirs(TMP + "/chains/clusters", exist_ok=True)
        subprocess.check_call(["make", "-C", HERE, "ref"])
        json.dump(maximum_cases(os.path.join(HERE, "_ref", "ref_driver_inject")), open(os.path.join(GOLD, "ref_maximum.json"), "w"), indent=1)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "grades":     # refresh only the grade cases of ref_injected.json
        os.makedirs(TMP + "/chains/clusters", exist_ok=True)
        subprocess.check_call(["make", "-C", HERE, "ref"])
        path = os.path.join(GOLD, "ref_injected.json")
        injected = [c for c in json.load(open(path)) if "grade_dims" not in c]
        for c in injected:
            c.pop("nlike_grades", None)
        injected += grade_cases(os.path.join(HERE, "_ref", "ref_driver_inject"))
        json.dump(injected, open(path, "w"), indent=1)
        return
    os.makedirs(TMP, exist_ok=True)
    subprocess.check_call(["make", "-C", HERE, "ref", "_ref/ref_units", "_ref/ref_priors"])
    units = json.loads(subprocess.check_output([os.path.join(HERE, "_ref", "ref_units")], text=True))
    json.dump(units, open(os.path.join(GOLD, "ref_units.json"), "w"), indent=1)
    # prior transforms of the reference's priors_module (oracle/ref_priors.f90)
    priors = json.loads(subprocess.check_output([os.path.join(HERE, "_ref", "ref_priors")], text=True))
    json.dump(priors, open(os.path.join(GOLD, "ref_priors.json"), "w"), indent=1)

    inj = os.path.join(HERE, "_ref", "ref_driver_inject")
    nat = os.path.join(HERE, "_ref", "ref_driver")
    cases = [  # like nDims nDerived nlive nrepeats seed clustering
        ("gaussian", 20, 2, 500, 40, 7, 0), ("gaussian", 20, 2, 100, 20, 1, 0), ("gaussian", 4, 1, 100, 20, 2, 0),
        ("gaussian", 20, 2, 200, 40, 3, 1),
        ("rastrigin", 2, 0, 1000, 6, 7, 1), ("rastrigin", 2, 0, 300, 6, 2, 1), ("rastrigin", 4, 0, 200, 12, 5, 1),
        ("twin_gaussian", 10, 1, 200, 20, 3, 1), ("twin_gaussian", 6, 1, 150, 12, 4, 1),
        # other kernel variants of the engine: nDims > 32, num_repeats > 64, small clustered problems
        ("gaussian", 33, 0, 60, 40, 5, 0), ("gaussian", 6, 2, 80, 70, 6, 0), ("rastrigin", 3, 0, 150, 9, 8, 1),
        ("twin_gaussian", 4, 1, 120, 8, 9, 1),
    ]
    injected = []
    for c in cases:
        j = last_json(sh(f"{inj} {c[0]} {c[1]} {c[2]} {c[3]} {c[4]} {c[5]} {c[6]} {TMP}/chains inj 0"))
        j.update(nDerived=c[2], clustering=c[6])
        j.pop("wall", None); j.pop("nlike_grades", None)
        injected.append(j)
        print("injected", j)
    # dynamic nlive + nprior > nlive (environment of ref_driver: REF_NPRIOR, REF_NLIVES)
    for c, nprior, nlv in ((("gaussian", 4, 1, 100, 8, 9, 0), 160, "-20:60,0:150"), (("gaussian", 4, 1, 100, 8, 4, 0), -1, "-10:200")):
        env = f"REF_NLIVES={nlv} " + (f"REF_NPRIOR={nprior} " if nprior > 0 else "")
        j = last_json(sh(f"{env}{inj} {c[0]} {c[1]} {c[2]} {c[3]} {c[4]} {c[5]} {c[6]} {TMP}/chains inj 0"))
        j.update(nDerived=c[2], clustering=c[6], nprior=nprior, nlives=nlv)
        j.pop("wall", None); j.pop("nlike_grades", None)
        injected.append(j)
        print("injected", j)
    injected += grade_cases(inj)
    json.dump(injected, open(os.path.join(GOLD, "ref_injected.json"), "w"), indent=1)
    json.dump(maximum_cases(inj), open(os.path.join(GOLD, "ref_maximum.json"), "w"), indent=1)
    file_cases(inj)

    native = []
    for seed in range(1, 9):
        j = last_json(sh(f"{nat} gaussian 20 2 500 40 {seed} 0 {TMP}/chains nat 0"))
        j.update(nDerived=2, clustering=0)
        native.append(j)
        print("native", j)
    for seed in (1, 2, 3):
        j = last_json(sh(f"{nat} rastrigin 2 0 1000 6 {seed} 1 {TMP}/chains nat 0"))
        j.update(nDerived=0, clustering=1)
        native.append(j)
        print("native", j)
    j = last_json(sh(f"{nat} gaussian 20 2 2000 40 7 0 {TMP}/chains nat 0"))
    j.update(nDerived=2, clustering=0)
    native.append(j)
    print("native", j)
    json.dump(native, open(os.path.join(GOLD, "ref_native.json"), "w"), indent=1)

    # evidence replay trace: small run with dead-birth output
    j = last_json(sh(f"{nat} gaussian 4 1 100 20 11 0 {TMP}/chains rep 1"))
    rows = [l.split() for l in open(f"{TMP}/chains/rep_dead-birth.txt")]
    logL = [float(r[-2]) for r in rows]
    birth = [float(r[-1]) for r in rows]
    json.dump({"stats": j, "logL": logL, "birth": birth}, open(os.path.join(GOLD, "ref_replay.json"), "w"))
    print("replay rows", len(rows))

    # resume grammar: the .resume file of the reference in the middle of a clustered run (4 live clusters,
    # 4 dead ones, phantoms, posterior stacks) -- data written by the reference's write_resume_file
    import shutil
    sh(f"rm -rf {TMP}/rs && {nat} rastrigin 2 0 40 6 3 1 {TMP}/rs r 0 300")
    shutil.copy(f"{TMP}/rs/r.resume_mid", os.path.join(GOLD, "ref_rastrigin2d_mid.resume"))


if __name__ == "__main__":
    main()
