"""dev: one run of the metric configuration on the SLICE_DBG build (PCHIP_LIB), the section counters of k_slice read from the
engine's PC_DEBUG=4 line -> JSON on stdout.  Sections (pc_sample.hip k_slice, chain 0 of every nursery, summed over the run):
  0 direction take-over + Philox call (every 4th slice), 1 chord coefficients (three wave sums) + initial bracket,
  2 stepping out, 3 shrinkage (four speculative trials + loop), 4 stores of the baby."""
import ctypes as C, json, os, re, subprocess, sys
sys.path.insert(0, ".")
if os.environ.get("SLICE_DBG_CHILD") != "1":
    env = dict(os.environ, SLICE_DBG_CHILD="1")
    p = subprocess.run([sys.executable, __file__] + sys.argv[1:], env=env, capture_output=True, text=True)
    line = [l for l in p.stderr.splitlines() if "dbg par: stage+search" in l][-1]
    v = [int(x) for x in re.findall(r"(\d+)", line.split("stage+search")[1])][:7]
    res = json.loads(p.stdout.strip().splitlines()[-1])
    nsl = res["nbatches"] * res["num_repeats"]
    names = ["take_over_and_philox", "coefficients_and_initial_bracket", "stepping_out", "shrinkage", "stores"]
    out = {"workload": "BASELINE configs[1]: 20-D Gaussian, nlive 2000, num_repeats 40, B 1000", "slices_of_chain_0": nsl,
           "cycles_per_slice": {n: v[i] / nsl for i, n in enumerate(names)}, "cycles_per_slice_total": sum(v[:5]) / nsl,
           "shrink_loop_evaluations_beyond_the_four_speculative_per_slice": v[5] / nsl,
           "evaluations_per_slice": res["nlike"] / res["niter"] / res["num_repeats"], "k_slice_us_per_launch_hip_events": res["k_slice_us"],
           "note": "s_memtime at the section boundaries (each read costs the wave ~40 cycles: included)"}
    if len(sys.argv) > 1 and os.path.exists(sys.argv[1]):
        out["ubench_single_wave"] = open(sys.argv[1]).read().strip().splitlines()
    print(json.dumps(out, indent=1))
    sys.exit(0)
from polychordlite_amd import _ctypes_api as api
lib = api.load()
s = api.Settings(); lib.pchip_settings_default(C.byref(s), 20, 2)
s.nlive, s.num_repeats, s.seed = 2000, 40, 1001
s.profile = 1 << 2
L, P, keep = api.make_problem("gaussian", 20, 2)
api.run(s, L, P)
s.seed = 1002
r = api.run(s, L, P)
kt = r["kernel_time"]["k_slice"]
print(json.dumps({"nbatches": r["nbatches"], "num_repeats": 40, "nlike": r["nlike"], "niter": r["niter"], "k_slice_us": kt["total_s"] / kt["launches"] * 1e6}))
