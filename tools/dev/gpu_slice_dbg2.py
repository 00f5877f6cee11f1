"""dev: whole-kernel view of k_slice on the SLICE_DBG=2 build (PCHIP_LIB; `hipcc -DSLICE_DBG=2 -c pc_sample.hip`, pc_par.hip with
-DPAR_NO_DBG): cycles of every chain before / inside / after the slice loop, the prologue by parts -> JSON (profiles/rNN_slice_kernel_phases.json)"""
import ctypes as C, json, os, re, subprocess, sys
sys.path.insert(0, ".")
if os.environ.get("SLICE_DBG_CHILD") != "1":
    p = subprocess.run([sys.executable, __file__], env=dict(os.environ, SLICE_DBG_CHILD="1", PC_DEBUG="4"), capture_output=True, text=True)
    line = [l for l in p.stderr.splitlines() if "dbg par: stage+search" in l][-1]
    v = [int(x) for x in re.findall(r"(\d+)", line.split("stage+search")[1])][:7]
    res = json.loads(p.stdout.strip().splitlines()[-1])
    n = res["nbatches"] * res["B"]
    names = ["loads_issued_and_seed_choice", "start_point_and_prior_box", "deck_shuffle", "whitening_of_all_directions", "slice_loop", "derived_parameters_and_counts"]
    cyc = {k: v[i] / n for i, k in enumerate(names)}
    out = {"workload": "BASELINE configs[1]: 20-D Gaussian, nlive 2000, num_repeats 40, B 1000", "chains": n, "cycles_per_chain": cyc,
           "cycles_per_chain_total": sum(cyc.values()), "us_at_2.4GHz": {k: c / 2400.0 for k, c in cyc.items()},
           "longest_chain_us_100MHz_clock": v[6] / 100.0,
           "note": "SLICE_DBG=2 build: s_memtime at the phase boundaries of k_slice<1,2,false,1,24>, all chains of all nurseries of one run (atomics into the control block at the end slow the launch itself: its HIP-event time is not comparable)"}
    print(json.dumps(out, indent=1))
    sys.exit(0)
from polychordlite_amd import _ctypes_api as api
lib = api.load()
s = api.Settings(); lib.pchip_settings_default(C.byref(s), 20, 2)
s.nlive, s.num_repeats, s.seed = 2000, 40, 1001
L, P, keep = api.make_problem("gaussian", 20, 2)
api.run(s, L, P)
s.seed = 1002
r = api.run(s, L, P)
print(json.dumps({"nbatches": r["nbatches"], "B": r["batch"]}))
