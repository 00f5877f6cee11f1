#!/bin/bash
# collect_issue_profile.sh <tag> -- run ON THE GPU BOX (gpurun): the instruction counters of the metric configuration's kernels (one run of
# tools/dev/gpu_one_run.py; --pmc with --kernel-trace only) and the single-wave issue intervals of tools/dev/ubench_fp64.hip:
# gpurun_out/<tag>_issue.json (copy into profiles/).  k_slice runs one wavefront per SIMD: its time is its instruction count times the
# interval at which ONE wavefront issues, whatever the dependences -- that is the bound bench.py's roofline.latency quotes.
set -u
tag=${1:-r06}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/prof_issue_$tag; rm -rf "$out"; mkdir -p "$out"
hipcc -O2 --offload-arch=gfx950 tools/dev/ubench_fp64.hip -o /tmp/ubench_fp64 2>/dev/null && /tmp/ubench_fp64 | head -1 > "$out/ubench_fp64.txt"
for c in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/$c" -o p -- python tools/dev/gpu_one_run.py 8192 > "$out/$c.log" 2>&1
done
python - "$out" "$tag" <<'PY'
import csv, glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
res = collections.defaultdict(dict)
for d in sorted(glob.glob(out + "/SQ_*")):
    if not d.split("/")[-1].startswith("SQ_") or d.endswith(".log"): continue
    c = d.split("/")[-1]
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs: continue
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(fs[0])):
        if r.get("Counter_Name") != c: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
    for k, (n, v) in acc.items(): res[k][c] = v / n; res[k]["launches"] = n
keep = {k: v for k, v in res.items() if k.startswith(("k_slice", "k_consume_par", "k_nhats", "k_apply_pool", "k_upd"))}
rec = {"command": "rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -- python tools/dev/gpu_one_run.py 8192 (one pass per counter; per launch; settings.ablate bit 13: ONE wavefront a chain, so that the counters are the chain's own instruction stream -- the product kernel gives the deck, the whitening and nine of ten Philox calls to a helper wavefront)",
       "ubench_fp64": open(out + "/ubench_fp64.txt").read().strip() if glob.glob(out + "/ubench_fp64.txt") else None, "kernels": keep}
json.dump(rec, open("gpurun_out/%s_issue.json" % tag, "w"), indent=1)
print(json.dumps({k: v for k, v in keep.items() if k.startswith("k_slice")}, indent=1))
PY
