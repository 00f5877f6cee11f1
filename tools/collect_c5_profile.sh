#!/bin/bash
# collect_c5_profile.sh <tag> -- ON THE GPU BOX: BASELINE configs[4] (100-D correlated Gaussian, nlive 5000,
# num_repeats 200, one full run): kernel-trace stats, then one PMC pass with the matrix-core counters
# (--kernel-trace only beside --pmc).  Summaries -> gpurun_out/ (copy into profiles/ afterwards).
set -u
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/prof_c5_$tag
rm -rf "$out"; mkdir -p "$out"
CMD="python tools/dev/gpu_c5_full.py 5000 200"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o p -- $CMD > "$out/stats.log" 2>&1
cp "$(find "$out/stats" -name '*kernel_stats.csv' | head -1)" "gpurun_out/${tag}_c5_kernel_stats.csv"
tail -1 "$out/stats.log" > "gpurun_out/${tag}_c5_run.txt"
rocprofv3 -L 2>/dev/null | grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u > "$out/mfma_counters.txt"
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace --output-format csv -d "$out/mfma" -o p -- $CMD > "$out/mfma.log" 2>&1
python - "$out/mfma" "gpurun_out/${tag}_c5_mfma.json" <<'PY'
import csv, glob, json, os, sys
acc = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        d = acc.setdefault(k, {})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d["_dur_ns"] = d.get("_dur_ns", 0.0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / max(1, len([1]))
        d["_rows"] = d.get("_rows", 0) + 1
json.dump({k: v for k, v in acc.items() if any("MFMA" in c and v[c] > 0 for c in v if not c.startswith("_"))}, open(sys.argv[2], "w"), indent=1)
PY
cat "$out/mfma_counters.txt" | head -20
head -8 "gpurun_out/${tag}_c5_kernel_stats.csv"; cat "gpurun_out/${tag}_c5_mfma.json" | head -40; tail -3 "$out/mfma.log"
