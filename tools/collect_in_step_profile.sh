#!/bin/bash
# collect_in_step_profile.sh <tag> <c3|c4> <R> [B] -- run ON THE GPU BOX (gpurun): kernel-trace stats of R runs of a clustered BASELINE
# configuration in step (tools/dev/gpu_c34_in_step.py); summary to gpurun_out/<tag>_<cfg>_concurrent<R>_kernel_stats.csv (copy into profiles/).
set -u
tag=${1:-r04}; cfg=${2:-c3}; R=${3:-16}; B=${4:-0}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/prof_${cfg}_conc${R}_$tag
rm -rf "$out"; mkdir -p "$out"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o p -- python tools/dev/gpu_c34_in_step.py $cfg $R $B 1 > "$out/stats.log" 2>&1
cp "$(find "$out/stats" -name '*kernel_stats.csv' | head -1)" "gpurun_out/${tag}_${cfg}_concurrent${R}_kernel_stats.csv"
head -24 "gpurun_out/${tag}_${cfg}_concurrent${R}_kernel_stats.csv" | cut -c1-150
tail -3 "$out/stats.log"
