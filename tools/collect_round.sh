#!/bin/bash
# collect_round.sh <tag> -- run ON THE GPU BOX (gpurun): every profile of a round in one call; summaries land in gpurun_out/<tag>_*
# (copy them into profiles/).  PMC passes carry --kernel-trace only, one counter per pass.
set -u
tag=${1:-r04}
root="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
cd "$root"
mkdir -p gpurun_out
bash tools/collect_profiles.sh $tag > gpurun_out/${tag}_collect_c2.log 2>&1                 # C2: kernel stats, FETCH / WRITE passes, the bench line
bash tools/collect_c34_profile.sh $tag > gpurun_out/${tag}_collect_c34.log 2>&1             # C3, C4 one run each
bash tools/collect_in_step_profile.sh $tag c3 16 > gpurun_out/${tag}_collect_c3_16.log 2>&1  # sixteen clustered runs in step
bash tools/collect_in_step_profile.sh $tag c4 16 > gpurun_out/${tag}_collect_c4_16.log 2>&1
bash tools/collect_c5_profile.sh $tag > gpurun_out/${tag}_collect_c5.log 2>&1               # C5: kernel stats + matrix-core counters
# sixteen runs of the metric configuration in step: kernel stats and HBM traffic
out=gpurun_out/prof_conc16_$tag; rm -rf "$out"; mkdir -p "$out"
CMD="python tools/dev/gpu_repeats_trace.py 16"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o p -- $CMD > "$out/stats.log" 2>&1
cp "$(find "$out/stats" -name '*kernel_stats.csv' | head -1)" "gpurun_out/${tag}_concurrent16_kernel_stats.csv"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/fetch" -o p -- $CMD > "$out/fetch.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/write" -o p -- $CMD > "$out/write.log" 2>&1
python tools/pmc_summary.py "$out/fetch" "$out/write" "gpurun_out/${tag}_concurrent16_pmc.json" "rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -- $CMD (one pass per counter)"
tail -3 "$out/stats.log"
bash tools/collect_slice_dbg.sh $tag > /dev/null 2>> gpurun_out/${tag}_collect_c2.log
bash tools/collect_issue_profile.sh $tag > /dev/null 2>> gpurun_out/${tag}_collect_c2.log     # instruction counters of the metric configuration's kernels, single-wave issue intervals
# the other BASELINE configurations through the same harness: their own line each (with the CPU baseline, the live PMC pass and -- c3, c4:
# what north_star shards -- sixteen runs in step + the exchange as roofline.in_step_multi)
for w in c3 c4 c5; do timeout 900 python bench.py --workload $w --steps 3 --warmup 1 --concurrent "" --full-out gpurun_out/${tag}_bench_${w}_full.json > gpurun_out/${tag}_bench_$w.json 2> /dev/null; done
find gpurun_out -maxdepth 1 -name "${tag}_*" -size 0 -print -delete | sed "s/^/collect_round.sh: EMPTY (removed): /"      # (nothing empty is carried into profiles/)
ls gpurun_out | grep "^$tag" 
