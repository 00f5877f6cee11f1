#!/bin/bash
# collect_profiles.sh <tag> -- run ON THE GPU BOX (gpurun): kernel-trace stats and the two PMC passes of the bench
# command, summaries copied to profiles/<tag>_* (rocprofv3 output itself stays under gpurun_out/, which is scratch).
# PMC passes carry --kernel-trace only (no other trace domain), one counter per pass.
set -u
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out" profiles
BENCH="python bench.py --no-cpu --no-extras --steps 3 --warmup 1"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o p -- $BENCH > "$out/stats.log" 2>&1
cp "$(find "$out/stats" -name '*kernel_stats.csv' | head -1)" "profiles/${tag}_kernel_stats.csv"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/fetch" -o p -- $BENCH > "$out/fetch.log" 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/write" -o p -- $BENCH > "$out/write.log" 2>&1
python tools/pmc_summary.py "$out/fetch" "$out/write" "profiles/${tag}_pmc.json" "rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -- $BENCH (one pass per counter)"
cp "profiles/${tag}_pmc.json" "profiles/${tag}_kernel_stats.csv" gpurun_out/
# the same three passes for the path ANY device functor takes: the built-in Gaussian evaluated with one reduction per trial (settings.ablate
# bit 0 through PC_ABLATE=1: k_slice<.., LEAN = 5>), i.e. without the closed form along the chord that `value` enjoys
timeout 400 env PC_ABLATE=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/gstats" -o p -- $BENCH > "$out/gstats.log" 2>&1
cp "$(find "$out/gstats" -name '*kernel_stats.csv' | head -1)" "profiles/${tag}_general_kernel_stats.csv"
timeout 400 env PC_ABLATE=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/gfetch" -o p -- $BENCH > "$out/gfetch.log" 2>&1
timeout 400 env PC_ABLATE=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/gwrite" -o p -- $BENCH > "$out/gwrite.log" 2>&1
python tools/pmc_summary.py "$out/gfetch" "$out/gwrite" "profiles/${tag}_general_pmc.json" "PC_ABLATE=1 rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -- $BENCH (one pass per counter; the general-functor path)"
cp "profiles/${tag}_general_pmc.json" "profiles/${tag}_general_kernel_stats.csv" gpurun_out/
# the bench line itself (same command plus the CPU baseline and the concurrent-runs capacity figure), with the PMC file in place
timeout 600 python bench.py --steps 20 --warmup 5 --full-out "gpurun_out/${tag}_bench_full.json" > "profiles/${tag}_bench.json" 2> "$out/bench.log"      # (the compact record the driver parses; the full one beside it)
[ -s "profiles/${tag}_bench.json" ] || { rm -f "profiles/${tag}_bench.json"; echo "collect_profiles.sh: bench.py printed no line:" >&2; tail -5 "$out/bench.log" >&2; }
cp "profiles/${tag}_bench.json" gpurun_out/ 2>/dev/null; cp "gpurun_out/${tag}_bench_full.json" profiles/ 2>/dev/null
head -5 "profiles/${tag}_kernel_stats.csv"
