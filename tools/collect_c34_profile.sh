#!/bin/bash
# collect_c34_profile.sh <tag> -- run ON THE GPU BOX (gpurun): kernel-trace stats of one bench step at BASELINE configs[2]
# (Rastrigin, clustering) and configs[3] (twin Gaussian, clustering); summaries to gpurun_out/<tag>_c3_kernel_stats.csv and
# <tag>_c4_kernel_stats.csv (copy them into profiles/).
set -u
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for c in c3 c4; do
  out=gpurun_out/prof_${c}_$tag
  rm -rf "$out"; mkdir -p "$out"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o p -- python bench.py --workload $c --steps 2 --warmup 1 --no-cpu --no-extras > "$out/stats.log" 2>&1
  cp "$(find "$out/stats" -name '*kernel_stats.csv' | head -1)" "gpurun_out/${tag}_${c}_kernel_stats.csv"
  head -6 "gpurun_out/${tag}_${c}_kernel_stats.csv" | cut -c1-140
done
