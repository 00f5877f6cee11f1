cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --workload c3 --steps 3 --warmup 1 > gpurun_out/r02_bench_c3.json 2> gpurun_out/c3err.log
timeout 300 python bench.py --workload c4 --steps 3 --warmup 1 > gpurun_out/r02_bench_c4.json 2> gpurun_out/c4err.log
timeout 400 python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu > gpurun_out/r02_bench_c5.json 2> gpurun_out/c5err.log
bash tools/collect_c5_profile.sh r02 > gpurun_out/c5prof.log 2>&1
for c in c3 c4 c5; do python -c "
import json; d=json.load(open('gpurun_out/r02_bench_$c.json')); print('$c', d['ms_per_step'], d['value'], d['logZ'])"; done
head -6 gpurun_out/r02_c5_kernel_stats.csv | cut -c1-160
