cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not high_dim" > gpurun_out/ab_pytest.log 2>&1; echo "pytest rc $?" > gpurun_out/ab_par.log
grep -E "passed|failed" gpurun_out/ab_pytest.log | tail -2 >> gpurun_out/ab_par.log
for i in 1 2 3; do
for which in base new; do
if [ $which = base ]; then export PCHIP_LIB=$PWD/polychordlite_amd/libpc_base.so; else unset PCHIP_LIB; fi
python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$which', round(d['ms_per_step'],3), [(k['kernel'], round(k['avg_launch_us'],1)) for k in d['roofline'].get('kernels', [])], d['logZ'][:2])" >> gpurun_out/ab_par.log
done; done
unset PCHIP_LIB
PC_DEBUG=4 python bench.py --steps 2 --warmup 1 --no-cpu 2>&1 >/dev/null | grep "dbg par" | tail -1 >> gpurun_out/ab_par.log
cat gpurun_out/ab_par.log
