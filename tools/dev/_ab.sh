cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/full_gpu.log 2>&1; echo "pytest rc $?"
grep -E "passed|failed" gpurun_out/full_gpu.log | tail -2
python bench.py --no-cpu --no-extras --steps 20 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], [round(x,2) for x in d['step_ms']])"
