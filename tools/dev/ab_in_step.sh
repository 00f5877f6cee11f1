run() { env "$@" timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu --other-configs '' --concurrent-configs '' --concurrent 16,64 --no-live-pmc 2>/dev/null | tail -1 > gpurun_out/ab_tmp.json
python - "$*" <<PY
import json,sys
d=json.load(open("gpurun_out/ab_tmp.json"))
print(sys.argv[1], d["ms_per_step"], [(x["runs"],round(x["value"]/1e9,3)) for x in d["roofline"]["in_step"]])
PY
}
for i in 1 2; do
run A=1
run PC_BASES_OWN=1
run PC_COHORT_AHEAD=2 PC_BASES_OWN=1
run PC_COHORT_AHEAD=2
done
