# usage: ab_env.sh VAR  -- bench.py's in-step figures with and without VAR=1 in the environment, twice each, in one call (one box)
v=${1:?variable}
for i in 1 2; do
for a in "" 1; do
if [ -n "$a" ]; then export $v=1; else unset $v; fi
timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu --other-configs '' --concurrent-configs '' --concurrent 16,64 --no-live-pmc 2>/dev/null | tail -1 > gpurun_out/ab_tmp.json
python - <<PY
import json
d=json.load(open("gpurun_out/ab_tmp.json"))
print("$v=$a", d["ms_per_step"], [(x["runs"],round(x["value"]/1e9,3)) for x in d["roofline"]["in_step"]])
PY
done; done
