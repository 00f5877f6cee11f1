for i in 1 2 3; do
for l in libpolychord_hip.so libpolychord_hip_base.so; do
PCHIP_LIB=$PWD/polychordlite_amd/$l timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu --other-configs '' --concurrent-configs '' --concurrent 16,64 --no-live-pmc 2>/dev/null | tail -1 > gpurun_out/ab_tmp.json
python - "$l" <<PY
import json,sys
d=json.load(open("gpurun_out/ab_tmp.json"))
print(sys.argv[1], [(x["runs"],round(x["value"]/1e9,3)) for x in d["roofline"]["in_step"]])
PY
done; done
