import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for c in j['concurrent']: print(c['runs'], round(c['wall_ms'],1), round(c['value']/1e9,2), round(c['value_min']/1e9,2), round(c['value_max']/1e9,2))
