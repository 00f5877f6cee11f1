"""dev: rocprofv3 kernel_stats.csv -> short table (name, calls, total ms, average us); argv[2] = divide totals by this many calls of the script's loop"""
import csv, re, sys
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 18]:
    n = re.sub(r"\(.*", "", r["Name"]).replace("void ", "").replace("(anonymous namespace)::", "")
    print(f"{n[:34]:34s} {int(r['Calls']):6d} calls {float(r['TotalDurationNs'])/1e6/div:8.2f} ms {float(r['AverageNs'])/1e3:8.1f} us  {r['Percentage']}%")
