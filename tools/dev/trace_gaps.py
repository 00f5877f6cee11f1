"""dev: timeline of ONE run from a rocprofv3 --kernel-trace CSV: busy time per kernel, and the idle gaps of the device
between consecutive kernels grouped by (previous kernel -> next kernel).  usage: trace_gaps.py <kernel_trace.csv> [run_index_from_end]"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")) for r in rows))
# split into runs at k_init_state
starts = [i for i, e in enumerate(ev) if e[2].startswith("k_init_state")]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
a = starts[-k]; b = starts[-k + 1] if k > 1 else len(ev)
run = ev[a:b]
# cut at k_post_moments (end of the run)
for i, e in enumerate(run):
    if e[2].startswith("k_post_moments"):
        run = run[:i + 1]; break
span = run[-1][1] - run[0][0]
busy = collections.Counter(); cnt = collections.Counter()
gaps = collections.Counter(); gcnt = collections.Counter()
cur_end = run[0][0]; union = 0
prev = None
for s, e, n in run:
    busy[n] += e - s; cnt[n] += 1
    if s > cur_end:
        if prev: gaps[(prev, n)] += s - cur_end; gcnt[(prev, n)] += 1
        union += e - s; cur_end = e; prev = n
    else:
        if e > cur_end: union += e - cur_end; cur_end = e; prev = n
print(f"span {span/1e6:.2f} ms, device busy (union) {union/1e6:.2f} ms, idle {(span-union)/1e6:.2f} ms, kernels {len(run)}")
for n, t in busy.most_common(20): print(f"  {t/1e6:7.3f} ms  {cnt[n]:5d} x {t/cnt[n]/1e3:7.1f} us  {n[:70]}")
print("idle gaps:")
for (p, n), t in gaps.most_common(16): print(f"  {t/1e6:7.3f} ms  {gcnt[(p,n)]:5d} x {t/gcnt[(p,n)]/1e3:6.1f} us  {p[:32]} -> {n[:32]}")
