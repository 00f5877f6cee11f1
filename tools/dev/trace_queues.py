# per hardware queue of the last call in a rocprofv3 kernel trace: busy time, idle, per-kernel totals -- and the launches of one queue over a stretch
# from the middle, in order with the gaps in front of them.  usage: trace_queues.py <p_kernel_trace.csv> [window_ms_from_end] [stretch_ms]
import csv, sys, re, collections
fn = sys.argv[1]; win = float(sys.argv[2]) if len(sys.argv) > 2 else 700.0; stretch = float(sys.argv[3]) if len(sys.argv) > 3 else 6.0
rows = []
for r in csv.DictReader(open(fn)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Queue_Id"]), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"])))
rows.sort()
t_end = max(r[1] for r in rows); t0 = t_end - int(win * 1e6)
W = [r for r in rows if r[0] >= t0]
short = lambda n: re.sub(r"\(.*", "", re.sub(r"^void ", "", n).replace("(anonymous namespace)::", ""))[:40]
span = W[-1][1] - W[0][0]
byq = collections.defaultdict(list)
for r in W: byq[r[3]].append(r)
print("window %.1f ms, %d queues" % (span / 1e6, len(byq)))
main_q = None
for q, L in sorted(byq.items()):
    busy = sum(e - s for s, e, *_ in L)
    tot = collections.Counter(); cnt = collections.Counter()
    for s, e, n, *_ in L: tot[short(n)] += e - s; cnt[short(n)] += 1
    print("queue %d: %d launches, busy %.1f ms (%.0f%% of window); first +%.1f ms, last +%.1f ms" % (q, len(L), busy / 1e6, 100.0 * busy / span, (L[0][0] - W[0][0]) / 1e6, (L[-1][1] - W[0][0]) / 1e6))
    for n, t in tot.most_common(int(sys.argv[4]) if len(sys.argv) > 4 else 9): print("    %-42s %6d x %8.1f us = %8.2f ms" % (n, cnt[n], t / cnt[n] / 1e3, t / 1e6))
    if main_q is None and any("consume" in short(n) for n in tot): main_q = q
    # idle time of the queue by what ran in front of the gap / behind it (gaps of more than 3 us)
    gb = collections.Counter(); ga = collections.Counter(); gn = collections.Counter(); idle = 0
    for (s0, e0, n0, *_), (s1, e1, n1, *_) in zip(L, L[1:]):
        g = s1 - e0
        if g > 3000: gb[short(n0)] += g; ga[short(n1)] += g; gn[short(n0)] += 1; idle += g
    print("    idle in gaps > 3 us: %.1f ms; by the kernel in front: %s" % (idle / 1e6, ", ".join("%s %.1f ms/%d" % (k, v / 1e6, gn[k]) for k, v in gb.most_common(8))))
    print("    by the kernel behind: %s" % ", ".join("%s %.1f ms" % (k, v / 1e6) for k, v in ga.most_common(8)))
if main_q is not None:
    L = byq[main_q]; mid = L[len(L) // 2][0]
    print("queue %d, %.1f ms from the middle (gap in front, duration, grid):" % (main_q, stretch))
    prev = None
    for s, e, n, q, gx, gy in L:
        if s < mid or s > mid + stretch * 1e6: prev = e; continue
        print("  +%8.1f us  gap %7.1f  %7.1f us  grid %5d x %3d  %s" % ((s - mid) / 1e3, (s - prev) / 1e3 if prev else 0.0, (e - s) / 1e3, gx, gy, short(n)))
        prev = e
