# timeline of the last call in a rocprofv3 kernel trace: per-kernel totals inside the window, union busy time, idle, and one round's launches in order.
# usage: trace_timeline.py <p_kernel_trace.csv> [window_ms_from_end] [round_marker_kernel_substring]
import csv, sys, re, collections
fn = sys.argv[1]; win = float(sys.argv[2]) if len(sys.argv) > 2 else 700.0; mark = sys.argv[3] if len(sys.argv) > 3 else "k_slice_many"
rows = []
for r in csv.DictReader(open(fn)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Queue_Id"]), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"])))
rows.sort()
t_end = max(r[1] for r in rows); t0 = t_end - int(win * 1e6)
W = [r for r in rows if r[0] >= t0]
short = lambda n: re.sub(r"\(.*", "", re.sub(r"^void ", "", n).replace("(anonymous namespace)::", ""))[:44]
tot = collections.Counter(); cnt = collections.Counter()
for s, e, n, q, gx, gy in W: tot[short(n)] += e - s; cnt[short(n)] += 1
ev = sorted([(s, 1) for s, e, *_ in W] + [(e, -1) for s, e, *_ in W]); busy = 0; depth = 0; last = None
for t, d in ev:
    if depth > 0: busy += t - last
    depth += d; last = t
span = W[-1][1] - W[0][0]
print("window %.1f ms, union busy %.1f ms (%.0f%%), sum of kernels %.1f ms" % (span / 1e6, busy / 1e6, 100.0 * busy / span, sum(tot.values()) / 1e6))
for n, t in tot.most_common(28): print("  %-46s %6d x %8.1f us = %8.2f ms (%4.1f%% of window)" % (n, cnt[n], t / cnt[n] / 1e3, t / 1e6, 100.0 * t / span))
idx = [i for i, r in enumerate(W) if mark in r[2]]
if len(idx) > 12:
    a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 2]
    base = W[a][0]
    print("two rounds from the middle:")
    for s, e, n, q, gx, gy in W[a:b + 1]: print("  +%8.1f us  %7.1f us  q%d  grid %5d x %3d  %s" % ((s - base) / 1e3, (e - s) / 1e3, q, gx, gy, short(n)))
