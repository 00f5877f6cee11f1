"""dev: how much do the kernels of a rocprofv3 --kernel-trace CSV overlap?  Per queue: kernels and busy time; overall: sum of
durations, union of busy intervals, histogram of the concurrency level weighted by time.  usage: trace_overlap.py <kernel_trace.csv> [t0_frac t1_frac]"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", ""), r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows]
t_lo, t_hi = min(e[0] for e in ev), max(e[1] for e in ev)
f0, f1 = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.0, 1.0)
a, b = t_lo + f0 * (t_hi - t_lo), t_lo + f1 * (t_hi - t_lo)
ev = [e for e in ev if e[0] >= a and e[1] <= b]
print(f"{len(ev)} kernels in {(b - a) / 1e6:.1f} ms")
q = collections.defaultdict(lambda: [0, 0])
for s, e, n, qi, si in ev: q[(qi, si)][0] += 1; q[(qi, si)][1] += e - s
for k, (c, t) in sorted(q.items()): print(f"  queue {k[0]} stream {k[1]}: {c} kernels, {t / 1e6:.2f} ms")
pts = sorted([(s, 1) for s, e, *_ in ev] + [(e, -1) for s, e, *_ in ev])
lvl = 0; last = pts[0][0]; hist = collections.Counter()
for t, d in pts:
    hist[lvl] += t - last; last = t; lvl += d
tot = sum(e - s for s, e, *_ in ev); union = sum(v for k, v in hist.items() if k > 0)
print(f"sum of kernel durations {tot / 1e6:.2f} ms, union {union / 1e6:.2f} ms, idle {hist[0] / 1e6:.2f} ms, mean concurrency when busy {tot / max(union, 1):.2f}")
for k in sorted(hist): print(f"  {k} kernels at once: {hist[k] / 1e6:.2f} ms")
by = collections.Counter(); cnt = collections.Counter()
for s, e, n, *_ in ev: by[n] += e - s; cnt[n] += 1
for n, t in by.most_common(8): print(f"  {t / 1e6:7.2f} ms {cnt[n]:6d} x {t / cnt[n] / 1e3:7.1f} us {n[:60]}")
# the last batch of runs: from the first k_init_state of the last group (groups = k_init_state launches closer than 5 ms) to the end
inits = sorted(s for s, e, n, *_ in ev if n.startswith("k_init_state"))
if inits:
    g0 = inits[-1]
    for t in reversed(inits):
        if g0 - t < 5e6: g0 = t
    ev2 = [x for x in ev if x[0] >= g0]
    pts = sorted([(s, 1) for s, e, *_ in ev2] + [(e, -1) for s, e, *_ in ev2])
    lvl = 0; last = pts[0][0]; hist = collections.Counter()
    for t, d in pts: hist[lvl] += t - last; last = t; lvl += d
    span = pts[-1][0] - pts[0][0]; tot = sum(e - s for s, e, *_ in ev2)
    print(f"LAST BATCH: span {span / 1e6:.2f} ms, {len(ev2)} kernels, sum {tot / 1e6:.2f} ms, idle {hist[0] / 1e6:.2f} ms; time at concurrency 1..: " + ", ".join(f"{k}:{hist[k] / 1e6:.1f}" for k in sorted(hist) if k > 0))
    by = collections.Counter(); cnt = collections.Counter()
    for s, e, n, *_ in ev2: by[n] += e - s; cnt[n] += 1
    for n, t in by.most_common(6): print(f"  {t / 1e6:7.2f} ms {cnt[n]:6d} x {t / cnt[n] / 1e3:7.1f} us {n[:60]}")
