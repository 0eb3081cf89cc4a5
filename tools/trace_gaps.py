#!/usr/bin/env python3
"""Timeline of the LAST `window_ms` of a rocprofv3 kernel trace (…_kernel_trace.csv): busy time per kernel name, idle time
between kernels, the largest gaps.  Usage: trace_gaps.py <kernel_trace.csv> <window_ms> | <from_ms> <to_ms> (relative to the
first kernel of the trace)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) * 1e6
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
t_end = max(e for _, e, _ in ev)
if len(sys.argv) > 3:
    t0 = ev[0][0]
    ev = [x for x in ev if t0 + float(sys.argv[2]) * 1e6 <= x[0] < t0 + float(sys.argv[3]) * 1e6]
else:
    ev = [x for x in ev if x[0] >= t_end - win]
busy = collections.Counter(); calls = collections.Counter()
gaps = []
cur = ev[0][0]
for s, e, nm in ev:
    busy[nm[:60]] += e - s; calls[nm[:60]] += 1
    if s > cur:
        gaps.append((s - cur, nm[:50]))
    cur = max(cur, e)
span = cur - ev[0][0]
print(f"window {span/1e6:.2f} ms, kernels {len(ev)}, busy {sum(busy.values())/1e6:.2f} ms, idle {sum(g for g, _ in gaps)/1e6:.2f} ms in {len(gaps)} gaps")
for nm, t in busy.most_common(25):
    print(f"  {t/1e6:9.3f} ms  {calls[nm]:5d} x  {nm}")
gaps.sort(reverse=True)
print("largest gaps (us, before kernel):", [(round(g / 1e3, 1), nm[:24]) for g, nm in gaps[:12]])
hist = collections.Counter(min(int(g / 1e3) // 5 * 5, 100) for g, _ in gaps)
print("gap histogram (us bucket: count):", sorted(hist.items()))
