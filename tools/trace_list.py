#!/usr/bin/env python3
"""Ordered listing of the `window_ms` up to 2 ms past the last VM kernel of a rocprofv3 kernel trace, one line per launch: start (ms from the trace's first kernel), duration,
idle time of its queue before it, queue, grid, kernel name.  `min_us` hides launches shorter than that unless a gap of min_us precedes them.
Usage: trace_list.py <kernel_trace.csv> <window_ms> [min_us]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) * 1e6
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r["Queue_Id"], r["Grid_Size_X"]) for r in rows)
t0 = ev[0][0]
t_end = max(e for _, e, nm, *_ in ev if nm.startswith("vm_asm")) + 2_000_000      # (teardown kernels follow long after the last call)
ev = [x for x in ev if t_end - win <= x[0] <= t_end]
last = {}
for s, e, nm, q, g in ev:
    gap = (s - last.get(q, s)) / 1e3
    if (e - s) / 1e3 >= min_us or gap >= max(min_us, 30.0):
        print(f"{(s - t0) / 1e6:9.3f}  {(e - s) / 1e3:9.1f} us  gap {gap:8.1f}  q{q}  g{g:>6}  {nm}")
    last[q] = e
print(f"window: {(t_end - ev[0][0]) / 1e6:.2f} ms, {len(ev)} launches")
