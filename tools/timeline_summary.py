#!/usr/bin/env python3
"""Reads a rocprofv3 kernel-trace CSV and prints the LAST render's launches of the streaming renderer in start order: kernel, queue, start, duration,
the gap to the previous launch's end on the same queue, and how much of the launch ran beside another trace launch.
usage: python tools/timeline_summary.py kernel_trace.csv"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    kind = "T" if "pt_trace" in name else "F" if "fold" in name else None
    if kind is None:
        continue
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, r.get("Queue_Id", "?")))
rows.sort()
# the last second of the trace: steady state
t_end = rows[-1][1]
rows = [x for x in rows if x[0] > t_end - 70e6]     # the last ~70 ms: two renders
t0 = rows[0][0]
last_end = {}
busy_T = sum(e - s for s, e, k, q in rows if k == "T")
busy_F = sum(e - s for s, e, k, q in rows if k == "F")
# union of the trace launches' intervals
iv = sorted((s, e) for s, e, k, q in rows if k == "T")
union, cs, ce = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > ce:
        union += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
union += ce - cs
print("window %.3f ms: %d trace launches, sum %.3f ms, union %.3f ms (not covered by any trace launch: %.3f ms); %d folds, sum %.3f ms"
      % ((rows[-1][1] - t0) / 1e6, len(iv), busy_T / 1e6, union / 1e6, ((rows[-1][1] - t0) - union) / 1e6, sum(1 for x in rows if x[2] == "F"), busy_F / 1e6))
print("kind queue   start_us   dur_us  gap_same_queue_us  gap_any_trace_us")
prev_T_end = None
for s, e, k, q in rows[:120]:
    g = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
    ga = (s - prev_T_end) / 1e3 if (k == "T" and prev_T_end is not None) else float("nan")
    print("%s    %5s %10.1f %8.1f %10.1f %10.1f" % (k, q, (s - t0) / 1e3, (e - s) / 1e3, g, ga))
    last_end[q] = e
    if k == "T":
        prev_T_end = e if prev_T_end is None else max(prev_T_end, e)
