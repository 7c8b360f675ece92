#!/usr/bin/env python3
"""Prints the kernel timeline of ONE step out of a rocprofv3 --kernel-trace CSV: name, start offset, duration and the gap
to the previous kernel's end (all queues merged), for a step in the middle of the run.  A step starts at every dispatch
of --first (a kernel-name prefix, default the fBm kernel).
  tools/trace_timeline.py <kernel_trace.csv> [--first fractal_simplex] [--step -3]"""
import argparse
import csv

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--first", default="fractal_simplex")
ap.add_argument("--step", type=int, default=-3)
ap.add_argument("--width", type=int, default=70)
a = ap.parse_args()
rows = []
with open(a.csv) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
starts = [i for i, r in enumerate(rows) if a.first in r[2]]
# a step may launch the first kernel several times (one per local stripe): group dispatches closer than 20 us
groups = []
for i in starts:
    if groups and rows[i][0] - rows[groups[-1][-1]][0] < 20000 and i - groups[-1][-1] < 3:
        groups[-1].append(i)
    else:
        groups.append([i])
g = groups[a.step]
nxt = groups[a.step + 1][0] if a.step + 1 < 0 or a.step + 1 < len(groups) and a.step >= 0 else len(rows)
t0 = rows[g[0]][0]
prev_end = None
print("%9s %9s %8s  %s" % ("start_us", "dur_us", "gap_us", "kernel (queue)"))
for r in rows[g[0]:nxt]:
    gap = "" if prev_end is None else "%.1f" % ((r[0] - prev_end) / 1e3)
    print("%9.1f %9.1f %8s  %s (q%s)" % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, gap, r[2][:a.width], r[3]))
    prev_end = r[1] if prev_end is None else max(prev_end, r[1])
print("step: %.1f us from first start to last end" % ((max(r[1] for r in rows[g[0]:nxt]) - t0) / 1e3))
# every step: period (start to next step's start), busy span, idle gap before the next step
print("\n%5s %10s %10s %10s" % ("step", "period_us", "span_us", "gap_us"))
for k in range(len(groups) - 1):
    a0, a1 = groups[k][0], groups[k + 1][0]
    last_end = max(r[1] for r in rows[a0:a1])
    print("%5d %10.1f %10.1f %10.1f" % (k, (rows[a1][0] - rows[a0][0]) / 1e3, (last_end - rows[a0][0]) / 1e3,
                                      (rows[a1][0] - last_end) / 1e3))
