#!/usr/bin/env python3
"""Concurrency picture of a rocprofv3 --kernel-trace CSV: over the last `frac` of the trace, the share of time with
0, 1, 2, ... kernels in flight, each kernel's summed duration, and its duration per launch.
usage: trace_overlap.py <kernel_trace.csv> [frac=0.5]"""
import collections
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"\bk_[a-z0-9_]+", r["Kernel_Name"])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(0) if m else r["Kernel_Name"][:24], r.get("Queue_Id", "?")))
rows.sort()
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
t0 = rows[0][0] + (rows[-1][1] - rows[0][0]) * (1 - frac)
rows = [r for r in rows if r[0] >= t0]
ev = []
for s, e, _, _ in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth = 0; last = ev[0][0]; hist = collections.Counter()
for t, d in ev:
    hist[depth] += t - last; last = t; depth += d
span = ev[-1][0] - ev[0][0]
print(f"window {span / 1e3:.1f} us, {len(rows)} launches, queues {sorted(set(r[3] for r in rows))}")
print("kernels in flight: " + ", ".join(f"{k}: {100 * v / span:.1f} %" for k, v in sorted(hist.items())))
per = collections.defaultdict(list)
for s, e, n, _ in rows:
    per[n].append((e - s) / 1e3)
print("| kernel | launches | total_us | share of window | avg_us |\n|---|---:|---:|---:|---:|")
for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    print(f"| {n} | {len(v)} | {sum(v):.0f} | {100 * sum(v) / (span / 1e3):.0f} % | {sum(v) / len(v):.1f} |")
