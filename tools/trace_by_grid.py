#!/usr/bin/env python3
"""Average kernel duration per (kernel, grid size) from a rocprofv3 --kernel-trace CSV: separates the
launches of one kernel over different pyramid levels.  usage: trace_by_grid.py <kernel_trace.csv>"""
import collections
import csv
import re
import sys

acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"\bk_[a-z0-9_]+", r["Kernel_Name"])
    if not m:
        continue
    key = (m.group(0), int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("| kernel | grid | launches | avg_us | min_us |\n|---|---|---:|---:|---:|")
for k, v in sorted(acc.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
    print(f"| {k[0]} | {k[1]}x{k[2]}x{k[3]} | {len(v)} | {sum(v)/len(v):.2f} | {min(v):.2f} |")
