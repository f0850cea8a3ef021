#!/usr/bin/env python3
"""Condense a rocprofv3 `--kernel-trace --stats` kernel_stats.csv into a short table for profiles/.
usage: summarize_prof.py <kernel_stats.csv> <out.md> [title]"""
import csv
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
title = sys.argv[3] if len(sys.argv) > 3 else src
rows = list(csv.DictReader(open(src)))
with open(dst, "w") as f:
    f.write(f"# {title}\n\nSource: rocprofv3 --kernel-trace --stats (kernel_stats.csv); durations in microseconds.\n\n")
    f.write("| kernel | calls | total_us | avg_us | min_us | max_us | % |\n|---|---:|---:|---:|---:|---:|---:|\n")
    for r in rows:
        m = re.search(r"\bk_[a-z0-9_]+(<[^>]*>)?", r["Name"])
        name = m.group(0) if m else re.sub(r"\(.*", "", r["Name"])[:70]
        f.write(f"| {name} | {r['Calls']} | {int(r['TotalDurationNs'])/1e3:.1f} | {float(r['AverageNs'])/1e3:.2f} | "
                f"{int(r['MinNs'])/1e3:.2f} | {int(r['MaxNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |\n")
print("wrote", dst)
