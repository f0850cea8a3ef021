#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel launch.
usage: pmc_summary.py <dir with pass subdirs> [out.md]"""
import collections
import csv
import glob
import re
import sys

root = sys.argv[1]
tab = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"\bk_[a-z0-9_]+", r["Kernel_Name"])
        if not m:
            continue
        name = m.group(0)
        tab[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
counters = sorted({c for k in tab.values() for c in k})
lines = ["| kernel | launches | " + " | ".join(counters) + " |", "|---|---:|" + "---:|" * len(counters)]
for k, d in sorted(tab.items()):
    n = max(len(v) for v in d.values())
    lines.append(f"| {k} | {n} | " + " | ".join(f"{sum(d[c])/len(d[c]):.4g}" if c in d else "-" for c in counters) + " |")
text = "\n".join(lines)
print(text)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write("# rocprofv3 --pmc, mean per launch\n\n" + text + "\n")
