#!/usr/bin/env python3
"""HBM-side bytes per launch of every extraction kernel from two rocprofv3 --pmc passes over
`bench.py --contexts 1` (one launch of a kernel covers the whole 256-frame batch):
  pass 1: TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum   (read requests by size)
  pass 2: TCC_EA0_WRREQ_WRITE_DRAM_32B_sum                                    (written 32-byte units)
  optional pass: SQ_INSTS_VALU GRBM_GUI_ACTIVE                                 (vector-ALU issue share per kernel)
bytes read = 32*n32 + 64*n64 + 128*n128, bytes written = 32*units.  (FETCH_SIZE tallies gfx950's 128-byte
requests at 64 bytes; the per-size counters do not.)  Writes profiles/traffic.json, keyed by bench.py's stages.
usage: pmc_traffic.py <dir with the passes' counter_collection.csv> <out.json> [resize launches per step] [build stamp]"""
import collections
import csv
import glob
import json
import re
import sys

root, out = sys.argv[1], sys.argv[2]
resizes = int(sys.argv[3]) if len(sys.argv) > 3 else 7
stamp = sys.argv[4] if len(sys.argv) > 4 else None      # orbx_build_stamp() of the library the passes ran on
tab = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"\bk_[a-z0-9_]+", r["Kernel_Name"])
        if m:
            tab[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
mean = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in tab.items()}
kern = {}
for k, d in mean.items():
    rd = 32 * d.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * d.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * d.get("TCC_EA0_RDREQ_128B_sum", 0)
    wr = 32 * d.get("TCC_EA0_WRREQ_WRITE_DRAM_32B_sum", 0)
    kern[k] = {"read_bytes": int(rd), "write_bytes": int(wr)}
    # share of the kernel's cycles in which its SIMDs issue vector ALU instructions, pricing every wave64 instruction at
    # 4 cycles: an upper estimate (measured on this chip, profiles/r02b_valu_rates.md: simple 32-bit instructions issue
    # every 2 cycles once two waves share a SIMD, the byte / packed / three-operand class every 3.3 to 4.5).  The chip has
    # 1024 SIMDs, GRBM_GUI_ACTIVE is summed over the 8 XCDs
    if "SQ_INSTS_VALU" in d and d.get("GRBM_GUI_ACTIVE"):
        kern[k]["valu_issue_frac"] = round(d["SQ_INSTS_VALU"] * 4 / 1024 / (d["GRBM_GUI_ACTIVE"] / 8), 3)
tot = lambda k, n=1: n * (kern.get(k, {}).get("read_bytes", 0) + kern.get(k, {}).get("write_bytes", 0))
# level 0: the batches of the timed region read their frames from the input view, where no kernel of its own runs (the frame of
# level 0 is written inside the first resize launch; the copy kernel appears in the passes once, when the view is filled)
level0 = tot("k_level0_frame")
stages = {"pyramid": level0 + tot("k_resize", resizes) + tot("k_borders"), "fast": tot("k_fast_cells"),
          "octree": tot("k_octree"), "blur": tot("k_blur"), "describe": tot("k_describe")}
json.dump({"build_stamp": stamp, "source": "rocprofv3 --pmc TCC_EA0_RDREQ_{32B,64B,128B}_sum / TCC_EA0_WRREQ_WRITE_DRAM_32B_sum over "
                     "bench.py --contexts 1 (640x480, 256 frames per launch); k_resize is the mean of its launches "
                     f"times {resizes}", "kernels_mean_per_launch": kern, "stages": stages}, open(out, "w"), indent=1)
print(json.dumps(stages))
