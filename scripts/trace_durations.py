#!/usr/bin/env python3
"""Per-dispatch durations of kernels matching a substring from a rocprofv3 kernel_trace CSV, grouped by grid size.
usage: trace_durations.py <dir> <substring> [algorithmic MB per group, comma separated, in first-seen order]"""
import csv
import glob
import os
import sys

d, sub = sys.argv[1], sys.argv[2]
mbs = [float(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else []
groups, order = {}, []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if sub not in row["Kernel_Name"]:
            continue
        key = (row["Kernel_Name"][:60], row.get("Grid_Size_X", row.get("Grid_Size", "")), row.get("Grid_Size_Y", ""),
               row.get("LDS_Block_Size", ""), row.get("VGPR_Count", ""))
        if key not in groups:
            groups[key] = []
            order.append(key)
        groups[key].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
tot = 0.0
for i, k in enumerate(order):
    v = sorted(groups[k])
    med = v[len(v) // 2]
    tot += med
    extra = f" {mbs[i] / med * 1e3:7.0f} GB/s" if i < len(mbs) else ""
    print(f"{k[0][:44]:44s} grid {k[1]:>8s}x{k[2]:>3s} lds {k[3]:>6s} vgpr {k[4]:>4s} n={len(v):3d} median {med:8.1f} us min {v[0]:8.1f}{extra}")
print(f"sum of medians {tot:.1f} us" + (f"  => {sum(mbs[:len(order)]) / tot * 1e3:.0f} GB/s" if mbs else ""))
