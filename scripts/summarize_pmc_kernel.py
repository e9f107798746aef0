#!/usr/bin/env python3
"""Per-dispatch PMC table for one kernel name substring from rocprofv3 counter_collection CSVs (several passes).
usage: summarize_pmc_kernel.py <dir> <kernel substring>"""
import csv
import glob
import os
import sys
from collections import defaultdict

d, sub = sys.argv[1], sys.argv[2]
tab = defaultdict(dict)  # (dispatch order within file) -> counter -> value
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    order = {}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if sub not in row["Kernel_Name"]:
                continue
            did = row["Dispatch_Id"]
            if did not in order:
                order[did] = len(order)
            key = order[did]
            tab[key][row["Counter_Name"]] = tab[key].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            tab[key]["grid"] = row.get("Grid_Size", "")
            tab[key]["lds"] = row.get("LDS_Block_Size", "")
            tab[key]["vgpr"] = row.get("VGPR_Count", "")
names = sorted({k for v in tab.values() for k in v} - {"grid", "lds", "vgpr"})
print("disp grid lds vgpr " + " ".join(names))
for k in sorted(tab):
    v = tab[k]
    print(k, v.get("grid"), v.get("lds"), v.get("vgpr"), " ".join(f"{v.get(n, float('nan')):.4g}" for n in names))
