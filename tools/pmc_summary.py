#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc csv outputs per kernel (mean per dispatch)."""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-60:]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    if "blind_rotate" not in k and "keyswitch_kernel" not in k:
        continue
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"   {c:32s} mean/dispatch {sum(v)/len(v):16.1f}  (n={len(v)})")
