#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter CSVs per kernel name: usage pmc_summary.py DIR [substring]."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
tot = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub and sub not in k:
            continue
        k = k.split("(")[0][-48:]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
for k, v in tot.items():
    print(k, "dispatches", len(calls[k]))
    for c, x in sorted(v.items()):
        print(f"   {c:32s} {x:.6g}")
