#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection CSV: per kernel and counter, dispatch count, mean and sum.
Usage: pmc_summary.py <counter_collection.csv> [kernel-substring] > summary.csv"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
needle = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: [0, 0.0])
with open(path, newline="") as f:
    for row in csv.DictReader(f):
        name = row.get("Kernel_Name", "")
        if needle and needle not in name:
            continue
        key = (name[:80], row.get("Counter_Name", ""))
        a = acc[key]
        a[0] += 1
        a[1] += float(row.get("Counter_Value", 0) or 0)
print("kernel,counter,dispatches,mean_per_dispatch,sum")
for (k, c), (n, s) in sorted(acc.items()):
    print(f"\"{k}\",{c},{n},{s / max(1, n):.3f},{s:.0f}")
