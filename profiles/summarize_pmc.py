#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter CSVs per kernel (za_k_* only) and print per-launch averages."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for path in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row.get("Kernel_Name", "")
            if k.startswith("void "):            # template instantiations are printed with their return type
                k = k[5:]
            if not k.startswith("za_k_"):
                continue
            k = k.split("(")[0]
            c = row["Counter_Name"]
            acc[k][c] += float(row["Counter_Value"])
            calls[k][c] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        n = calls[k][c]
        print(f"   {c:24s} total {acc[k][c]:16.0f}  launches {n:4d}  per-launch {acc[k][c] / n:16.1f}")
