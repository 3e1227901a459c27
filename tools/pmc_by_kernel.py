#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counters per kernel name: pmc_by_kernel.py <dir> [name-substring]"""
import csv, glob, os, sys, collections, re
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
f = [p for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.Counter()
for p in f:
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if sub not in k: continue
        k = re.sub(r"\(.*", "", k)
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[(k, r["Counter_Name"])] += 1
for k, c in sorted(acc.items()):
    n = max(disp[(k, x)] for x in c)
    print(f"{k}  launches {n}: " + "  ".join(f"{x} {v / 1e6:.2f}M" for x, v in sorted(c.items())))
