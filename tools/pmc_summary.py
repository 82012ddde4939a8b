#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection CSV: mean counter value per launch
for kernels whose name contains a substring (optionally only a given grid size)."""
import csv, sys, collections
path, sub = sys.argv[1], sys.argv[2]
grid = sys.argv[3] if len(sys.argv) > 3 else None
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(path)):
    if sub not in r["Kernel_Name"]:
        continue
    if grid and r.get("Grid_Size", r.get("Grid_Size_X")) != grid:
        continue
    a = acc[r["Counter_Name"]]
    a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (s, n) in sorted(acc.items()):
    print(f"{k:32s} launches={n:6d} mean={s/n:.6g}")
