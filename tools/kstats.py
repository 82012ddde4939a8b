#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv compactly: tools/kstats.py <csv> [max rows]"""
import csv, sys
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for i, r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i >= n:
        break
    print(f"{r['Name'][:70]:70s} calls={int(r['Calls']):6d} avg={float(r['AverageNs'])/1e3:10.1f} us  total={float(r['TotalDurationNs'])/1e6:9.2f} ms")
