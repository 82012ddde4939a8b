#!/usr/bin/env python3
"""Short view of rocprofv3 kernel_stats CSVs: tools/kshort.py file.csv [...]"""
import csv, sys
for f in sys.argv[1:]:
    print(f)
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("splpak::", "").replace("(anonymous namespace)::", "").replace("void ", "")
        print("  %-58s %6s x %10.1f us  %6.2f%%" % (n[:58], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
