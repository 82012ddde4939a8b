#!/usr/bin/env python3
"""Idle gaps (no kernel running on any queue) inside the LAST nested-dissection factorisation of a rocprofv3 kernel-trace
CSV: total, histogram, and the largest gaps with the launches around them.   tools/gap_report.py <kernel_trace.csv> [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(nm):
    nm = nm.replace("splpak::", "").replace("(anonymous namespace)::", "").replace("void ", "")
    return nm.split("(")[0][:44]
ia = max(i for i, r in enumerate(rows) if "nd_assemble" in r["Kernel_Name"])
it = min(i for i, r in enumerate(rows) if i > ia and "nd_trinv" in r["Kernel_Name"])
sel = [r for r in rows[ia:] if int(r["Start_Timestamp"]) <= int(rows[it]["End_Timestamp"])]
t0 = int(sel[0]["Start_Timestamp"])
end = int(sel[0]["End_Timestamp"]); last = sel[0]
gaps = []
for r in sel[1:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > end: gaps.append((s - end, end - t0, last, r))
    if e > end: end, last = e, r
tot = sum(g[0] for g in gaps)
print(f"factorisation span {(end - t0) / 1e6:.2f} ms, {len(sel)} launches, idle {tot / 1e6:.2f} ms in {len(gaps)} gaps")
for lo, hi in ((0, 5), (5, 20), (20, 50), (50, 200), (200, 1e9)):
    g = [x[0] for x in gaps if lo * 1e3 <= x[0] < hi * 1e3]
    print(f"  gaps {lo:>4} .. {hi if hi < 1e8 else 'inf':>4} us: {len(g):5d}, {sum(g) / 1e6:7.2f} ms")
for g in sorted(gaps, key=lambda x: -x[0])[:nshow]:
    print(f"  at {g[1] / 1e6:8.2f} ms: {g[0] / 1e3:8.1f} us idle after {short(g[2]['Kernel_Name'])} (q{g[2]['Queue_Id']}) before {short(g[3]['Kernel_Name'])} (q{g[3]['Queue_Id']}, grid {g[3]['Grid_Size_X']})")
