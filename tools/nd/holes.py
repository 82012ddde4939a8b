#!/usr/bin/env python3
"""Where the matrix cores wait in the LAST nested-dissection factorisation of a rocprofv3 kernel-trace CSV: the intervals in which no
Schur pass (nd_syrk_kernel<..., true, ...>) is running, longest first, with what ran in them.   tools/nd/holes.py <kernel_trace.csv> [min_us]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
for r in rows: r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
def short(nm):
    if "nd_syrk_kernel" in nm: return "schur" if "true" in nm.split("nd_syrk_kernel")[1].split(">")[0] else "upd"
    for k in ("nd_potrf", "nd_trsm", "nd_extend_add", "nd_trinv", "nd_init", "nd_zero", "fillBuffer", "nd_dot", "nd_fwd", "nd_bwd", "nd_mv"):
        if k in nm: return k[3:] if k.startswith("nd_") else k
    return nm.split("(")[0][-24:]
it = max(i for i, r in enumerate(rows) if "nd_trinv" in r["Kernel_Name"])
ia = max(i for i, r in enumerate(rows[:it]) if "stencil_gather" in r["Kernel_Name"] or "nd_assemble" in r["Kernel_Name"])
sel = rows[ia + 1:it + 1]
t0, t1 = sel[0]["s"], sel[-1]["e"]
sch = sorted((r["s"], r["e"]) for r in sel if short(r["Kernel_Name"]) == "schur")
holes, ce = [], t0
for s, e in sch:
    if s > ce: holes.append((ce, s))
    ce = max(ce, e)
if ce < t1: holes.append((ce, t1))
tot = sum(b - a for a, b in holes)
print(f"factorisation {(t1 - t0) / 1e6:.2f} ms; a Schur pass runs for {(t1 - t0 - tot) / 1e6:.2f} ms; {len(holes)} holes, {tot / 1e6:.2f} ms")
for a, b in sorted(holes, key=lambda h: h[0]):
    if (b - a) / 1e3 < min_us: continue
    inside = collections.Counter(); n = collections.Counter()
    for r in sel:
        if r["e"] > a and r["s"] < b:
            k = short(r["Kernel_Name"]); inside[k] += min(r["e"], b) - max(r["s"], a); n[k] += 1
    print(f"  at {(a - t0) / 1e6:8.2f} ms: {(b - a) / 1e3:8.1f} us | " + " ".join(f"{k}:{n[k]}x{v / 1e3:.0f}" for k, v in inside.most_common(6)))
