#!/usr/bin/env python3
"""Timeline of the LAST nested-dissection factorisation in a rocprofv3 kernel-trace CSV: per launch start offset,
duration, queue, grid and kernel; then per tree depth (delimited by the extend-add launches) the span and the busy
time of the Schur passes, the panel updates and the chain kernels.   tools/nd_timeline.py <kernel_trace.csv> [rows]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(nm):
    if "nd_syrk_kernel" in nm: return "schur" if "true>" in nm else "upd"
    for k in ("nd_potrf", "nd_trsm", "nd_extend_add", "nd_trinv", "nd_assemble", "fillBuffer", "nd_dot", "nd_fwd", "nd_bwd", "nd_mv"):
        if k in nm: return k[3:] if k.startswith("nd_") else k
    return nm.split("(")[0][-28:]
# last factorisation: from the last nd_assemble to the following nd_trinv
ia = max(i for i, r in enumerate(rows) if "nd_assemble" in r["Kernel_Name"])
it = min(i for i, r in enumerate(rows) if i > ia and "nd_trinv" in r["Kernel_Name"])
sel = [r for r in rows[ia:] if int(r["Start_Timestamp"]) <= int(rows[it]["End_Timestamp"])]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel[:nshow]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:10.1f} us dur {(e-s)/1e3:9.1f} q={r['Queue_Id']:>2} grid={r['Grid_Size_X']:>9} {short(r['Kernel_Name'])}")
print(f"factorisation span {(int(rows[it]['End_Timestamp']) - t0)/1e6:.1f} ms")
# depth segments: the diagonal-block launches carry one workgroup per front, so their grid size (256 threads each) tells
# the depth: a new segment starts when it drops
depth_rows = collections.OrderedDict()
d = -1
cur = None
for r in sel:
    k = short(r["Kernel_Name"])
    if k == "potrf":
        g = int(r["Grid_Size_X"]) // 256
        if cur is None or g < cur:
            d += 1
            cur = g
    depth_rows.setdefault(max(d, 0), []).append(r)
for d, rs in depth_rows.items():
    a = min(int(r["Start_Timestamp"]) for r in rs); b = max(int(r["End_Timestamp"]) for r in rs)
    tot = collections.Counter(); cnt = collections.Counter()
    for r in rs:
        tot[short(r["Kernel_Name"])] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[short(r["Kernel_Name"])] += 1
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rs)
    busy, ce = 0, a
    for s0, e0 in iv:
        if e0 > ce: busy += e0 - max(s0, ce); ce = e0
    nf = [int(r["Grid_Size_X"]) // 256 for r in rs if short(r["Kernel_Name"]) == "potrf"]
    print(f"segment {d:2d} ({max(nf) if nf else 0:4d} fronts): span {(b-a)/1e6:7.2f} ms busy {busy/1e6:7.2f} | " + " ".join(f"{k}:{cnt[k]}x{v/1e6:.2f}" for k, v in tot.most_common(7)))
