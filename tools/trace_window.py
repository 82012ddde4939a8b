#!/usr/bin/env python3
"""Print a window of a rocprofv3 kernel-trace CSV (start offset, duration, queue, name)
and per-kernel busy/overlap summaries for the factorisation phase."""
import csv, sys, collections
path = sys.argv[1]
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
def short(nm):
    for k in ("syrk", "trsm", "potrf", "inv64", "trtri", "blockmv", "fwd_update", "bwd_update", "gram", "residual"):
        if k in nm: return k
    return nm[:30]
for r in rows[lo:lo + n]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:12.1f} us  dur {(e-s)/1e3:9.1f} us  q={r.get('Queue_Id','?'):>3} grid={r.get('Grid_Size_X', r.get('Grid_Size','?')):>8} {short(r['Kernel_Name'])}")
# union-busy time of syrk and of panel kernels between first and last potrf
f = [r for r in rows if "potrf" in r["Kernel_Name"]]
if f:
    a, b = int(f[0]["Start_Timestamp"]), int(f[-1]["End_Timestamp"])
    print(f"factor span {(b-a)/1e6:.1f} ms")
    tot = collections.Counter()
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s >= a and e <= b + 1:
            tot[short(r["Kernel_Name"])] += e - s
    for k, v in tot.most_common(8):
        print(f"  {k:12s} {v/1e6:9.1f} ms")
