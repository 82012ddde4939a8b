#!/usr/bin/env python3
"""Timeline of the panel chain in a rocprofv3 kernel trace: per kernel family, count / mean duration, and for
potrf the mean distance between consecutive launches (the chain step) of the LAST fit in the trace.
    tools/chain_trace.py <kernel_trace.csv>"""
import csv, sys, statistics as st
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
def fam(nm):
    for k in ("potrf_strip", "potrf_block", "trsm", "syrk32", "trinv", "sweepmat", "combine", "expand2", "fwd_step", "bwd_step",
              "fwd_update", "bwd_update", "blockmv"):
        if k in nm: return k
    if "syrk64_kernel<16, 1, 4" in nm: return "bulk"
    if "syrk64_kernel" in nm: return "colpiece"
    return None
pot = [r for r in rows if fam(r["Kernel_Name"]) in ("potrf_strip", "potrf_block")]
# last fit: potrf launches after the last gap > 3 ms
start = 0
for i in range(1, len(pot)):
    if pot[i]["s"] - pot[i - 1]["e"] > 3e6: start = i
pot = pot[start:]
t0, t1 = pot[0]["s"], pot[-1]["e"]
print(f"last fit: {len(pot)} potrf launches over {(t1 - t0) / 1e6:.2f} ms")
sel = [r for r in rows if t0 - 1e5 <= r["s"] <= t1 + 2e6 and fam(r["Kernel_Name"])]
by = {}
for r in sel: by.setdefault(fam(r["Kernel_Name"]), []).append(r)
for k, v in sorted(by.items(), key=lambda kv: -sum(r["e"] - r["s"] for r in kv[1])):
    d = [r["e"] - r["s"] for r in v]
    print(f"  {k:12s} n={len(v):5d} mean {st.mean(d)/1e3:8.1f} us  median {st.median(d)/1e3:8.1f}  max {max(d)/1e3:8.1f}  total {sum(d)/1e6:8.2f} ms")
# chain step: potrf launches are from one or two chains; split by queue / stream id when present
key = "Queue_Id" if "Queue_Id" in pot[0] else None
chains = {}
for r in pot: chains.setdefault(r.get(key, "0") if key else "0", []).append(r)
for q, v in chains.items():
    if len(v) < 3: continue
    steps = [b["s"] - a["s"] for a, b in zip(v, v[1:])]
    print(f"  chain on queue {q}: {len(v)} potrf, step mean {st.mean(steps)/1e3:.1f} us median {st.median(steps)/1e3:.1f} us; potrf mean {st.mean([r['e']-r['s'] for r in v])/1e3:.1f} us")
# which queues carry which kernel families, and when each queue's first / last kernel of the fit ran
if key:
    qs = {}
    for r in sel: qs.setdefault(r[key], []).append(r)
    for q, v in sorted(qs.items()):
        fams = {}
        for r in v: fams[fam(r["Kernel_Name"])] = fams.get(fam(r["Kernel_Name"]), 0) + 1
        print(f"  queue {q}: first +{(v[0]['s'] - t0)/1e6:6.2f} ms last +{(v[-1]['e'] - t0)/1e6:6.2f} ms  {fams}")
