#!/usr/bin/env python3
"""Gaps between consecutive bulk trailing-update launches in a rocprofv3 kernel trace,
and which kernel finished last before each delayed launch (the dependency that held it)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
def short(nm):
    if "syrk64_kernel<16, 1, 4" in nm or "syrk64_kernel<16, 1, 2052" in nm: return "bulk"
    if "syrk64_kernel<16, 1, 12" in nm: return "colpiece"
    for k in ("trsm", "potrf", "trtri", "sweepmat", "fwd_step", "bwd_step", "gram", "residual", "expand"):
        if k in nm: return k
    return nm[:24]
bulk = [r for r in rows if short(r["Kernel_Name"]) == "bulk"]
# split into fits: a new fit starts when the gap is > 20 ms
fits, cur = [], [bulk[0]]
for a, b in zip(bulk, bulk[1:]):
    if b["s"] - a["e"] > 20e6: fits.append(cur); cur = []
    cur.append(b)
fits.append(cur)
print(f"{len(bulk)} bulk launches in {len(fits)} fits")
f = fits[-1]
others = [r for r in rows if short(r["Kernel_Name"]) != "bulk" and f[0]["s"] - 2e6 <= r["s"] <= f[-1]["e"]]
gaps = [b["s"] - a["e"] for a, b in zip(f, f[1:])]
durs = [r["e"] - r["s"] for r in f]
import statistics as st
n = len(f)
print(f"last fit: {n} launches, span {(f[-1]['e']-f[0]['s'])/1e6:.1f} ms, sum of durations {sum(durs)/1e6:.1f} ms, sum of gaps {sum(gaps)/1e6:.1f} ms")
for lo, hi in ((0, 50), (50, n - 60), (n - 60, n - 1)):
    g = gaps[lo:hi]; d = durs[lo:hi]
    if g: print(f"  launches {lo:4d}..{hi:4d}: mean dur {st.mean(d)/1e3:7.1f} us, mean gap {st.mean(g)/1e3:6.1f} us, median {st.median(g)/1e3:6.1f}, p90 {sorted(g)[int(0.9*len(g))]/1e3:6.1f}, max {max(g)/1e3:6.1f}")
# the kernel that ended last before each delayed launch
why = collections.Counter(); lat = collections.defaultdict(list)
oi = 0
others.sort(key=lambda r: r["e"])
ends = [r["e"] for r in others]
import bisect
for a, b in zip(f[50:n - 60], f[51:n - 59]):
    if b["s"] - a["e"] < 5e3: why["(back to back)"] += 1; continue
    i = bisect.bisect_right(ends, b["s"]) - 1
    if i >= 0:
        k = short(others[i]["Kernel_Name"]) + f" grid={others[i].get('Grid_Size_X', others[i].get('Grid_Size','?'))}"
        why[k] += 1; lat[k].append(b["s"] - others[i]["e"])
for k, v in why.most_common(8):
    extra = f"  (launch starts {st.mean(lat[k])/1e3:.1f} us after it ends)" if lat[k] else ""
    print(f"  {v:5d}  {k}{extra}")
# busy time of the other kernels inside the steady phase
a, b = f[50]["s"], f[n - 60]["e"]
tot = collections.Counter(); cnt = collections.Counter()
for r in others:
    if r["s"] >= a and r["e"] <= b:
        k = short(r["Kernel_Name"]) + f" grid={r.get('Grid_Size_X', r.get('Grid_Size','?'))}"
        tot[k] += r["e"] - r["s"]; cnt[k] += 1
print(f"steady phase {(b-a)/1e6:.1f} ms; other kernels (mean duration):")
for k, v in tot.most_common(10):
    print(f"  {k:32s} n={cnt[k]:5d} mean {v/cnt[k]/1e3:8.1f} us")
