#!/usr/bin/env python3
"""Every launch of the LAST fit in a rocprofv3 kernel-trace CSV (from its first binning kernel on): start offset, duration, gap to
the previous launch's end on any queue, queue, workgroups, kernel.   tools/last_fit_trace.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
i0 = max(i for i, r in enumerate(rows) if "keys_kernel" in r["Kernel_Name"] or "sp_count_kernel" in r["Kernel_Name"])
sel = rows[i0:]
t0 = int(sel[0]["Start_Timestamp"])
end = t0
busy = 0
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"].replace("splpak::", "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    wg = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - end) / 1e3:7.1f}  q={r['Queue_Id']:>2} wg={wg:>6}  {nm[:50]}")
    busy += max(0, e - max(s, end))
    end = max(end, e)
print(f"span {(end - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us, launches {len(sel)}")
