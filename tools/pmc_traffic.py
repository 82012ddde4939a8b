#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) to the per-launch
HBM traffic of the syrk kernels, with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE
counts 64 B per 128-B request for wide streaming reads: doubled)."""
import csv, json, sys
fetch_csv, write_csv, out = sys.argv[1], sys.argv[2], sys.argv[3]
def mean(path, counter):
    s = n = 0
    sb = nb = 0
    for r in csv.DictReader(open(path)):
        if "syrk" not in r["Kernel_Name"] or r["Counter_Name"] != counter:
            continue
        v = float(r["Counter_Value"]); s += v; n += 1
        if int(r["Grid_Size"]) > 1000000:
            sb += v; nb += 1
    return s / max(n, 1), n, sb / max(nb, 1), nb
f, nf, fb, nfb = mean(fetch_csv, "FETCH_SIZE")
w, nw, wb, nwb = mean(write_csv, "WRITE_SIZE")
res = {
    "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes), --kernel-include-regex syrk, bench.py C3",
    "launches": nf,
    "fetch_size_kb_per_launch_raw": f, "write_size_kb_per_launch": w,
    "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
    "bulk_launches": nfb, "bulk_fetch_kb_raw": fb, "bulk_write_kb": wb,
    "bulk_hbm_bytes_per_launch": (2.0 * fb + wb) * 1024.0,
    "note": "FETCH_SIZE x 2 is the gfx950 correction for wide streaming reads; for the 8-B/lane loads of syrk64_kernel it is an upper bound (see profiles/r01_syrk_pmc.json)",
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
