#!/usr/bin/env python3
"""Time the one-shot HOST entry point (what the Fortran module calls) on the C3 workload:
PCIe-inclusive fit rate (DESIGN.md notes it next to the resident-data rate of bench.py)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from splpak_amd import capi
from splpak_amd.synth import synth_points
nd, nod, m = 3, int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
x, y, w = synth_points(nd, m)
for rep in range(3):
    t0 = time.perf_counter()
    coef, ierr, _, info = capi.fit(nd, x, y, w, [0.0] * nd, [1.0] * nd, [nod] * nd, 1.0)
    dt = time.perf_counter() - t0
    print(f"host-pointer fit #{rep}: ierror={ierr} {dt*1e3:.1f} ms total -> {m/dt:.3e} points/s "
          f"(assembly {info[5]*1e3:.1f} factor {info[6]*1e3:.1f} solve {info[7]*1e3:.1f} ms; rest = alloc + PCIe)", flush=True)
