#!/usr/bin/env python3
"""Fit time on an anisotropic node grid with and without the internal dimension reordering
(SPLPAK_NO_REORDER=1 disables it):  python tools/aniso_bench.py 64 64 16"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from splpak_amd import capi

nodes = [int(a) for a in sys.argv[1:]] or [64, 64, 16]
nd, m = len(nodes), 1_000_000
dev = torch.device("cuda", 0)
x = torch.empty((m, nd), dtype=torch.float64, device=dev)
y = torch.empty(m, dtype=torch.float64, device=dev)
w = torch.empty(m, dtype=torch.float64, device=dev)
capi.synth_points_dev(nd, 0, m, x, y, w, 0)
ncol = 1
for n in nodes:
    ncol *= n
coef = torch.zeros(ncol, dtype=torch.float64, device=dev)
plan = capi.Plan(nd, nodes, [0.0] * nd, [1.0] * nd, 1.0, m)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ierr, info = plan.fit(x, y, w, coef, 0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"nodes {nodes} reorder={'off' if os.environ.get('SPLPAK_NO_REORDER') else 'on'}: ierror {ierr}, {1e3 * dt:.1f} ms per fit "
      f"(factor {1e3 * info[6]:.1f} ms), checksum {float(coef.abs().sum()):.12e}")
plan.close()
