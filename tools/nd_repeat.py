#!/usr/bin/env python3
"""N consecutive fits through one plan, each one reported: tools/nd_repeat.py ndim nodes ndata [nfits] [timing]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from splpak_amd import capi
nd, nod, m = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 6
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
x = torch.empty((m, nd), dtype=torch.float64, device=dev)
y = torch.empty(m, dtype=torch.float64, device=dev)
w = torch.empty(m, dtype=torch.float64, device=dev)
capi.synth_points_dev(nd, 0, m, x, y, w, st)
coef = torch.zeros(nod ** nd, dtype=torch.float64, device=dev)
plan = capi.Plan(nd, [nod] * nd, [0.0] * nd, [1.0] * nd, 1.0, m)
if len(sys.argv) > 5:
    plan.enable_kernel_timing(True)
ref = None
for i in range(n):
    ierr, info = plan.fit(x, y, w, coef, st)
    c = coef.cpu()
    if ref is None:
        ref = c.clone()
    import hashlib
    print(f"fit {i}: ierr {ierr} steps {info[2]:.0f} last {info[3]:.2e} omega {info[9]:.2e} factor {info[6]*1e3:.2f} ms solve {info[7]*1e3:.2f} ms  max diff to fit 0: {float((c - ref).abs().max()):.2e}  "
          f"sha {hashlib.sha1(c.numpy().tobytes()).hexdigest()[:12]}", flush=True)
plan.close()
