#!/usr/bin/env python3
"""A few fits of BASELINE config 2 (2-D, 64x64 nodes, 1e6 points, splcc) or another small grid, for rocprofv3 / phase times.
    python tools/c2_profile.py [ndim] [nodes] [ndata]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from splpak_amd import capi
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nod = int(sys.argv[2]) if len(sys.argv) > 2 else 64
m = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
dev = torch.device("cuda", 0)
if os.environ.get("C2_OWN_STREAM"):            # a created (non-NULL) stream instead of the legacy default stream
    _own = torch.cuda.Stream()
    torch.cuda.set_stream(_own)
st = torch.cuda.current_stream().cuda_stream
x = torch.empty((m, nd), dtype=torch.float64, device=dev)
y = torch.empty(m, dtype=torch.float64, device=dev)
w = torch.empty(m, dtype=torch.float64, device=dev)
capi.synth_points_dev(nd, 0, m, x, y, w, st)
coef = torch.zeros(nod ** nd, dtype=torch.float64, device=dev)
plan = capi.Plan(nd, [nod] * nd, [0.0] * nd, [1.0] * nd, 1.0, m)
ww = None if nd == 2 else w
for _ in range(int(os.environ.get("C2_WARM", 3))):
    plan.fit(x, y, ww, coef, st)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = int(os.environ.get("C2_REPS", 10))
for _ in range(n):
    ierr, info = plan.fit(x, y, ww, coef, st)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"{nd}-D {nod}^{nd} m={m}: {dt*1e3:.3f} ms per fit; phases {info[5]*1e3:.3f} / {info[6]*1e3:.3f} / {info[7]*1e3:.3f} ms; steps {info[2]:.0f} ierr {ierr}")
if os.environ.get("C2_STAGES"):
    plan.enable_kernel_timing(True)
    plan.fit(x, y, ww, coef, st)
    print("  stages (ms):", {k: round(v, 3) for k, v in plan.stage_timing().items()})
plan.close()
