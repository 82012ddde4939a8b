#!/usr/bin/env python3
"""Fit time of the distributed-band path against the single-GPU path on resident data.

    python tools/dist_bench.py [nodes] [ndata] [ngpus ...]        (SPLPAK_VIRTUAL_GPUS=1: all ranks on device 0)
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from splpak_amd import capi

nod = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
rs = [int(a) for a in sys.argv[3:]] or [2, 4]
chunk = int(os.environ.get("SPLPAK_DIST_CHUNK", "0"))
nd = 3
nodes, lo, hi = [nod] * nd, [0.0] * nd, [1.0] * nd
ndev = torch.cuda.device_count()
virt = bool(os.environ.get("SPLPAK_VIRTUAL_GPUS"))


def points(dev, first, n):
    torch.cuda.set_device(dev)
    x = torch.empty((n, nd), dtype=torch.float64, device=dev)
    y = torch.empty(n, dtype=torch.float64, device=dev)
    w = torch.empty(n, dtype=torch.float64, device=dev)
    capi.synth_points_dev(nd, first, n, x, y, w, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return x, y, w


d0 = torch.device("cuda", 0)
x, y, w = points(d0, 0, m)
coef1 = torch.zeros(nod ** nd, dtype=torch.float64, device=d0)
plan = capi.Plan(nd, nodes, lo, hi, 1.0, m)
st = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    ierr, info = plan.fit(x, y, w, coef1, st)
torch.cuda.synchronize()
t0 = time.perf_counter(); ierr, info = plan.fit(x, y, w, coef1, st); torch.cuda.synchronize(); t1 = time.perf_counter() - t0
print(f"{nod}^3 m={m}: single GPU {t1*1e3:8.1f} ms  phases {info[5]*1e3:.1f} / {info[6]*1e3:.1f} / {info[7]*1e3:.1f} ms  ierr={ierr}", flush=True)
plan.close()
for R in rs:
    if not virt and R > ndev:
        print(f"   {R} GPUs: only {ndev} device(s)"); continue
    per = (m + R - 1) // R
    mp = capi.MultiPlan(R, nd, nodes, lo, hi, 1.0, per, chunk=chunk)
    xs, ys, ws = [], [], []
    for r in range(R):
        n = max(0, min(per, m - r * per))
        xr, yr, wr = points(torch.device("cuda", mp.device(r)), r * per, n)
        xs.append(xr); ys.append(yr); ws.append(wr)
    torch.cuda.set_device(0)
    coef = torch.zeros(nod ** nd, dtype=torch.float64, device=torch.device("cuda", mp.device(0)))
    for _ in range(2):
        ierr, info = mp.fit(xs, ys, ws, coef)
    t0 = time.perf_counter(); ierr, info = mp.fit(xs, ys, ws, coef); t2 = time.perf_counter() - t0
    err = float((coef.to(d0) - coef1).abs().max() / coef1.abs().max())
    print(f"   {R} {'virtual ' if virt else ''}GPUs chunk {chunk}: {t2*1e3:8.1f} ms  phases {info[5]*1e3:.1f} / {info[6]*1e3:.1f} / {info[7]*1e3:.1f} ms  "
          f"ierr={ierr} vs single {err:.1e} steps={info[2]:.0f}", flush=True)
    mp.close()
