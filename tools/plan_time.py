import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from splpak_amd import capi
torch.cuda.init(); torch.zeros(1, device="cuda")
for nd, nod, m in ((1, 16, 1000), (2, 16, 10000), (2, 64, 1000000), (3, 32, 1000000), (3, 64, 10000000)):
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        p = capi.Plan(nd, [nod] * nd, [0.0] * nd, [1.0] * nd, 1.0, m)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        p.close()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1))
    print(f"{nd}-D {nod}^{nd} m={m}: create {min(t[0] for t in ts)*1e3:.2f} ms (first {ts[0][0]*1e3:.2f}), destroy {min(t[1] for t in ts)*1e3:.2f} ms")
# the one-shot pattern: a fresh plan per fit (what a caller with a new grid per data set pays)
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
for nd, nod, m in ((2, 16, 10000), (2, 64, 1000000), (3, 32, 1000000)):
    x = torch.empty((m, nd), dtype=torch.float64, device=dev); y = torch.empty(m, dtype=torch.float64, device=dev)
    w = torch.empty(m, dtype=torch.float64, device=dev)
    capi.synth_points_dev(nd, 0, m, x, y, w, st)
    coef = torch.zeros(nod ** nd, dtype=torch.float64, device=dev)
    ts = []
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        p = capi.Plan(nd, [nod] * nd, [0.0] * nd, [1.0] * nd, 1.0, m)
        ierr, info = p.fit(x, y, w, coef, st)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        ierr, info = p.fit(x, y, w, coef, st)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        p.close(); torch.cuda.synchronize(); t3 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1, t3 - t2))
    print(f"{nd}-D {nod}^{nd}: fresh plan + first fit {min(t[0] for t in ts)*1e3:.2f} ms, second fit {min(t[1] for t in ts)*1e3:.2f} ms, destroy {min(t[2] for t in ts)*1e3:.2f} ms")
