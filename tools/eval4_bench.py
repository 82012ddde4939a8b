#!/usr/bin/env python3
"""4-D 32^4 evaluation rate (BASELINE config 5's evaluation half): tools/eval4_bench.py [nq]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from splpak_amd import capi
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
nd, nod = 4, 32
dev = torch.device("cuda", 0)
coef = torch.randn(nod ** nd, dtype=torch.float64, device=dev)
xq = torch.empty((nq, nd), dtype=torch.float64, device=dev)
out = torch.empty(nq, dtype=torch.float64, device=dev)
capi.synth_queries_dev(nd, 0, 0, nq, xq, 0)
for pat in (None, [1, 0, 0, 0]):
    for _ in range(2):
        capi.evaluate_dev(nd, xq, pat, coef, [0.0] * nd, [1.0] * nd, [nod] * nd, out, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        capi.evaluate_dev(nd, xq, pat, coef, [0.0] * nd, [1.0] * nd, [nod] * nd, out, 0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"4-D 32^4 nderiv={pat}: {dt*1e3:.2f} ms per {nq} queries = {nq/dt/1e9:.2f} Gevals/s")
