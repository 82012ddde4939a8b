#!/usr/bin/env python3
"""Throughput of the fused value/gradient/Hessian evaluation vs separate single-pattern calls."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from splpak_amd import capi
nd, nod, nq = 3, 64, 20_000_000
dev = torch.device("cuda", 0)
nodes, lo, hi = [nod] * nd, [0.0] * nd, [1.0] * nd
coef = torch.randn(nod ** nd, dtype=torch.float64, device=dev)
xq = torch.empty((nq, nd), dtype=torch.float64, device=dev)
capi.synth_queries_dev(nd, 0, 0, nq, xq, 0)
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for order in (1, 2):
    nout = capi.derivs_nout(nd, order)
    out = torch.empty((nq, nout), dtype=torch.float64, device=dev)
    ms = timeit(lambda: capi.evaluate_derivs_dev(nd, xq, order, coef, lo, hi, nodes, out, 0))
    o1 = torch.empty(nq, dtype=torch.float64, device=dev)
    pats = [[0] * nd] + [[int(e == d) for e in range(nd)] for d in range(nd)]
    if order == 2:
        pats += [[int(e == d) + int(e == f) for e in range(nd)] for d in range(nd) for f in range(d, nd)]
    ms_sep = sum(timeit(lambda p=p: capi.evaluate_dev(nd, xq, p, coef, lo, hi, nodes, o1, 0)) for p in pats)
    print(f"order {order}: {nout} outputs per query, fused {ms:.2f} ms ({nq / ms / 1e6:.2f} G queries/s), "
          f"{len(pats)} separate single-pattern calls {ms_sep:.2f} ms")
