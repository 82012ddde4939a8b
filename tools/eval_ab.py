#!/usr/bin/env python3
"""Evaluation throughput at C3 (5e7 random queries, 64^3 nodes) for A/B runs of two library builds:
    SPLPAK_LIB=/path/to/other/libsplpak_hip.so python tools/eval_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from splpak_amd import capi
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
nd, nod, nq = 3, 64, 50_000_000
coef = torch.randn(nod ** nd, dtype=torch.float64, device=dev)
xq = torch.empty((nq, nd), dtype=torch.float64, device=dev)
out = torch.empty(nq, dtype=torch.float64, device=dev)
capi.synth_queries_dev(nd, 10_000_000, 0, nq, xq, st)
lo, hi, nodes = [0.0] * nd, [1.0] * nd, [nod] * nd
for mode, name in ((capi.EVAL_AUTO, "binned"), (capi.EVAL_DIRECT, "direct")):
    capi.set_eval_mode(mode)
    for _ in range(3):
        capi.evaluate_dev(nd, xq, None, coef, lo, hi, nodes, out, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(4):
        e0.record()
        for _ in range(5):
            capi.evaluate_dev(nd, xq, None, coef, lo, hi, nodes, out, st)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    print(f"{os.path.basename(capi.LIB_PATH)} {name}: {best:.3f} ms per 5e7 queries = {nq / best / 1e6:.2f}e9 evals/s")
