#!/usr/bin/env python3
"""Binned and direct evaluation batches (EVAL_PROFILE_REPS of each, default 3), for rocprofv3 --kernel-trace --stats / --pmc."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from splpak_amd import capi
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 3
nod = int(sys.argv[2]) if len(sys.argv) > 2 else 64
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 50_000_000
chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = torch.device("cuda", 0)
nodes, lo, hi = [nod] * nd, [0.0] * nd, [1.0] * nd
coef = torch.randn(nod ** nd, dtype=torch.float64, device=dev)
xq = torch.empty((nq, nd), dtype=torch.float64, device=dev)
out = torch.empty(nq, dtype=torch.float64, device=dev)
capi.synth_queries_dev(nd, 0, 0, nq, xq, 0)
for mode in (capi.EVAL_BINNED, capi.EVAL_DIRECT):
    capi.set_eval_mode(mode, chunk)
    for _ in range(int(os.environ.get("EVAL_PROFILE_REPS", "3"))):
        capi.evaluate_dev(nd, xq, None, coef, lo, hi, nodes, out, 0)
    torch.cuda.synchronize()
