#!/usr/bin/env python3
"""Time the binned evaluation of one batch (tools; experiment switches through the environment)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from splpak_amd import capi
nd, nod = 3, 64
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
dev = torch.device("cuda", 0)
nodes, lo, hi = [nod] * nd, [0.0] * nd, [1.0] * nd
coef = torch.randn(nod ** nd, dtype=torch.float64, device=dev)
xq = torch.empty((nq, nd), dtype=torch.float64, device=dev)
out = torch.empty(nq, dtype=torch.float64, device=dev)
capi.synth_queries_dev(nd, 0, 0, nq, xq, 0)
capi.set_eval_mode(capi.EVAL_BINNED, 0)
for _ in range(20):
    capi.evaluate_dev(nd, xq, None, coef, lo, hi, nodes, out, 0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(40):
    capi.evaluate_dev(nd, xq, None, coef, lo, hi, nodes, out, 0)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 40
print(f"exp={os.environ.get('SPLPAK_PR_EXP', '0')}: {ms:.3f} ms per {nq} queries = {nq / ms / 1e6:.2f} Gevals/s", flush=True)
