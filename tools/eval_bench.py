#!/usr/bin/env python3
"""Evaluation throughput of the direct and the binned path (tools; not part of the test tier).

    python tools/eval_bench.py [ndim] [nodes] [nq]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from splpak_amd import capi

nd = int(sys.argv[1]) if len(sys.argv) > 1 else 3
nod = int(sys.argv[2]) if len(sys.argv) > 2 else 64
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 50_000_000
dev = torch.device("cuda", 0)
nodes = [nod] * nd
lo, hi = [0.0] * nd, [1.0] * nd
coef = torch.randn(nod ** nd, dtype=torch.float64, device=dev)
xq = torch.empty((nq, nd), dtype=torch.float64, device=dev)
out = torch.empty(nq, dtype=torch.float64, device=dev)
ref = torch.empty(nq, dtype=torch.float64, device=dev)
capi.synth_queries_dev(nd, 0, 0, nq, xq, 0)


def run(mode, chunk, dst, reps=5):
    capi.set_eval_mode(mode, chunk)
    capi.evaluate_dev(nd, xq, None, coef, lo, hi, nodes, dst, 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        capi.evaluate_dev(nd, xq, None, coef, lo, hi, nodes, dst, 0)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ms = run(capi.EVAL_DIRECT, 0, ref)
print(f"{nd}-D {nod}^{nd} nq={nq}: direct {ms:8.3f} ms  {nq / ms / 1e6:8.2f} Gevals/s", flush=True)
for sh in (18, 20, 21, 22, 23, 24, 26):
    ms = run(capi.EVAL_BINNED, 1 << sh, out)
    same = bool(torch.equal(out, ref))
    print(f"   binned chunk 2^{sh}: {ms:8.3f} ms  {nq / ms / 1e6:8.2f} Gevals/s  identical={same}", flush=True)
# sorted queries (a best case for the direct path: neighbouring threads share lines)
xs, _ = torch.sort(xq[:, nd - 1])
xq[:, nd - 1] = xs
ms = run(capi.EVAL_DIRECT, 0, ref)
print(f"   last coordinate sorted: direct {ms:8.3f} ms  {nq / ms / 1e6:8.2f} Gevals/s", flush=True)
ms = run(capi.EVAL_BINNED, 0, out)
print(f"   last coordinate sorted: binned {ms:8.3f} ms  {nq / ms / 1e6:8.2f} Gevals/s identical={bool(torch.equal(out, ref))}", flush=True)
capi.set_eval_mode(capi.EVAL_AUTO)
