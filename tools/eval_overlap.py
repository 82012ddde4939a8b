#!/usr/bin/env python3
"""Do two evaluations on two streams overlap?  (tools; experiment)  Two host threads, each with its own scratch and stream,
evaluate half a batch each, NREP times; against one thread evaluating the whole batch."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from splpak_amd import capi
nd, nod = 3, 64
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
nparts = int(sys.argv[2]) if len(sys.argv) > 2 else 2
NREP = 40
dev = torch.device("cuda", 0)
nodes, lo, hi = [nod] * nd, [0.0] * nd, [1.0] * nd
coef = torch.randn(nod ** nd, dtype=torch.float64, device=dev)
xq = torch.empty((nq, nd), dtype=torch.float64, device=dev)
out = torch.empty(nq, dtype=torch.float64, device=dev)
capi.synth_queries_dev(nd, 0, 0, nq, xq, 0)
capi.set_eval_mode(capi.EVAL_BINNED, 0)
torch.cuda.synchronize()

def worker(k, n, reps, stream):
    capi.set_eval_mode(capi.EVAL_BINNED, 0)
    a, b = k * (nq // n), (k + 1) * (nq // n)
    for _ in range(reps):
        capi.evaluate_dev(nd, xq[a:b], None, coef, lo, hi, nodes, out[a:b], stream.cuda_stream)

def run(n):
    streams = [torch.cuda.Stream() for _ in range(n)]
    for reps in (10, NREP):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=worker, args=(k, n, reps, streams[k])) for k in range(n)]
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
    return dt * 1e3

for n in (1, nparts, 1, nparts):
    ms = run(n)
    print(f"{n} stream(s): {ms:.3f} ms per {nq} queries = {nq / ms / 1e6:.2f} Gevals/s", flush=True)
