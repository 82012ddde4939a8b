"""A/B of the half-stage schedule of the nested-dissection factorisation (SPLPAK_ND_HALVES = deepest tree depth that is split by the
root's two subtrees; 0 = whole stages).  usage: ab_halves.py [nd,nodes,points] [halves values ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
from splpak_amd import capi

def run(nd, nodes, m, halves, fits=6):
    os.environ["SPLPAK_SOLVER"] = "direct"
    os.environ["SPLPAK_ND_HALVES"] = str(halves)
    dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
    x = torch.empty((m, nd), dtype=torch.float64, device=dev); y = torch.empty(m, dtype=torch.float64, device=dev); w = torch.empty(m, dtype=torch.float64, device=dev)
    capi.synth_points_dev(nd, 0, m, x, y, w, st)
    plan = capi.Plan(nd, nodes, [0.0] * nd, [1.0] * nd, 1.0, m)
    try:
        coef = torch.zeros(int(np.prod(nodes)), dtype=torch.float64, device=dev)
        ts = []
        for _ in range(fits):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ierr, info = plan.fit(x, y, w, coef, st)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"{nd}-D {nodes} m={m} halves={halves}: ierror {ierr}, fits {' '.join(f'{1e3 * t:.1f}' for t in ts)} ms, factor {1e3 * info[6]:.1f} ms, "
              f"backward error {info[9]:.1e}, {plan.device_bytes() / 1e9:.1f} GB", flush=True)
        return coef.cpu().numpy()
    finally:
        plan.close()

nd, nod, m = 3, 64, 10_000_000
vals = [0, 1, 2, 3, 4, 9]
args = sys.argv[1:]
if args and "," in args[0]:
    nd, nod, m = (int(float(v)) for v in args[0].split(","))
    args = args[1:]
if args: vals = [int(v) for v in args]
ref = None
for h in vals:
    c = run(nd, [nod] * nd, m, h)
    if ref is None: ref = c
    else: print("   same bits as the first" if np.array_equal(c, ref) else f"   DIFFERENT: {np.abs(c - ref).max() / np.abs(ref).max():.2e}", flush=True)
