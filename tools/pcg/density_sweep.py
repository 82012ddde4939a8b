"""Where the iterative solve works: 4-D grids, iteration alone (SPLPAK_SOLVER=pcg), points per grid cell swept.
usage: density_sweep.py nodes_per_dim ppc [ppc ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from splpak_amd import capi
os.environ["SPLPAK_SOLVER"] = "pcg"
nd, nod = 4, int(sys.argv[1])
dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
for arg in sys.argv[2:]:
    xtrap = 1.0
    if arg.endswith("x0"):
        xtrap, arg = 0.0, arg[:-2]
    ppc = float(arg)
    m = int(ppc * (nod - 1) ** nd)
    x = torch.empty((m, nd), dtype=torch.float64, device=dev); y = torch.empty(m, dtype=torch.float64, device=dev); w = torch.empty(m, dtype=torch.float64, device=dev)
    capi.synth_points_dev(nd, 0, m, x, y, w, st)
    plan = capi.Plan(nd, [nod] * nd, [0.0] * nd, [1.0] * nd, xtrap, m)
    try:
        coef = torch.zeros(nod ** nd, dtype=torch.float64, device=dev)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ierr, info = plan.fit(x, y, w, coef, st)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        ps = plan.pcg_stats()
        print(f"{nod}^4 xtrap {xtrap:.0f} points/cell {ppc:5.1f} (m = {m}): sparse nodes {info[1] / 10 / nod ** nd:6.3f} of all, constraint rows per column {info[1] / nod ** nd:5.2f}; "
              f"ierror {ierr}, {ps['iterations']} iterations in {ps['solves']} solves, last residual {ps['last_residual']:.1e}, {dt:.2f} s, backward error {info[9]:.1e}", flush=True)
    finally:
        plan.close()
    del x, y, w
