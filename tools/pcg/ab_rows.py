"""A/B of the tiled residual pass (csrc/rowsop.hip) against the cell-by-cell passes: same fit, SPLPAK_ROWS_TILES=0/1, 4-D grids."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from splpak_amd import capi

def fit(nd, nod, m, solver, tiles):
    os.environ["SPLPAK_SOLVER"] = solver; os.environ["SPLPAK_ROWS_TILES"] = tiles
    dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
    x = torch.empty((m, nd), dtype=torch.float64, device=dev); y = torch.empty(m, dtype=torch.float64, device=dev); w = torch.empty(m, dtype=torch.float64, device=dev)
    capi.synth_points_dev(nd, 0, m, x, y, w, st)
    plan = capi.Plan(nd, [nod] * nd, [0.0] * nd, [1.0] * nd, 1.0, m)
    try:
        coef = torch.zeros(nod ** nd, dtype=torch.float64, device=dev)
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ierr, info = plan.fit(x, y, w, coef, st)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{nd}-D {nod}^{nd} m={m} solver={solver} tiles={tiles}: ierror {ierr} {dt:.3f} s, steps {info[2]:.0f}, last dx {info[3]:.1e}, backward error {info[9]:.1e}, reserr {info[8]:.12e}, pcg {plan.pcg_stats()}", flush=True)
        return coef.cpu().numpy()
    finally:
        plan.close()

for nd, nod, m in ((4, 6, 5000), (4, 12, 158122), (4, 16, 546750)):
    a = fit(nd, nod, m, "direct", "0"); b = fit(nd, nod, m, "direct", "1")
    print("   direct: tiles vs cells", np.abs(a - b).max() / np.abs(a).max())
    c = fit(nd, nod, m, "pcg", "0"); d = fit(nd, nod, m, "pcg", "1")
    print("   pcg: tiles vs cells", np.abs(c - d).max() / np.abs(c).max(), " pcg vs direct", np.abs(d - a).max() / np.abs(a).max())
fit(4, 32, 10_000_000, "pcg", "1")
