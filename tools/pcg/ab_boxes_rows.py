"""A/B: boxes built from the rows (scaled mass + exact constraint part; nothing assembled) against boxes extracted from the assembled N (SPLPAK_PCG_ASSEMBLE=1)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from splpak_amd import capi

def fit(nd, nodes, m, assemble, xtrap=1.0):
    os.environ["SPLPAK_SOLVER"] = "pcg"
    if assemble: os.environ["SPLPAK_PCG_ASSEMBLE"] = "1"
    else: os.environ.pop("SPLPAK_PCG_ASSEMBLE", None)
    dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
    x = torch.empty((m, nd), dtype=torch.float64, device=dev); y = torch.empty(m, dtype=torch.float64, device=dev); w = torch.empty(m, dtype=torch.float64, device=dev)
    capi.synth_points_dev(nd, 0, m, x, y, w, st)
    plan = capi.Plan(nd, nodes, [0.0] * nd, [1.0] * nd, xtrap, m)
    try:
        coef = torch.zeros(int(np.prod(nodes)), dtype=torch.float64, device=dev)
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ierr, info = plan.fit(x, y, w, coef, st)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        ps = plan.pcg_stats()
        print(f"{nd}-D {nodes} m={m} xtrap={xtrap} assembled={assemble}: ierror {ierr} {dt:.3f} s (assembly {info[5]:.3f}, solve {info[7]:.3f}), rows {info[0]:.0f}+{info[1]:.0f}, steps {info[2]:.0f}, "
              f"backward error {info[9]:.1e}, reserr {info[8]:.10e}, {plan.device_bytes() / 1e9:.1f} GB, {ps['iterations']} iterations in {ps['solves']} solves", flush=True)
        return coef.cpu().numpy(), ierr
    finally:
        plan.close()

for nd, nodes, m, xt in ((4, [6] * 4, 5000, 1.0), (4, [12] * 4, 158122, 1.0), (4, [12] * 4, 366025, 1.0), (4, [13, 12, 14, 11], 200000, 1.0), (4, [12] * 4, 158122, 0.0), (4, [16] * 4, 546750, 1.0),
                      (4, [24] * 4, 5596820, 1.0), (4, [32] * 4, 10_000_000, 1.0), (4, [28] * 4, 10_000_000, 1.0)):
    a, ea = fit(nd, nodes, m, True, xt); b, eb = fit(nd, nodes, m, False, xt)
    if ea == 0 and eb == 0: print("   from the rows vs assembled:", np.abs(a - b).max() / np.abs(a).max())
