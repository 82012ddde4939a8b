"""A/B of the rows-only assembly of iteration-only 4-D plans (no normal equations) against the assembled form (SPLPAK_PCG_ASSEMBLE=1)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from splpak_amd import capi

def fit(nd, nod, m, assemble, xtrap=1.0, want_hist=False):
    os.environ["SPLPAK_SOLVER"] = "pcg"
    if assemble: os.environ["SPLPAK_PCG_ASSEMBLE"] = "1"
    else: os.environ.pop("SPLPAK_PCG_ASSEMBLE", None)
    dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
    x = torch.empty((m, nd), dtype=torch.float64, device=dev); y = torch.empty(m, dtype=torch.float64, device=dev); w = torch.empty(m, dtype=torch.float64, device=dev)
    capi.synth_points_dev(nd, 0, m, x, y, w, st)
    plan = capi.Plan(nd, [nod] * nd, [0.0] * nd, [1.0] * nd, xtrap, m)
    try:
        coef = torch.zeros(nod ** nd, dtype=torch.float64, device=dev)
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ierr, info = plan.fit(x, y, w, coef, st)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{nd}-D {nod}^{nd} m={m} xtrap={xtrap} assemble={assemble}: ierror {ierr} {dt:.3f} s (assembly {info[5]:.3f}), rows {info[0]:.0f}+{info[1]:.0f}, steps {info[2]:.0f}, backward error {info[9]:.1e}, "
              f"reserr {info[8]:.12e}, {plan.device_bytes() / 1e9:.1f} GB, pcg {plan.pcg_stats()['iterations']}", flush=True)
        return coef.cpu().numpy()
    finally:
        plan.close()

for nd, nod, m, xt in ((4, 6, 5000, 1.0), (4, 12, 158122, 1.0), (4, 12, 158122, 0.0), (4, 16, 546750, 1.0)):
    a = fit(nd, nod, m, True, xt); b = fit(nd, nod, m, False, xt)
    print("   rows-only vs assembled:", np.abs(a - b).max() / np.abs(a).max())
fit(4, 32, 10_000_000, False)
