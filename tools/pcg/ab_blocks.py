"""A/B of the block-Jacobi component of the iterative solve (SPLPAK_PCG_NO_BLOCKS=1 switches it off; that plan is rows-only)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from splpak_amd import capi

def fit(nd, nodes, m, blocks, xtrap=1.0, solver="pcg"):
    os.environ["SPLPAK_SOLVER"] = solver
    if blocks: os.environ.pop("SPLPAK_PCG_NO_BLOCKS", None)
    else: os.environ["SPLPAK_PCG_NO_BLOCKS"] = "1"
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
        print(f"{nd}-D {nodes} m={m} xtrap={xtrap} solver={solver} blocks={blocks}: ierror {ierr} {dt:.3f} s (assembly {info[5]:.3f}, solve {info[7]:.3f}), rows/col {info[1] / np.prod(nodes):.2f}, steps {info[2]:.0f}, "
              f"backward error {info[9]:.1e}, {plan.device_bytes() / 1e9:.1f} GB, {ps['iterations']} iterations in {ps['solves']} solves", flush=True)
        return coef.cpu().numpy(), ierr
    finally:
        plan.close()

cases = [(4, [12] * 4, 158122), (4, [12] * 4, 366025), (4, [13, 12, 14, 11], 200000), (4, [16] * 4, 546750), (3, [24] * 3, 486680), (2, [64, 64], 100000), (3, [32] * 3, 300000)]
if len(sys.argv) > 1: cases = []
for nd, nodes, m in cases:
    a, ea = fit(nd, nodes, m, False); b, eb = fit(nd, nodes, m, True); c, ec = fit(nd, nodes, m, True, solver="direct")
    if ea == 0: print("   separable alone vs factorisation:", np.abs(a - c).max() / np.abs(c).max())
    if eb == 0: print("   with boxes vs factorisation:     ", np.abs(b - c).max() / np.abs(c).max())
for arg in sys.argv[1:]:
    nd, nod, m = (int(float(v)) for v in arg.split(","))
    fit(nd, [nod] * nd, m, True)
