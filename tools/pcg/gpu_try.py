"""Round-6 bring-up of the iterative solve on the GPU: iteration counts, times, agreement with the factorisation.
usage: gpu_try.py [ndim nodes ndata [compare]] ..."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from splpak_amd import capi  # noqa: E402


def log(*a):
    print(time.strftime("%H:%M:%S"), *a, flush=True)


def run(nd, nod, m, compare, xtrap=1.0):
    nodes, lo, hi = [nod] * nd, [0.0] * nd, [1.0] * nd
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    x = torch.empty((m, nd), dtype=torch.float64, device=dev)
    y = torch.empty(m, dtype=torch.float64, device=dev)
    w = torch.empty(m, dtype=torch.float64, device=dev)
    capi.synth_points_dev(nd, 0, m, x, y, w, st)
    ncol = nod ** nd
    res = {}
    for solver in (["direct", "pcg"] if compare else ["auto"]):
        if solver == "auto":
            os.environ.pop("SPLPAK_SOLVER", None)
        else:
            os.environ["SPLPAK_SOLVER"] = solver
        t0 = time.perf_counter()
        plan = capi.Plan(nd, nodes, lo, hi, xtrap, m)
        t1 = time.perf_counter()
        log(f"{nd}-D {nod}^{nd} m={m} solver={solver}: plan in {t1 - t0:.1f} s: {plan.factorisation()}; {plan.device_bytes() / 1e9:.1f} GB")
        try:
            coef = torch.zeros(ncol, dtype=torch.float64, device=dev)
            for rep in range(2):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                ierr, info = plan.fit(x, y, w, coef, st)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                log(f"  fit {rep}: ierror {ierr}, {t2 - t1:.3f} s (assembly {info[5]:.3f}, factorisation {info[6]:.3f}, solve {info[7]:.3f}); rows {info[0]:.0f} + {info[1]:.0f}, "
                    f"steps {info[2]:.0f}, last dx {info[3]:.1e}, backward error {info[9]:.1e}, reserr {info[8]:.9e}; pcg {plan.pcg_stats()}")
            res[solver] = coef.cpu().numpy()
        finally:
            plan.close()
    if compare:
        a, b = res["direct"], res["pcg"]
        log(f"  pcg vs direct: max |dcoef| / max |coef| = {np.abs(a - b).max() / np.abs(a).max():.2e}")


if __name__ == "__main__":
    args = sys.argv[1:]
    while args:
        nd, nod, m = int(args[0]), int(args[1]), int(float(args[2]))
        compare = len(args) > 3 and args[3] == "compare"
        args = args[4:] if compare else args[3:]
        run(nd, nod, m, compare)
