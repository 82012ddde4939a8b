"""BASELINE config 5's fit half at the largest 4-D grid one MI355X holds (round 5: 28^4 = 614 656 columns; the postorder
schedule with packed Schur buffers, csrc/ndtree.hpp NdSchedule): (a) 1e7 points sampled from a spline with random
coefficients, xtrap = 0 -> the coefficients come back; (b) the seeded weighted workload with xtrap = 1 and the HOST-side
backward error over the reference's rows (oracle_rows_gradient).  usage: c5_fit.py [nodes_per_dim] [ndata] [a|b|ab]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from splpak_amd import capi  # noqa: E402


def log(*a):
    print(time.strftime("%H:%M:%S"), *a, flush=True)


def main():
    nod = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    m = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
    what = sys.argv[3] if len(sys.argv) > 3 else "ab"
    nd = 4
    nodes, lo, hi = [nod] * nd, [0.0] * nd, [1.0] * nd
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    free, total = torch.cuda.mem_get_info()
    log(f"device memory: {free / 1e9:.1f} GB free of {total / 1e9:.1f}")
    tree = capi.debug_nd_tree(nodes, check=False)
    log(f"{nod}^4: {tree['fronts']:.0f} fronts, {tree['factor_bytes'] / 1e9:.1f} GB of panels, level-order arenas {tree['arena_bytes'] / 1e9:.1f} GB, "
        f"{tree['flop']:.3e} flop")
    x = torch.empty((m, nd), dtype=torch.float64, device=dev)
    y = torch.empty(m, dtype=torch.float64, device=dev)
    w = torch.empty(m, dtype=torch.float64, device=dev)
    capi.synth_points_dev(nd, 0, m, x, y, w, st)
    ncol = nod ** nd
    if "a" in what:
        gen = torch.Generator(device=dev)
        gen.manual_seed(11)
        ctrue = torch.randn(ncol, dtype=torch.float64, device=dev, generator=gen)
        ya = torch.empty(m, dtype=torch.float64, device=dev)
        capi.evaluate_dev(nd, x, None, ctrue, lo, hi, nodes, ya, st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan = capi.Plan(nd, nodes, lo, hi, 0.0, m)
        t1 = time.perf_counter()
        log(f"plan created in {t1 - t0:.1f} s: {plan.factorisation()[1]}; plan holds {plan.device_bytes() / 1e9:.1f} GB")
        try:
            coef = torch.zeros(ncol, dtype=torch.float64, device=dev)
            for rep in range(2):
                t1 = time.perf_counter()
                ierr, info = plan.fit(x, ya, None, coef, st)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                err = float((coef - ctrue).abs().max() / ctrue.abs().max())
                log(f"(a) projection fit {rep}: ierror {ierr}, {t2 - t1:.2f} s (assembly {info[5]:.2f}, factorisation {info[6]:.2f}, solve {info[7]:.2f}); "
                    f"coefficient error {err:.2e}, steps {info[2]:.0f}, backward error {info[9]:.1e}; {m / (t2 - t1):.3e} points/s; "
                    f"factorisation {tree['flop'] / max(info[6], 1e-9) / 1e12:.1f} TFLOP/s (padded flop)")
        finally:
            plan.close()
        del ctrue, ya
    if "b" in what:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan = capi.Plan(nd, nodes, lo, hi, 1.0, m)
        t1 = time.perf_counter()
        log(f"plan (xtrap = 1) created in {t1 - t0:.1f} s")
        try:
            coef = torch.zeros(ncol, dtype=torch.float64, device=dev)
            ierr, info = plan.fit(x, y, w, coef, st)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            log(f"(b) weighted seeded fit: ierror {ierr}, {t2 - t1:.2f} s (assembly {info[5]:.2f}, factorisation {info[6]:.2f}, solve {info[7]:.2f}); rows {info[0]:.0f} + "
                f"{info[1]:.0f} constraint rows, steps {info[2]:.0f}, backward error {info[9]:.1e}, reserr {info[8]:.9e}")
            c = coef.cpu().numpy()
        finally:
            plan.close()
        from oracle.binding import Port
        xh, yh, wh = x.cpu().numpy(), y.cpu().numpy(), w.cpu().numpy()
        t3 = time.perf_counter()
        omega, reserr, nrow, ncons = Port().rows_gradient(nd, xh, yh, wh, lo, hi, nodes, 1.0, c)
        log(f"(b) host rows_gradient in {time.perf_counter() - t3:.1f} s: backward error {omega:.2e}, rows {nrow} + {ncons}, reserr {reserr:.9e}")


if __name__ == "__main__":
    main()
