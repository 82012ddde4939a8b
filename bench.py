#!/usr/bin/env python3
"""Headline benchmark: SPLPAK least-squares spline FIT (splcw) + EVALUATION (splfe)
on MI355X, BASELINE.json config 3: 3-D, 1e7 scattered weighted points per GPU,
64x64x64 nodes, real64.

    python bench.py --gpus N --steps K --warmup W

A "step" is one complete fit of this rank's points that are already resident in
HBM: window binning, Gram assembly, (RCCL all-reduce of the normal equations when
N > 1), derivative-constraint rows, Cholesky of the normal equations on the f64
matrix cores (nested-dissection multifrontal for this grid, csrc/ndchol.hip),
solve and iterative refinement against the rows, coefficients left in HBM.  For
N > 1 the points are sharded (weak scaling: every rank holds --ndata points;
--gpus 8 defaults to BASELINE config 4: 1e8 points in all) and the histogram, the
normal equations and each refinement residual are all-reduced over RCCL -- through the
library's NATIVE hook (splpak_plan_set_rccl: ncclAllReduce on the fit's stream, no Python
and no host synchronisation in the loop; torch.distributed is the fallback and carries the
bench's own barriers) -- and the nested-dissection factorisation is distributed by subtrees
(the top of the tree replicated).  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline      the dominant kernel (nd_syrk_kernel<4,2,true,1,1>: f64-MFMA Schur-buffer passes of the
                multifrontal factorisation): algorithmic flop / HIP-event time measured inside the timed region
  cpu_baseline  the reference itself (oracle/_ref, 1 core; it is single-threaded)
                on a bounded sample -- the dense reference algorithm cannot run the
                64^3 grid at all (550 GB workspace)
  eval          batched evaluation throughput (one thread per query) + its HBM roofline
  strong        (N > 1) config 3's 1e7 points in all, sharded: the fixed-total line beside the weak headline
  c2, grid32, c5_eval, c5_fit, fit_incl_h2d, multi_gpu_one_process   guarded side legs (N = 1; the last one also at N > 1)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Before the HIP runtime starts (DESIGN 4a "One more stream is not free"): how the process's streams fall onto the hardware
# queues changes the factorisation by up to 11 % (one unused stream created before the plan: 32^3 12.2 -> 14.2 ms; five: 64^3
# 222 -> 246 ms) -- with two hardware queues for the normal-priority streams the time is the best one whatever else the
# process has created (RCCL and torch.distributed bring streams of their own).  The caller's setting wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")

import numpy as np
import torch

def committed_profile(*names):
    """The newest of the committed rocprofv3 summaries under profiles/ (the counters are collected in runs of their own)."""
    for n in names:
        q = os.path.join(ROOT, "profiles", n)
        if os.path.exists(q):
            return q
    return None


F64_MFMA_PEAK_TFLOPS = 78.6    # MI355X FP64 matrix peak (AMD datasheet; = 256 CU * 4 SIMD * 32 flop/clk * 2.4 GHz)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--ndim", type=int, default=3)
    ap.add_argument("--nodes", type=int, default=64, help="nodes per dimension")
    ap.add_argument("--ndata", type=int, default=None,
                    help="points per GPU (default: 1e7 = BASELINE config 3; with --gpus 8 on the 3-D 64^3 grid 1.25e7 = config 4's 1e8 points in all)")
    ap.add_argument("--neval", type=int, default=50_000_000, help="evaluation queries per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-side-legs", action="store_true", help="skip the C2 and host-pointer (PCIe-inclusive) legs")
    ap.add_argument("--dist-child", type=str, default="", help=argparse.SUPPRESS)   # internal: "ngpus,nod,m,virtual"
    return ap.parse_args()


def default_ndata(world, nd, nod):
    """Points per GPU when --ndata is not given: BASELINE config 4 (1e8 points over 8 GPUs) at N = 8 on the
    config's grid, config 3's 1e7 per GPU otherwise."""
    return 12_500_000 if (world == 8 and nd == 3 and nod == 64) else 10_000_000


def workload_label(world, nd, nod, m):
    """config.workload: which BASELINE config the line is quoted on."""
    nodes = "x".join([str(nod)] * nd)
    base = (f"{nd}-D splcw least-squares spline fit, {{pts}} weighted scattered points (Park-Miller stream, seed 42), "
            f"{nodes} nodes, xtrap=1, real64")
    if nd == 3 and nod == 64 and world == 1 and m == 10_000_000:
        return "C3: " + base.format(pts=f"{m}")
    if nd == 3 and nod == 64 and world * m == 100_000_000 and world == 8:
        return "C4: " + base.format(pts=f"{world * m} (= {m} per GPU, sharded over {world} GPUs)")
    if nd == 3 and nod == 64:
        return (f"C3's grid with {m} points on each of {world} GPUs (weak scaling between config 3 and config 4; config 4 itself "
                f"is --gpus 8): " + base.format(pts=f"{world * m}"))
    return base.format(pts=f"{world * m} ({m} per GPU)")


def guarded(fn, *a, **k):
    """A side leg must never take the headline line with it: -> its result, or {"error": ...}."""
    try:
        return fn(*a, **k)
    except Exception as exc:
        return {"error": f"{type(exc).__name__}: {exc}"}


def cpu_baseline(ndim):
    """Reference (oracle/_ref) or port on a bounded sample, 1 core."""
    from oracle import binding
    from splpak_amd.synth import synth_points, synth_queries
    if binding.ref_available():
        impl, kind = binding.Reference(), "reference"
    else:
        if not binding.port_available():
            import subprocess
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "port"])
        impl, kind = binding.Port(), "port"
    nod, m = (12, 1500) if ndim == 3 else ((32, 8000) if ndim == 2 else (6, 3000))
    nodes = [nod] * ndim
    x, y, w = synth_points(ndim, m)
    lo, hi = [0.0] * ndim, [1.0] * ndim
    t0 = time.perf_counter()
    coef, ierr, _ = impl.fit(ndim, x, y, w, lo, hi, nodes, 1.0)
    t_fit = time.perf_counter() - t0
    nq = 200_000
    q = synth_queries(ndim, nq, m)
    t0 = time.perf_counter()
    impl.evaluate(ndim, q, None, coef, lo, hi, nodes)
    t_ev = time.perf_counter() - t0
    ncol = nod ** ndim
    # "best CPU": the same rows solved as banded normal equations + Cholesky + refinement on all host cores
    # (oracle/splpak_banded.c, pinned to the reference goldens) -- SURVEY 8d's honest comparator, bounded sample
    best = None
    try:
        P = binding.Port()
        nb, mb = (24, 100_000) if ndim == 3 else ((64, 200_000) if ndim == 2 else (8, 50_000))
        xb, yb, wb = synth_points(ndim, mb)
        t0 = time.perf_counter()
        _, eb, ib = P.fit_banded(ndim, xb, yb, wb, lo, hi, [nb] * ndim, 1.0)
        tb = time.perf_counter() - t0
        best = {"value": mb / tb, "unit": "points/s", "cores": int(ib[9]), "kind": "port (banded normal equations + Cholesky + refinement, OpenMP)",
                "sample": f"{ndim}-D splcw fit of {mb} weighted points on a {'x'.join([str(nb)] * ndim)} node grid, xtrap=1, {tb:.1f} s "
                          f"(rows+assembly {ib[5]:.1f} s, factorisation {ib[6]:.1f} s, solve+refine {ib[7]:.1f} s); its factorisation "
                          f"cost grows as ncol*halfbw^2, i.e. ~{(64 / nb) ** 7:.0f}x from this grid to 64^{ndim}", "ierror": int(eb)}
    except Exception as exc:
        best = {"error": f"{type(exc).__name__}: {exc}"}
    return {
        "best_cpu": best,
        "value": m / t_fit, "unit": "points/s", "cores": 1, "kind": kind,
        "sample": (f"{ndim}-D splcw fit of {m} weighted points of the same stream on a "
                   f"{'x'.join([str(nod)] * ndim)} node grid ({ncol} columns), xtrap=1, {t_fit:.1f} s; "
                   f"the dense reference algorithm needs ncol*(ncol+1) reals of workspace (550 GB at 64^3) "
                   f"and O(ncol^2) flop per row, so the full workload cannot run on it; its cost per "
                   f"point grows ~(262144/{ncol})^2 = {(262144 / ncol) ** 2:.0f}x from this sample to 64^3"),
        "ierror": int(ierr),
        "evals_per_s": nq / t_ev,
        "eval_sample": f"{nq} scalar splfe calls, {ndim}-D, same grid",
    }


def pin_to_gpu_numa_node(local_rank):
    """Run this process on the CPUs of the NUMA node the GPU hangs off (HIP events and completion
    signals live in host memory: a remote socket costs ~1 % of the fit).  Best effort."""
    try:
        bus = torch.cuda.get_device_properties(local_rank).pci_bus_id
        dom = getattr(torch.cuda.get_device_properties(local_rank), "pci_domain_id", 0)
        dev = getattr(torch.cuda.get_device_properties(local_rank), "pci_device_id", 0)
        path = f"/sys/bus/pci/devices/{dom:04x}:{bus:02x}:{dev:02x}.0/numa_node"
        node = int(open(path).read())
        if node < 0:
            return None
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if cpus:
            os.sched_setaffinity(0, cpus)
            return node
    except Exception:
        pass
    return None


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh child processes, one rank per GPU
    (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in their environment), BEFORE this process has touched the
    GPU -- nothing is ever exec'ed from a process that initialised HIP.  Returns the worst exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    return max(p.wait() for p in procs)


def bench_small(capi, dev, stream, steps, nd, nod, m, weighted, label):
    """A chain-bound (narrow-band) grid with resident data: BASELINE config 2 or the 32^3 grid VERDICT r01 names."""
    import torch
    nodes = [nod] * nd
    x = torch.empty((m, nd), dtype=torch.float64, device=dev)
    y = torch.empty(m, dtype=torch.float64, device=dev)
    w = torch.empty(m, dtype=torch.float64, device=dev) if weighted else None
    capi.synth_points_dev(nd, 0, m, x, y, w, stream)
    coef = torch.zeros(nod ** nd, dtype=torch.float64, device=dev)
    plan = capi.Plan(nd, nodes, [0.0] * nd, [1.0] * nd, 1.0, m)
    fact = plan.factorisation()[1]
    for _ in range(3):
        ierr, info = plan.fit(x, y, w, coef, stream)
        assert ierr == 0, f"{label} fit failed with ierror {ierr}"
    torch.cuda.synchronize()
    n = max(steps, 10)
    t0 = time.perf_counter()
    phase = np.zeros(3)
    for _ in range(n):
        ierr, info = plan.fit(x, y, w, coef, stream)
        phase += info[5:8]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    plan.close()
    assert ierr == 0 and info[9] < 1e-9, f"{label}: ierror {ierr}, optimality residual {info[9]:.2e}"
    return {"workload": label, "value": m / dt, "unit": "points/s", "ms_per_fit": 1e3 * dt, "fits_timed": n,
            "phase_ms": {"assembly": 1e3 * phase[0] / n, "factor": 1e3 * phase[1] / n, "solve_refine": 1e3 * phase[2] / n},
            "factorisation": fact,
            "refine_steps": int(info[2]), "optimality_residual": float(info[9])}


def bench_c5_fit(capi, dev, stream):
    """BASELINE config 5, fit half, AT ITS OWN SIZE on one GPU (round 6): 4-D, 32^4 = 1 048 576 columns, 1e7 weighted scattered
    points, xtrap = 1.  No factorisation of that grid fits the device (476 GB of nested-dissection panels, band 851 GB), so the
    plan takes the iterative solve by itself (csrc/pcg.hip: conjugate gradients on the rows, separable + block-Jacobi preconditioner,
    inside the same refinement against the rows).  The largest grid that still has a factorisation, 28^4, is timed beside it."""
    import torch
    nd, nod, m = 4, 32, 10_000_000
    nodes = [nod] * nd
    x = torch.empty((m, nd), dtype=torch.float64, device=dev)
    y = torch.empty(m, dtype=torch.float64, device=dev)
    w = torch.empty(m, dtype=torch.float64, device=dev)
    capi.synth_points_dev(nd, 0, m, x, y, w, stream)
    coef = torch.zeros(nod ** nd, dtype=torch.float64, device=dev)
    t0 = time.perf_counter()
    plan = capi.Plan(nd, nodes, [0.0] * nd, [1.0] * nd, 1.0, m)
    t_plan = time.perf_counter() - t0
    try:
        code, fact = plan.factorisation()
        plan_gb = plan.device_bytes() / 1e9
        ierr, info = plan.fit(x, y, w, coef, stream)                  # warm-up (first touch of the plan's buffers)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ierr, info = plan.fit(x, y, w, coef, stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ps = plan.pcg_stats()
    finally:
        plan.close()
    assert ierr == 0 and info[9] < 1e-9, f"config 5 fit: ierror {ierr}, optimality residual {info[9]:.2e}"
    del x, y, w, coef
    out = {"workload": "C5 (fit half) at its own size on ONE GPU: 4-D splcw fit, 1e7 weighted scattered points, 32^4 nodes (1048576 columns), "
                       "xtrap=1, real64, resident data",
           "value": m / dt, "unit": "points/s", "seconds_per_fit": dt, "plan_seconds": t_plan, "factorisation": fact, "solver_code": code,
           "plan_GB": plan_gb, "phase_seconds": {"assembly": float(info[5]), "factor": float(info[6]), "solve_refine": float(info[7])},
           "iterations": ps["iterations"], "solves": ps["solves"],
           "ms_per_iteration": 1e3 * float(info[7]) / max(ps["iterations"], 1),
           "refine_steps": int(info[2]), "optimality_residual": float(info[9]), "constraint_rows": int(info[1])}
    # roofline of an iteration: algorithmic bytes = one pass over the sorted points (4 coordinates + weight) + the vector in and out,
    # against the HBM peak; the counters' traffic of the same workload from the committed PMC passes (not measured in this run)
    alg = m * 8.0 * (nd + 1) + 2 * 8.0 * nod ** nd
    traffic = None
    try:
        pm = json.load(open(committed_profile("r06_c5_pcg_pmc.json")))
        traffic = {"hbm_bytes_per_iteration": pm["hbm_bytes_per_iteration_all_kernels"], "source": "profiles/r06_c5_pcg_pmc.json (rocprofv3 --pmc "
                   "FETCH_SIZE / WRITE_SIZE passes of the same workload; not measured in this run)"}
    except Exception:
        pass
    ms_it = out["ms_per_iteration"]
    out["roofline"] = {"bound": "hbm", "kernel": "one iteration of the solve (rows4_tile_kernel = half of it: issue bound -- window tables + f64 matrix-pipe products --, not HBM bound)",
                       "achieved": alg / (ms_it * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (ms_it * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "algorithmic_bytes_per_iteration": alg, "traffic": traffic}
    torch.cuda.empty_cache()
    out["largest_grid_with_a_factorisation"] = guarded(bench_c5_fit_nd28, capi, dev, stream)
    return out


def bench_c5_fit_nd28(capi, dev, stream):
    """Config 5's points on the largest 4-D grid whose nested-dissection factorisation fits ONE MI355X (288 GiB =
    309 GB of HBM): since round 5 that is 28^4 nodes (614 656 columns, 1.1e15 flop: 205 GB of factor panels + 63 GB of Schur
    arena in the postorder schedule with packed buffers; the level-by-level order of rounds 3-4 held 24^4), 1e7 points of the
    seeded stream.  Config 5's own 32^4 grid needs 476 GB of factor panels + 131 GB of arena: the plan is refused on one GPU
    (and the boxes' 322 GB host-memory cgroup rules out parking the difference in host memory); it is the 8-GPU route's."""
    import torch
    nd, nod, m = 4, 28, 10_000_000
    nodes = [nod] * nd
    free, total = torch.cuda.mem_get_info()
    if free < 285e9:                                                   # (another process on the device: the grid of rounds 3-4)
        nod, nodes = 24, [24] * nd
    x = torch.empty((m, nd), dtype=torch.float64, device=dev)
    y = torch.empty(m, dtype=torch.float64, device=dev)
    w = torch.empty(m, dtype=torch.float64, device=dev)
    capi.synth_points_dev(nd, 0, m, x, y, w, stream)
    coef = torch.zeros(nod ** nd, dtype=torch.float64, device=dev)
    t0 = time.perf_counter()
    os.environ["SPLPAK_SOLVER"] = "direct"                            # (left to itself a grid of this size tries the iteration first)
    try:
        plan = capi.Plan(nd, nodes, [0.0] * nd, [1.0] * nd, 1.0, m)
    finally:
        os.environ.pop("SPLPAK_SOLVER", None)
    t_plan = time.perf_counter() - t0
    try:
        fact = plan.factorisation()[1]
        plan_gb = plan.device_bytes() / 1e9
        torch.cuda.synchronize()
        t0 = time.perf_counter()                                       # ONE fit, no warm-up: the first and the second fit of a plan take
        ierr, info = plan.fit(x, y, w, coef, stream)                  # the same 18.4 s (tools/c5_fit.py), and the leg has its own time budget
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        plan.close()
    assert ierr == 0 and info[9] < 1e-9, f"4-D fit: ierror {ierr}, optimality residual {info[9]:.2e}"
    tree = capi.debug_nd_tree(nodes, check=False)
    big = capi.debug_nd_tree([32] * 4, check=False)
    big_arena = min(capi.debug_nd_schedule([32] * 4, cut=c)["arena_bytes"] for c in range(0, 5))
    ranks8, summ8 = capi.debug_nd_partition([32] * 4, 8)
    return {"workload": f"C5's points on the largest 4-D grid with a factorisation on one GPU: 4-D splcw fit, 1e7 weighted scattered points, {nod}^4 nodes "
                        f"({nod ** 4} columns), xtrap=1, real64, resident data",
            "value": m / dt, "unit": "points/s", "seconds_per_fit": dt, "plan_seconds": t_plan, "factorisation": fact, "plan_GB": plan_gb,
            "device_memory_GB": total / 1e9,
            "phase_seconds": {"assembly": float(info[5]), "factor": float(info[6]), "solve_refine": float(info[7])},
            "factor_tflops": tree["flop"] / max(float(info[6]), 1e-9) / 1e12,
            "factor_frac_of_f64_mfma_peak": tree["flop"] / max(float(info[6]), 1e-9) / 1e12 / 78.6,
            "refine_steps": int(info[2]), "optimality_residual": float(info[9]), "constraint_rows": int(info[1]),
            "fronts": int(tree["fronts"]), "factor_GB": tree["factor_bytes"] / 1e9, "schur_arenas_GB_level_order": tree["arena_bytes"] / 1e9,
            "config5_32^4_needs": {"factor_GB": big["factor_bytes"] / 1e9, "schur_arena_GB_postorder_packed": big_arena / 1e9,
                                   "schur_arenas_GB_level_order": big["arena_bytes"] / 1e9, "flop": big["flop"],
                                   "note": "refused on one GPU (SPLPAK_E_NOMEM): 476 GB of panels + 131 GB of arena against 309 GB of HBM; an "
                                           "out-of-core form would have to park >= 300 GB in host memory and the GPU boxes of this pool give a job "
                                           "322 GB (cgroup memory.max; tools/host_probe.py, DESIGN 4a)",
                                   "on_8_gpus_one_process": {
                                       "factorisation_GB_per_gpu": [round(r["bytes"] / 1e9, 1) for r in ranks8],
                                       "normal_equations_GB_per_gpu": round(summ8["normal_eq_bytes"] / 1e9, 1),
                                       "flop_per_gpu": [r["flop_subtrees"] + r["flop_top"] for r in ranks8],
                                       "note": "splpak_mplan_* distribute the nested-dissection factorisation: a subtree per GPU, the 7 fronts above "
                                               "them by block columns (host-only partition, splpak_debug_nd_partition; this pool has one GPU per box: "
                                               "rehearsed at 24^4 on 4 virtual GPUs in tests/test_dist.py, never run at 32^4)"}}}


def bench_c2(capi, dev, stream, steps):
    """BASELINE config 2 (2-D, 1e6 scattered points, 64x64 nodes, equal weights = splcc), resident data."""
    return bench_small(capi, dev, stream, steps, 2, 64, 1_000_000, False,
                       "C2: 2-D splcc fit, 1e6 scattered points (same stream), 64x64 nodes, xtrap=1, real64, resident data")


def bench_dist_band(capi, ngpus, nd, nod, m_total, virtual, steps):
    """The fit with the FACTORISATION DISTRIBUTED over `ngpus` GPUs, driven from this one process (splpak_mplan_*, what a
    Fortran caller reaches through set_gpus): the points sharded, the nested-dissection factorisation distributed -- a subtree
    per GPU, the fronts above them by block columns with peer-copied panels (round 4; grids below the nested-dissection
    threshold: the distributed band of round 2).  m_total points in all.  virtual=True places every rank on this GPU (a
    rehearsal of the protocol, not a speed-up)."""
    import torch
    nodes, lo, hi = [nod] * nd, [0.0] * nd, [1.0] * nd
    if virtual:
        os.environ["SPLPAK_VIRTUAL_GPUS"] = "1"
    try:
        per = (m_total + ngpus - 1) // ngpus
        mp = capi.MultiPlan(ngpus, nd, nodes, lo, hi, 1.0, per)
        xs, ys, ws = [], [], []
        cur = torch.cuda.current_device()
        for r in range(ngpus):
            n = max(0, min(per, m_total - r * per))
            d = torch.device("cuda", mp.device(r))
            torch.cuda.set_device(d)
            x = torch.empty((n, nd), dtype=torch.float64, device=d)
            y = torch.empty(n, dtype=torch.float64, device=d)
            w = torch.empty(n, dtype=torch.float64, device=d)
            capi.synth_points_dev(nd, r * per, n, x, y, w, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            xs.append(x); ys.append(y); ws.append(w)
        torch.cuda.set_device(cur)
        coef = torch.zeros(nod ** nd, dtype=torch.float64, device=torch.device("cuda", mp.device(0)))
        code, what = mp.factorisation()
        rank_gb = [round(mp.rank_bytes(r) / 1e9, 2) for r in range(ngpus)]
        ierr, info = mp.fit(xs, ys, ws, coef)                      # warm-up
        assert ierr == 0, f"distributed fit failed with ierror {ierr}"
        n = max(1, min(steps, 3))
        t0 = time.perf_counter()
        for _ in range(n):
            ierr, info = mp.fit(xs, ys, ws, coef)
        dt = (time.perf_counter() - t0) / n
        mp.close()
        assert ierr == 0 and info[9] < 1e-9, f"distributed fit: ierror {ierr}, optimality residual {info[9]:.2e}"
        return {"workload": f"{nd}-D splcw fit, {m_total} points in all, {nod}^{nd} nodes, factorisation distributed over "
                            f"{ngpus} {'virtual GPUs (all ranks on this device: protocol rehearsal)' if virtual else 'GPUs, one process, peer copies over xGMI'}",
                "factorisation": {"code": code, "what": what}, "device_gb_per_rank": rank_gb,
                "value": m_total / dt, "unit": "points/s", "ms_per_fit": 1e3 * dt, "n_gpus": ngpus, "virtual": bool(virtual),
                "phase_seconds": {"assembly": float(info[5]), "factor": float(info[6]), "solve_refine": float(info[7])},
                "refine_steps": int(info[2]), "optimality_residual": float(info[9])}
    except Exception as exc:      # the headline line must survive a failure of this leg
        return {"error": f"{type(exc).__name__}: {exc}"}
    finally:
        if virtual:
            os.environ.pop("SPLPAK_VIRTUAL_GPUS", None)


def c5_traffic():
    """Fabric bytes of the 4-D evaluation passes from the committed PMC profile (not measured in this run)."""
    try:
        src = committed_profile("r06_eval_pmc.json", "r05_eval_pmc.json", "r04_eval_pmc.json")
        pm = json.load(open(src))["4d_32"]
        kname, kv = next((k, v) for k, v in pm["kernels"].items() if k.startswith(("pr_eval_kernel<4", "eval_binned_kernel<4, true")))
        return {"kernel": kname, "bytes_per_launch": kv["hbm_bytes"],
                "queries_per_launch": pm["queries_per_launch"], "bytes_per_query_all_passes": pm["hbm_bytes_per_query_all_passes"],
                "algorithmic_bytes_per_query": 40.0,
                "source": os.path.relpath(src, ROOT) + " (rocprofv3 --pmc passes of tools/eval_profile.py 4 32 100000000; not measured in this run)"}
    except Exception:
        return None


def bench_c5_eval(capi, dev, stream):
    """BASELINE config 5, evaluation half at full size: 4-D 32^4 coefficients (8 MB), 1e8 queries of the
    seeded stream, splfe and two splde derivative patterns; real64 resident data.  (The fit half of
    config 5: bench_c5_fit -- 32^4 on one GPU by the iterative solve, 28^4 by nested dissection beside it.)"""
    import torch
    nd, nod, nq = 4, 32, 100_000_000
    nodes, lo, hi = [nod] * nd, [0.0] * nd, [1.0] * nd
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    coef = torch.randn(nod ** nd, dtype=torch.float64, device=dev, generator=gen)
    xq = torch.empty((nq, nd), dtype=torch.float64, device=dev)
    out = torch.empty(nq, dtype=torch.float64, device=dev)
    capi.synth_queries_dev(nd, 10_000_000, 0, nq, xq, stream)
    res = {}
    for label, pat in (("splfe", None), ("splde_1000", [1, 0, 0, 0]), ("splde_0201", [0, 2, 0, 1])):
        capi.evaluate_dev(nd, xq, pat, coef, lo, hi, nodes, out, stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        capi.evaluate_dev(nd, xq, pat, coef, lo, hi, nodes, out, stream)
        torch.cuda.synchronize()
        res[label] = nq / (time.perf_counter() - t0)
    # the tolerance sweep's other half: the same 1e8 queries in REAL32 storage (float queries, coefficients, results; double
    # arithmetic), and how far its values are from the real64 ones
    capi.evaluate_dev(nd, xq, None, coef, lo, hi, nodes, out, stream)
    x32, c32 = xq.float(), coef.float()
    o32 = torch.empty(nq, dtype=torch.float32, device=dev)
    capi.evaluate_dev(nd, x32, None, c32, lo, hi, nodes, o32, stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    capi.evaluate_dev(nd, x32, None, c32, lo, hi, nodes, o32, stream)
    torch.cuda.synchronize()
    res32 = nq / (time.perf_counter() - t0)
    dev32 = float((o32[:10_000_000].double() - out[:10_000_000]).abs().max() / out[:10_000_000].abs().max())
    del xq, out, x32, c32, o32
    return {"workload": "C5 (evaluation half): 4-D, 32^4 nodes, 1e8 queries, real64, resident data", "unit": "evals/s",
            "value": res["splfe"], **{k + "_evals_per_s": v for k, v in res.items()},
            "real32": {"splfe_evals_per_s": res32, "max_rel_deviation_from_real64": dev32,
                       "roofline": {"bound": "hbm", "achieved": 20.0 * res32 / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": 20.0 * res32 / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_query": 20.0}},
            "roofline": {"bound": "hbm", "achieved": 40.0 * res["splfe"] / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": 40.0 * res["splfe"] / 1e9 / HBM_PEAK_GBS, "traffic": c5_traffic()}}


def dist_band_in_child(ngpus, nd, nod, m_total, virtual, steps, timeout):
    """Run bench_dist_band in a CHILD process with a time limit: the distributed-band leg drives several GPUs
    from one process, and nothing it does -- a failure, a hang on an untested interconnect -- may take the
    headline line with it.  (A child is started, nothing is exec'ed over this process.)"""
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE",
              "TORCHELASTIC_RUN_ID", "ROLE_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.abspath(__file__), "--ndim", str(nd), "--steps", str(steps),
           "--dist-child", f"{ngpus},{nod},{m_total},{int(bool(virtual))}"]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
        for ln in reversed(r.stdout.strip().splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)
        return {"error": f"distributed-band child exited with {r.returncode}: {(r.stderr or r.stdout)[-300:]}"}
    except subprocess.TimeoutExpired:
        return {"error": f"distributed-band child exceeded {timeout} s"}
    except Exception as exc:
        return {"error": f"{type(exc).__name__}: {exc}"}


def bench_incl_h2d(capi, x, y, w, lo, hi, nodes):
    """The host-pointer entry (what the Fortran module binds): pageable host arrays in, coefficients
    out, PCIe transfers included.  Never `value`."""
    xh, yh, wh = x.cpu().numpy(), y.cpu().numpy(), w.cpu().numpy()
    nd = xh.shape[1]
    capi.fit(nd, xh, yh, wh, lo, hi, nodes, 1.0)              # first call: plan allocation (cached afterwards)
    t0 = time.perf_counter()
    coef, ierr, _, info = capi.fit(nd, xh, yh, wh, lo, hi, nodes, 1.0)
    dt = time.perf_counter() - t0
    assert ierr == 0
    return {"value": xh.shape[0] / dt, "unit": "points/s", "seconds": dt,
            "what": "splpak_fit_f64 on host arrays: H2D of xdata/ydata/wdata + fit + D2H of coef, second call of the process"}


def main():
    args = parse()
    if os.environ.get("SPLPAK_BENCH_WATCHDOG"):       # a hang ends with the Python stacks of every thread instead of silence
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["SPLPAK_BENCH_WATCHDOG"]), exit=True)
    if args.dist_child:
        from splpak_amd import capi
        ng, nod_c, m_c, virt = (int(v) for v in args.dist_child.split(","))
        print(json.dumps(bench_dist_band(capi, ng, args.ndim, nod_c, m_c, bool(virt), args.steps)), flush=True)
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the splpak HIP path has no CPU fallback")
    # rehearsal aids for a one-GPU box: all ranks on device 0 and/or a gloo process group
    if os.environ.get("SPLPAK_BENCH_SINGLE_DEVICE"):
        local_rank = 0
    numa_node = pin_to_gpu_numa_node(local_rank)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SPLPAK_BENCH_BACKEND", "nccl")          # nccl = RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from splpak_amd import capi

    nd, nod = args.ndim, args.nodes
    m = args.ndata if args.ndata is not None else default_ndata(world, nd, nod)
    nodes = [nod] * nd
    lo, hi = [0.0] * nd, [1.0] * nd
    ncol = nod ** nd
    stream = torch.cuda.current_stream().cuda_stream

    # ---- synthetic inputs, generated on the device, resident in HBM -----------
    x = torch.empty((m, nd), dtype=torch.float64, device=dev)
    y = torch.empty(m, dtype=torch.float64, device=dev)
    w = torch.empty(m, dtype=torch.float64, device=dev)
    capi.synth_points_dev(nd, rank * m, m, x, y, w, stream)
    coef = torch.zeros(ncol, dtype=torch.float64, device=dev)

    from splpak_amd.dist import ShardedFit
    sharded = ShardedFit(nd, nodes, lo, hi, 1.0, m, dev, dist if world > 1 else None)
    plan = sharded.plan
    plan.enable_kernel_timing(not args.no_kernel_timing)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    info = None
    for _ in range(args.warmup):
        ierr, info = plan.fit(x, y, w, coef, stream)
        assert ierr == 0, f"fit failed with ierror {ierr}"
    barrier()
    t0 = time.perf_counter()
    kt_sum = dict(syrk_launches=0.0, syrk_ms=0.0, syrk_flop=0.0, factor_ms=0.0, total_flop=0.0,
                  bulk_launches=0.0, bulk_flop=0.0)
    phase = np.zeros(3)
    for _ in range(args.steps):
        ierr, info = plan.fit(x, y, w, coef, stream)
        kt = plan.kernel_timing()
        for k in kt_sum:
            kt_sum[k] += kt[k]
        phase += info[5:8]
    barrier()
    elapsed = time.perf_counter() - t0
    assert ierr == 0, f"fit failed with ierror {ierr}"
    stages = plan.stage_timing()
    # parity gate at the size that is timed: the MEASURED optimality residual of the returned coefficients
    # (gradient of the least-squares functional, relative to |A^T W^2 y|) must be at rounding level
    assert info[9] < 1e-9, f"optimality residual {info[9]:.2e} of the timed fit exceeds 1e-9"
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    value = world * m * args.steps / elapsed
    fact_code, fact_name = plan.factorisation()
    # the roofline kernel with the chip to itself: one extra fit OUTSIDE the timed region with every launch of the
    # factorisation on one stream (the plan's option no_lookahead), so that no chain kernel shares the CUs with it
    kt_alone = None
    if world == 1 and not args.no_kernel_timing:
        plan.set_option("no_lookahead", "1")                   # (a per-fit option of the plan: round 6, no getenv on the fit path)
        try:
            ierr_a, _ = plan.fit(x, y, w, coef, stream)
            kt_alone = plan.kernel_timing() if ierr_a == 0 else None
        finally:
            plan.set_option("no_lookahead", None)
        ierr, info = plan.fit(x, y, w, coef, stream)          # (the plan's streams and the reported diagnostics: back to the timed form)

    # strong scaling beside the weak headline (N > 1): config 3's 1e7 points IN ALL, sharded over the ranks
    strong = None
    if world > 1:
        ms_tot = 10_000_000
        from splpak_amd.dist import shard_range
        first, cnt = shard_range(ms_tot, rank, world)
        if cnt <= m:
            capi.synth_points_dev(nd, first, cnt, x[:cnt], y[:cnt], w[:cnt], stream)
            ierr_s, _ = plan.fit(x[:cnt], y[:cnt], w[:cnt], coef, stream)
            barrier()
            ts = time.perf_counter()
            for _ in range(args.steps):
                ierr_s, info_s = plan.fit(x[:cnt], y[:cnt], w[:cnt], coef, stream)
            barrier()
            el = time.perf_counter() - ts
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            strong = {"workload": f"C3's 1e7 points in all, sharded over {world} GPUs (fixed total)", "scaling": "strong",
                      "value": ms_tot * args.steps / float(t.item()), "unit": "points/s",
                      "ms_per_step": 1e3 * float(t.item()) / args.steps, "ierror": int(ierr_s)}
            capi.synth_points_dev(nd, rank * m, m, x, y, w, stream)      # back to the weak-scaling shard
            ierr, info = plan.fit(x, y, w, coef, stream)                 # coefficients of the headline workload for the evaluation leg

    # ---- evaluation throughput (splfe), queries sharded, no collectives ---------
    nq = args.neval
    xq = torch.empty((nq, nd), dtype=torch.float64, device=dev)
    out = torch.empty(nq, dtype=torch.float64, device=dev)
    capi.synth_queries_dev(nd, world * m, rank * nq, nq, xq, stream)
    def time_eval(mode):
        capi.set_eval_mode(mode)
        capi.evaluate_dev(nd, xq, None, coef, lo, hi, nodes, out, stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        barrier()
        e0.record()
        for _ in range(reps):
            capi.evaluate_dev(nd, xq, None, coef, lo, hi, nodes, out, stream)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        if world > 1:
            t = torch.tensor([ms], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ms = float(t.item())
        return ms

    ev_direct_ms = time_eval(capi.EVAL_DIRECT)     # one thread per query, global (L2) gathers
    ev_ms = time_eval(capi.EVAL_AUTO)              # library default: LDS-binned for batches like this one
    evals_per_s = world * nq / (ev_ms * 1e-3)
    ev_bytes = 8.0 * (nd + 1) * nq

    # fabric bytes of the evaluation passes: PMC passes of an earlier run of the same workload, kept under profiles/ --
    # NOT measured in this run
    eval_traffic = None
    eval_pmc = committed_profile("r06_eval_pmc.json", "r05_eval_pmc.json", "r04_eval_pmc.json")
    try:
        pm = json.load(open(eval_pmc))["3d_64"]
        eval_traffic = {"kernel": "pr_eval_kernel<3,16,true,double> (evaluation pass of the persistent region path; all three passes in bytes_per_query_all_passes)", "bytes_per_launch": next(v for k, v in pm["kernels"].items() if k.startswith(("pr_eval_kernel<3", "eval_runs_kernel<3, true", "eval_binned_kernel<3, true")))["hbm_bytes"],
                        "queries_per_launch": pm["queries_per_launch"],
                        "bytes_per_query_all_passes": pm["hbm_bytes_per_query_all_passes"], "algorithmic_bytes_per_query": 8.0 * (nd + 1),
                        "source": os.path.relpath(eval_pmc, ROOT) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same workload; not measured in this run)"}
    except Exception:
        pass
    dist_leg = None
    if world > 1 and not args.no_side_legs:
        # strong scaling with a DISTRIBUTED factorisation: rank 0's process drives all `world` GPUs of the node
        # (native peer copies, no Python in the loop); the other ranks free their plans and wait
        if rank != 0:
            plan.close()
            torch.cuda.empty_cache()
        barrier()
        # the other ranks wait on the rendezvous store, not in a collective: an RCCL barrier would keep a kernel
        # spinning on every GPU the child is about to drive
        from datetime import timedelta
        store = dist.distributed_c10d._get_default_store()
        if rank == 0:
            try:
                # the headline's workload (every rank's m points: m * world in all) through the one-process route
                dist_leg = dist_band_in_child(world, nd, nod, m * world, bool(os.environ.get("SPLPAK_BENCH_SINGLE_DEVICE")),
                                              args.steps, 150)
            finally:
                store.set("splpak_dist_leg_done", "1")
        else:
            store.wait(["splpak_dist_leg_done"], timedelta(seconds=400))
        barrier()
    if rank == 0:
        line = {
            "metric": (f"fitted points/sec (splcw) + evals/sec (splfe), {nd}-D {m:.0e} pts {nod}^{nd} nodes"
                       .replace("e+0", "e").replace("e+", "e")),
            # N > 1 (round-5 verdict): the headline is the STRONG-scaling number -- BASELINE's own workload, 1e7 points in all,
            # sharded over the ranks; the factorisation (most of a fit) does not depend on the point count, so the weak number
            # (1e7 points PER GPU, kept as `weak` below) rises ~N-fold by construction whatever the GPUs do
            "value": strong["value"] if strong else value,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": strong["ms_per_step"] if strong else 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": workload_label(world, nd, nod, m),
                "ncol": ncol, "points_per_gpu": m, "points_total": world * m, "host_numa_node": numa_node,
                "parallelism": ("points sharded per GPU; RCCL all-reduce of histogram, normal equations and refinement residuals; "
                                + ("factorisation replicated" if (fact_code != 4 or os.environ.get("SPLPAK_ND_DIST") == "0") else
                                   "nested-dissection factorisation distributed by subtrees (a rank eliminates its own subtrees below "
                                   "tree depth ceil(log2 ranks), their Schur complements are all-reduced, the top of the tree is "
                                   "factored by every rank; the tree solves follow the same split)")) if world > 1 else "single GPU",
                "factorisation": fact_name,
                # "rccl-native": the library's own hook (ncclAllReduce on the fit's stream); "torch.distributed:<backend>": the callback
                "collective_backend": (sharded.collective if world > 1 else None),
                "native_hook_error": (sharded.native_error if world > 1 else None),
                "rccl_ranks": (dist.get_world_size() if world > 1 and dist.get_backend() == "nccl" else 0),
                "refine_steps": int(info[2]), "last_correction_rel": float(info[3]),
                "optimality_residual": float(info[9]), "residual_norm": float(info[8]),
                "data_rows": float(info[0]), "constraint_rows": float(info[1]),
                "phase_seconds_per_step": {"assembly": phase[0] / args.steps, "factor": phase[1] / args.steps,
                                           "solve_refine": phase[2] / args.steps},
            },
            "evals_per_s": evals_per_s,
            "eval": {
                "value": evals_per_s, "unit": "evals/s", "queries_per_gpu": nq, "ms_per_batch": ev_ms,
                "path": "auto (3-D: persistent region path -- place pass, persistent per-region workers with the tile in LDS, unsort pass)",
                "direct_path_evals_per_s": world * nq / (ev_direct_ms * 1e-3),
                "roofline": {"bound": "hbm", "achieved": ev_bytes / (ev_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": ev_bytes / (ev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             # scalars (the driver's parser keeps scalars only): HBM bytes per query over ALL passes, by the counters
                             "traffic": (eval_traffic or {}).get("bytes_per_query_all_passes"),
                             "algorithmic_bytes_per_query": 8.0 * (nd + 1),
                             "traffic_source": (eval_traffic or {}).get("source"),
                             "traffic_detail": eval_traffic},
            },
            # the evaluation half of the headline metric as scalars of the line itself
            "eval_roofline_frac": ev_bytes / (ev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "eval_hbm_bytes_per_query": (eval_traffic or {}).get("bytes_per_query_all_passes"),
        }
        if stages["bin_ms"] > 0:
            # the HBM-bound stages around the factorisation: algorithmic bytes (8*(d+1+[weighted]) per point
            # and streaming pass) over the HIP-event time of the stage in the last timed fit
            bpp = 8.0 * (nd + 2)
            hst = (7 ** nd + 1) // 2                                  # stored stencil entries per row
            halfbw = 3 * sum(nod ** k for k in range(nd))
            ldband = (min(-(-halfbw // 256), max(-(-ncol // 256) - 1, 0)) + 1) * 256 + 17   # column stride of the band storage
            def hbm(nbytes, ms):
                gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
                return {"ms": ms, "algorithmic_GB": nbytes / 1e9, "achieved_GBs": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS}
            line["assembly"] = {
                "binning (stable partition: count, prefix, scatter, second level)": hbm(m * (2 * bpp + bpp + 4.0), stages["bin_ms"]),
                "gram blocks + stencil gather": hbm(m * bpp + 8.0 * ncol * hst, stages["gram_ms"]),
                "constraint rows": {"ms": stages["constraints_ms"]},
                "refinement residual pass": hbm(m * bpp, stages["residual_pass_ms"]),
                "note": "the Gram stage moves 3.9 GB of per-cell blocks through HBM on top of its algorithmic bytes (PMC: profiles/r06_fit_pmc.json)"
                        if nd == 3 and nod == 64 else "",
            }
            if fact_code == 4:
                fbytes = capi.debug_nd_tree(nodes, check=False)["factor_bytes"]
                # (round 5: the panels are written stage by stage from the half stencil -- zeros and entries, nd_init_kernel; the
                #  stamped interval holds the stages that are alive when the factorisation starts, the others are written inside
                #  the factor phase on the update stream.  SPLPAK_ND_STAGED_INIT=0 / the distributed forms: clear all, then scatter)
                line["assembly"]["panels of the first stages (zeros + entries of the half stencil)"] = {
                    "ms": stages["expand_ms"], "stencil_GB": 8.0 * ncol * hst / 1e9, "arena_GB_written_stage_by_stage": fbytes / 1e9}
                line["assembly"]["one solve (forward + backward tree sweep)"] = hbm(2 * fbytes, stages["solve_ms"])
            else:
                line["assembly"]["band memset + expansion"] = hbm(8.0 * ncol * hst + 8.0 * ncol * ldband, stages["expand_ms"])
                line["assembly"]["one solve (two band sweeps)"] = hbm(2 * 8.0 * ncol * ldband, stages["solve_ms"])
        if kt_sum["syrk_ms"] > 0:
            ach = kt_sum["syrk_flop"] / (kt_sum["syrk_ms"] * 1e-3) / 1e12
            nd_path = fact_code == 4
            # fabric bytes per launch of the roofline kernel: PMC passes of an earlier run of the same workload, kept
            # under profiles/ -- NOT measured in this run (counters and kernel timing do not share a run)
            traffic = None
            pmc = committed_profile("r06_fit_pmc.json", "r05_fit_pmc.json", "r04_fit_pmc.json") if nd_path else committed_profile("r02_fit_pmc.json")
            if nd == 3 and nod == 64 and pmc:
                try:
                    pj = json.load(open(pmc))
                    traffic = {"hbm_bytes_per_launch": pj.get("schur_hbm_bytes_per_launch" if nd_path else "bulk_hbm_bytes_per_launch"),
                               "algorithmic_bytes_per_launch": pj.get("schur_algorithmic_bytes_per_launch"),
                               "source": os.path.relpath(pmc, ROOT) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same "
                                         "workload; not measured in this run)"}
                except Exception:
                    traffic = None
            line["roofline"] = {
                "kernel": ("nd_syrk_kernel<4,2,true,1,1> (Schur-buffer passes S -= L21 L21^T of the nested-dissection fronts, K = 1024 per pass, "
                           "v_mfma_f64_16x16x4_f64; one launch at a time per tree depth and block group)") if nd_path else
                          "syrk64_kernel<16,1,4,256> (bulk trailing update C -= P P^T of the band Cholesky, v_mfma_f64_16x16x4_f64; one launch per block step)",
                "bound": "mfma", "achieved": ach, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / F64_MFMA_PEAK_TFLOPS,
                # scalars (the driver's parser keeps scalars only): HBM / fabric bytes per launch by the PMC counters of a
                # separate run of the same workload, beside the algorithmic bytes of the tiles
                "traffic": (traffic or {}).get("hbm_bytes_per_launch"),
                "algorithmic_bytes_per_launch": (traffic or {}).get("algorithmic_bytes_per_launch"),
                "traffic_source": (traffic or {}).get("source"),
                "timed_launches": kt_sum["syrk_launches"], "launches": kt_sum["bulk_launches"],
                "timing": "HIP start/stop event pair carried by every launch of this kernel inside the timed region (hipExtLaunchKernelGGL, on the launch stream)",
                "avg_launch_ms": kt_sum["syrk_ms"] / max(kt_sum["syrk_launches"], 1),
                "flop_per_launch": kt_sum["syrk_flop"] / max(kt_sum["syrk_launches"], 1),
                "factor_ms_per_step": kt_sum["factor_ms"] / args.steps,
                "flop_share_of_factorisation": kt_sum["bulk_flop"] / max(kt_sum["total_flop"], 1.0),
                "factorisation_tflops": kt_sum["total_flop"] / max(kt_sum["factor_ms"], 1e-9) / 1e9,
                "factorisation_flop": kt_sum["total_flop"] / args.steps,
            }
            if kt_alone is not None and kt_alone["syrk_ms"] > 0:
                al = kt_alone["syrk_flop"] / (kt_alone["syrk_ms"] * 1e-3) / 1e12
                line["roofline"]["kernel_alone_tflops"] = al
                line["roofline"]["kernel_alone_frac"] = al / F64_MFMA_PEAK_TFLOPS
                line["roofline"]["kernel_alone_factor_ms"] = kt_alone["factor_ms"]
                line["roofline"]["kernel_alone_what"] = ("the same launches in one extra fit outside the timed region, whole factorisation on "
                                                         "ONE stream (SPLPAK_NO_LOOKAHEAD=1): every launch has the chip to itself")
            # the evaluation half of the headline metric, as scalars of THIS object (the driver's record keeps the scalars of
            # `roofline` and `config`; top-level extras only by name -- VERDICT r04)
            line["roofline"]["evals_per_s"] = evals_per_s
            line["roofline"]["eval_frac"] = ev_bytes / (ev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            line["roofline"]["eval_ms_per_batch"] = ev_ms
            line["roofline"]["eval_queries_per_batch"] = nq
            line["roofline"]["eval_bytes_per_query"] = (eval_traffic or {}).get("bytes_per_query_all_passes")
            line["roofline"]["eval_algorithmic_bytes_per_query"] = 8.0 * (nd + 1)
            line["roofline"]["eval_direct_kernel_evals_per_s"] = world * nq / (ev_direct_ms * 1e-3)
            line["roofline"]["eval_direct_kernel_frac"] = ev_bytes / (ev_direct_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            if nd_path:
                line["roofline"]["note"] = ("the panel updates of the chain (nd_syrk_kernel<4,2,false>, K = 256) and the diagonal-block / panel-solve "
                                            "kernels run beside these launches on other streams and share the CUs with them; "
                                            "the band factorisation of round 2 needed 4.1e13 flop for this grid")
        if strong is not None:
            line["strong"] = strong
            line["weak"] = {"workload": workload_label(world, nd, nod, m), "scaling": "weak", "value": value, "unit": "points/s",
                            "ms_per_step": 1e3 * elapsed / args.steps, "points_per_gpu": m,
                            "note": "the timed K steps of the contract ran on this workload too; the factorisation does not depend on the point count"}
            line["config"]["workload"] = strong["workload"] + "; 3-D, 64^3 nodes, xtrap=1, real64 (BASELINE config 3's total)"
            line["config"]["points_total"] = 10_000_000
        if world == 1 and not args.no_side_legs:
            plan.close()                      # the side legs have the GPU to themselves
            # every side leg is guarded: a failure becomes {"error": ...} inside the line, the headline survives
            # rehearsal of the one-process multi-GPU route on this one GPU: the headline workload on 2 virtual ranks
            # (nested dissection distributed: a subtree per rank, the root by block columns)
            line["multi_gpu_one_process"] = guarded(dist_band_in_child, 2, nd, nod, m, True, args.steps, 180)
            line["c2"] = guarded(bench_c2, capi, dev, stream, args.steps)
            line["grid32"] = guarded(bench_small, capi, dev, stream, args.steps, 3, 32, 1_000_000, True,
                                     "3-D splcw fit, 1e6 weighted scattered points, 32x32x32 nodes, xtrap=1, real64, resident data")
            del xq, out
            torch.cuda.empty_cache()
            line["fit_incl_h2d"] = guarded(bench_incl_h2d, capi, x, y, w, lo, hi, nodes)
            del x, y, w
            capi.shutdown()                   # the one-shot entry's cached plan (35 GB): the 4-D legs need the room
            torch.cuda.empty_cache()
            line["c5_eval"] = guarded(bench_c5_eval, capi, dev, stream)
            torch.cuda.empty_cache()
            line["c5_fit"] = guarded(bench_c5_fit, capi, dev, stream)
            if "roofline" in line:
                c5e, c5f = line.get("c5_eval") or {}, line.get("c5_fit") or {}
                line["roofline"]["eval4d_evals_per_s"] = c5e.get("value")
                line["roofline"]["eval4d_frac"] = (c5e.get("roofline") or {}).get("frac")
                line["roofline"]["c5_fit_points_per_s"] = c5f.get("value")
                line["roofline"]["c5_fit_seconds"] = c5f.get("seconds_per_fit")
                line["roofline"]["c5_fit_iterations"] = c5f.get("iterations")
                line["roofline"]["c5_fit_factor_frac"] = (c5f.get("largest_grid_with_a_factorisation") or {}).get("factor_frac_of_f64_mfma_peak")
                c2l = line.get("c2") or {}
                line["roofline"]["c2_ms_per_fit"] = c2l.get("ms_per_fit")
        if dist_leg is not None:
            line["multi_gpu_one_process"] = dist_leg
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = guarded(cpu_baseline, nd)
        print(json.dumps(line), flush=True)

    sharded.close()                       # (the plan, and the native RCCL communicator if one was made)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
