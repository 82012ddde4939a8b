"""Predicted per-rank cost of BASELINE config 3 (and 4) on 1, 2, 4, 8 MI355X -- from the host-only partition of the nested-dissection
tree (splpak_debug_nd_partition) and the rates MEASURED on one GPU, so that the first real SCALE run has something to be compared
with (round-5 verdict, item 6c).  No GPU needed.      python tools/scale_prediction.py > profiles/r06_scale_prediction.json

Model (every number's source is in the output):
  assembly      points of the rank / measured points rate of binning + Gram + gather on one GPU
  factorisation route (b), one process per node (splpak_mplan_*): max over ranks of (subtree flop + its share of the top fronts) / rate;
                route (a), one process per GPU (the driver's launch): max over ranks of (subtree flop) + ALL top flop (replicated) / rate
  solves        HBM bytes of the rank's panels read twice per solve / measured solve bandwidth, (1 + refinement steps) solves
  collectives   ring all-reduce of the histogram, the normal equations and each residual: 2 (N-1)/N x bytes / link bandwidth
                (xGMI is point to point: 7 links x 153 GB/s per GPU; one ring is bound by ONE link each way; RCCL runs several
                rings over distinct links -- `links_used` brackets it between 1 and 4)
  xGMI panels   route (b) only: every solved top-front block column is copied to the ranks that update with it
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from splpak_amd import capi  # noqa: E402

RATE_TFLOPS = 45.7          # in-pipeline rate of the factorisation at 64^3 on one GPU (BENCH_r05.json: 1.10e13 flop in 0.2429 s fit, 217 ms of it factorisation -> 5.07e13 flop/s; the Schur kernel alone 45.7)
FACTOR_RATE = 1.18e13 / 0.217
ASSEMBLY_PTS_PER_S = 1e7 / 0.0045       # 4.5 ms of binning + Gram + gather + constraint rows for 1e7 points (r05 stage timing)
SOLVE_GBS = 4600.0          # 28 GB of panels per tree solve in 6.07 ms (round-5 verdict)
LINK_GBS = 153.0 * 0.8      # one xGMI link, 80 % achievable
REFINE_SOLVES = 3

def predict(nodes, npoints, label):
    out = {"workload": label, "ranks": {}}
    _, s1 = capi.debug_nd_partition(nodes, 1)
    total_flop = float(s1["flop"])
    for R in (1, 2, 4, 8):
        ranks, summ = capi.debug_nd_partition(nodes, R)
        sub = [float(r["flop_subtrees"]) for r in ranks]
        top = [float(r["flop_top"]) for r in ranks]
        top_all = sum(top)
        pts = npoints / R
        asm_ms = 1e3 * pts / ASSEMBLY_PTS_PER_S
        fb_ms = 1e3 * max(a + b for a, b in zip(sub, top)) / FACTOR_RATE
        fa_ms = 1e3 * (max(sub) + top_all) / FACTOR_RATE if R > 1 else 1e3 * total_flop / FACTOR_RATE
        panel = [float(r["panel_bytes"]) + float(r["top_bytes"]) for r in ranks]
        solve_ms = 1e3 * REFINE_SOLVES * 2 * max(panel) / (SOLVE_GBS * 1e9)
        neq = float(summ["normal_eq_bytes"])
        ncol = 1
        for n in nodes:
            ncol *= n
        ar_bytes = neq + 8.0 * ncol + REFINE_SOLVES * 8.0 * ncol
        ring = 2.0 * (R - 1) / R * ar_bytes if R > 1 else 0.0
        coll = {f"links_used_{k}": 1e3 * ring / (k * LINK_GBS * 1e9) for k in (1, 4)}
        xgmi_panels = float(summ["top_steps"]) * float(summ["max_panel_bytes"]) if R > 1 else 0.0
        out["ranks"][str(R)] = {
            "points_per_rank": pts, "assembly_ms": asm_ms,
            "flop_subtrees_max": max(sub), "flop_top_total": top_all, "flop_top_share_max": max(top),
            "factorisation_ms_route_b_one_process": fb_ms, "factorisation_ms_route_a_process_per_gpu": fa_ms,
            "factor_speedup_route_b": (1e3 * total_flop / FACTOR_RATE) / fb_ms, "factor_speedup_route_a": (1e3 * total_flop / FACTOR_RATE) / fa_ms,
            "panel_bytes_max": max(panel), "solve_refine_ms": solve_ms,
            "allreduce_bytes": ar_bytes, "allreduce_ms": coll, "xgmi_top_panel_bytes_upper_bound": xgmi_panels,
            "xgmi_top_panel_ms": 1e3 * xgmi_panels / (LINK_GBS * 1e9),
            "fit_ms_route_b": asm_ms + fb_ms + solve_ms + coll["links_used_1"] + 1e3 * xgmi_panels / (LINK_GBS * 1e9),
            "fit_ms_route_a": asm_ms + fa_ms + solve_ms + coll["links_used_1"],
            "memory_bytes_per_rank_route_b": max(float(r["bytes"]) for r in ranks),
        }
    return out

if __name__ == "__main__":
    doc = {"what": "PREDICTION, not a measurement: per-rank cost of the fit on 1/2/4/8 MI355X from the nested-dissection partition and single-GPU rates",
           "rates": {"factorisation_flop_per_s": FACTOR_RATE, "assembly_points_per_s": ASSEMBLY_PTS_PER_S, "solve_GBs": SOLVE_GBS,
                     "xgmi_link_GBs_achievable": LINK_GBS, "solves_per_fit": REFINE_SOLVES,
                     "sources": "BENCH_r05.json (242.9 ms per fit, 217 ms factorisation, 1.18e13 padded flop), profiles/r05_c3_bench_kernel_stats.csv, "
                                "VERDICT round 5 (tree solve 28 GB in 6.07 ms), MI355X_MICROARCH.md (7 x 153 GB/s xGMI)"},
           "c3": predict([64, 64, 64], 1e7, "BASELINE config 3: 3-D, 1e7 points in all (strong scaling), 64^3 nodes"),
           "c4": predict([64, 64, 64], 1e8, "BASELINE config 4: 3-D, 1e8 points in all, 64^3 nodes")}
    json.dump(doc, sys.stdout, indent=1)
    print()
