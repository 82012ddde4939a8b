#!/usr/bin/env python3
"""profiles/r06_fit_pmc.json and profiles/r06_eval_pmc.json from the counter CSVs tools/r06_profiles.sh leaves under
gpurun_out/r06/ (FETCH_SIZE and WRITE_SIZE in separate passes).  Per kernel: mean counter value per launch, hbm_bytes =
(2 FETCH_SIZE + WRITE_SIZE) * 1024 (the gfx950 correction of MI355X_MICROARCH.md) and the uncorrected sum; for the roofline
kernel of the fit the keys bench.py reads.   tools/r06_pmc_json.py [gpurun_out/r06] [profiles]"""
import collections, csv, json, os, re, sys
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r06"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles"

def kernels(paths):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for p in paths:
        for r in csv.DictReader(open(p)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("splpak::", "")
            name = re.sub(r"\(.*", "", name)
            a = acc[name][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    res = {}
    for k, cs in sorted(acc.items()):
        d = {c: s / n for c, (s, n) in cs.items()}
        d["launches"] = max(n for _, n in cs.values())
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["hbm_bytes"] = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
            d["hbm_bytes_uncorrected"] = (d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
        res[k] = d
    return res

corr = ("hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md (a wide coalesced streaming read is tallied at half "
        "its bytes on gfx950); for the 8-byte-per-lane operand loads of the update kernels the factor is uncalibrated, so hbm_bytes "
        "is an UPPER bound and hbm_bytes_uncorrected (FETCH_SIZE + WRITE_SIZE) a lower one; Infinity-Cache hits are included "
        "(fabric traffic)")
fit = kernels([os.path.join(src, f) for f in ("fit_fetch.counters.csv", "fit_write.counters.csv")])
schur = next(v for k, v in fit.items() if k.startswith("nd_syrk_kernel<4, 2, true"))
upd = next((v for k, v in fit.items() if k.startswith("nd_syrk_kernel<4, 2, false")), None)
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, no tracing beside them; tools/r06_profiles.sh), "
              "kernels of one C3 fit (tools/c2_profile.py 3 64 10000000, one fit: 3-D 64^3, 1e7 weighted points, nested-dissection "
              "factorisation); means per launch; KB as the profiler reports them",
    "correction": corr,
    "schur_launches_per_fit": schur["launches"],
    "schur_hbm_bytes_per_launch": schur["hbm_bytes"],
    "schur_hbm_bytes_per_launch_uncorrected": schur["hbm_bytes_uncorrected"],
    "schur_write_bytes_per_launch": schur["WRITE_SIZE"] * 1024.0,
    "schur_algorithmic_bytes_per_launch": 2.0 * schur["WRITE_SIZE"] * 1024.0,
    "schur_algorithmic_note": "algorithmic = every 64x64 tile of the lower triangle of a Schur buffer written once per pass (WRITE_SIZE is "
                              "the tile bytes) and read at most once (first passes start from zero, the fused last passes add into the parent "
                              "instead of storing): 2 x WRITE_SIZE is the upper bound used; the operands (panel blocks, 8 flop per byte from "
                              "L2 / Infinity Cache) are what the rest of the fetches are",
    "panel_update_hbm_bytes_per_launch": upd["hbm_bytes"] if upd else None,
    "kernels": fit,
}
json.dump(out, open(os.path.join(dst, "r06_fit_pmc.json"), "w"), indent=1)
def eval_set(files, nq_launch, alg, extra=None):
    ks = kernels([os.path.join(src, f) for f in files])
    binned = {k: v for k, v in ks.items() if k.startswith(("bin_", "eval_binned", "run_place", "eval_runs", "pr_"))}
    tot = sum(v.get("hbm_bytes", 0.0) for v in binned.values())
    totu = sum(v.get("hbm_bytes_uncorrected", 0.0) for v in binned.values())
    res = {"queries_per_launch": nq_launch, "hbm_bytes_per_query_all_passes": tot / nq_launch,
           "hbm_bytes_per_query_all_passes_uncorrected": totu / nq_launch, "algorithmic_bytes_per_query": alg, "kernels": ks}
    if extra:
        ex = kernels([os.path.join(src, extra)])
        for k, v in ex.items():
            if k in ks and "SQ_INSTS_VALU" in v:
                ks[k]["valu_instructions_per_query"] = v["SQ_INSTS_VALU"] * 64.0 / nq_launch
                ks[k]["lds_bank_conflict_fraction"] = v.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(v.get("SQ_LDS_IDX_ACTIVE", 1.0), 1.0)
                ks[k].update({c: v[c] for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE") if c in v})
    return res
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes; tools/r06_profiles.sh) of tools/eval_profile.py 3 64 50000000 "
                     "and 4 32 100000000 (three binned and three direct batches each; 3-D and (round 5) 4-D: the persistent region path, ONE launch of each of "
                     "its three kernels per batch of 5e7 / 1e8 queries); means per launch; "
                     "valu_instructions_per_query = SQ_INSTS_VALU (wave instructions) x 64 lanes / queries of a launch; kernel times: profiles/r06_eval3_kernel_stats.csv, r06_eval4_kernel_stats.csv",
           "correction": corr,
           "3d_64": eval_set(("eval3_fetch.counters.csv", "eval3_write.counters.csv"), 5e7, 32, "eval3_valu.counters.csv"),
           "4d_32": eval_set(("eval4_fetch.counters.csv", "eval4_write.counters.csv"), 1e8, 40, "eval4_valu.counters.csv")},
          open(os.path.join(dst, "r06_eval_pmc.json"), "w"), indent=1)
print("schur per launch: hbm %.3g B (uncorrected %.3g), write %.3g, launches/fit %d" % (schur["hbm_bytes"], schur["hbm_bytes_uncorrected"], schur["WRITE_SIZE"] * 1024, schur["launches"]))
# config 5's fit by the iterative solve (round 6): traffic of its kernels, per launch and per iteration
try:
    pcg = kernels([os.path.join(src, f) for f in ("c5_pcg_fetch.counters.csv", "c5_pcg_write.counters.csv")])
    tile = next(v for k, v in pcg.items() if k.startswith("rows4_tile_kernel"))
    per_iter = sum(v.get("hbm_bytes", 0.0) * v["launches"] for v in pcg.values()) / max(tile["launches"], 1)
    try:        # issue counters of the tile kernel (own pass): vector instructions per point, the matrix pipe's share of the SIMD cycles
        vv = kernels([os.path.join(src, "c5_pcg_valu.counters.csv")])
        tv = next(v for k, v in vv.items() if k.startswith("rows4_tile_kernel"))
        tile["valu_wave_instructions_per_point"] = tv["SQ_INSTS_VALU"] / 1e7
        tile.update({c: tv[c] for c in ("SQ_INSTS_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_LDS") if c in tv})
    except Exception as exc:
        print("no issue counters of the tile kernel:", exc)
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes; tools/r06_profiles.sh) of ONE fit of BASELINE config 5 at its own size "
                         "(tools/c2_profile.py 4 32 10000000: 4-D, 32^4 nodes, 1e7 weighted points, iterative solve); means per launch",
               "correction": corr,
               "iterations_plus_residual_passes": tile["launches"],
               "hbm_bytes_per_iteration_all_kernels": per_iter,
               "algorithmic_bytes_per_iteration": 1e7 * 8.0 * 5 + 2 * 8.0 * 32 ** 4,
               "algorithmic_note": "one pass over the sorted points (4 coordinates + weight, 40 B each) + the vector in and out",
               "kernels": pcg}, open(os.path.join(dst, "r06_c5_pcg_pmc.json"), "w"), indent=1)
    print("c5 pcg: %.3g B per iteration (tile kernel %.3g B per launch)" % (per_iter, tile.get("hbm_bytes", 0.0)))
except Exception as exc:
    print("no c5 pcg counters:", exc)
