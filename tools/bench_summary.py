#!/usr/bin/env python3
"""Compact view of a bench.py JSON line: tools/bench_summary.py <file>"""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
r = d["roofline"]
print("value", f"{d['value']:.4g}", "ms/step", round(d["ms_per_step"], 2), "phases", {k: round(v * 1e3, 2) for k, v in d["config"]["phase_seconds_per_step"].items()})
print("roofline", round(r["achieved"], 2), round(r["frac"], 3), "launches", r["timed_launches"], "alone", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in (r.get("kernel_alone") or {}).items() if k != "what"})
for k in ("c2", "grid32"):
    print(k, round(d[k].get("ms_per_fit", -1), 3), d[k].get("factorisation", d[k].get("error", ""))[:40])
print("eval", f"{d['evals_per_s']:.4g}", "c5_eval", {k: (f"{v:.4g}" if isinstance(v, float) else v) for k, v in d["c5_eval"].items() if k.endswith("per_s") or k == "error"}, "real32", (d["c5_eval"].get("real32") or {}).get("splfe_evals_per_s"), (d["c5_eval"].get("real32") or {}).get("max_rel_deviation_from_real64"))
print("c5_fit", d["c5_fit"].get("seconds_per_fit"), d["c5_fit"].get("error"))
print("h2d", d["fit_incl_h2d"].get("seconds"), "multi_gpu_one_process", (d.get("multi_gpu_one_process") or {}).get("ms_per_fit", (d.get("multi_gpu_one_process") or {}).get("error")))
print("cpu_baseline", d["cpu_baseline"].get("value"), d["cpu_baseline"].get("kind"))
