"""The bench line the driver parses: the committed output of `python bench.py` on MI355X (profiles/r05_c3_bench.json)
must carry the contract's keys, BASELINE.json's metric, a roofline object for the dominant kernel whose numbers are
consistent with each other, and a CPU baseline -- checked on the CPU tier so that a change to bench.py that drops a
key is caught before the GPU run."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


import pytest


def _line():
    txt = open(os.path.join(ROOT, "profiles", "r05_c3_bench.json")).read().strip().splitlines()
    lines = [ln for ln in txt if ln.startswith("{")]
    assert len(lines) == 1, "bench.py prints ONE JSON line"
    return json.loads(lines[0])


def test_contract_keys_and_metric():
    d = _line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    norm = lambda s: s.replace("³", "^3").replace(" ", "")
    assert norm(d["metric"]) == norm(base["metric"])
    assert d["unit"] == "points/s" and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = points of all ranks / time of the timed steps
    pts = d["config"]["points_total"]
    assert abs(d["value"] - pts / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert d["config"]["optimality_residual"] < 1e-9


def test_roofline_object_is_consistent():
    r = _line()["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # achieved = algorithmic flops per launch / average launch duration (HIP events inside the timed region)
    assert abs(r["achieved"] - r["flop_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12) < 1e-6 * r["achieved"]
    assert 0.3 < r["frac"] < 1.0
    # scalars only (the driver's parser drops nested objects): counter traffic per launch beside the algorithmic bytes
    assert r["traffic"] is None or (r["traffic"] > 0 and "not measured in this run" in r["traffic_source"])
    assert all(not isinstance(v, (dict, list)) for v in r.values())
    assert 0.3 < r["kernel_alone_frac"] < 1.0
    # round 5: the evaluation half of the headline metric and the side configurations as scalars of THIS
    # object, which the driver's record keeps
    assert r["evals_per_s"] > 1e10 and abs(r["eval_frac"] - 32.0 * r["evals_per_s"] / 8e12) < 1e-9
    assert r["eval_bytes_per_query"] is None or r["eval_bytes_per_query"] > r["eval_algorithmic_bytes_per_query"] == 32.0
    assert 0 < r["eval_direct_kernel_frac"] < r["eval_frac"]
    assert r["eval4d_evals_per_s"] > 1e10 and abs(r["eval4d_frac"] - 40.0 * r["eval4d_evals_per_s"] / 8e12) < 1e-9
    assert r["c5_fit_points_per_s"] > 0 and 0.3 < r["c5_fit_factor_frac"] < 1.0 and r["c2_ms_per_fit"] > 0


def test_cpu_baseline_and_side_objects():
    d = _line()
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0
    assert d["eval"]["roofline"]["bound"] == "hbm" and 0 < d["eval"]["roofline"]["frac"] < 1
    for k in ("c2", "grid32", "multi_gpu_one_process", "fit_incl_h2d", "c5_eval", "c5_fit", "assembly"):
        assert k in d, k
    for v in d["assembly"].values():                       # no stage claims more than the HBM peak (round 3: 1.07)
        if isinstance(v, dict) and "frac_of_hbm_peak" in v:
            assert 0 < v["frac_of_hbm_peak"] < 1
    mg = d["multi_gpu_one_process"]
    assert mg["factorisation"]["code"] == 5 and mg["optimality_residual"] < 1e-9
    assert "nested-dissection" in d["config"]["factorisation"]
    assert d["c5_fit"]["optimality_residual"] < 1e-9
    # BASELINE config 5's fit half on one GPU: 28^4 since round 5 (24^4 only if the device was shared), 32^4's refusal explained
    assert ("28^4" in d["c5_fit"]["workload"] or "24^4" in d["c5_fit"]["workload"]) and "packed" in d["c5_fit"]["factorisation"]
    assert d["c5_fit"]["config5_32^4_needs"]["factor_GB"] > 400 and "refused" in d["c5_fit"]["config5_32^4_needs"]["note"]
    assert "collective_backend" in d["config"]
    assert d["c2"]["optimality_residual"] < 1e-9 and d["grid32"]["optimality_residual"] < 1e-9


def test_multi_gpu_defaults_name_the_baseline_configs():
    """`bench.py --gpus 8` without --ndata is BASELINE config 4 (1e8 points in all = 1.25e7 per GPU) and says so;
    one GPU is config 3; other rank counts say what they are."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.default_ndata(8, 3, 64) == 12_500_000 and 8 * bench.default_ndata(8, 3, 64) == 100_000_000
    for n in (1, 2, 4):
        assert bench.default_ndata(n, 3, 64) == 10_000_000
    assert bench.workload_label(1, 3, 64, 10_000_000).startswith("C3: ")
    lab8 = bench.workload_label(8, 3, 64, bench.default_ndata(8, 3, 64))
    assert lab8.startswith("C4: ") and "100000000" in lab8 and "12500000 per GPU" in lab8
    lab2 = bench.workload_label(2, 3, 64, 10_000_000)
    assert not lab2.startswith("C3: ") and not lab2.startswith("C4: ") and "20000000" in lab2
    # a guarded side leg turns an exception into an error object instead of losing the line
    assert "error" in bench.guarded(lambda: 1 / 0)


@pytest.mark.gpu
def test_live_bench_line_keeps_the_contract():
    """A line produced NOW by `python bench.py --no-side-legs` on the GPU (not the committed profile): the contract's keys,
    a roofline object of scalars whose numbers are consistent, the CPU baseline."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-side-legs", "--neval", "20000000"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-500:], r.stderr[-500:])
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["steps"] == 2 and d["warmup"] == 1 and d["n_gpus"] == 1
    assert abs(d["value"] - d["config"]["points_total"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r_ = d["roofline"]
    assert all(not isinstance(v, (dict, list)) for v in r_.values())
    assert abs(r_["achieved"] - r_["flop_per_launch"] / (r_["avg_launch_ms"] * 1e-3) / 1e12) < 1e-6 * r_["achieved"]
    assert 0.3 < r_["frac"] < 1.0 and d["config"]["optimality_residual"] < 1e-9
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] in ("reference", "port")
    assert 0 < d["eval_roofline_frac"] < 1
    assert r_["evals_per_s"] > 1e10 and 0 < r_["eval_frac"] < 1 and "collective_backend" in d["config"]
