"""Options of the library (round 6, include/splpak_hip.h): named switches instead of environment variables read at every fit.
CPU tier: the process defaults (no GPU needed).  GPU tier: a plan's snapshot, per-fit options, and the goldens over every
documented boolean option."""
import os

import numpy as np
import pytest

from splpak_amd import capi
from tests.cases import CASES, make_inputs
from tests.conftest import load_golden, relmax


def test_default_options_know_their_names():
    assert capi.set_default_option("nd_kb", "2") == 0
    assert capi.set_default_option("SPLPAK_ND_KB", None) == 0
    assert capi.set_default_option("Solver", "direct") == 0
    assert capi.set_default_option("solver", None) == 0
    with pytest.raises(capi.SplpakError) as e:
        capi.set_default_option("no_such_option", "1")
    assert "unknown option" in str(e.value)


def test_no_getenv_on_the_fit_path():
    """The library's sources read the environment in ONE place (csrc/options.hip: the snapshot a plan takes at creation and the
    fallback outside of any plan); everything else goes through the calling thread's current options."""
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "splpak_amd", "csrc")
    hits = []
    for f in sorted(os.listdir(src)):
        if not f.endswith((".hip", ".hpp", ".inc")) or f.startswith("options."):
            continue
        for i, line in enumerate(open(os.path.join(src, f)), 1):
            if "getenv(" in line and "opt_get" not in line:
                hits.append(f"{f}:{i}")
    assert hits == [], hits


@pytest.mark.gpu
def test_plan_keeps_its_snapshot_and_takes_per_fit_options():
    os.environ["SPLPAK_PCG_TOL2"] = "1e-4"
    try:
        plan = capi.Plan(2, [8, 8], [0.0, 0.0], [1.0, 1.0], 1.0, 100)
    finally:
        os.environ.pop("SPLPAK_PCG_TOL2")
    try:
        assert plan.get_option("pcg_tol2") == "1e-4"          # as the environment was when the plan was created
        assert plan.get_option("debug") is None
        plan.set_option("debug", "1")
        assert plan.get_option("SPLPAK_DEBUG") == "1"
        plan.set_option("debug", None)
        assert plan.get_option("debug") is None
        with pytest.raises(capi.SplpakError) as e:
            plan.set_option("nd_kb", "2")                      # shapes the plan: creation time only
        assert "-4" in str(e.value) and "splpak_set_default_option" in str(e.value)
        with pytest.raises(capi.SplpakError):
            plan.set_option("nonsense", "2")
    finally:
        plan.close()


BOOLEAN_OPTIONS = [("solver", "direct"), ("solver", "pcg+direct"), ("nd", "1"), ("nd", "0"), ("no_reorder", "1"), ("no_plan_cache", "1"),
                   ("nd_kb", "2"), ("nd_cut", "1"), ("nd_split", "5"), ("nd_res_cus", "0"), ("gram_scratch_mb", "1"), ("pcg_tol1", "1e-12")]


@pytest.mark.gpu
@pytest.mark.parametrize("name,value", BOOLEAN_OPTIONS)
def test_goldens_under_every_documented_option(name, value):
    """Each documented option set as a process default through the API (not the environment): the reference's goldens at 1e-10."""
    capi.set_default_option(name, value)
    try:
        for case in ("2d16", "3d8_sparse", "3d_aniso", "4d5_cc", "c1_1d16", "2d_aniso_box"):
            gold = load_golden(case)
            inp = make_inputs(CASES[case])
            c, rc, _, info = capi.fit(inp["ndim"], inp["xdata"], inp["ydata"], inp["wdata"], inp["xmin"], inp["xmax"], inp["nodes"], inp["xtrap"])
            assert rc == 0, (name, value, case)
            assert relmax(c, gold["coef"]) < 1e-10, (name, value, case)
    finally:
        capi.set_default_option(name, None)
        capi.shutdown()
