"""The iterative solve of round 6 (splpak_amd/csrc/pcg.hip): conjugate gradients on the rows with the separable
preconditioner, in place of suprls (src/splpak.F90:1375-1695) for the grids no factorisation fits -- BASELINE config 5's
4-D 32^4 grid on ONE GPU -- and, on request, before the factorisation everywhere else.

What is checked, and against what:
  * every golden of the reference (tests/golden/*.npz, outputs of the unmodified reference) at 1e-10 with the iteration
    switched on in front of the factorisation (SPLPAK_SOLVER=pcg+direct): the iteration answers where it converges, the
    stagnation rule hands over to the factorisation where it does not -- never a silent miss;
  * iteration only (SPLPAK_SOLVER=pcg): either the golden at 1e-10 or the reference's 107, nothing in between;
  * the nested-dissection factorisation at 24^4 (and at 28^4 inside tests/test_nd.py) at 1e-10, and the HOST-side
    backward error over the reference's rows (oracle_rows_gradient) below 1e-12;
  * config 5 itself: 32^4 nodes, 1e7 weighted points, one GPU.
"""
import os
import time

import numpy as np
import pytest

from splpak_amd import capi
from tests.cases import CASES, make_inputs
from tests.conftest import load_golden, relmax

COEF_TOL = 1e-10        # north_star: 1e-10 relative, max norm


def _fit_env(inp, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return capi.fit(inp["ndim"], inp["xdata"], inp["ydata"], inp["wdata"], inp["xmin"], inp["xmax"],
                        inp["nodes"], inp["xtrap"], want_hist=False)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _iteration_answered(info):
    """No factorisation ran in the fit: its seconds and its smallest pivot stay 0 (include/splpak_hip.h, info[4], info[6])."""
    return info[4] == 0.0 and info[6] == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_every_golden_with_the_iteration_in_front_of_the_factorisation(name):
    gold = load_golden(name)
    inp = make_inputs(CASES[name])
    # (pcg_always: also in the regime -- between 0 and 1.6 constraint rows per column -- in which a plan with a factorisation behind the
    #  iteration would not even try it: the hand-over after a stagnating attempt is what is tested here)
    c, rc, _, info = _fit_env(inp, {"SPLPAK_SOLVER": "pcg+direct", "SPLPAK_PCG_ALWAYS": "1"})
    c0, rc0, _, info0 = _fit_env(inp, {"SPLPAK_SOLVER": "direct"})
    assert rc == 0 and rc0 == 0
    err = relmax(c, gold["coef"])
    print(f"{name}: {'iteration' if _iteration_answered(info) else 'factorisation after the iteration gave up'}; vs golden {err:.2e}, "
          f"vs the factorisation alone {relmax(c, c0):.2e}, steps {info[2]:.0f}, backward error {info[9]:.1e}")
    assert err < COEF_TOL
    assert info[9] < 1e-9
    assert info[0] == info0[0] and info[1] == info0[1]
    assert abs(info[8] - info0[8]) <= 1e-9 * info0[8] + 1e-13 * np.linalg.norm(inp["ydata"])     # (ref_linear: an exact fit, reserr ~ 1e-16)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["2d16_sparse", "3d8_cc_clust", "1d_sparse", "2d_aniso_box", "4d6", "4d4", "4d5_cc", "3d8", "c1_1d16_xt0", "2d32_cc_xt0"])
def test_iteration_alone_is_right_or_107(name):
    """Without a factorisation behind it (the plans of grids beyond the device) the iteration either meets the bar or the fit
    returns the reference's 107 with the library's explanation.  The 4-D cases run the rows-only form (nothing assembled: histogram,
    right-hand side, boxes and denominators from the rows); 4d5_cc has points outside the grid (the histogram's :899 quirk) and no weights."""
    gold = load_golden(name)
    inp = make_inputs(CASES[name])
    c, rc, _, info = _fit_env(inp, {"SPLPAK_SOLVER": "pcg"})
    assert rc in (0, 107)
    if rc == 0:
        err = relmax(c, gold["coef"])
        print(f"{name}: converged, vs golden {err:.2e}, steps {info[2]:.0f}, backward error {info[9]:.1e}")
        assert err < COEF_TOL and info[9] < 1e-9
        assert _iteration_answered(info)
    else:
        msg = capi.last_error()
        print(f"{name}: 107: {msg}")
        assert "iterative solve did not converge" in msg
        assert np.all(c == 0.0)


@pytest.mark.gpu
def test_known_bad_regime_goes_straight_to_the_factorisation():
    """A plan with a factorisation behind the iteration does not spend an attempt where the iteration is known to stagnate (0 < constraint
    rows per column < 1.6: DESIGN section 4c): 2d32 (0.2 rows per column) is answered by the factorisation with no iteration at all, 4d6
    (2.6) by the iteration."""
    for name, expect_iteration in (("2d32", False), ("4d6", True)):
        inp = make_inputs(CASES[name])
        gold = load_golden(name)
        c, rc, _, info = _fit_env(inp, {"SPLPAK_SOLVER": "pcg+direct"})
        assert rc == 0 and relmax(c, gold["coef"]) < COEF_TOL
        assert _iteration_answered(info) == expect_iteration, (name, info[1] / np.prod(inp["nodes"]))


@pytest.mark.gpu
def test_singular_normal_equations_are_107_from_the_iteration_too():
    """xtrap = 0 and nodes without any data: the normal equations have zero columns, the reference's "system is singular" (suprls 34 ->
    107, src/splpak.F90:1662-1667) and the factorisations' failed pivot test.  The iteration would run on to a minimiser that is
    arbitrary where the rows see nothing; since a box of the assembled normal equations fails the same pivot test, it says 107 as
    well -- alone and in front of the factorisation (found by tools/pcg/fuzz_pcg.py, seed 9 trial 99)."""
    rng = np.random.default_rng(3)
    nodes, m = [150], 361
    x = np.concatenate([0.45 * rng.random(m // 2), 0.55 + 0.45 * rng.random(m - m // 2)])[:, None]      # nothing in [0.45, 0.55]
    y = np.sin(3.0 * x[:, 0])
    w = 0.5 + rng.random(m)
    inp = dict(ndim=1, xdata=x, ydata=y, wdata=w, xmin=[0.0], xmax=[1.0], nodes=nodes, xtrap=0.0)
    for solver in ("direct", "pcg", "pcg+direct"):
        c, rc, _, info = _fit_env(inp, {"SPLPAK_SOLVER": solver, "SPLPAK_PCG_ALWAYS": "1"})
        assert rc == 107, solver
        assert "suprls 34" in capi.last_error(), (solver, capi.last_error())
    # with constraint rows the same data are a regular problem for all three
    inp["xtrap"] = 1.0
    ref = None
    for solver in ("direct", "pcg+direct"):
        c, rc, _, info = _fit_env(inp, {"SPLPAK_SOLVER": solver, "SPLPAK_PCG_ALWAYS": "1"})
        assert rc == 0 and info[9] < 1e-9, solver
        ref = c if ref is None else ref
        assert relmax(c, ref) < COEF_TOL


@pytest.mark.gpu
def test_plans_with_both_solvers_assemble_the_normal_equations_only_for_the_factorisation():
    """A 4-D plan with the iteration in front of a factorisation starts its fit as an iteration-only plan does -- histogram, right-hand
    side and boxes from the rows -- and assembles the half stencil only where the factorisation is going to run: the iteration is
    not tried (the known bad regime) or gave up.  Then the factorisation sees exactly what the eager order (pcg_eager) gives it:
    same bits as solver = direct.  Where the iteration answers, the two orders agree to rounding (their boxes differ: built from the
    rows / cut out of the assembled equations)."""
    nd, nodes = 4, [12] * 4
    lo, hi = [0.0] * nd, [1.0] * nd
    # (a) the iteration answers: 2.5 constraint rows per column
    x, y, w, st = _device_points(nd, 158122)
    ref, e0, i0, _, _, _ = _plan_fit(nd, nodes, lo, hi, 1.0, x, y, w, st, "direct")
    assert e0 == 0
    got = {}
    for name, env in (("lazy", {}), ("eager", {"SPLPAK_PCG_EAGER": "1"})):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            c, e, info, fac, ps, dt = _plan_fit(nd, nodes, lo, hi, 1.0, x, y, w, st, "pcg+direct")
        finally:
            for k, v in old.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
        assert e == 0 and ps["iterations"] > 0 and relmax(c, ref) < COEF_TOL and info[9] < 1e-9, name
        assert info[0] == i0[0] and info[1] == i0[1] and abs(info[8] - i0[8]) <= 1e-9 * i0[8], name
        got[name] = (c, info[5], ps["iterations"])
    print(f"12^4: assembly {1e3 * got['lazy'][1]:.2f} ms (nothing assembled) against {1e3 * got['eager'][1]:.2f} ms; iterations {got['lazy'][2]} / {got['eager'][2]}")
    assert relmax(got["lazy"][0], got["eager"][0]) < 1e-12
    assert got["lazy"][1] < got["eager"][1]
    # (b) the bad regime (1.0 constraint rows per column): no attempt, the assembly follows, the factorisation answers -- the bits of solver = direct
    x, y, w, st = _device_points(nd, 366025 * 2)
    ref, e0, i0, _, _, _ = _plan_fit(nd, nodes, lo, hi, 1.0, x, y, w, st, "direct")
    c, e, info, fac, ps, dt = _plan_fit(nd, nodes, lo, hi, 1.0, x, y, w, st, "pcg+direct")
    assert e0 == 0 and e == 0 and 0.0 < info[1] / 12 ** 4 < 1.6, info[1] / 12 ** 4
    assert ps["iterations"] == 0 and np.array_equal(c, ref)
    # (c) the iteration is tried there all the same (pcg_always), gives up or not: the answer stays within the bar
    os.environ["SPLPAK_PCG_ALWAYS"] = "1"
    try:
        c, e, info, fac, ps, dt = _plan_fit(nd, nodes, lo, hi, 1.0, x, y, w, st, "pcg+direct")
    finally:
        os.environ.pop("SPLPAK_PCG_ALWAYS", None)
    assert e == 0 and ps["iterations"] > 0 and relmax(c, ref) < COEF_TOL


def _device_points(nd, m):
    import torch
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    x = torch.empty((m, nd), dtype=torch.float64, device=dev)
    y = torch.empty(m, dtype=torch.float64, device=dev)
    w = torch.empty(m, dtype=torch.float64, device=dev)
    capi.synth_points_dev(nd, 0, m, x, y, w, st)
    return x, y, w, st


def _plan_fit(nd, nodes, lo, hi, xtrap, x, y, w, st, solver):
    import torch
    old = os.environ.get("SPLPAK_SOLVER")
    if solver is None:
        os.environ.pop("SPLPAK_SOLVER", None)
    else:
        os.environ["SPLPAK_SOLVER"] = solver
    try:
        plan = capi.Plan(nd, nodes, lo, hi, xtrap, x.shape[0])
    finally:
        if old is None:
            os.environ.pop("SPLPAK_SOLVER", None)
        else:
            os.environ["SPLPAK_SOLVER"] = old
    try:
        coef = torch.zeros(int(np.prod(nodes)), dtype=torch.float64, device=x.device)
        t0 = time.perf_counter()
        ierr, info = plan.fit(x, y, w, coef, st)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return coef.cpu().numpy(), ierr, info, plan.factorisation(), plan.pcg_stats(), dt
    finally:
        plan.close()


@pytest.mark.gpu
def test_iteration_agrees_with_nested_dissection_at_4d_24(port):
    """24^4 = 331 776 columns at config 5's density of points (10.8 per grid cell, a quarter of the nodes data sparse):
    the iteration alone against the nested-dissection factorisation, and against the reference's rows on the host."""
    import torch
    capi.shutdown()
    torch.cuda.empty_cache()
    nd, nod = 4, 24
    m = int(10.8 * (nod - 1) ** nd)
    nodes, lo, hi = [nod] * nd, [0.0] * nd, [1.0] * nd
    x, y, w, st = _device_points(nd, m)
    c_nd, e_nd, i_nd, f_nd, _, t_nd = _plan_fit(nd, nodes, lo, hi, 1.0, x, y, w, st, "direct")
    c_it, e_it, i_it, f_it, ps, t_it = _plan_fit(nd, nodes, lo, hi, 1.0, x, y, w, st, "pcg")
    assert e_nd == 0 and e_it == 0 and f_nd[0] == 4 and f_it[0] == 6
    err = relmax(c_it, c_nd)
    omega, reserr, nrow, ncons = port.rows_gradient(nd, x.cpu().numpy(), y.cpu().numpy(), w.cpu().numpy(), lo, hi, nodes, 1.0, c_it)
    print(f"24^4, {m} points: iteration {t_it:.2f} s ({ps['iterations']} iterations in {ps['solves']} solves) against {t_nd:.2f} s nested dissection; "
          f"coefficients {err:.2e} apart; host backward error {omega:.2e} (GPU {i_it[9]:.1e}); rows {nrow}+{ncons}")
    assert err < COEF_TOL
    assert omega < 1e-12 and i_it[9] < 1e-9
    assert nrow == i_it[0] == i_nd[0] and ncons == i_it[1] == i_nd[1]
    assert abs(reserr - i_it[8]) <= 1e-9 * reserr
    assert 0 < ps["iterations"] < 2000


@pytest.mark.gpu
@pytest.mark.parametrize("nodes,m,xtrap", [([12] * 4, 158122, 1.0), ([13, 12, 14, 11], 200000, 1.0), ([12] * 4, 366025, 1.0), ([12] * 4, 158122, 0.0)])
def test_forms_of_the_preconditioner_agree_with_the_factorisation(nodes, m, xtrap):
    """The iteration-only plan of a 4-D grid in its three forms -- nothing assembled, boxes = scaled mass + exact constraint part (the
    default); normal equations assembled, boxes extracted from them (pcg_assemble); separable part alone (pcg_no_blocks) -- against
    the nested-dissection factorisation: every form that converges agrees at 1e-10 with the same row counts; the default and the
    assembled form take the same number of iterations where constraint rows are present (the data part of a box does not matter
    there: DESIGN section 4c); a grid whose node counts are not multiples of the box edge has partial boxes."""
    import torch
    nd = 4
    lo, hi = [0.0] * nd, [1.0] * nd
    x, y, w, st = _device_points(nd, m)
    ref, e0, i0, _, _, _ = _plan_fit(nd, nodes, lo, hi, xtrap, x, y, w, st, "direct")
    assert e0 == 0
    its, coefs, diag = {}, {}, {}
    # (the separable part's mode products: pairs on the matrix pipe (default), pairs on the vector unit, one mode per launch)
    for form, env in (("rows", {}), ("assembled", {"SPLPAK_PCG_ASSEMBLE": "1"}), ("separable", {"SPLPAK_PCG_NO_BLOCKS": "1"}),
                      ("pairs_valu", {"SPLPAK_PCG_PAIRS_VALU": "1"}), ("no_pairs", {"SPLPAK_PCG_NO_PAIRS": "1"}),
                      # the constraint rows' passes behind the data rows' tile kernel instead of beside it (a stream of their own)
                      ("one_stream", {"SPLPAK_ROWS_ONE_STREAM": "1"}),
                      # the final residual pass and reserr cell by cell (assemble.hip) instead of tile by tile
                      ("cells", {"SPLPAK_RESIDUAL_CELLS": "1"})):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            c, e, info, fac, ps, dt = _plan_fit(nd, nodes, lo, hi, xtrap, x, y, w, st, "pcg")
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        assert fac[0] == 6 and e in (0, 107), (form, e)
        its[form] = (e, ps["iterations"])
        coefs[form] = c
        diag[form] = (info[8], info[9])
        if e == 0:
            assert relmax(c, ref) < COEF_TOL and info[9] < 1e-9, form
            assert info[0] == i0[0] and info[1] == i0[1], form
            assert abs(info[8] - i0[8]) <= 1e-9 * i0[8], form
    print(f"{nodes}, {m} points, xtrap {xtrap}: iterations rows-only {its['rows']}, assembled {its['assembled']}, separable alone {its['separable']}")
    assert its["rows"][0] == 0 and its["assembled"][0] == 0
    # the two forms of the final pass see the same coefficients: the same residual norm and backward error to rounding
    assert np.array_equal(coefs["cells"], coefs["rows"])
    assert abs(diag["cells"][0] - diag["rows"][0]) <= 1e-12 * diag["rows"][0] and max(diag["cells"][1], diag["rows"][1]) < 1e-13, diag
    # the side stream only re-times the same operations: identical bits
    assert its["one_stream"] == its["rows"] and np.array_equal(coefs["one_stream"], coefs["rows"])
    # the same transform computed three ways: the iteration counts differ by rounding at most
    assert its["pairs_valu"][0] == 0 and its["no_pairs"][0] == 0
    assert abs(its["rows"][1] - its["pairs_valu"][1]) <= 5 and abs(its["rows"][1] - its["no_pairs"][1]) <= 5, its
    if xtrap != 0.0:
        assert abs(its["rows"][1] - its["assembled"][1]) <= 10
        assert its["rows"][1] < its["separable"][1] or its["separable"][0] == 107


@pytest.mark.gpu
def test_config5_fit_4d_32_on_one_gpu(port):
    """BASELINE config 5's fit half at its own size: 4-D, 32^4 = 1 048 576 columns, 1e7 weighted scattered points, xtrap = 1, ONE
    GPU.  No factorisation fits (476 GB of nested-dissection panels, band 851 GB: SPLPAK_SOLVER=direct is refused with the
    out-of-memory status); the plan takes the iteration by itself.  Checked on the HOST against the reference's rows
    (src/splpak.F90:788-855, :862-1046): componentwise backward error < 1e-12, row counts and residual norm as the GPU reports."""
    import torch
    capi.shutdown()
    torch.cuda.empty_cache()
    nd, nod, m = 4, 32, 10_000_000
    nodes, lo, hi = [nod] * nd, [0.0] * nd, [1.0] * nd
    x, y, w, st = _device_points(nd, m)
    c, ierr, info, fac, ps, dt = _plan_fit(nd, nodes, lo, hi, 1.0, x, y, w, st, None)
    assert fac[0] == 6, fac
    assert ierr == 0
    omega, reserr, nrow, ncons = port.rows_gradient(nd, x.cpu().numpy(), y.cpu().numpy(), w.cpu().numpy(), lo, hi, nodes, 1.0, c)
    print(f"32^4, 1e7 points on one GPU: {dt:.2f} s (assembly {info[5]:.2f}, solve {info[7]:.2f}); {ps['iterations']} iterations in {ps['solves']} solves; "
          f"host backward error {omega:.2e} (GPU {info[9]:.1e}); rows {nrow}+{ncons}; reserr host {reserr:.9e} GPU {info[8]:.9e}")
    assert omega < 1e-12 and info[9] < 1e-9
    assert nrow == info[0] == m and ncons == info[1] and ncons > 1_000_000
    assert abs(reserr - info[8]) <= 1e-9 * reserr
    assert ps["iterations"] < 2000
    # the same coefficients from a second fit with the same plan settings: the iteration has no atomics, fixed summation orders
    c2, ierr2, _, _, _, _ = _plan_fit(nd, nodes, lo, hi, 1.0, x, y, w, st, None)
    assert ierr2 == 0 and np.array_equal(c, c2)
