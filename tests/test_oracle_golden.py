"""CPU: pin the oracle (C restatement) against golden vectors from the REAL reference.

The fixtures in tests/golden/ were produced by oracle/gen_golden.py from the
unmodified reference (oracle/_ref).  Tolerances: coefficients 1e-10 max-norm
relative (BASELINE.json north_star); in practice the port agrees to ~1e-12.
"""
import numpy as np
import pytest

from tests.cases import CASES, make_inputs, make_queries, nderiv_patterns
from tests.conftest import load_golden, relmax

COEF_TOL = 1e-10
EVAL_TOL = 1e-11

# the slowest port runs (dense O(m n^2) algorithm) are marked slow
FAST = [n for n, s in CASES.items() if int(np.prod(s["nodes"])) <= 700 and not s.get("slow")]
SLOW = [n for n, s in CASES.items() if int(np.prod(s["nodes"])) > 700 and not s.get("slow")]


def _fit_and_compare(port, name):
    spec = CASES[name]
    gold = load_golden(name)
    inp = make_inputs(spec)
    coef, ierr, work = port.fit(inp["ndim"], inp["xdata"], inp["ydata"], inp["wdata"], inp["xmin"],
                                inp["xmax"], inp["nodes"], inp["xtrap"])
    ncol = int(np.prod(inp["nodes"]))
    assert ierr == int(gold["ierror"]) == 0
    assert relmax(coef[:ncol], gold["coef"]) < COEF_TOL
    if inp["xtrap"] != 0.0:
        # the sparse-area histogram the reference leaves in work(1:ncol) (:879-907)
        assert relmax(work[:ncol], gold["hist"]) < 1e-13


@pytest.mark.parametrize("name", FAST)
def test_port_fit_matches_reference_golden(port, name):
    _fit_and_compare(port, name)


@pytest.mark.parametrize("name", ["2d32", "3d12"])
def test_port_fit_matches_reference_golden_large(port, name):
    _fit_and_compare(port, name)


@pytest.mark.parametrize("name", [n for n in CASES if not CASES[n].get("slow")])
def test_port_eval_matches_reference_golden(port, name):
    spec = CASES[name]
    gold = load_golden(name)
    inp = make_inputs(spec)
    q = make_queries(spec)
    pats = nderiv_patterns(inp["ndim"])
    assert np.array_equal(np.array(pats), gold["patterns"])
    for i, p in enumerate(pats):
        v, ierr = port.evaluate(inp["ndim"], q, p, gold["coef"], inp["xmin"], inp["xmax"], inp["nodes"])
        assert ierr == 0
        scale = max(np.max(np.abs(gold["values"][i])), 1e-300)
        assert np.max(np.abs(v - gold["values"][i])) / scale < EVAL_TOL, (name, p)
    v0, _ = port.evaluate(inp["ndim"], q, None, gold["coef"], inp["xmin"], inp["xmax"], inp["nodes"])
    assert np.max(np.abs(v0 - gold["values"][0])) <= EVAL_TOL * max(np.max(np.abs(gold["values"][0])), 1e-300)


def test_reference_known_answer_linear(port):
    """test/splpak_test_linear.f90:79-83 -- slope of the fitted y=2x is 2 within 1e-12,
    and the fit error bound of :73 (1e-1)."""
    spec = CASES["ref_linear"]
    inp = make_inputs(spec)
    coef, ierr, _ = port.fit(1, inp["xdata"], inp["ydata"], inp["wdata"], inp["xmin"], inp["xmax"],
                             inp["nodes"], inp["xtrap"])
    assert ierr == 0
    for x0 in (0.0, 1.0):
        v, ie = port.evaluate(1, np.array([[x0]]), [1], coef, inp["xmin"], inp["xmax"], inp["nodes"])
        assert ie == 0 and abs(v[0] - 2.0) <= 1e-12
    xs = (np.arange(100) / 100.0).reshape(-1, 1)
    v, _ = port.evaluate(1, xs, None, coef, inp["xmin"], inp["xmax"], inp["nodes"])
    assert np.max(np.abs(v - 2.0 * xs[:, 0])) <= 1e-1


def test_port_error_codes(port):
    """ierror conventions of splcw (:716-781, :850-854, :1053-1058) and splde (:1166-1194)."""
    x = np.linspace(0, 1, 50).reshape(-1, 1)
    y = np.sin(x[:, 0])
    w = np.ones(50)
    f = lambda **k: port.fit(k.get("ndim", 1), x, y, k.get("w", w), k.get("xmin", [0.0]),
                             k.get("xmax", [1.0]), k.get("nodes", [10]), k.get("xtrap", 1.0),
                             nwrk=k.get("nwrk"), ncf=k.get("ncf"), ndata=k.get("ndata"))[1]
    assert f(ndim=0) == 101
    assert f(nodes=[3]) == 102
    assert f(xmax=[0.0]) == 103
    assert f(ncf=9) == 104
    assert f(ndata=0) == 105
    assert f(nwrk=10) == 106           # nwrk - (ncol+1) + 1 < 1
    assert f(nwrk=10, xtrap=0.0) == 107  # passes 106, fails suprls' own check (32 -> 107)
    assert f(w=np.zeros(50)) == 107    # all-zero weights (33 -> 107)
    assert f(w=np.zeros(50), xtrap=0.0) == 107
    # 5 points, 10 nodes: xtrap=0 -> too few rows; xtrap=1 -> constraint rows regularise
    x5 = np.linspace(0.1, 0.9, 5).reshape(-1, 1)
    assert port.fit(1, x5, np.sin(x5[:, 0]), None, [0.0], [1.0], [10], 0.0)[1] == 107
    assert port.fit(1, x5, np.sin(x5[:, 0]), None, [0.0], [1.0], [10], 1.0)[1] == 0
    coef = np.ones(10)
    e = lambda **k: port.evaluate(k.get("ndim", 1), np.array([[0.3]]), k.get("nd", [0]), coef,
                                  k.get("xmin", [0.0]), k.get("xmax", [1.0]), k.get("nodes", [10]))
    assert e(ndim=0) == (pytest.approx([0.0]), 101)
    assert e(nodes=[3])[1] == 102
    assert e(xmax=[0.0])[1] == 103
    assert e(nd=[3])[1] == 104


def test_port_matches_live_reference(port, reference):
    """Where the reference itself can be built (this container), compare directly on
    inputs that are not in the golden set."""
    if reference is None:
        pytest.skip("oracle/_ref not available")
    rng = np.random.default_rng(7)
    for nd, nodes, m in [(1, [9], 40), (2, [6, 11], 700), (3, [5, 4, 7], 900), (4, [4, 4, 5, 4], 1500)]:
        x = rng.random((m, nd)) * 1.2 - 0.1
        y = np.cos(x.sum(axis=1) * 2.0) + 0.05 * rng.standard_normal(m)
        w = rng.random(m)
        w[rng.random(m) < 0.1] = 0.0
        for wd, xtrap in [(w, 1.0), (None, 0.3)]:
            c0, e0, w0 = reference.fit(nd, x, y, wd, [0.0] * nd, [1.0] * nd, nodes, xtrap)
            c1, e1, w1 = port.fit(nd, x, y, wd, [0.0] * nd, [1.0] * nd, nodes, xtrap)
            n = int(np.prod(nodes))
            assert e0 == e1 == 0
            assert relmax(c1[:n], c0[:n]) < COEF_TOL
            assert relmax(w1[:n], w0[:n]) < 1e-13


def test_histogram_out_of_range_quirk(port):
    """an out-of-range coordinate skips its own dimension in the Horner
    address but the point is still counted (:899).  2-D 4x4 nodes, 36 in-range points
    + one point at (5.0, 0.5): the histogram sums to 37 and the outlier lands at
    0-based address 2 (the dim-2 index alone)."""
    g = (np.arange(6) + 0.5) / 6.0
    xx, yy = np.meshgrid(g, g, indexing="ij")
    x = np.column_stack([xx.ravel(), yy.ravel()])
    x = np.vstack([x, [5.0, 0.5]])
    y = x[:, 0] + x[:, 1]
    _, ierr, work = port.fit(2, x, y, None, [0.0, 0.0], [1.0, 1.0], [4, 4], 1.0)
    assert ierr == 0
    assert work[:16].sum() == 37.0
    base = port.fit(2, x[:-1], y[:-1], None, [0.0, 0.0], [1.0, 1.0], [4, 4], 1.0)[2]
    diff = work[:16] - base[:16]
    assert diff[2] == 1.0 and np.count_nonzero(diff) == 1


@pytest.mark.parametrize("name", ["2d8", "3d8", "4d4"])
def test_real32_goldens_are_single_precision_images_of_the_real64_ones(name):
    """tests/golden/*_r32.npz come from the reference built with -DREAL32 (src/splpak.F90:33-41) on the
    inputs rounded to single precision: stored as float32, and within single-precision conditioning of
    the real64 goldens (they are the oracle of the GPU tier's REAL32 tests)."""
    g32, g64 = load_golden(name + "_r32"), load_golden(name)
    assert g32["coef"].dtype == np.float32 and g32["values"].dtype == np.float32
    assert int(g32["ierror"]) == 0
    assert np.array_equal(g32["patterns"], g64["patterns"])
    assert 1e-9 < relmax(g32["coef"], g64["coef"]) < 1e-3
    assert relmax(g32["values"][0], g64["values"][0]) < 2e-3
    if g64["hist"].size:
        assert relmax(g32["hist"], g64["hist"]) < 1e-5


@pytest.mark.parametrize("name", FAST + ["2d32", "3d12", "2d64_c2grid", "3d16"])
def test_banded_cpu_solver_matches_reference_golden(port, name):
    """oracle/splpak_banded.c (the reference's rows -> banded normal equations -> Cholesky -> refinement,
    all host cores) is pinned to the same reference goldens as the port: it is the "best CPU" comparator
    of bench.py and the independent answer for grids beyond the dense algorithm's reach."""
    spec = CASES[name]
    gold = load_golden(name)
    inp = make_inputs(spec)
    coef, ierr, info = port.fit_banded(inp["ndim"], inp["xdata"], inp["ydata"], inp["wdata"], inp["xmin"],
                                       inp["xmax"], inp["nodes"], inp["xtrap"])
    assert ierr == 0
    assert relmax(coef, gold["coef"]) < COEF_TOL
    nd_rows = inp["xdata"].shape[0] if inp["wdata"] is None else int(np.count_nonzero(inp["wdata"]))
    assert info[0] == nd_rows


@pytest.mark.parametrize("name", ["c1_1d16", "2d16_sparse", "3d8", "3d8_cc_clust", "3d12", "4d4", "2d16_outside", "2d16_zero_w"])
def test_host_rows_gradient_vanishes_at_the_reference_coefficients(port, name):
    """oracle_rows_gradient (the host optimality check the GPU tier uses at BASELINE's full size) pinned to the
    reference: at the golden coefficients the gradient of the least-squares functional over the reference's rows
    is at rounding level, the row counts are the reference's, and a perturbed coefficient is seen."""
    from tests.cases import CASES, make_inputs
    from tests.conftest import load_golden
    inp = make_inputs(CASES[name])
    g = load_golden(name)
    om, reserr, nrow, ncons = port.rows_gradient(inp["ndim"], inp["xdata"], inp["ydata"], inp["wdata"], inp["xmin"], inp["xmax"],
                                                 inp["nodes"], inp["xtrap"], g["coef"])
    assert om < 1e-12, om
    w = inp["wdata"]
    assert nrow == (int(np.count_nonzero(w)) if w is not None else inp["xdata"].shape[0])
    c2 = g["coef"].copy()
    c2[0] += 1e-6 * np.abs(c2).max()
    assert port.rows_gradient(inp["ndim"], inp["xdata"], inp["ydata"], inp["wdata"], inp["xmin"], inp["xmax"], inp["nodes"],
                              inp["xtrap"], c2)[0] > 1e-9
