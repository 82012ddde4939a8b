"""CPU: the C-ABI library loads, exports every symbol include/splpak_hip.h declares,
and its host-side argument validation follows the reference's ierror order.
No compute entry point is exercised here (there is no GPU in the CPU tier)."""
import os
import re

import numpy as np
import pytest

from splpak_amd import capi
from tests.conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "splpak_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(splpak_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    declared = _declared_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/splpak_hip.h but not exported"
    assert sorted(capi.SYMBOLS) == declared


def test_fit_validation_order_without_gpu():
    """101..106 are decided on the host before any device work (:716-781)."""
    x = np.linspace(0, 1, 20).reshape(-1, 1)
    y = x[:, 0].copy()
    f = lambda **k: capi.fit(k.get("ndim", 1), x, y, None, k.get("xmin", [0.0]), k.get("xmax", [1.0]),
                             k.get("nodes", [10]), k.get("xtrap", 1.0), ncf=k.get("ncf"),
                             nwrk=k.get("nwrk", -1), ndata=k.get("ndata"))[1]
    assert f(ndim=0) == 101
    assert f(nodes=[3]) == 102
    assert f(xmax=[0.0]) == 103
    assert f(ncf=9) == 104
    assert f(ndata=0) == 105
    assert f(nwrk=10) == 106
    # first failing check wins
    assert f(ndim=0, nodes=[3]) == 101
    assert f(nodes=[3], ncf=1) == 102


def test_eval_validation_without_gpu():
    coef = np.ones(10)
    e = lambda **k: capi.evaluate(k.get("ndim", 1), np.array([[0.3]]), k.get("nd"), coef,
                                  k.get("xmin", [0.0]), k.get("xmax", [1.0]), k.get("nodes", [10]))
    v, rc = e(ndim=0)
    assert rc == 101 and v[0] == 0.0
    assert e(nodes=[3])[1] == 102
    assert e(xmax=[0.0])[1] == 103


def test_eval_rejects_short_leading_dimension():
    """xq(ldxq, nq) with ldxq < ndim cannot hold a query: argument error before any device work."""
    with pytest.raises(capi.SplpakError) as e:
        capi.evaluate(2, np.array([[0.3], [0.4]]), None, np.ones(16), [0.0, 0.0], [1.0, 1.0], [4, 4])
    assert "-3" in str(e.value)
    assert capi.lib().splpak_set_eval_mode(9, 0) == capi.E_BADARG
    assert capi.lib().splpak_set_eval_mode(0, 0) == 0


def test_eval_derivs_validation_without_gpu():
    q = np.array([[0.3, 0.4]])
    c = np.ones(16)
    assert capi.evaluate_derivs(0, q, 1, c, [0.0, 0.0], [1.0, 1.0], [4, 4])[1] == 101
    assert capi.evaluate_derivs(2, q, 1, c, [0.0, 0.0], [1.0, 1.0], [4, 3])[1] == 102
    assert capi.evaluate_derivs(2, q, 1, c, [0.0, 0.0], [1.0, 0.0], [4, 4])[1] == 103
    with pytest.raises(capi.SplpakError) as e:
        capi.evaluate_derivs(2, q, 0, c, [0.0, 0.0], [1.0, 1.0], [4, 4])
    assert "-3" in str(e.value)
    assert capi.derivs_nout(3, 1) == 4 and capi.derivs_nout(3, 2) == 10 and capi.derivs_nout(4, 2) == 15


def test_no_cpu_fallback_without_gpu():
    """On a box without a GPU the compute path must fail loudly, not fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    x = np.linspace(0, 1, 20).reshape(-1, 1)
    with pytest.raises(capi.SplpakError):
        capi.fit(1, x, x[:, 0], None, [0.0], [1.0], [10], 1.0)
    with pytest.raises(capi.SplpakError):
        capi.evaluate(1, x, None, np.ones(10), [0.0], [1.0], [10])


def test_comm_len_formula():
    nodes = np.array([8, 8, 8], dtype=np.int32)
    n = 512
    npad = 512
    assert capi.lib().splpak_plan_comm_len(3, capi._p(nodes, capi._ip)) == n * 172 + n + 8 + n + 8 + npad + 8


@pytest.mark.parametrize("nodes", [4, 7, 8, 9, 16, 64, 301])
def test_basis_table_forms_match_the_reference_basis(port, nodes):
    """The value table of the evaluation kernels (csrc/basis.hpp: closed form of an interior window, closed form with
    the end functions put in next to an end of the grid, general form elsewhere and on grids of fewer than 8 nodes)
    against the reference's bascmp (oracle, :206-389) entry by entry: inside, on the nodes, one ulp either side of them,
    in the end cells and outside the grid.  Host code only -- the same functions the kernels compile."""
    import ctypes as C
    rng = np.random.default_rng(nodes)
    xmin, xmax = -1.25, 3.5
    dx = (xmax - xmin) / (nodes - 1)
    grid = xmin + dx * np.arange(-2, nodes + 2)
    x = np.concatenate([rng.uniform(xmin - 2 * dx, xmax + 2 * dx, 20000), grid, np.nextafter(grid, 1e9), np.nextafter(grid, -1e9),
                        xmin + dx * rng.uniform(0, 3, 3000), xmax - dx * rng.uniform(0, 3, 3000), [xmin, xmax, xmin - 40.0, xmax + 1e6]])
    ws, used, gen, form = capi.debug_window_values(nodes, xmin, xmax, x)
    if nodes >= 8:
        assert set(np.unique(form)) == {0, 1, 2}
        assert np.all(form[x < xmin] == 2) and np.all(form[x > xmax] == 2)
        assert np.all(form[(x >= xmin) & (x < xmax - 1e-9)] <= 1)
    else:
        assert np.all(form == 2)
    # the forms among themselves: rounding of dxin (x - x_node) only
    assert np.max(np.abs(used - gen)) <= 1e-12 * max(1.0, np.max(np.abs(gen)))
    # against the reference's basis function of every window node
    L = port.lib if hasattr(port, "lib") else port._lib
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    nd0 = (C.c_int * 1)(0)
    nodes_c = (C.c_int * 1)(nodes)
    xmin_c, dx_c = (C.c_double * 1)(xmin), (C.c_double * 1)(dx)
    icol = (C.c_int * 1)(0)
    worst = 0.0
    for i in range(0, x.size, 7):
        # window rule :1201-1209: entries outside [ibmn, ibmx] do not enter the sum (value 0 in the table)
        it = int(np.trunc((1.0 / dx) * (x[i] - xmin))) if abs((x[i] - xmin) / dx) < 2e9 else 0
        ibmn = min(max(it - 1, 0), nodes - 2)
        ibmx = max(min(it + 2, nodes - 1), 1)
        for k in range(4):
            ib = int(ws[i]) + k
            xi = (C.c_double * 1)(x[i])
            ibc = (C.c_int * 1)(ib)
            ref = L.oracle_bascmp(1, xi, nd0, ibc, xmin_c, dx_c, nodes_c, icol) if ibmn <= ib <= ibmx else 0.0
            worst = max(worst, abs(used[i, k] - ref) / max(1.0, abs(ref)))
    assert worst <= 1e-12, worst


# ---------------------------------------------------------------------------------------------------------------
# the RCCL rendezvous file (csrc/rccl.hip): host logic, no GPU and no RCCL call needed for the reader side
def _id_file(path, job, age_s=0.0, magic=b"SPLPAKID"):
    import struct
    import time
    h = 1469598103934665603
    for c in job.encode():
        h = ((h ^ c) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    with open(path, "wb") as f:
        f.write(magic + struct.pack("<Qq", h, int((time.time() - age_s) * 1e9)) + bytes(range(128)))


def test_rccl_id_file_of_another_run_is_ignored(tmp_path):
    """A rank > 0 must not pick up an id file that another run left at the same path (it would enter ncclCommInitRank with a
    stale id and hang): files with another job tag, files older than the wait, short files and files of the round-4 layout
    (128 bare bytes) all end in the timeout with SPLPAK_E_COMM -- never in a communicator."""
    path = tmp_path / "id"
    for make in (lambda: _id_file(path, "another job"),
                 lambda: _id_file(path, "this job", age_s=3600.0),
                 lambda: _id_file(path, "this job", magic=b"XXXXXXXX"),
                 lambda: path.write_bytes(bytes(128)),
                 lambda: path.write_bytes(b"short")):
        make()
        with pytest.raises(capi.SplpakError) as ei:
            capi.rccl_comm_create_from_file(path, 1, 2, timeout_s=0.3, job="this job")
        assert "-5" in str(ei.value) and "timed out" in str(ei.value)
        assert "another run" in str(ei.value)
    path.unlink()
    with pytest.raises(capi.SplpakError) as ei:
        capi.rccl_comm_create_from_file(path, 1, 2, timeout_s=0.2, job="this job")
    assert "timed out" in str(ei.value) and "another run" not in str(ei.value)
    with pytest.raises(capi.SplpakError):
        capi.rccl_comm_create_from_file(path, 2, 2, timeout_s=0.2)          # rank out of range


def test_rccl_library_missing_is_a_clean_error():
    """with no loadable librccl the hook entry points return SPLPAK_E_COMM and a message (rccl_load used to build
    its message from two dlerror() calls -- the second returns NULL -- and crashed)."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from splpak_amd import capi\n"
            "try:\n"
            "    capi.rccl_unique_id()\n"
            "except capi.SplpakError as e:\n"
            "    print('ERR', e)\n") % ROOT
    env = dict(os.environ, SPLPAK_RCCL_LIB="/nonexistent/librccl.so")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-400:]
    assert "ERR" in r.stdout and "-5" in r.stdout and "librccl" in r.stdout
