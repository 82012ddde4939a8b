"""The Fortran drop-in `splpak_module` (splpak_amd/fortran) over the C ABI.

CPU tier : the module and its test programs build with amdflang, link against the HIP
           library, and FAIL LOUDLY without a GPU (the GPU path never falls back); the module's separate HOST solver
           (`set_host(.true.)`, splpak_host.F90; the only path of a -DREAL128 build) is held to the reference's goldens.
GPU tier : the three test programs -- re-creations of the reference's own tests
           (test/splpak_test_linear.f90, test/splpak_test.f90) plus an API/ierror test --
           pass on the MI355X.
"""
import os
import re
import subprocess

import pytest

from tests.conftest import ROOT

FDIR = os.path.join(ROOT, "splpak_amd", "fortran")
BUILD = os.path.join(FDIR, "build")
PROGS = ["test_linear", "test_noisy", "test_api", "test_info"]
ALL_PROGS = PROGS + ["test_evalfix", "test_hostfit"]


def _ensure_built():
    if all(os.path.exists(os.path.join(BUILD, p)) for p in ALL_PROGS):
        return
    if not os.path.exists("/opt/rocm/bin/amdflang"):
        pytest.skip("amdflang not available")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "splpak_amd", "csrc")])
    subprocess.check_call(["make", "-C", FDIR])


def test_fortran_module_builds():
    _ensure_built()
    assert os.path.exists(os.path.join(BUILD, "splpak_module.mod"))
    assert os.path.exists(os.path.join(BUILD, "libsplpak.so"))


def test_fortran_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    _ensure_built()
    r = subprocess.run([os.path.join(BUILD, "test_linear")], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "HIP library failure" in r.stdout + r.stderr


def test_fortran_scalar_evaluate_matches_reference_goldens_3d_4d():
    """The host scalar `evaluate` (product code that re-derives splde, src/splpak.F90:1089-1240) against the
    reference's own values on 3-D / 4-D grids, incl. node / boundary / outside queries and nderiv = 2, at 1e-12
    (VERDICT r02 #6a).  No GPU needed."""
    _ensure_built()
    fx = [os.path.join(ROOT, "tests", "golden", f"eval_{n}.txt") for n in ("3d12", "4d6", "3d_aniso", "5d4")]   # 5d4: ndim > 4, also the batch call
    r = subprocess.run([os.path.join(BUILD, "test_evalfix")] + fx, capture_output=True, text=True, timeout=300)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PASS test_evalfix" in r.stdout


def _fit_fixtures():
    import glob
    return sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "fit_*.txt")))


def test_fortran_host_solver_matches_reference_goldens():
    """`call solver%set_host(.true.)`: the from-scratch host solver inside the Fortran package (banded normal equations,
    Cholesky, refinement against the rows) against the reference's coefficients and sparse-area histograms
    of 15 golden cases (1-D .. 4-D, weighted / unweighted, xtrap 0 / /= 0, zero weights, points outside the box,
    clustered data): 1e-10 max-norm on the coefficients, 1e-12 on the histogram.  No GPU, and not the oracle."""
    _ensure_built()
    fx = _fit_fixtures()
    assert len(fx) >= 16 and any("fit_5d4" in f for f in fx)      # 5d4: ndim = 5, routed to the host solver by the module itself
    r = subprocess.run([os.path.join(BUILD, "test_hostfit")] + fx, capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PASS test_hostfit" in r.stdout


@pytest.mark.parametrize("prog", ["test_linear", "test_noisy", "test_api"])
def test_fortran_reference_scenarios_on_the_host_solver(prog):
    """The reference's own test scenarios (test/splpak_test_linear.f90: slope 2 +- 1e-12; test/splpak_test.f90) and the
    API / ierror test, run with `host` = set_host(.true.): they pass on a machine without a GPU."""
    _ensure_built()
    r = subprocess.run([os.path.join(BUILD, prog), "host"], capture_output=True, text=True, timeout=300)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"PASS {prog}" in r.stdout


@pytest.mark.parametrize("prog", ["test_linear", "test_noisy"])
def test_unchanged_caller_runs_without_gpu_when_opted_in(prog):
    """SURVEY 8f-4 / round-5 verdict: an UNCHANGED caller (no set_host in its source) on a machine without a GPU.  By default the
    call fails loudly (test_fortran_fails_loudly_without_gpu); with SPLPAK_HOST_IF_NO_GPU=1 in the environment the module says so
    once and runs the host solver -- the reference's own scenarios pass."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    _ensure_built()
    env = dict(os.environ, SPLPAK_HOST_IF_NO_GPU="1")
    r = subprocess.run([os.path.join(BUILD, prog)], capture_output=True, text=True, timeout=300, env=env)
    print(r.stdout[-1500:], r.stderr[-1500:])
    assert r.returncode == 0 and f"PASS {prog}" in r.stdout
    assert r.stdout.count("SPLPAK_HOST_IF_NO_GPU=1: running on the host solver") == 1
    env["SPLPAK_HOST_IF_NO_GPU"] = "0"
    r = subprocess.run([os.path.join(BUILD, prog)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "HIP library failure" in r.stdout + r.stderr


def test_fortran_real128_build_runs_on_the_host_solver():
    """-DREAL128 (reference: README.md:24-36, src/splpak.F90:33-41): no GPU arithmetic exists for quad precision, the
    module is built on the host solver alone (no HIP library linked) and reproduces the real64 goldens to their own
    rounding, with a backward error at quad-precision level."""
    if not os.path.exists("/opt/rocm/bin/amdflang"):
        pytest.skip("amdflang not available")
    subprocess.check_call(["make", "-C", FDIR, "build/r128/test_hostfit"])
    fx = [f for f in _fit_fixtures() if any(k in f for k in ("fit_3d8.txt", "fit_2d16_sparse.txt", "fit_c1_1d16.txt"))]
    r = subprocess.run([os.path.join(BUILD, "r128", "test_hostfit")] + fx, capture_output=True, text=True, timeout=600)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0 and "PASS test_hostfit" in r.stdout
    import re as _re
    for om in _re.findall(r"backward error\s+(\S+)", r.stdout):
        assert float(om) < 1e-25


def test_consumer_example_builds_without_fpm():
    """examples/consumer: a caller's program compiled against the module files and linked to build/libsplpak.so with a
    plain Makefile (fpm is absent from the image), run on the host solver."""
    _ensure_built()
    ex = os.path.join(ROOT, "examples", "consumer")
    subprocess.check_call(["make", "-C", ex])
    r = subprocess.run([os.path.join(ex, "fit_surface"), "host"], capture_output=True, text=True, timeout=300)
    print(r.stdout[-1000:], r.stderr[-1000:])
    assert r.returncode == 0 and "OK fit_surface" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("prog", PROGS)
def test_fortran_program_on_gpu(prog):
    _ensure_built()
    env = dict(os.environ, SPLPAK_VIRTUAL_GPUS="1")      # test_info also runs its fit on 2 (virtual) GPUs
    r = subprocess.run([os.path.join(BUILD, prog)], capture_output=True, text=True, timeout=300, env=env)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"PASS {prog}" in r.stdout
    if prog == "test_linear":
        # the reference's own 100-call scalar loop (test/splpak_test_linear.f90:66-72) is a host computation
        secs = float(re.search(r"scalar evaluate loop seconds =\s*(\S+)", r.stdout).group(1))
        assert secs < 1e-3, secs
    if prog == "test_info":
        # `reserr` surfaced by last_fit_info against the oracle's value for the same inputs (case 2d16)
        from oracle import binding
        from tests.cases import CASES, make_inputs
        if not binding.port_available():
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "port"])
        inp = make_inputs(CASES["2d16"])
        P = binding.Port()
        _, e0, _ = P.fit(2, inp["xdata"], inp["ydata"], inp["wdata"], inp["xmin"], inp["xmax"], inp["nodes"], 1.0)
        got = float(re.search(r"reserr =\s*(\S+)", r.stdout).group(1))
        assert e0 == 0 and abs(got - P.last_reserr) <= 1e-9 * P.last_reserr, (got, P.last_reserr)
        assert "2-GPU vs 1-GPU coefficients" in r.stdout


@pytest.mark.gpu
def test_fortran_set_gpus_reaches_the_distributed_nested_dissection():
    """what a Fortran caller reaches through `call solver%set_gpus(n)` (splpak_fit_multi_f64) factors through
    the nested-dissection tree, distributed over the GPUs, whenever the single-GPU fit of the grid would (forced here with
    SPLPAK_ND=1 on test_info's 16 x 16 grid; by default from 4 096 columns on).  The program itself checks the 2-GPU fit
    against the 1-GPU fit at 1e-12; the library's debug line shows which factorisation ran."""
    _ensure_built()
    env = dict(os.environ, SPLPAK_VIRTUAL_GPUS="1", SPLPAK_ND="1", SPLPAK_DEBUG="1")
    r = subprocess.run([os.path.join(BUILD, "test_info")], capture_output=True, text=True, timeout=300, env=env)
    print(r.stdout[-1500:], r.stderr[-1500:])
    assert r.returncode == 0 and "PASS test_info" in r.stdout
    assert "one rank of a multi-GPU fit" in r.stderr
    assert "2-GPU vs 1-GPU coefficients" in r.stdout
