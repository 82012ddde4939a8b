"""The Fortran drop-in `splpak_module` (splpak_amd/fortran) over the C ABI.

CPU tier : the module and its test programs build with amdflang, link against the HIP
           library, and FAIL LOUDLY without a GPU (no host fallback).
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
ALL_PROGS = PROGS + ["test_evalfix"]


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
    fx = [os.path.join(ROOT, "tests", "golden", f"eval_{n}.txt") for n in ("3d12", "4d6", "3d_aniso")]
    r = subprocess.run([os.path.join(BUILD, "test_evalfix")] + fx, capture_output=True, text=True, timeout=300)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PASS test_evalfix" in r.stdout


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
