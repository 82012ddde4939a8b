"""The Fortran drop-in `splpak_module` (splpak_amd/fortran) over the C ABI.

CPU tier : the module and its test programs build with amdflang, link against the HIP
           library, and FAIL LOUDLY without a GPU (no host fallback).
GPU tier : the three test programs -- re-creations of the reference's own tests
           (test/splpak_test_linear.f90, test/splpak_test.f90) plus an API/ierror test --
           pass on the MI355X.
"""
import os
import subprocess

import pytest

from tests.conftest import ROOT

FDIR = os.path.join(ROOT, "splpak_amd", "fortran")
BUILD = os.path.join(FDIR, "build")
PROGS = ["test_linear", "test_noisy", "test_api"]


def _ensure_built():
    if all(os.path.exists(os.path.join(BUILD, p)) for p in PROGS):
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


@pytest.mark.gpu
@pytest.mark.parametrize("prog", PROGS)
def test_fortran_program_on_gpu(prog):
    _ensure_built()
    r = subprocess.run([os.path.join(BUILD, prog)], capture_output=True, text=True, timeout=300)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"PASS {prog}" in r.stdout
