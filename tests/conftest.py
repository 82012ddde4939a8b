"""pytest configuration: the `gpu` marker and shared fixtures.

`-m "not gpu"`: oracle vs golden vectors, host logic, C-ABI symbol checks (no GPU).
`-m gpu`     : the parity tests proper, through the C ABI on a real MI355X.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch  # noqa: F401  -- before the HIP library is loaded: one HIP runtime per process (see splpak_amd/capi.py)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: long-running CPU oracle case")


@pytest.fixture(scope="session")
def port():
    """The C restatement of the reference algorithm (oracle/splpak_oracle.c)."""
    from oracle import binding
    if not binding.port_available():
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "port"])
    return binding.Port()


@pytest.fixture(scope="session")
def reference():
    """The real reference built into oracle/_ref (None when it cannot be built here)."""
    from oracle import binding
    if not binding.ref_available() and os.path.exists("/root/reference/src/splpak.F90"):
        subprocess.call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    return binding.Reference() if binding.ref_available() else None


def load_golden(name):
    path = os.path.join(GOLDEN, name + ".npz")
    if not os.path.exists(path):
        pytest.skip(f"golden fixture {name}.npz not generated")
    return np.load(path)


def relmax(a, b):
    """max-norm relative error, the parity metric of SURVEY.md section 0.3."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (den if den > 0 else 1.0))
