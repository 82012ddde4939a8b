#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- text fixtures for the Fortran HOST solver test (test_hostfit.f90).

Re-writes the reference's golden OUTPUTS (tests/golden/<case>.npz: coefficients and, for xtrap /= 0, the
sparse-area histogram the reference leaves in work(1:ncol)) as plain text together with the case's
parameters.  The INPUTS are not stored: the Fortran program regenerates them from the seeded Park-Miller
stream of SURVEY 8d exactly as tests/cases.py does (variants dense / zero_w / outside / clustered on the unit
box).  No reference code runs here: data only.

    tests/golden/fit_<case>.txt:  ndim | nodes | m weighted(0/1) xtrap variant | ncol | coef(1:ncol) | nhist | hist

    python oracle/gen_fit_fixture.py [case ...]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests.cases import CASES as _GPU_CASES, HOST_CASES  # noqa: E402

CASES = {**_GPU_CASES, **HOST_CASES}

VARIANT = {"dense": 0, "zero_w": 1, "outside": 2, "clustered": 3}
DEFAULT = ["c1_1d16", "c1_1d16_xt0", "1d_sparse", "2d8_cc", "2d16", "2d16_sparse", "2d16_zero_w", "2d16_outside",
           "2d32_cc_xt0", "3d8", "3d8_sparse", "3d8_cc_clust", "3d12", "4d4", "4d5_cc", "5d4"]


def write_case(name):
    spec = CASES[name]
    if spec["variant"] not in VARIANT:
        raise SystemExit(f"{name}: variant {spec['variant']} has no Fortran generator")
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    path = os.path.join(ROOT, "tests", "golden", f"fit_{name}.txt")
    with open(path, "w") as f:
        f.write(f"{spec['ndim']}\n")
        f.write(" ".join(str(int(v)) for v in spec["nodes"]) + "\n")
        f.write(f"{spec['m']} {1 if spec['weighted'] else 0} {spec['xtrap']:.17g} {VARIANT[spec['variant']]}\n")
        f.write(f"{g['coef'].size}\n")
        for c in g["coef"]:
            f.write(f"{c:.17g}\n")
        f.write(f"{g['hist'].size}\n")
        for h in g["hist"]:
            f.write(f"{h:.17g}\n")
    print(f"{path}: {os.path.getsize(path) / 1024:.0f} KB")


if __name__ == "__main__":
    for n in (sys.argv[1:] or DEFAULT):
        write_case(n)
