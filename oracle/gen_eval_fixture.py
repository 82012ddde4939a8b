#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- text fixtures for the Fortran scalar `evaluate` (splfe / splde).

The reference's outputs are already committed as tests/golden/<case>.npz (oracle/gen_golden.py, from the
unmodified reference built into oracle/_ref).  The Fortran test program cannot read .npz, so this script
re-writes a few of them as plain text, inputs included:

    tests/golden/eval_<case>.txt
        ndim
        nodes(1:ndim)
        xmin(1:ndim)
        xmax(1:ndim)
        ncol, npat, nq
        coef(1:ncol)                        one per line   (the reference's fitted coefficients)
        per pattern:  nderiv(1:ndim)        then nq lines  x(1:ndim)  value   (the reference's splde)

Queries are tests.cases.make_queries(spec): exact node and boundary locations first, points outside the
grid, then the seeded stream on [-0.25, 1.25]^d of the box.  No reference code runs here: data only.

    python oracle/gen_eval_fixture.py [case ...]        default: 3d12 4d6 3d_aniso 5d4
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests.cases import CASES as _GPU_CASES, HOST_CASES, make_inputs, make_queries  # noqa: E402

CASES = {**_GPU_CASES, **HOST_CASES}

NQ_KEEP = 96          # the node / boundary / outside queries come first in make_queries


def write_case(name):
    spec = CASES[name]
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    inp = make_inputs(spec)
    q = make_queries(spec)[:NQ_KEEP]
    pats, vals = g["patterns"], g["values"][:, :NQ_KEEP]
    nd = inp["ndim"]
    path = os.path.join(ROOT, "tests", "golden", f"eval_{name}.txt")
    with open(path, "w") as f:
        f.write(f"{nd}\n")
        f.write(" ".join(str(int(v)) for v in inp["nodes"]) + "\n")
        f.write(" ".join(f"{v:.17g}" for v in inp["xmin"]) + "\n")
        f.write(" ".join(f"{v:.17g}" for v in inp["xmax"]) + "\n")
        f.write(f"{g['coef'].size} {len(pats)} {q.shape[0]}\n")
        for c in g["coef"]:
            f.write(f"{c:.17g}\n")
        for p, v in zip(pats, vals):
            f.write(" ".join(str(int(k)) for k in p) + "\n")
            for x, y in zip(q, v):
                f.write(" ".join(f"{t:.17g}" for t in x) + f" {y:.17g}\n")
    print(f"{path}: {os.path.getsize(path) / 1024:.0f} KB, {len(pats)} patterns x {q.shape[0]} queries")


if __name__ == "__main__":
    for n in (sys.argv[1:] or ["3d12", "4d6", "3d_aniso", "5d4"]):
        write_case(n)
