"""Shared definition of the parity-test cases (inputs only).

Inputs are regenerated from the seeded Park-Miller stream of ``splpak_amd.synth``
(SURVEY.md section 8d), so the golden fixtures under ``tests/golden/`` hold only
the reference's OUTPUTS.  ``oracle/gen_golden.py`` and the tests both build their
inputs from :func:`make_inputs`, which keeps the two in lock-step.
"""
from __future__ import annotations

import numpy as np

from splpak_amd.synth import synth_points, synth_queries

# name -> spec.  `variant` tweaks the synthetic data:
#   dense     : the plain stream on [0,1]^d
#   zero_w    : every third weight is exactly 0 (skipped rows, src/splpak.F90:799)
#   outside   : points stretched to [-0.15,1.15]^d (linear-extrapolation basis, histogram quirk :899)
#   box       : non-unit, anisotropic domain
#   clustered : x -> x^2 so that large parts of the grid are data sparse (constraint rows :921-1046)
#   linear    : the reference's own known-answer test (test/splpak_test_linear.f90): y = 2x, w = 1
CASES = {
    # BASELINE config 1
    "c1_1d16":        dict(ndim=1, nodes=[16], m=1000, weighted=True, xtrap=1.0, variant="dense"),
    "c1_1d16_xt0":    dict(ndim=1, nodes=[16], m=1000, weighted=True, xtrap=0.0, variant="dense"),
    "ref_linear":     dict(ndim=1, nodes=[10], m=20, weighted=True, xtrap=1.0, variant="linear"),
    "1d_sparse":      dict(ndim=1, nodes=[12], m=9, weighted=False, xtrap=1.0, variant="clustered"),
    "2d8":            dict(ndim=2, nodes=[8, 8], m=3000, weighted=True, xtrap=1.0, variant="dense"),
    "2d8_cc":         dict(ndim=2, nodes=[8, 8], m=3000, weighted=False, xtrap=1.0, variant="dense"),
    "2d16":           dict(ndim=2, nodes=[16, 16], m=10000, weighted=True, xtrap=1.0, variant="dense"),
    "2d16_sparse":    dict(ndim=2, nodes=[16, 16], m=300, weighted=True, xtrap=1.0, variant="dense"),
    "2d16_zero_w":    dict(ndim=2, nodes=[16, 16], m=6000, weighted=True, xtrap=1.0, variant="zero_w"),
    "2d16_outside":   dict(ndim=2, nodes=[16, 16], m=6000, weighted=True, xtrap=0.5, variant="outside"),
    "2d_aniso_box":   dict(ndim=2, nodes=[7, 19], m=4000, weighted=True, xtrap=2.0, variant="box"),
    "2d32":           dict(ndim=2, nodes=[32, 32], m=20000, weighted=True, xtrap=1.0, variant="dense"),
    "2d32_cc_xt0":    dict(ndim=2, nodes=[32, 32], m=20000, weighted=False, xtrap=0.0, variant="dense"),
    "2d_min_nodes":   dict(ndim=2, nodes=[4, 5], m=500, weighted=True, xtrap=1.0, variant="outside"),
    "3d8":            dict(ndim=3, nodes=[8, 8, 8], m=10000, weighted=True, xtrap=1.0, variant="dense"),
    "3d8_sparse":     dict(ndim=3, nodes=[8, 8, 8], m=600, weighted=True, xtrap=1.0, variant="dense"),
    "3d8_cc_clust":   dict(ndim=3, nodes=[8, 8, 8], m=8000, weighted=False, xtrap=1.0, variant="clustered"),
    "3d_aniso":       dict(ndim=3, nodes=[5, 9, 6], m=4000, weighted=True, xtrap=1.0, variant="box"),
    "3d12":           dict(ndim=3, nodes=[12, 12, 12], m=4000, weighted=True, xtrap=1.0, variant="dense"),
    # SURVEY 8c's 3-D 16^3 (4096 columns) case: the largest 3-D grid the dense reference reaches (~2 h)
    "3d16":           dict(ndim=3, nodes=[16, 16, 16], m=10000, weighted=True, xtrap=1.0, variant="dense", slow=True),
    "4d4":            dict(ndim=4, nodes=[4, 4, 4, 4], m=3000, weighted=True, xtrap=1.0, variant="dense"),
    "4d5_cc":         dict(ndim=4, nodes=[5, 5, 5, 5], m=4000, weighted=False, xtrap=1.0, variant="outside"),
    "4d6":            dict(ndim=4, nodes=[6, 6, 6, 6], m=5000, weighted=True, xtrap=1.0, variant="dense"),
    # BASELINE config 2's node grid (64x64 = 4096 columns); slow on the reference (~1 h)
    "2d64_c2grid":    dict(ndim=2, nodes=[64, 64], m=12000, weighted=False, xtrap=1.0, variant="dense", slow=True),
}

# More than four dimensions: the reference takes any ndim >= 1 (src/splpak.F90:716-722); the HIP kernels are written for 1..4, the
# Fortran module routes such calls to its host solver (round 6).  Kept apart from CASES: the GPU tests iterate over those.
HOST_CASES = {
    "5d4":            dict(ndim=5, nodes=[4, 5, 4, 4, 4], m=4000, weighted=True, xtrap=1.0, variant="dense"),
}

NQ = 256  # evaluation queries stored per nderiv pattern


def domain(spec):
    nd = spec["ndim"]
    if spec["variant"] == "box":
        xmin = np.array([-1.0, 2.0, 10.0, -3.0][:nd])
        xmax = xmin + np.array([2.0, 0.5, 7.0, 1.25][:nd])
    else:
        xmin = np.zeros(nd)
        xmax = np.ones(nd)
    return xmin, xmax


def make_inputs(spec):
    """-> dict(ndim, xdata (m,ndim), ydata, wdata|None, xmin, xmax, nodes, xtrap)."""
    nd, m = spec["ndim"], spec["m"]
    xmin, xmax = domain(spec)
    var = spec["variant"]
    if var == "linear":
        # test/splpak_test_linear.f90:44-49
        x = (np.arange(m, dtype=np.float64) / float(m - 1)).reshape(m, 1)
        y = 2.0 * x[:, 0]
        w = np.ones(m)
    else:
        x, y, w = synth_points(nd, m)
        if var == "outside":
            x = x * 1.3 - 0.15
        elif var == "clustered":
            x = x * x
        elif var == "box":
            x = xmin + x * (xmax - xmin)
        elif var == "zero_w":
            w = w.copy()
            w[::3] = 0.0
    return dict(ndim=nd, xdata=np.ascontiguousarray(x), ydata=y,
                wdata=(np.ascontiguousarray(w) if spec["weighted"] else None),
                xmin=xmin, xmax=xmax, nodes=np.array(spec["nodes"], dtype=np.int32),
                xtrap=float(spec["xtrap"]))


def make_queries(spec, nq=NQ):
    """Queries in and outside the grid, plus exact node / boundary locations."""
    nd, m = spec["ndim"], spec["m"]
    xmin, xmax = domain(spec)
    u = synth_queries(nd, nq, m)
    q = xmin + (u * 1.5 - 0.25) * (xmax - xmin)
    nodes = np.array(spec["nodes"])
    dx = (xmax - xmin) / (nodes - 1)
    # first rows: exact node locations (incl. both boundaries) to pin the open/closed
    # ends of the piecewise definitions
    k = 0
    for idx in range(min(nq // 4, int(nodes.max()))):
        q[k] = xmin + np.minimum(idx, nodes - 1) * dx
        k += 1
    q[k] = xmin
    q[k + 1] = xmax
    q[k + 2] = xmin - 2.5 * dx
    q[k + 3] = xmax + 2.5 * dx
    return np.ascontiguousarray(q)


def nderiv_patterns(nd):
    """All 3^d patterns for d <= 2, a fixed sample for d >= 3."""
    if nd == 1:
        return [[0], [1], [2]]
    if nd == 2:
        return [[a, b] for b in range(3) for a in range(3)]
    if nd == 3:
        return [[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [2, 0, 0], [0, 2, 0], [0, 0, 2],
                [1, 1, 0], [1, 0, 1], [0, 1, 1], [2, 1, 0], [1, 1, 1], [2, 2, 2]]
    if nd == 4:
        return [[0, 0, 0, 0], [1, 0, 0, 0], [0, 0, 0, 1], [0, 2, 0, 0], [1, 1, 0, 0],
                [0, 1, 0, 1], [1, 1, 1, 1], [2, 0, 1, 0], [2, 2, 2, 2]]
    return [[0] * nd, [1] + [0] * (nd - 1), [0] * (nd - 1) + [2], [1, 1] + [0] * (nd - 2), [2] * nd]
