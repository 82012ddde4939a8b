#!/usr/bin/env python3
"""Randomised sweep of the iterative solve against the factorisation on the GPU (one-off, not part of the test tier): random dimension
counts, node counts, boxes, weights (some zero), xtrap, points outside the box, a leading dimension larger than ndim, REAL32.  The
iteration alone (solver = pcg: 4-D plans of it never assemble the normal equations) must return the factorisation's coefficients
(<= 1e-10) or 107; in front of the factorisation (pcg+direct with pcg_always) the fit must always succeed.
    tools/pcg/fuzz_pcg.py [seed] [trials]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from splpak_amd import capi

def fit_env(env, *a, **k):
    old = {q: os.environ.get(q) for q in env}
    os.environ.update(env)
    try:
        return capi.fit(*a, **k)
    finally:
        for q, v in old.items():
            if v is None: os.environ.pop(q, None)
            else: os.environ[q] = v

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 6)
worst, fails, gaveup = 0.0, 0, 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    nd = int(rng.choice([1, 2, 3, 4, 4, 4]))
    hi_nodes = {1: 200, 2: 40, 3: 14, 4: 9}[nd]
    nodes = [int(rng.integers(4, hi_nodes + 1)) for _ in range(nd)]
    ncol = int(np.prod(nodes))
    dense = rng.random() < 0.6
    m = int(rng.integers(6 * ncol, 40 * ncol)) if dense else int(rng.integers(max(ncol // 2, 8), 6 * ncol + 50))
    lo = rng.normal(size=nd)
    hi = lo + 0.2 + 3.0 * rng.random(nd)
    spread = 1.0 + 0.3 * rng.random()
    ldx = nd + int(rng.integers(0, 3))
    xf = rng.random((m, ldx))
    x = xf.copy()
    x[:, :nd] = lo + (hi - lo) * (0.5 + spread * (xf[:, :nd] - 0.5))
    if rng.random() < 0.25:
        x[: m // 2, :nd] = lo + (hi - lo) * 0.3 * rng.random((m // 2, nd))
    y = np.sin(3.0 * ((x[:, :nd] - lo) / (hi - lo)).sum(axis=1)) + 0.1 * rng.standard_normal(m)
    w = None if rng.random() < 0.3 else 0.2 + rng.random(m)
    if w is not None and rng.random() < 0.5:
        w[rng.random(m) < 0.1] = 0.0
    xtrap = float(rng.choice([0.0, 0.3, 1.0, 1.0, 2.5]))
    r32 = rng.random() < 0.15
    args = (nd, x, y, w, lo, hi, nodes, xtrap)
    kw = dict(l1xdat=ldx, want_hist=True, real32=r32)
    c0, e0, h0, i0 = fit_env({"SPLPAK_SOLVER": "direct"}, *args, **kw)
    c1, e1, h1, i1 = fit_env({"SPLPAK_SOLVER": "pcg"}, *args, **kw)
    c2, e2, h2, i2 = fit_env({"SPLPAK_SOLVER": "pcg+direct", "SPLPAK_PCG_ALWAYS": "1"}, *args, **kw)
    tag = f"trial {trial:3d} nd={nd} nodes={nodes} m={m} ldx={ldx} xtrap={xtrap} weighted={w is not None} real32={r32} rows/col {i0[1] / ncol:.2f}"
    tol = 2e-4 if r32 else 1e-10
    if e0 != 0:
        ok = e1 in (e0, 107) and e2 == e0
        print(tag, f"factorisation ierror {e0}; iteration {e1}; in front {e2}", "" if ok else "<-- FAIL")
        fails += 0 if ok else 1
        continue
    sc = max(np.max(np.abs(c0[:ncol])), 1e-300)
    illposed = np.max(np.abs(c0[:ncol])) > 100.0 * np.max(np.abs(y))
    r2 = np.max(np.abs(c2[:ncol] - c0[:ncol])) / sc if e2 == 0 else np.inf
    htol = (2e-6 if r32 else 1e-12) * max(np.max(np.abs(h0[:ncol])), 1)
    hbad = (h1 is not None and np.max(np.abs(h1[:ncol] - h0[:ncol])) > htol) or (e2 == 0 and h2 is not None and np.max(np.abs(h2[:ncol] - h0[:ncol])) > htol)
    if e1 == 107:
        gaveup += 1
        r1 = 0.0
    elif e1 != 0:
        r1 = np.inf
    else:
        r1 = np.max(np.abs(c1[:ncol] - c0[:ncol])) / sc
    if illposed:
        print(tag, f"ill-conditioned (max|coef| {sc:.1e}): {r1:.1e} / {r2:.1e} not judged; ierror {e1} {e2}")
        if e2 != 0: fails += 1
        continue
    worst = max(worst, r1, r2)
    bad = r1 > tol or r2 > tol or hbad or e2 != 0
    if bad or e1 == 107:
        print(tag, f"iteration alone: ierror {e1} {r1:.1e} (backward error {i1[9]:.1e}); in front: ierror {e2} {r2:.1e}; histogram {'differs' if hbad else 'same'}", "<-- FAIL" if bad else "(gave up: 107)")
    fails += 1 if bad else 0
print(f"worst coefficient deviation {worst:.2e}; iteration alone gave up {gaveup} times; failures {fails}")
sys.exit(1 if fails else 0)
