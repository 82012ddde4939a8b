#!/usr/bin/env python3
"""One-off randomized parity sweep of the HIP fit / evaluation against the CPU oracle (not part of the
test tier): random dimensions, node counts, boxes, weights, xtrap, points outside the box.
    tools/fuzz_parity.py [seed] [trials] [big]
"big": larger grids (up to ~6000 columns, several 256-column blocks: the two-ended factorisation of narrow bands,
separators of 1..8 blocks, padded and unpadded orders) against the BANDED CPU restatement (oracle/splpak_banded.c,
itself pinned to the reference's goldens), which finishes these sizes in seconds."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from splpak_amd import capi
from oracle import binding

port = binding.Port()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 2026)
worst, fails = 0.0, 0
BIG = len(sys.argv) > 3 and sys.argv[3] == "big"
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    nd = int(rng.integers(1, 5))
    hi_nodes = ({1: 3000, 2: 70, 3: 18, 4: 8} if BIG else {1: 40, 2: 14, 3: 8, 4: 6})[nd]
    nodes = [int(rng.integers(4, hi_nodes + 1)) for _ in range(nd)]
    ncol = int(np.prod(nodes))
    m = int(rng.integers(max(ncol // 2, 8), 6 * ncol + 50)) if not BIG else int(rng.integers(2 * ncol, 8 * ncol + 50))
    lo = rng.normal(size=nd)
    hi = lo + 0.2 + 3.0 * rng.random(nd)
    spread = 1.0 + 0.3 * rng.random()
    x = lo + (hi - lo) * (0.5 + spread * (rng.random((m, nd)) - 0.5))
    if rng.random() < 0.3:                       # clustered data -> sparse areas
        x[: m // 2] = lo + (hi - lo) * 0.3 * rng.random((m // 2, nd))
    y = np.sin(3.0 * ((x - lo) / (hi - lo)).sum(axis=1)) + 0.1 * rng.standard_normal(m)
    w = None if rng.random() < 0.3 else 0.2 + rng.random(m)
    if w is not None and rng.random() < 0.5:
        w[rng.random(m) < 0.1] = 0.0
    xtrap = float(rng.choice([0.0, 0.3, 1.0, 2.5]))
    if os.environ.get("FUZZ_ONLY") and trial != int(os.environ["FUZZ_ONLY"]) and not os.environ.get("FUZZ_EXACT"):
        rng.random((300, nd)); [rng.integers(0, 3) for _ in range(nd)]        # (the draws the comparison below would make)
        continue
    # generous workspace: the reference's own size check (suprls 32 -> 107, :1443-1454) is not under test
    if BIG:
        c0, e0, ib = port.fit_banded(nd, x, y, w, lo, hi, nodes, xtrap)
        w0 = None
    else:
        c0, e0, w0 = port.fit(nd, x, y, w, lo, hi, nodes, xtrap, nwrk=ncol * (ncol + 1) + 64)
    c1, e1, h1, i1 = capi.fit(nd, x, y, w, lo, hi, nodes, xtrap, want_hist=True)
    tag = f"trial {trial:3d} nd={nd} nodes={nodes} m={m} xtrap={xtrap} weighted={w is not None}"
    if e0 != e1:
        # the reference reports 107 only on an EXACT zero pivot and otherwise returns whatever a numerically
        # singular system gives (documented deviation): accept hip = 107 when the oracle's coefficients blew up
        singular = e0 == 0 and e1 == 107 and np.max(np.abs(c0[:ncol])) > 1e4 * np.max(np.abs(y))
        print(tag, f"ierror oracle {e0} hip {e1}", "(numerically singular: oracle max|coef| = %.1e)" % np.max(np.abs(c0[:ncol])) if singular else "<-- FAIL")
        fails += 0 if singular else 1
        if not singular:
            print(f"    hip says: {capi.last_error()!r}; info steps {int(i1[2])} last correction {i1[3]:.2e} omega {i1[9]:.2e}; oracle max|coef| {np.max(np.abs(c0[:ncol])):.2e}, max|y| {np.max(np.abs(y)):.2e}")
        continue
    if e0 != 0:
        continue
    rel = np.max(np.abs(c1[:ncol] - c0[:ncol])) / max(np.max(np.abs(c0[:ncol])), 1e-300)
    q = lo + (hi - lo) * (rng.random((300, nd)) * 1.4 - 0.2)
    p = [int(rng.integers(0, 3)) for _ in range(nd)]
    v1, _ = capi.evaluate(nd, q, p, c0, lo, hi, nodes)
    v0, _ = port.evaluate(nd, q, p, c0, lo, hi, nodes)
    dxin = (np.array(nodes) - 1) / (hi - lo)
    scale = max(np.max(np.abs(v0)), np.max(np.abs(c0)) * float(np.prod(dxin ** np.array(p))))
    erel = np.max(np.abs(v1 - v0)) / scale
    # coefficients far larger than the data: an ill-conditioned (or numerically singular, non-unique) problem,
    # where the reference's own result is only accurate to cond*eps -- compare residual-equivalent answers loosely
    illposed = np.max(np.abs(c0[:ncol])) > 100.0 * np.max(np.abs(y))
    if illposed:
        print(tag, f"ill-conditioned (oracle max|coef| {np.max(np.abs(c0[:ncol])):.1e}): coef rel {rel:.2e} not judged")
        rel = 0.0
    if BIG and rel > 1e-10 and (abs(i1[8] - ib[8]) <= 1e-12 * ib[8] or (abs(i1[8] - ib[8]) <= 1e-11 * ib[8] and i1[9] < 1e-12)):
        # two solutions with the same least-squares objective to 12 digits (11 when the GPU's measured backward error is at
        # rounding level): a flat direction of an ill-conditioned problem (both solvers stop at cond*eps), not a discrepancy
        # between them
        print(tag, f"flat direction: coef rel {rel:.2e} at equal residual norm {ib[8]:.12e}; hip steps {int(i1[2])}, last correction {i1[3]:.1e}")
        rel = 0.0
    if BIG and rel > 1e-10 and i1[8] < ib[8] * (1.0 - 1e-9) and i1[9] < 1e-12:
        # the GPU's solution has the SMALLER residual norm and a componentwise backward error at rounding level: the banded CPU
        # comparator stopped short on this problem (1-D, ~2 000 nodes, xtrap 2.5: round 3, seed 778 trial 51)
        print(tag, f"comparator short of the minimum: coef rel {rel:.2e}, residual norm hip {i1[8]:.9e} < cpu {ib[8]:.9e}, hip backward error {i1[9]:.1e}")
        rel = 0.0
    worst = max(worst, rel)
    bad = rel > 1e-10 or erel > 1e-12 or (w0 is not None and xtrap != 0 and np.max(np.abs(h1[:ncol] - w0[:ncol])) > 1e-12 * max(np.max(np.abs(w0[:ncol])), 1))
    if bad:
        fails += 1
        print(tag, f"coef rel {rel:.2e} eval rel {erel:.2e}  <-- FAIL")
        if BIG:
            print(f"    hip: ierr {e1} steps {int(i1[2])} last correction {i1[3]:.2e} reserr {i1[8]:.12e} omega {i1[9]:.2e}; banded CPU reserr {ib[8]:.12e}")
print(f"worst coefficient deviation {worst:.2e}; failures {fails}")
sys.exit(1 if fails else 0)
