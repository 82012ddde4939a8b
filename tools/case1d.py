#!/usr/bin/env python3
"""1-D many-node cases where the fuzz sweep saw the GPU fit and the banded CPU restatement disagree: compare both
with the dense (Householder) restatement of the reference.  tools/case1d.py [nodes] [m] [xtrap] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from splpak_amd import capi
from oracle import binding
port = binding.Port()
nodes = [int(sys.argv[1]) if len(sys.argv) > 1 else 1806]
m = int(sys.argv[2]) if len(sys.argv) > 2 else 4995
xtrap = float(sys.argv[3]) if len(sys.argv) > 3 else 2.5
rng = np.random.default_rng(int(sys.argv[4]) if len(sys.argv) > 4 else 5)
nd, ncol = 1, nodes[0]
lo = rng.normal(size=nd); hi = lo + 0.2 + 3.0 * rng.random(nd)
x = lo + (hi - lo) * (0.5 + 1.2 * (rng.random((m, nd)) - 0.5))
y = np.sin(3.0 * ((x - lo) / (hi - lo)).sum(axis=1)) + 0.1 * rng.standard_normal(m)
cb, eb, ib = port.fit_banded(nd, x, y, None, lo, hi, nodes, xtrap)
cd, ed, wd = port.fit(nd, x, y, None, lo, hi, nodes, xtrap, nwrk=ncol * (ncol + 1) + 64)
cg, eg, hg, info = capi.fit(nd, x, y, None, lo, hi, nodes, xtrap, want_hist=True)
den = np.max(np.abs(cd[:ncol]))
print(f"ierror dense {ed} banded {eb} gpu {eg}; max|coef| {den:.3e}, max|y| {np.max(np.abs(y)):.3e}")
print(f"gpu    vs dense : {np.max(np.abs(cg[:ncol]-cd[:ncol]))/den:.3e}")
print(f"banded vs dense : {np.max(np.abs(cb[:ncol]-cd[:ncol]))/den:.3e}")
print(f"gpu    vs banded: {np.max(np.abs(cg[:ncol]-cb[:ncol]))/den:.3e}")
print("gpu info:", info)
print("banded info:", ib)
