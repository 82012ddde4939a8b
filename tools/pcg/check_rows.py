import sys, numpy as np, scipy.sparse as sp, scipy.sparse.linalg as sla
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools/pcg')
from rows import data_rows, constraint_rows
from tests.cases import CASES, make_inputs
import os
for name in ['2d16', '2d16_sparse', '3d8', '3d8_sparse', '4d6', '1d_sparse','3d8_cc_clust']:
    spec = CASES[name]; inp = make_inputs(spec)
    x = inp['xdata']; w = inp['wdata'] if inp['wdata'] is not None else np.ones(len(x))
    A = data_rows(x, w, inp['xmin'], inp['xmax'], inp['nodes'])
    C, hist, spn = constraint_rows(x, w, inp['xmin'], inp['xmax'], inp['nodes'], inp['xtrap'])
    M = sp.vstack([A, C]).toarray()
    b = np.concatenate([w * inp['ydata'], np.zeros(C.shape[0])])
    sol = np.linalg.lstsq(M, b, rcond=None)[0]
    g = np.load(f'/root/repo/tests/golden/{name}.npz')
    coef = g['coef'] if 'coef' in g else g[g.files[0]]
    print(name, C.shape[0], np.abs(sol - coef).max() / np.abs(coef).max(), np.linalg.cond(M))
