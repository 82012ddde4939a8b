"""Research: FD-preconditioned CG on N = A^T A + C^T C applied matrix-free (rows in CSR), larger grids.
usage: exp5.py ndim nodes ppc [tol]"""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools/pcg')
from rows import data_rows, constraint_rows
from fd import FD
from splpak_amd.synth import synth_points

d, nod, ppc = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
tol = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-12
nodes = np.array([nod] * d); m = int(ppc * (nod - 1) ** d)
t = time.time()
x, y, w = synth_points(d, m)
xmin = np.zeros(d); xmax = np.ones(d)
A = data_rows(x, w, xmin, xmax, nodes); At = A.T.tocsr()
C, hist, spn = constraint_rows(x, w, xmin, xmax, nodes, 1.0); Ct = C.T.tocsr()
n = A.shape[1]
print(f'build {time.time()-t:.1f}s n={n} m={m} nnzA={A.nnz} cons rows={C.shape[0]}', flush=True)
r = At @ (w * y)
sub = np.array(np.unravel_index(np.arange(n), nodes[::-1])).T[:, ::-1]
onb = ((sub == 0) | (sub == nodes - 1)).sum(1)
wt = w.sum() / np.prod(nodes - 1)
expect = wt * 0.5 ** onb
dcw2 = np.where(spn, (expect - hist) ** 2, 0.0)
rho = (w ** 2).sum()
lam = dcw2[onb == 0].mean(); lam1 = dcw2[onb == 1].mean(); qb = lam1 / lam
print(f'sparse frac {spn.mean():.3f} rho {rho:.4g} lam {lam:.4g} qb {qb:.3f}', flush=True)
fd = FD(list(nodes), rho, lam, qb)
op = lambda v: At @ (A @ v) + Ct @ (C @ v)
xs = np.zeros(n); res = r.copy(); z = fd.solve(res); p = z.copy(); rz = res @ z; rz0 = rz
marks = {}
t = time.time()
for it in range(1, 4001):
    Np = op(p); a = rz / (p @ Np); xs += a * p; res -= a * Np
    z = fd.solve(res); rz2 = res @ z
    rel = np.sqrt(rz2 / rz0)
    for th in (1e-2, 1e-4, 1e-6, 1e-8, 1e-10, 1e-12):
        if rel < th and th not in marks: marks[th] = it
    if it % 50 == 0: print(it, f'{rel:.3e}', f'{time.time()-t:.0f}s', flush=True)
    if rel < tol: break
    p = z + (rz2 / rz) * p; rz = rz2
print('its', it, 'marks', marks, flush=True)
# true gradient residual of the rows at xs
g = At @ (w * y - A @ xs) - Ct @ (C @ xs)
den = abs(At) @ (abs(A) @ abs(xs)) + abs(Ct) @ (abs(C) @ abs(xs)) + abs(At @ (w * y))
print('backward error omega', np.max(np.abs(g) / den))
