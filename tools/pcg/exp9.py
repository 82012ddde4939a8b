"""Research: overlapping boxes (two lattices of 4^4 boxes, the second shifted by 2 nodes in every dimension) in the regime where aligned boxes
+ separable still stagnate (0.5 .. 1.4 constraint rows per column).  4-D 12^4."""
import sys, time, numpy as np, scipy.sparse as sp, scipy.linalg as la
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools/pcg')
from rows import data_rows, constraint_rows
from exp6 import FD2
from splpak_amd.synth import synth_points

d, nod, ppc = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
nodes = np.array([nod] * d); m = int(ppc * (nod - 1) ** d)
x, y, w = synth_points(d, m)
xmin = np.zeros(d); xmax = np.ones(d)
A = data_rows(x, w, xmin, xmax, nodes); At = A.T.tocsr()
C, hist, spn = constraint_rows(x, w, xmin, xmax, nodes, 1.0); Ct = C.T.tocsr()
n = A.shape[1]
N = (At @ A + Ct @ C).tocsr()
r = At @ (w * y)
sub = np.array(np.unravel_index(np.arange(n), nodes[::-1])).T[:, ::-1]
onb = ((sub == 0) | (sub == nodes - 1)).sum(1)
wt = w.sum() / np.prod(nodes - 1)
expect = wt * 0.5 ** onb
dcw2 = np.where(spn, (expect - hist) ** 2, 0.0)
rho = (w ** 2).sum(); lam = dcw2.mean()
print(f'n={n} m={m} sparse frac {spn.mean():.3f} rows/col {C.shape[0] / n:.2f}', flush=True)
fd = FD2(list(nodes), rho, lam, 0.5, 'K0')
def lattice(shift, bs=4):
    bid = np.zeros(n, dtype=np.int64); mul = 1
    for k in range(d):
        bid += ((sub[:, k] + shift) // bs) * mul; mul *= (nod + shift) // bs + 2
    order = np.argsort(bid, kind='stable'); bounds = np.flatnonzero(np.diff(bid[order])) + 1
    groups = np.split(order, bounds)
    chol = [la.cho_factor(N[g][:, g].toarray()) for g in groups]
    def ap(v):
        out = np.zeros_like(v)
        for g, c in zip(groups, chol): out[g] = la.cho_solve(c, v[g])
        return out
    return ap
def run(name, Minv, tol=1e-10, maxit=1500):
    xs = np.zeros(n); res = r.copy(); z = Minv(res); p = z.copy(); rz = res @ z; rz0 = rz; marks = {}
    for it in range(1, maxit + 1):
        Np = N @ p; a = rz / (p @ Np); xs += a * p; res -= a * Np
        z = Minv(res); rz2 = res @ z; rel = np.sqrt(abs(rz2) / rz0)
        for th in (1e-2, 1e-4, 1e-6, 1e-8, 1e-10):
            if rel < th and th not in marks: marks[th] = it
        if rel < tol: break
        p = z + (rz2 / rz) * p; rz = rz2
    print(f'{name}: its {it} final {rel:.1e} {marks}', flush=True)
b0 = lattice(0); b2 = lattice(2)
run('separable + boxes', lambda v: fd.solve(v) + b0(v))
run('separable + boxes + shifted boxes', lambda v: fd.solve(v) + b0(v) + b2(v))
b1 = lattice(1); b3 = lattice(3)
run('separable + four lattices', lambda v: fd.solve(v) + b0(v) + b1(v) + b2(v) + b3(v))
