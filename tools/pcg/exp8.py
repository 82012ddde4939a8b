"""Research: must the boxes hold the EXACT data part?  Boxes = rho M_box + P_box (expected data part, exact constraint part) against exact boxes,
added to the separable preconditioner.  4-D 12^4."""
import sys, time, numpy as np, scipy.sparse as sp, scipy.linalg as la
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools/pcg')
from rows import data_rows, constraint_rows
from exp6 import FD2
from fd import one_d
from splpak_amd.synth import synth_points

d, nod, ppc, bs = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
xtrap = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
nodes = np.array([nod] * d); m = int(ppc * (nod - 1) ** d)
x, y, w = synth_points(d, m)
xmin = np.zeros(d); xmax = np.ones(d)
A = data_rows(x, w, xmin, xmax, nodes); At = A.T.tocsr()
C, hist, spn = constraint_rows(x, w, xmin, xmax, nodes, xtrap); Ct = C.T.tocsr()
n = A.shape[1]
Dm = (At @ A).tocsr(); Pm = (Ct @ C).tocsr() if C.shape[0] else sp.csr_matrix((n, n))
N = (Dm + Pm).tocsr()
r = At @ (w * y)
sub = np.array(np.unravel_index(np.arange(n), nodes[::-1])).T[:, ::-1]
onb = ((sub == 0) | (sub == nodes - 1)).sum(1)
wt = w.sum() / np.prod(nodes - 1)
expect = wt * 0.5 ** onb
dcw2 = np.where(spn, (xtrap * (expect - hist)) ** 2, 0.0)
rho = (w ** 2).sum(); lam = dcw2.mean()
print(f'n={n} m={m} sparse frac {spn.mean():.3f} rows/col {C.shape[0] / n:.2f}', flush=True)
fd = FD2(list(nodes), rho, max(lam, 1e-300), 0.5, 'K0')
M1 = one_d(nod)[3]
bid = np.zeros(n, dtype=np.int64); mul = 1
for k in range(d):
    bid += (sub[:, k] // bs) * mul; mul *= -(-nod // bs)
order = np.argsort(bid, kind='stable'); bounds = np.flatnonzero(np.diff(bid[order])) + 1
groups = np.split(order, bounds)
def mass_box(g):
    # kron of the 1-D mass sub-blocks (dimension 0 fastest)
    K = np.ones((1, 1))
    for k in range(d):
        idx = np.unique(sub[g, k])
        K = np.kron(M1[np.ix_(idx, idx)], K)
    return K
Ddiag = Dm.diagonal()
def make(kind):
    ch = []
    for g in groups:
        Pb = Pm[g][:, g].toarray()
        if kind == 'exact': B = Dm[g][:, g].toarray() + Pb
        elif kind == 'rhoM': B = rho * mass_box(g) + Pb
        elif kind == 'traceM':
            Mb = mass_box(g); B = (Ddiag[g].sum() / np.trace(Mb)) * Mb + Pb
        elif kind == 'diagD': B = np.diag(Ddiag[g]) + Pb
        ch.append(la.cho_factor(B))
    def ap(v):
        out = np.empty_like(v)
        for g, c in zip(groups, ch): out[g] = la.cho_solve(c, v[g])
        return out
    return ap
def run(name, Minv, tol=1e-10, maxit=1200):
    xs = np.zeros(n); res = r.copy(); z = Minv(res); p = z.copy(); rz = res @ z; rz0 = rz; marks = {}
    for it in range(1, maxit + 1):
        Np = N @ p; a = rz / (p @ Np); xs += a * p; res -= a * Np
        z = Minv(res); rz2 = res @ z; rel = np.sqrt(abs(rz2) / rz0)
        for th in (1e-4, 1e-8, 1e-10):
            if rel < th and th not in marks: marks[th] = it
        if rel < tol: break
        p = z + (rz2 / rz) * p; rz = rz2
    print(f'{name}: its {it} final {rel:.1e} {marks}', flush=True)
for kind in ('exact', 'rhoM', 'traceM', 'diagD'):
    bj = make(kind)
    run('separable + boxes ' + kind, lambda v: fd.solve(v) + bj(v))
