"""Research: additive Schwarz over subdomains that CONTAIN the rows of their core nodes -- an aligned core box of `bs` nodes per
dimension plus `ov` nodes on every side (ov = 1: every constraint row of a core node, 3 nodes wide, lies inside) -- in the regime where
aligned boxes + separable stagnate (about 1 constraint row per column).  Variants: plain additive (symmetric), and additive with the
corrections weighted by 1 / multiplicity of a node.  4-D 12^4.   exp10.py ndim nodes points_per_cell [bs] [ov]"""
import sys, time, numpy as np, scipy.sparse as sp, scipy.linalg as la
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools/pcg')
from rows import data_rows, constraint_rows
from exp6 import FD2
from splpak_amd.synth import synth_points

d, nod, ppc = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
bs = int(sys.argv[4]) if len(sys.argv) > 4 else 4
ov = int(sys.argv[5]) if len(sys.argv) > 5 else 1
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
print(f'n={n} m={m} sparse frac {spn.mean():.3f} rows/col {C.shape[0] / n:.2f}  core {bs}^d, overlap {ov}', flush=True)
fd = FD2(list(nodes), rho, lam, 0.5, 'K0')
def schwarz(bs, ov):
    nb = (nod + bs - 1) // bs
    groups = []
    for b in np.ndindex(*([nb] * d)):
        ok = np.ones(n, bool)
        for k in range(d):
            lo, hi = b[k] * bs - ov, b[k] * bs + bs - 1 + ov
            ok &= (sub[:, k] >= lo) & (sub[:, k] <= hi)
        g = np.flatnonzero(ok)
        if g.size: groups.append(g)
    mult = np.zeros(n)
    for g in groups: mult[g] += 1
    t0 = time.time()
    chol = [la.cho_factor(N[g][:, g].toarray()) for g in groups]
    print(f'  {len(groups)} subdomains of up to {max(g.size for g in groups)} nodes, factored in {time.time() - t0:.1f} s, multiplicity up to {mult.max():.0f}', flush=True)
    def ap(v, weighted=False):
        out = np.zeros_like(v)
        if weighted:
            sq = 1.0 / np.sqrt(mult)
            for g, c in zip(groups, chol): out[g] += sq[g] * la.cho_solve(c, sq[g] * v[g])
        else:
            for g, c in zip(groups, chol): out[g] += la.cho_solve(c, v[g])
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
s1 = schwarz(bs, ov)
run(f'separable + Schwarz(core {bs}, overlap {ov})', lambda v: fd.solve(v) + s1(v))
run(f'Schwarz(core {bs}, overlap {ov}) alone', lambda v: s1(v))
run(f'separable + weighted Schwarz', lambda v: fd.solve(v) + s1(v, True))
