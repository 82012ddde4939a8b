"""Research: does a location-aware local component rescue the regime where the separable preconditioner stagnates (0.5 .. 1.7 constraint
rows per column)?  4-D 12^4, points per cell given; block Jacobi over non-overlapping b^4-node blocks of the assembled N, alone and combined
with the separable preconditioner (additively, and multiplicatively: symmetric two-stage)."""
import sys, time, numpy as np, scipy.sparse as sp, scipy.linalg as la
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools/pcg')
from rows import data_rows, constraint_rows
from exp6 import FD2  # noqa (runs nothing: guarded below)
from splpak_amd.synth import synth_points

d, nod, ppc, bs = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
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
# non-overlapping blocks of bs^d nodes
bid = np.zeros(n, dtype=np.int64); mul = 1
for k in range(d):
    bid += (sub[:, k] // bs) * mul; mul *= -(-nod // bs)
order = np.argsort(bid, kind='stable'); bounds = np.flatnonzero(np.diff(bid[order])) + 1
groups = np.split(order, bounds)
t = time.time()
chol = [la.cho_factor(N[g][:, g].toarray()) for g in groups]
print(f'{len(groups)} blocks of up to {max(len(g) for g in groups)} nodes factored in {time.time() - t:.1f} s', flush=True)
def bj(v):
    out = np.empty_like(v)
    for g, c in zip(groups, chol): out[g] = la.cho_solve(c, v[g])
    return out
op = lambda v: N @ v
def run(name, Minv, tol=1e-10, maxit=1500):
    xs = np.zeros(n); res = r.copy(); z = Minv(res); p = z.copy(); rz = res @ z; rz0 = rz; marks = {}
    for it in range(1, maxit + 1):
        Np = op(p); a = rz / (p @ Np); xs += a * p; res -= a * Np
        z = Minv(res); rz2 = res @ z; rel = np.sqrt(abs(rz2) / rz0)
        for th in (1e-2, 1e-4, 1e-6, 1e-8, 1e-10):
            if rel < th and th not in marks: marks[th] = it
        if rel < tol: break
        p = z + (rz2 / rz) * p; rz = rz2
    print(f'{name}: its {it} final {rel:.1e} {marks}', flush=True)
run('separable', fd.solve)
run('block jacobi', bj)
run('additive', lambda v: fd.solve(v) + bj(v))
def mult(v):            # symmetric multiplicative: block solve, separable on the residual, block solve again
    z1 = bj(v); r1 = v - N @ z1
    z2 = z1 + fd.solve(r1); r2 = v - N @ z2
    return z2 + bj(r2)
run('multiplicative bj-fd-bj', mult)
def mult2(v):
    z1 = fd.solve(v); r1 = v - N @ z1
    z2 = z1 + bj(r1); r2 = v - N @ z2
    return z2 + fd.solve(r2)
run('multiplicative fd-bj-fd', mult2)
