import sys, time, numpy as np, scipy.sparse as sp, scipy.sparse.linalg as sla
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools/pcg')
from exp1 import problem
from fd import FD

def pcg_res(N, r, Minv, tol=1e-13, maxit=3000):
    x = np.zeros_like(r); res = r.copy(); z = Minv(res); p = z.copy(); rz = res @ z; rz0 = rz
    hist = []
    for it in range(1, maxit + 1):
        Np = N @ p
        a = rz / (p @ Np)
        x += a * p; res -= a * Np
        z = Minv(res); rz2 = res @ z
        hist.append(np.sqrt(rz2 / rz0))
        if hist[-1] < tol: break
        p = z + (rz2 / rz) * p; rz = rz2
    return x, it, hist

d, nod, ppc = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
t = time.time(); P = problem(d, nod, ppc); print('build', time.time() - t)
N, r, nodes = P['N'], P['r'], P['nodes']
n = N.shape[0]
# empirical parameters of the expectation
A, C = P['A'], P['C']
w2 = (A.multiply(A)).sum()   # not needed
x_, y_, w_ = __import__('splpak_amd.synth', fromlist=['synth_points']).synth_points(d, P['m'])
rho = (w_ ** 2).sum()        # sum w^2 over unit volume
hist = P['hist']; spn = P['spn']
sub = np.array(np.unravel_index(np.arange(n), nodes[::-1])).T[:, ::-1]
onb = ((sub == 0) | (sub == nodes - 1)).sum(1)
wt = w_.sum() / np.prod(nodes - 1)
expect = wt * 0.5 ** onb
dcw2 = np.where(spn, (expect - hist) ** 2, 0.0)
lam = dcw2[onb == 0].mean() if (onb == 0).any() else dcw2.mean()
lam1 = dcw2[onb == 1].mean() if (onb == 1).any() else lam
qb = lam1 / lam
print(f'n={n} m={P["m"]} sparse frac {spn.mean():.3f} (interior {spn[onb==0].mean():.3f}) rho {rho:.4g} lam {lam:.4g} qb {qb:.3f}')
for name, (l, q) in {'fd': (lam, qb), 'fd_lam/3': (lam / 3, qb), 'fd_lam*3': (lam * 3, qb)}.items():
    fd = FD(list(nodes), rho, l, q)
    t = time.time(); x, it, h = pcg_res(N, r, fd.solve); dt = time.time() - t
    marks = [next((i + 1 for i, v in enumerate(h) if v < th), None) for th in (1e-4, 1e-8, 1e-12)]
    print(f'{name}: its {it} final {h[-1]:.2e} its to 1e-4/1e-8/1e-12: {marks}  ({dt:.1f}s)')
    xs = x
dg = N.diagonal()
x, it, h = pcg_res(N, r, lambda v: v / dg, maxit=2000)
print('jacobi its', it, 'final', h[-1], 'err vs fd solution', np.abs(x - xs).max() / np.abs(xs).max())
