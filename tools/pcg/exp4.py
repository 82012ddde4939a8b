import sys, time, numpy as np, scipy.sparse as sp, scipy.sparse.linalg as sla
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools/pcg')
from exp1 import problem
from fd import FD
from splpak_amd.synth import synth_points

def setup_fd(P, d):
    nodes = P['nodes']; n = P['N'].shape[0]
    x_, y_, w_ = synth_points(d, P['m'])
    rho = (w_ ** 2).sum()
    hist = P['hist']; spn = P['spn']
    sub = np.array(np.unravel_index(np.arange(n), nodes[::-1])).T[:, ::-1]
    onb = ((sub == 0) | (sub == nodes - 1)).sum(1)
    wt = w_.sum() / np.prod(nodes - 1)
    expect = wt * 0.5 ** onb
    dcw2 = np.where(spn, (expect - hist) ** 2, 0.0)
    lam = dcw2[onb == 0].mean() if (onb == 0).any() and dcw2[onb == 0].mean() > 0 else max(dcw2.mean(), 1e-300)
    lam1 = dcw2[onb == 1].mean() if (onb == 1).any() else lam
    return FD(list(nodes), rho, lam, lam1 / lam if lam > 0 else 1.0), w_, y_

d, nod, ppc = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
P = problem(d, nod, ppc)
A, C, N, r = P['A'], P['C'], P['N'], P['r']
fd, w_, y_ = setup_fd(P, d)
b = w_ * y_
lu = sla.splu(N.tocsc()); xd = lu.solve(r)
# refine direct solution against rows
for _ in range(3):
    rho = A.T @ (b - A @ xd) - C.T @ (C @ xd); xd = xd + lu.solve(rho)
op = lambda v: A.T @ (A @ v) + C.T @ (C @ v)
def pcg(rhs, tol, maxit=5000):
    x = np.zeros_like(rhs); res = rhs.copy(); z = fd.solve(res); p = z.copy(); rz = res @ z; rz0 = rz
    for it in range(1, maxit + 1):
        Np = op(p); a = rz / (p @ Np); x += a * p; res -= a * Np
        z = fd.solve(res); rz2 = res @ z
        if rz2 <= tol * tol * rz0: break
        p = z + (rz2 / rz) * p; rz = rz2
    return x, it
for tol in (1e-4, 1e-6, 1e-8):
    x = np.zeros_like(r); tot = 0; log = []
    rhs = r.copy()
    for outer in range(8):
        dx, it = pcg(rhs, tol); tot += it
        x = x + dx
        err = np.abs(x - xd).max() / np.abs(xd).max()
        log.append((it, float(f'{np.abs(dx).max()/np.abs(x).max():.2e}'), float(f'{err:.2e}')))
        if err < 1e-12: break
        rhs = A.T @ (b - A @ x) - C.T @ (C @ x)
    print(f'inner tol {tol}: total its {tot}; (its, |dx|/|x|, err) per outer:', log)
