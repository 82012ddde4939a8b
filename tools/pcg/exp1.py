import sys, time, numpy as np, scipy.sparse as sp, scipy.sparse.linalg as sla
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools/pcg')
from rows import data_rows, constraint_rows
from splpak_amd.synth import synth_points

def problem(d, nod, ppc, xtrap=1.0, weighted=True):
    nodes = np.array([nod] * d)
    m = int(ppc * (nod - 1) ** d)
    x, y, w = synth_points(d, m)
    if not weighted: w = np.ones(m)
    xmin = np.zeros(d); xmax = np.ones(d)
    A = data_rows(x, w, xmin, xmax, nodes)
    C, hist, spn = constraint_rows(x, w, xmin, xmax, nodes, xtrap)
    N = (A.T @ A + C.T @ C).tocsr()
    r = A.T @ (w * y)
    return dict(A=A, C=C, N=N, r=r, nodes=nodes, hist=hist, spn=spn, m=m)

def pcg(N, r, Minv, xref, tol=1e-11, maxit=5000):
    x = np.zeros_like(r); res = r.copy(); z = Minv(res); p = z.copy(); rz = res @ z
    hist = []
    for it in range(1, maxit + 1):
        Np = N @ p
        a = rz / (p @ Np)
        x += a * p; res -= a * Np
        err = np.abs(x - xref).max() / np.abs(xref).max()
        hist.append(err)
        if err < tol: break
        z = Minv(res); rz2 = res @ z; p = z + (rz2 / rz) * p; rz = rz2
    return it, hist

if __name__ == '__main__':
    d, nod, ppc = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
    P = problem(d, nod, ppc)
    N, r = P['N'], P['r']
    n = N.shape[0]
    print(f"d={d} nod={nod} n={n} m={P['m']} cons rows={P['C'].shape[0]} sparse nodes={P['spn'].sum()} ({P['spn'].mean():.3f})")
    t = time.time(); lu = sla.splu(N.tocsc()); xref = lu.solve(r); print('direct', time.time() - t)
    dg = N.diagonal()
    it, h = pcg(N, r, lambda v: v / dg, xref, maxit=3000)
    print('jacobi: its', it, 'err', h[-1], 'err@100,300,1000', [h[min(k, len(h)-1)] for k in (99, 299, 999)])
