"""Research: variants of the separable preconditioner at 4-D 12^4 (fast): pairing (K2,K0) vs (K2,M), fixed qb, Jacobi rescaling."""
import sys, time, numpy as np, scipy.linalg as la
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools/pcg')
from rows import data_rows, constraint_rows
from fd import one_d
from splpak_amd.synth import synth_points

class FD2:
    def __init__(self, nodes, rho, lam, qb, pair='K0'):
        self.nodes = nodes; d = len(nodes); self.d = d
        self.V = []; l2 = []; d1 = []; mu = []; k0 = []
        for nod in nodes:
            T0, T1, T2b, M = one_d(nod)
            q = np.ones(nod); q[0] = q[-1] = qb
            K0 = T0.T @ (q[:, None] * T0); K1 = T1.T @ (q[:, None] * T1); K2 = T2b.T @ (q[:, None] * T2b)
            w, V = la.eigh(K2, K0 if pair == 'K0' else M)
            self.V.append(V); l2.append(w)
            d1.append(np.einsum('ij,ik,kj->j', V, K1, V)); mu.append(np.einsum('ij,ik,kj->j', V, M, V)); k0.append(np.einsum('ij,ik,kj->j', V, K0, V))
        sh = lambda v, k: v.reshape([-1 if j == k else 1 for j in range(d)][::-1])
        prod = lambda arrs: np.prod(np.stack(np.broadcast_arrays(*arrs)), axis=0)
        Dg = rho * prod([sh(mu[k], k) for k in range(d)])
        pen = 0.0
        for i in range(d):
            pen = pen + prod([sh(l2[k] if k == i else k0[k], k) for k in range(d)])
            for j in range(i + 1, d):
                pen = pen + 4.0 * prod([sh(d1[k] if k in (i, j) else k0[k], k) for k in range(d)])
        self.diag = Dg + lam * pen
    def _mode(self, X, Mat, k):
        ax = self.d - 1 - k
        return np.moveaxis(np.tensordot(Mat, X, axes=([1], [ax])), 0, ax)
    def solve(self, v):
        X = v.reshape(self.nodes[::-1])
        for k in range(self.d): X = self._mode(X, self.V[k].T, k)
        X = X / self.diag
        for k in range(self.d): X = self._mode(X, self.V[k], k)
        return X.ravel()
    def mult_diag(self):
        """diagonal of the preconditioner matrix M = V^-T diag V^-1 (for the Jacobi rescaling)"""
        d = self.d
        W = [np.linalg.inv(V) for V in self.V]     # V^-1
        # diag(M)_i = sum_j W[j,i]^2 diag_j  -> mode products with (W^2)^T
        X = self.diag.copy()
        for k in range(d): X = self._mode(X, (W[k] ** 2).T, k)
        return X.ravel()

if __name__ == '__main__':
    d, nod, ppc = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
    nodes = np.array([nod] * d); m = int(ppc * (nod - 1) ** d)
    x, y, w = synth_points(d, m)
    xmin = np.zeros(d); xmax = np.ones(d)
    A = data_rows(x, w, xmin, xmax, nodes); At = A.T.tocsr()
    C, hist, spn = constraint_rows(x, w, xmin, xmax, nodes, 1.0); Ct = C.T.tocsr()
    n = A.shape[1]
    r = At @ (w * y)
    sub = np.array(np.unravel_index(np.arange(n), nodes[::-1])).T[:, ::-1]
    onb = ((sub == 0) | (sub == nodes - 1)).sum(1)
    wt = w.sum() / np.prod(nodes - 1)
    expect = wt * 0.5 ** onb
    dcw2 = np.where(spn, (expect - hist) ** 2, 0.0)
    rho = (w ** 2).sum(); lam = dcw2[onb == 0].mean(); lam1 = dcw2[onb == 1].mean(); qb = lam1 / lam
    lam_all = dcw2.mean()
    print(f'n={n} sparse frac {spn.mean():.3f} rho {rho:.4g} lam {lam:.4g} lam_all {lam_all:.4g} qb {qb:.3f}', flush=True)
    Ndiag = np.asarray(A.multiply(A).sum(0)).ravel() + np.asarray(C.multiply(C).sum(0)).ravel()
    op = lambda v: At @ (A @ v) + Ct @ (C @ v)
    def run(name, Minv, tol=1e-12):
        xs = np.zeros(n); res = r.copy(); z = Minv(res); p = z.copy(); rz = res @ z; rz0 = rz; marks = {}
        for it in range(1, 3001):
            Np = op(p); a = rz / (p @ Np); xs += a * p; res -= a * Np
            z = Minv(res); rz2 = res @ z; rel = np.sqrt(rz2 / rz0)
            for th in (1e-4, 1e-8, 1e-12):
                if rel < th and th not in marks: marks[th] = it
            if rel < tol: break
            p = z + (rz2 / rz) * p; rz = rz2
        print(name, 'its', it, marks, flush=True)
    for name, kw in [('K0 qb', dict(qb=qb, pair='K0')), ('K0 qb=.5', dict(qb=0.5, pair='K0')), ('K0 qb=1', dict(qb=1.0, pair='K0')), ('M qb', dict(qb=qb, pair='M')),
                     ('K0 lam_all', dict(qb=qb, pair='K0', lam=lam_all))]:
        l = kw.pop('lam', lam)
        fd = FD2(list(nodes), rho, l, **kw)
        run(name, fd.solve)
        if name in ('K0 qb', 'M qb'):
            s = np.sqrt(fd.mult_diag() / Ndiag)
            run(name + ' + jacobi rescale', lambda v: s * fd.solve(s * v))
