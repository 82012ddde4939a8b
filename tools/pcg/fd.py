"""Research: separable (fast-diagonalisation) preconditioner of the EXPECTED normal equations, applied as it would be on the GPU."""
import sys, time, numpy as np, scipy.sparse as sp, scipy.linalg as la
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools/pcg')
from rows import bas1


def one_d(nod, xmin=0.0, xmax=1.0):
    dx = (xmax - xmin) / (nod - 1)
    xn = xmin + np.arange(nod) * dx
    T = [np.zeros((nod, nod)) for _ in range(3)]
    for n in range(nod):
        for ib in range(max(0, n - 1), min(nod - 1, n + 1) + 1):
            for k in range(3):
                T[k][n, ib] = bas1(np.array([ib]), k, np.array([xn[n]]), xmin, dx, nod)[0]
    T2b = T[2].copy()
    T2b[0] = T[1][0]
    T2b[-1] = T[1][-1]
    # mass matrix of the basis on [xmin, xmax] by Gauss quadrature per interval
    gp, gw = np.polynomial.legendre.leggauss(6)
    M = np.zeros((nod, nod))
    for c in range(nod - 1):
        xs = xn[c] + (gp + 1) * 0.5 * dx
        ws = gw * 0.5 * dx
        B = np.array([bas1(np.full(xs.shape, ib), 0, xs, xmin, dx, nod) for ib in range(nod)])   # [nod, q]
        M += (B * ws) @ B.T
    return T[0], T[1], T2b, M


class FD:
    def __init__(self, nodes, rho, lam, qb):
        """rho: data density factor (sum w^2 per unit volume); lam: mean sparse*dcw^2 of interior nodes; qb: weight factor per boundary dim"""
        self.nodes = nodes
        self.V = []; self.l2 = []; self.d1 = []; self.mu = []
        for nod in nodes:
            T0, T1, T2b, M = one_d(nod)
            q = np.ones(nod); q[0] = q[-1] = qb
            K0 = T0.T @ (q[:, None] * T0); K1 = T1.T @ (q[:, None] * T1); K2 = T2b.T @ (q[:, None] * T2b)
            l2, V = la.eigh(K2, K0)           # V^T K0 V = I, V^T K2 V = diag(l2)
            self.V.append(V); self.l2.append(l2)
            self.d1.append(np.einsum('ij,ik,kj->j', V, K1, V)); self.mu.append(np.einsum('ij,ik,kj->j', V, M, V))
        d = len(nodes)
        sh = lambda v, k: v.reshape([-1 if j == k else 1 for j in range(d)][::-1])   # dim 0 fastest -> last numpy axis
        Dg = rho * np.prod(np.stack(np.broadcast_arrays(*[sh(self.mu[k], k) for k in range(d)])), axis=0)
        pen = 0.0
        for i in range(d):
            pen = pen + sh(self.l2[i], i)
            for j in range(i + 1, d):
                pen = pen + 4.0 * sh(self.d1[i], i) * sh(self.d1[j], j)
        self.diag = Dg + lam * pen
        self.d = d

    def _mode(self, X, Mat, k):
        # apply Mat along dimension k (numpy axis d-1-k)
        ax = self.d - 1 - k
        return np.moveaxis(np.tensordot(Mat, X, axes=([1], [ax])), 0, ax)

    def solve(self, v):
        X = v.reshape(self.nodes[::-1])
        for k in range(self.d): X = self._mode(X, self.V[k].T, k)
        X = X / self.diag
        for k in range(self.d): X = self._mode(X, self.V[k], k)
        return X.ravel()
