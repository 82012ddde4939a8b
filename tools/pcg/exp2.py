import sys, time, numpy as np, scipy.sparse as sp, scipy.linalg as la
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools/pcg')
from rows import data_rows, constraint_rows
from exp1 import problem, pcg
from splpak_amd.synth import synth_points

d, nod, ppc = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
P = problem(d, nod, ppc)
A, C, N, r = P['A'], P['C'], P['N'].toarray(), P['r']
n = N.shape[0]
D = (A.T @ A).toarray(); Pm = (C.T @ C).toarray()
print('n', n, 'cons rows', C.shape[0], 'diag D mean', D.diagonal().mean(), 'diag P max', Pm.diagonal().max(), 'cond N', np.linalg.cond(N))
# expected data Gram: many more points, rescaled
m2 = 30 * P['m']
x2, y2, w2 = synth_points(d, m2, first_point=P['m'] + 17)
A2 = data_rows(x2, w2, np.zeros(d), np.ones(d), P['nodes'])
De = (A2.T @ A2).toarray() * (P['m'] / m2)
def spec(M, name):
    ev = la.eigh(N, M, eigvals_only=True)
    print(f'{name}: min {ev.min():.3e} max {ev.max():.3e} cond {ev.max()/ev.min():.3e}; #ev<0.1: {(ev<0.1).sum()} #ev>10: {(ev>10).sum()}  quantiles', np.quantile(ev, [0.01, 0.1, 0.5, 0.9, 0.99]))
spec(np.diag(N.diagonal()), 'jacobi')
spec(De + Pm, 'E[D] + P')
# expected P: every node sparse with prob p, mean weight^2
nodes = P['nodes']; xmin = np.zeros(d); xmax = np.ones(d)
# build P with all nodes active & unit weight by faking hist=0, expect -> weight 1: use constraint_rows on empty-ish data
xx = np.full((1, d), 0.5); ww = np.array([1.0])
Call, _, _ = constraint_rows(xx, ww, xmin, xmax, nodes, 1.0)
# rows weighted by expect (~1e-300 scale) -> renormalise each row group: simpler: scale to match mean diag of actual P
Pall = (Call.T @ Call).toarray()
Pall *= Pm.diagonal().sum() / Pall.diagonal().sum()
spec(D + Pall, 'D + E[P]~')
spec(De + Pall, 'E[D] + E[P]~')
