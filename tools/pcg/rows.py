"""Research helper (not product, not the oracle): the least-squares rows of the fit as scipy sparse matrices,
vectorised numpy restatement of SURVEY.md appendix A + section 8a (window rule, basis functions, constraint rows).
Used to measure iteration counts of candidate iterative solvers before anything is written in HIP."""
import numpy as np
import scipy.sparse as sp


def bas1(ib, nder, x, xmin, dx, nod):
    """1-D basis function ib (array) at x (array), derivative nder (scalar)."""
    s = 1.0 / dx
    xb = xmin + ib * dx
    typ = np.where(ib <= 1, 1, np.where(ib >= nod - 2, 3, 2))
    out = np.zeros_like(x, dtype=np.float64)
    # chapeau
    c = typ == 2
    if c.any():
        u = (x - xb)[c]
        if nder == 0:
            z = np.abs(s * u) - 2.0
            b = np.where(z < 0, -0.25 * z**3, 0.0)
            b = b + np.where(z + 1 < 0, (z + 1) ** 3, 0.0)
        elif nder == 1:
            f = np.where(u >= 0, s, -s)
            z = f * u - 2.0
            b = np.where(z < 0, -0.75 * z * z, 0.0)
            b = b + np.where(z + 1 < 0, 3 * (z + 1) ** 2, 0.0)
            b = b * f
        else:
            z = s * np.abs(u) - 2.0
            b = np.where(z < 0, -1.5 * z, 0.0)
            b = b + np.where(z + 1 < 0, 6 * (z + 1), 0.0)
            b = b * s * s
        out[c] = b
    for t, sgn in ((3, 1.0), (1, -1.0)):
        c = typ == t
        if not c.any():
            continue
        f = sgn * s
        z = f * (x - xb)[c] + 2.0
        if nder == 0:
            b = np.where(z <= 0, 0.0, np.where(z < 2, 0.5 * z**3 - np.where(z > 1, (z - 1) ** 3, 0.0), 3 * z - 3))
        elif nder == 1:
            b = np.where(z <= 0, 0.0, np.where(z < 2, (1.5 * z * z - np.where(z > 1, 3 * (z - 1) ** 2, 0.0)) * f, 3 * f))
        else:
            b = np.where(np.abs(z - 1) < 1, (3 * z - np.where(z > 1, 6 * (z - 1), 0.0)) * f * f, 0.0)
        out[c] = b
    return out


def data_rows(x, w, xmin, xmax, nodes):
    m, d = x.shape
    nodes = np.asarray(nodes)
    dx = (xmax - xmin) / (nodes - 1)
    stride = np.concatenate([[1], np.cumprod(nodes[:-1])])
    ncol = int(np.prod(nodes))
    # per-dim windows: 4 candidates lo..lo+3 masked by <= hi
    vals = []
    cols = []
    for k in range(d):
        it = np.trunc((x[:, k] - xmin[k]) / dx[k]).astype(np.int64)
        lo = np.minimum(np.maximum(it - 1, 0), nodes[k] - 2)
        hi = np.maximum(np.minimum(it + 2, nodes[k] - 1), 1)
        vk = np.zeros((m, 4))
        ck = np.zeros((m, 4), dtype=np.int64)
        for j in range(4):
            ib = lo + j
            ok = ib <= hi
            ibc = np.minimum(ib, nodes[k] - 1)
            vk[:, j] = np.where(ok, bas1(ibc, 0, x[:, k], xmin[k], dx[k], nodes[k]), 0.0)
            ck[:, j] = ibc * stride[k]
        vals.append(vk)
        cols.append(ck)
    V = vals[0]
    Cc = cols[0]
    for k in range(1, d):
        V = (V[:, :, None] * vals[k][:, None, :]).reshape(m, -1)
        Cc = (Cc[:, :, None] + cols[k][:, None, :]).reshape(m, -1)
    V = V * w[:, None]
    rows = np.repeat(np.arange(m), V.shape[1])
    A = sp.csr_matrix((V.ravel(), (rows, Cc.ravel())), shape=(m, ncol))
    A.sum_duplicates()
    A.eliminate_zeros()
    return A


def constraint_rows(x, w, xmin, xmax, nodes, xtrap):
    m, d = x.shape
    nodes = np.asarray(nodes)
    dx = (xmax - xmin) / (nodes - 1)
    stride = np.concatenate([[1], np.cumprod(nodes[:-1])])
    ncol = int(np.prod(nodes))
    # histogram (in-range points only here; the :899 quirk does not matter for iteration counts)
    idx = np.zeros(m, dtype=np.int64)
    for k in range(d):
        ii = np.trunc((x[:, k] - xmin[k]) / dx[k] + 0.5).astype(np.int64)
        ii = np.clip(ii, 0, nodes[k] - 1)
        idx += ii * stride[k]
    hist = np.bincount(idx, weights=w, minlength=ncol)
    tot = w.sum()
    wtprrc = tot / np.prod(nodes - 1)
    sub = np.array(np.unravel_index(np.arange(ncol), nodes[::-1])).T[:, ::-1]  # [node, d], dim 0 fastest
    onb = (sub == 0) | (sub == nodes - 1)
    expect = wtprrc * 0.5 ** onb.sum(1)
    sparse = hist < 0.75 * expect
    dcw = xtrap * (expect - hist)
    sn = np.nonzero(sparse)[0]
    ns = sn.size
    R = []
    rowid = 0
    datas, rws, cls = [], [], []
    xs = xmin + sub[sn] * dx
    for i in range(d):
        for j in range(i, d):
            nder = np.zeros((ns, d), dtype=int)
            wt = np.where(i == j, 1.0, 2.0) * dcw[sn]
            if i == j:
                nder[:, i] = np.where(onb[sn, i], 1, 2)
            else:
                nder[:, i] = 1
                nder[:, j] = 1
            # 3^d window
            V = np.ones((ns, 1))
            Cc = np.zeros((ns, 1), dtype=np.int64)
            for k in range(d):
                vk = np.zeros((ns, 3))
                ck = np.zeros((ns, 3), dtype=np.int64)
                for o in (-1, 0, 1):
                    ib = sub[sn, k] + o
                    ok = (ib >= 0) & (ib <= nodes[k] - 1)
                    ibc = np.clip(ib, 0, nodes[k] - 1)
                    v = np.zeros(ns)
                    for nd in (0, 1, 2):
                        sel = nder[:, k] == nd
                        if sel.any():
                            v[sel] = bas1(ibc[sel], nd, xs[sel, k], xmin[k], dx[k], nodes[k])
                    vk[:, o + 1] = np.where(ok, v, 0.0)
                    ck[:, o + 1] = ibc * stride[k]
                V = (V[:, :, None] * vk[:, None, :]).reshape(ns, -1)
                Cc = (Cc[:, :, None] + ck[:, None, :]).reshape(ns, -1)
            V = V * wt[:, None]
            datas.append(V.ravel())
            rws.append(np.repeat(np.arange(ns) + rowid, V.shape[1]))
            cls.append(Cc.ravel())
            rowid += ns
    if rowid == 0:
        return sp.csr_matrix((0, ncol)), hist, sparse
    Cm = sp.csr_matrix((np.concatenate(datas), (np.concatenate(rws), np.concatenate(cls))), shape=(rowid, ncol))
    Cm.sum_duplicates()
    Cm.eliminate_zeros()
    return Cm, hist, sparse
