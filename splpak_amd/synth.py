"""Cross-language reproducible synthetic inputs.

Park-Miller "minimal standard" LCG: ``s <- 48271*s mod (2^31-1)``, ``u = s/(2^31-1)``,
seed 42.  Per data point, in this draw order: ``x_d = u()`` for d = 1..ndim,
``y = sum_d sin(3*x_d + d) + 0.01*(u() - 0.5)``, ``w = 0.5 + u()`` (drawn even when
the fit is unweighted).  Evaluation queries continue the same stream with
``x_d = u()``.  Skip-ahead ``s_k = s_0 * 48271^k mod (2^31-1)`` lets a shard start
anywhere in the stream, which is how multi-GPU ranks generate their own slice.

Host (numpy) version; the device version with the same stream is
``splpak_synth_points_f64`` in the HIP library.
"""
from __future__ import annotations

import numpy as np

LCG_A = 48271
LCG_M = 2147483647  # 2^31 - 1
SEED = 42


def lcg_skip(s0: int, k: int) -> int:
    """State after ``k`` steps from state ``s0``."""
    return (s0 * pow(LCG_A, k, LCG_M)) % LCG_M


def lcg_uniform(n: int, s0: int = SEED, skip: int = 0) -> np.ndarray:
    """``n`` consecutive draws u_1..u_n after skipping ``skip`` draws."""
    if n <= 0:
        return np.zeros(0, dtype=np.float64)
    block = 1 << 16
    # pw[j] = a^(j+1) mod m
    pw = np.empty(min(block, n), dtype=np.uint64)
    acc = 1
    for j in range(pw.size):
        acc = (acc * LCG_A) % LCG_M
        pw[j] = acc
    out = np.empty(n, dtype=np.float64)
    s = lcg_skip(s0, skip)
    a_blk = pow(LCG_A, block, LCG_M)
    pos = 0
    while pos < n:
        cnt = min(block, n - pos)
        st = (np.uint64(s) * pw[:cnt]) % np.uint64(LCG_M)
        out[pos:pos + cnt] = st.astype(np.float64) / float(LCG_M)
        s = (s * a_blk) % LCG_M
        pos += cnt
    return out


def draws_per_point(ndim: int) -> int:
    return ndim + 2


def synth_points(ndim: int, ndata: int, first_point: int = 0, seed: int = SEED):
    """Data points ``first_point .. first_point+ndata-1`` of the stream.

    Returns ``xdata (ndata, ndim)`` (row i = point i; this is the reference's
    column-major ``xdata(l1xdat=ndim, ndata)`` seen from C), ``ydata``, ``wdata``.
    """
    dpp = draws_per_point(ndim)
    u = lcg_uniform(ndata * dpp, seed, skip=first_point * dpp).reshape(ndata, dpp)
    x = np.ascontiguousarray(u[:, :ndim])
    y = np.zeros(ndata, dtype=np.float64)
    for d in range(ndim):
        y += np.sin(3.0 * x[:, d] + float(d + 1))
    y += 0.01 * (u[:, ndim] - 0.5)
    w = 0.5 + u[:, ndim + 1]
    return x, y, np.ascontiguousarray(w)


def synth_queries(ndim: int, nq: int, ndata_before: int, first_query: int = 0, seed: int = SEED):
    """Evaluation queries continuing the stream after ``ndata_before`` data points."""
    skip = ndata_before * draws_per_point(ndim) + first_query * ndim
    u = lcg_uniform(nq * ndim, seed, skip=skip).reshape(nq, ndim)
    return np.ascontiguousarray(u)
