/*
 * TEST INFRASTRUCTURE ONLY -- "best CPU" restatement of the fit: the reference's ROWS
 * (src/splpak.F90:788-855 data rows, :862-1046 derivative-constraint rows, generated here exactly as
 * oracle/splpak_oracle.c generates them, through the same oracle_bascmp) solved the way the GPU path
 * solves them -- banded normal equations, blocked band Cholesky, iterative refinement against the rows --
 * on all host cores (OpenMP).  Two uses, both outside the product path:
 *   * bench.py's cpu_baseline leg: the honest CPU comparator SURVEY 8d asks for beside the dense
 *     reference algorithm, which cannot reach the node grids the GPU path is built for;
 *   * tests: an independent answer on grids beyond the dense oracle's reach (24^3 ... 32^3).
 * Never linked into, imported by or executed from the product library.
 *
 * Parity pin: tests/test_oracle_golden.py holds this solver to the same reference goldens as the port.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

double oracle_bascmp(int mdim, const double *x, const int *nderiv, const int *ib,
                     const double *xmin, const double *dx, const int *nodes, int *icol_out);

#define BMAXD 4
#define BMAXNZ 256

typedef struct {
    long nrows, cap_rows, nnz, cap_nnz;
    long *ptr;
    int *col;
    double *val, *rhs;
} rows_t;

static int rows_push(rows_t *R, int nz, const int *col, const double *val, double rhs)
{
    if (R->nrows + 1 >= R->cap_rows) {
        R->cap_rows = R->cap_rows ? 2 * R->cap_rows : 1024;
        R->ptr = realloc(R->ptr, (size_t)(R->cap_rows + 1) * sizeof(long));
        R->rhs = realloc(R->rhs, (size_t)R->cap_rows * sizeof(double));
        if (!R->ptr || !R->rhs) return 1;
    }
    if (R->nnz + nz > R->cap_nnz) {
        R->cap_nnz = R->cap_nnz ? 2 * R->cap_nnz + nz : 65536;
        R->col = realloc(R->col, (size_t)R->cap_nnz * sizeof(int));
        R->val = realloc(R->val, (size_t)R->cap_nnz * sizeof(double));
        if (!R->col || !R->val) return 1;
    }
    if (R->nrows == 0) R->ptr[0] = 0;
    memcpy(R->col + R->nnz, col, (size_t)nz * sizeof(int));
    memcpy(R->val + R->nnz, val, (size_t)nz * sizeof(double));
    R->nnz += nz;
    R->rhs[R->nrows] = rhs;
    R->ptr[++R->nrows] = R->nnz;
    return 0;
}

/* A consumer of rows: 0 = go on.  build_rows stores them (rows_push); oracle_rows_gradient accumulates the
 * gradient of the least-squares functional without storing anything. */
typedef int (*row_sink)(void *ctx, int nz, const int *col, const double *val, double rhs);
static int sink_push(void *ctx, int nz, const int *col, const double *val, double rhs)
{
    return rows_push((rows_t *)ctx, nz, col, val, rhs);
}

/* rows of the reference's least-squares system, in the reference's order; data rows of the points [i0, i1),
 * the constraint rows (which need the histogram of ALL points) only when `with_constraints` */
static int emit_rows(int ndim, const double *xdata, int l1xdat, const double *ydata, const double *wdata,
                     int ndata, int i0, int i1, int with_constraints, const double *xmin, const double *xmax,
                     const int *nodes, double xtrap, row_sink sink, void *R, long *ncons);

static int build_rows(int ndim, const double *xdata, int l1xdat, const double *ydata, const double *wdata,
                      int ndata, const double *xmin, const double *xmax, const int *nodes, double xtrap,
                      rows_t *R, long *ncons)
{
    return emit_rows(ndim, xdata, l1xdat, ydata, wdata, ndata, 0, ndata, 1, xmin, xmax, nodes, xtrap, sink_push, R, ncons);
}

static int emit_rows(int ndim, const double *xdata, int l1xdat, const double *ydata, const double *wdata,
                     int ndata, int i0, int i1, int with_constraints, const double *xmin, const double *xmax,
                     const int *nodes, double xtrap, row_sink sink, void *R, long *ncons)
{
    double dx[BMAXD], dxin[BMAXD], x[BMAXD];
    int nderiv[BMAXD], ib[BMAXD], ibmn[BMAXD], ibmx[BMAXD], in[BMAXD], inmx[BMAXD];
    int col[BMAXNZ];
    double val[BMAXNZ];
    long ncol = 1;
    for (int d = 0; d < ndim; ++d) {
        ncol *= nodes[d];
        dx[d] = (xmax[d] - xmin[d]) / (double)(nodes[d] - 1);
        dxin[d] = 1.0 / dx[d];
        nderiv[d] = 0;
    }
    const int weighted = wdata && wdata[0] >= 0.0;
    double rowwt = 1.0;
    for (int idata = i0; idata < i1; ++idata) {                   /* :788-855 */
        if (weighted) {
            rowwt = wdata[idata];
            if (rowwt == 0.0) continue;
        }
        for (int d = 0; d < ndim; ++d) {
            x[d] = xdata[(long)idata * l1xdat + d];
            int nod = nodes[d];
            int it = (int)(dxin[d] * (x[d] - xmin[d]));
            int lo = it - 1; if (lo < 0) lo = 0; if (lo > nod - 2) lo = nod - 2;
            int hi = it + 2; if (hi > nod - 1) hi = nod - 1; if (hi < 1) hi = 1;
            ibmn[d] = lo; ib[d] = lo; ibmx[d] = hi;
        }
        int nz = 0;
        for (;;) {
            int icol;
            double basm = oracle_bascmp(ndim, x, nderiv, ib, xmin, dx, nodes, &icol);
            col[nz] = icol - 1;
            val[nz++] = rowwt * basm;
            int d = 0;
            for (; d < ndim; ++d) {
                if (++ib[d] <= ibmx[d]) break;
                ib[d] = ibmn[d];
            }
            if (d == ndim) break;
        }
        if (sink(R, nz, col, val, rowwt * ydata[idata])) return 1;
    }
    *ncons = 0;
    if (xtrap != 0.0 && with_constraints) {                        /* :862-1046 */
        long nrect = 1;
        for (int d = 0; d < ndim; ++d) { in[d] = 0; inmx[d] = nodes[d] - 1; nrect *= inmx[d]; }
        double *hist = calloc((size_t)ncol, sizeof(double));
        if (!hist) return 1;
        double totlwt = 0.0;
        for (int idata = 0; idata < ndata; ++idata) {
            double bump = weighted ? wdata[idata] : 1.0;
            if (bump == 0.0) continue;
            long iin = 0;
            for (int dc = 0; dc < ndim; ++dc) {
                int d = ndim - 1 - dc;
                int inidim = (int)(dxin[d] * (xdata[(long)idata * l1xdat + d] - xmin[d]) + 0.5);
                if (inidim < 0 || inidim > inmx[d]) continue;      /* :899 */
                iin = (long)(inmx[d] + 1) * iin + inidim;
            }
            hist[iin] += bump;
            totlwt += bump;
        }
        const double wtprrc = totlwt / (double)nrect;
        long iin = 0;
        for (;;) {
            double expect = wtprrc;
            for (int d = 0; d < ndim; ++d)
                if (in[d] == 0 || in[d] == inmx[d]) expect *= 0.5;
            if (hist[iin] < 0.75 * expect) {
                double dcwght = (expect - hist[iin]) * xtrap;
                for (int d = 0; d < ndim; ++d) {
                    x[d] = xmin[d] + (double)in[d] * dx[d];
                    ibmn[d] = in[d] - 1; ibmx[d] = in[d] + 1;
                    if (in[d] == 0) ibmn[d] = 0;
                    if (in[d] == inmx[d]) ibmx[d] = inmx[d];
                    ib[d] = ibmn[d];
                }
                for (int idm = 0; idm < ndim; ++idm)
                    for (int jdm = idm; jdm < ndim; ++jdm) {
                        for (int d = 0; d < ndim; ++d) nderiv[d] = 0;
                        int boundary = 1;
                        rowwt = 2.0 * dcwght;
                        if (jdm == idm) {
                            rowwt = dcwght;
                            nderiv[jdm] = 2;
                            if (in[idm] != 0 && in[idm] != inmx[idm]) boundary = 0;
                        }
                        if (boundary) { nderiv[idm] = 1; nderiv[jdm] = 1; }
                        int nz = 0;
                        for (;;) {
                            int icol;
                            double basm = oracle_bascmp(ndim, x, nderiv, ib, xmin, dx, nodes, &icol);
                            col[nz] = icol - 1;
                            val[nz++] = rowwt * basm;
                            int d = 0;
                            for (; d < ndim; ++d) {
                                if (++ib[d] <= ibmx[d]) break;
                                ib[d] = ibmn[d];
                            }
                            if (d == ndim) break;
                        }
                        if (sink(R, nz, col, val, 0.0)) { free(hist); return 1; }
                        ++*ncons;
                    }
                for (int d = 0; d < ndim; ++d) nderiv[d] = 0;
            }
            ++iin;
            int d = 0;
            for (; d < ndim; ++d) {
                if (++in[d] <= inmx[d]) break;
                in[d] = 0;
            }
            if (d == ndim) break;
        }
        free(hist);
    }
    return 0;
}

/* lower band storage, LAPACK 'L': A(i,j) = ab[(i - j) + j*ld], 0 <= i - j <= p, ld = p + 1 */
#define AB(i, j) ab[(size_t)((i) - (j)) + (size_t)(j) * ld]

/* blocked right-looking band Cholesky; the trailing update of every block step is shared by the cores */
/* (AVX2+FMA code path chosen at run time where the host has it: the library is built in one place and
 * timed in another) */
__attribute__((target_clones("avx2,fma", "default")))
static int band_cholesky(double *ab, long n, long p, double *minpiv)
{
    const long ld = p + 1, nb = 64;
    int bad = 0;
    double mp = INFINITY;
    for (long k0 = 0; k0 < n && !bad; k0 += nb) {
        const long kb = (k0 + nb < n) ? nb : n - k0;
        /* diagonal block: unblocked */
        for (long j = k0; j < k0 + kb; ++j) {
            double d = AB(j, j);
            if (!(d > 0.0)) { bad = 1; break; }
            if (d < mp) mp = d;
            d = sqrt(d);
            AB(j, j) = d;
            const long iend = (j + p < n - 1) ? j + p : n - 1;
            for (long i = j + 1; i <= iend; ++i) AB(i, j) /= d;
            /* update the remaining columns of the block only (the rest follows blockwise) */
            for (long c = j + 1; c < k0 + kb; ++c) {
                const double l = AB(c, j);
                if (l == 0.0) continue;
                const long ie = (j + p < n - 1) ? j + p : n - 1;
                for (long i = c; i <= ie; ++i) AB(i, c) -= AB(i, j) * l;
            }
        }
        if (bad) break;
        /* trailing update by the block's columns: columns c in (k0+kb, k0+kb+p) */
        const long cbeg = k0 + kb, cend = (k0 + kb - 1 + p < n - 1) ? k0 + kb - 1 + p : n - 1;
#pragma omp parallel for schedule(dynamic, 8)
        for (long c = cbeg; c <= cend; ++c) {
            for (long j = k0; j < k0 + kb; ++j) {
                if (c - j > p) continue;
                const double l = AB(c, j);
                if (l == 0.0) continue;
                const long ie = (j + p < n - 1) ? j + p : n - 1;
                double *restrict dst = &AB(c, c);
                const double *restrict src = &AB(c, j);
                const long len = ie - c + 1;
                for (long t = 0; t < len; ++t) dst[t] -= src[t] * l;
            }
        }
    }
    *minpiv = mp;
    return bad;
}

static void band_solve(const double *ab, long n, long p, double *x)
{
    const long ld = p + 1;
    for (long j = 0; j < n; ++j) {                        /* L y = b */
        x[j] /= AB(j, j);
        const double xj = x[j];
        const long ie = (j + p < n - 1) ? j + p : n - 1;
        for (long i = j + 1; i <= ie; ++i) x[i] -= AB(i, j) * xj;
    }
    for (long j = n - 1; j >= 0; --j) {                   /* L^T x = y */
        const long ie = (j + p < n - 1) ? j + p : n - 1;
        double s = x[j];
        for (long i = j + 1; i <= ie; ++i) s -= AB(i, j) * x[i];
        x[j] = s / AB(j, j);
    }
}

/* rho = A^T (b - A x) over all rows; returns ||b - A x||_2^2 */
static double row_residual(const rows_t *R, long n, const double *x, double *rho)
{
    memset(rho, 0, (size_t)n * sizeof(double));
    double ssq = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : ssq)
    for (long r = 0; r < R->nrows; ++r) {
        double e = R->rhs[r];
        for (long t = R->ptr[r]; t < R->ptr[r + 1]; ++t) e -= R->val[t] * x[R->col[t]];
        ssq += e * e;
        for (long t = R->ptr[r]; t < R->ptr[r + 1]; ++t) {
            const double v = R->val[t] * e;
#pragma omp atomic
            rho[R->col[t]] += v;
        }
    }
    return ssq;
}

/* info[0] data rows, [1] constraint rows, [2] refinement steps, [3] last correction, [4] min pivot,
 * [5] seconds rows+assembly, [6] seconds factorisation, [7] seconds solve+refine, [8] reserr, [9] threads */
int oracle_splcw_banded(int ndim, const double *xdata, int l1xdat, const double *ydata, const double *wdata,
                        int ndata, const double *xmin, const double *xmax, const int *nodes, double xtrap,
                        double *coef, int ncf, int nthreads, double *info)
{
    if (ndim < 1 || ndim > BMAXD) return 101;
    long n = 1, p = 0, stride = 1;
    for (int d = 0; d < ndim; ++d) {
        if (nodes[d] < 4) return 102;
        if (xmax[d] - xmin[d] == 0.0) return 103;
        p += 3 * stride;
        stride *= nodes[d];
        n *= nodes[d];
    }
    if (n > ncf) return 104;
    if (ndata < 1) return 105;
    if (p > n - 1) p = n - 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    const double t0 = omp_get_wtime();
#else
    const double t0 = 0.0;
#endif
    rows_t R;
    memset(&R, 0, sizeof R);
    long ncons = 0;
    int rc = 0;
    if (build_rows(ndim, xdata, l1xdat, ydata, wdata, ndata, xmin, xmax, nodes, xtrap, &R, &ncons)) rc = -2;
    const long ld = p + 1;
    double *ab = rc ? NULL : calloc((size_t)ld * (size_t)n, sizeof(double));
    double *rhs = rc ? NULL : calloc((size_t)n, sizeof(double));
    double *x = rc ? NULL : malloc((size_t)n * sizeof(double));
    double *rho = rc ? NULL : malloc((size_t)n * sizeof(double));
    if (!rc && (!ab || !rhs || !x || !rho)) rc = -2;
    if (!rc && R.nrows < n) rc = 107;
    double minpiv = 0.0, t1 = t0, t2 = t0, t3 = t0, last = 0.0, ssq = 0.0;
    int steps = 0;
    if (!rc) {
        /* N = sum of r r^T (lower band), A^T b */
#pragma omp parallel for schedule(static)
        for (long r = 0; r < R.nrows; ++r) {
            const long b = R.ptr[r], e = R.ptr[r + 1];
            for (long s = b; s < e; ++s) {
                const double vs = R.val[s];
                if (vs == 0.0) continue;
                const long cs = R.col[s];
                const double add = vs * R.rhs[r];
                if (add != 0.0) {
#pragma omp atomic
                    rhs[cs] += add;
                }
                for (long t = b; t < e; ++t) {
                    const long ct = R.col[t];
                    if (ct > cs) continue;                /* lower triangle: row cs >= column ct */
                    const double v = vs * R.val[t];
                    if (v == 0.0) continue;
#pragma omp atomic
                    AB(cs, ct) += v;
                }
            }
        }
#ifdef _OPENMP
        t1 = omp_get_wtime();
#endif
        if (band_cholesky(ab, n, p, &minpiv)) rc = 107;
#ifdef _OPENMP
        t2 = omp_get_wtime();
#endif
    }
    if (!rc) {
        memcpy(x, rhs, (size_t)n * sizeof(double));
        band_solve(ab, n, p, x);
        double prev = INFINITY;
        for (int it = 0; it < 8; ++it) {
            ssq = row_residual(&R, n, x, rho);
            band_solve(ab, n, p, rho);
            double mdx = 0.0, mx = 0.0;
            for (long i = 0; i < n; ++i) {
                x[i] += rho[i];
                if (fabs(rho[i]) > mdx) mdx = fabs(rho[i]);
                if (fabs(x[i]) > mx) mx = fabs(x[i]);
            }
            ++steps;
            last = mx > 0.0 ? mdx / mx : 0.0;
            if (!(last == last)) { rc = 107; break; }
            if (last <= 1e-14 || (it >= 1 && last > 0.5 * prev)) break;
            prev = last;
        }
        ssq = row_residual(&R, n, x, rho);
        memcpy(coef, x, (size_t)n * sizeof(double));
#ifdef _OPENMP
        t3 = omp_get_wtime();
#endif
    }
    if (info) {
        info[0] = (double)(R.nrows - ncons); info[1] = (double)ncons; info[2] = steps; info[3] = last;
        info[4] = minpiv; info[5] = t1 - t0; info[6] = t2 - t1; info[7] = t3 - t2; info[8] = sqrt(ssq);
#ifdef _OPENMP
        info[9] = (double)omp_get_max_threads();
#else
        info[9] = 1.0;
#endif
    }
    free(ab); free(rhs); free(x); free(rho);
    free(R.ptr); free(R.col); free(R.val); free(R.rhs);
    return rc;
}


/* ---------------------------------------------------------------------------------------------------------
 * Optimality of GIVEN coefficients with respect to the reference's rows, on the host, nothing stored:
 *     rho   = sum over rows  a_r (b_r - a_r . x)            (the gradient of 1/2 |A x - b|^2, sign flipped)
 *     denom = sum over rows  |a_r| (|a_r| . |x| + |b_r|)    (the size of the terms that cancel in rho)
 * omega = max_i |rho_i| / denom_i is a componentwise backward error of x: 0 at the reference's minimiser.
 * Rows are generated exactly as for oracle_splcw_banded (emit_rows: data rows :788-855, histogram and constraint
 * rows :862-1046); the data rows are spread over the host threads.  Used by the GPU test tier to check a fit at
 * BASELINE's full size independently of the GPU's own residual kernels.
 * Returns 0; *omega_out, *ssq_out (sum of squared row residuals), nrows_out[2] (data, constraint rows). */
typedef struct { const double *x; double *rho, *den; double ssq; long nrows; } grad_t;
static int sink_grad(void *ctx, int nz, const int *col, const double *val, double rhs)
{
    grad_t *G = (grad_t *)ctx;
    double dot = 0.0, adot = 0.0;
    for (int k = 0; k < nz; ++k) { dot += val[k] * G->x[col[k]]; adot += fabs(val[k]) * fabs(G->x[col[k]]); }
    const double res = rhs - dot;
    for (int k = 0; k < nz; ++k) {
        G->rho[col[k]] += val[k] * res;
        G->den[col[k]] += fabs(val[k]) * (adot + fabs(rhs));
    }
    G->ssq += res * res;
    ++G->nrows;
    return 0;
}

int oracle_rows_gradient(int ndim, const double *xdata, int l1xdat, const double *ydata, const double *wdata, int ndata,
                         const double *xmin, const double *xmax, const int *nodes, double xtrap, const double *coef,
                         int nthreads, double *omega_out, double *ssq_out, long *nrows_out)
{
    long ncol = 1;
    for (int d = 0; d < ndim; ++d) ncol *= nodes[d];
    if (nthreads < 1) nthreads = omp_get_max_threads();
    if (nthreads > 64) nthreads = 64;
    double *rho = calloc((size_t)ncol * (size_t)nthreads, sizeof(double));
    double *den = calloc((size_t)ncol * (size_t)nthreads, sizeof(double));
    double *ssq = calloc((size_t)nthreads, sizeof(double));
    long *nr = calloc((size_t)nthreads, sizeof(long));
    if (!rho || !den || !ssq || !nr) return 1;
    int bad = 0;
#pragma omp parallel num_threads(nthreads) reduction(|:bad)
    {
        const int t = omp_get_thread_num(), T = omp_get_num_threads();
        const long i0 = (long)ndata * t / T, i1 = (long)ndata * (t + 1) / T;
        grad_t G = {coef, rho + (size_t)ncol * t, den + (size_t)ncol * t, 0.0, 0};
        long nc = 0;
        bad |= emit_rows(ndim, xdata, l1xdat, ydata, wdata, ndata, (int)i0, (int)i1, 0, xmin, xmax, nodes, xtrap, sink_grad, &G, &nc);
        ssq[t] = G.ssq;
        nr[t] = G.nrows;
    }
    /* constraint rows (need the histogram of all points): one pass, thread 0's arrays */
    grad_t G = {coef, rho, den, 0.0, 0};
    long ncons = 0;
    bad |= emit_rows(ndim, xdata, l1xdat, ydata, wdata, ndata, 0, 0, 1, xmin, xmax, nodes, xtrap, sink_grad, &G, &ncons);
    double omega = 0.0, s2 = G.ssq;
    long ndat = 0;
    for (int t = 0; t < nthreads; ++t) { s2 += ssq[t]; ndat += nr[t]; }
    for (long i = 0; i < ncol; ++i) {
        double r = 0.0, d = 0.0;
        for (int t = 0; t < nthreads; ++t) { r += rho[(size_t)ncol * t + i]; d += den[(size_t)ncol * t + i]; }
        if (d > 0.0 && fabs(r) / d > omega) omega = fabs(r) / d;
    }
    free(rho); free(den); free(ssq); free(nr);
    if (omega_out) *omega_out = omega;
    if (ssq_out) *ssq_out = s2;
    if (nrows_out) { nrows_out[0] = ndat; nrows_out[1] = ncons; }
    return bad;
}
