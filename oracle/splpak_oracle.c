/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement ("port") of the reference algorithm.
 *
 * This file restates, in plain C, WHAT jacobwilliams/splpak computes on the
 * fit/evaluate hot path so that the HIP product path can be checked against it.
 * It is never linked into, imported by or executed from the product library:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Parity pin: checked against golden vectors generated from the UNMODIFIED
 * reference (oracle/_ref, built by `make ref` from /root/reference/src/splpak.F90)
 * -- see tests/test_oracle_golden.py and oracle/gen_golden.py -- and against the
 * reference's own known-answer test (test/splpak_test_linear.f90:79-83, slope 2
 * within 1e-12).
 *
 * Citations are into /root/reference/src/splpak.F90.
 *
 *   oracle_bascmp   <- bascmp  :206-389   (tensor-product basis value + column index)
 *   oracle_splcw    <- splcw   :512-1060  (row assembly, sparse-area histogram,
 *                                          derivative-constraint rows, solve)
 *   rls_*           <- suprls  :1375-1695 (row-streaming dense Householder/Givens LS)
 *   oracle_splde    <- splde   :1089-1240 (evaluation; splfe :1258-1275 is nderiv=0)
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* 1-D basis functions (bascmp :231-381; closed forms in SURVEY.md appendix A) */
/* ------------------------------------------------------------------------- */

/* kind: 1 = left-linear (ib <= 1), 2 = chapeau, 3 = right-linear (ib >= nodes-2)
 * (:231-240).  deriv in {0,1,2}.  s = dxin, xb = node location. */
static double basis_1d(int kind, int deriv, double x, double xb, double s)
{
    double b = 0.0, z, f;
    if (kind == 2) {
        if (deriv == 0) {                       /* ngo 4, :253-270 */
            z = fabs(s * (x - xb)) - 2.0;
            if (z < 0.0) {
                b = -0.25 * z * z * z;
                z += 1.0;
                if (z < 0.0) b += z * z * z;
            }
        } else if (deriv == 1) {                /* ngo 5, :272-286 */
            z = x - xb;
            f = s;
            if (z < 0.0) f = -f;
            z = f * z - 2.0;
            if (z < 0.0) {
                b = -0.75 * z * z;
                z += 1.0;
                if (z < 0.0) b += 3.0 * z * z;
                b *= f;
            }
        } else {                                /* ngo 6, :288-300 */
            f = s;
            z = f * fabs(x - xb) - 2.0;
            if (z < 0.0) {
                b = -1.5 * z;
                z += 1.0;
                if (z < 0.0) b += 6.0 * z;
                b *= f * f;
            }
        }
        return b;
    }
    /* kinds 1 and 3 share one shape; kind 1 is the mirror image (:345-356) */
    if (deriv == 0) {                           /* ngo 1 / 7, :345-379 */
        z = (kind == 1) ? s * (xb - x) + 2.0 : s * (x - xb) + 2.0;
        if (z > 0.0) {
            if (z < 2.0) {
                b = 0.5 * z * z * z;
                z -= 1.0;
                if (z > 0.0) b -= z * z * z;
            } else {
                b = 3.0 * z - 3.0;
            }
        }
    } else if (deriv == 1) {                    /* ngo 2 / 8, :302-322 */
        f = (kind == 1) ? -s : s;
        z = f * (x - xb) + 2.0;
        if (z > 0.0) {
            if (z < 2.0) {
                b = 1.5 * z * z;
                z -= 1.0;
                if (z > 0.0) b -= 3.0 * z * z;
                b *= f;
            } else {
                b = 3.0 * f;
            }
        }
    } else {                                    /* ngo 3 / 9, :324-340 */
        double z1;
        f = (kind == 1) ? -s : s;
        z = f * (x - xb) + 2.0;
        z1 = z - 1.0;
        if (fabs(z1) < 1.0) {
            b = 3.0 * z;
            if (z1 > 0.0) b -= 6.0 * z1;
            b *= f * f;
        }
    }
    return b;
}

/* bascmp :206-389.  ib[] are 0-based node indices; returns basm and the 1-based
 * column index in *icol (leftmost index fastest, :227-228, :387). */
double oracle_bascmp(int mdim, const double *x, const int *nderiv, const int *ib,
                     const double *xmin, const double *dx, const int *nodes, int *icol)
{
    int col = 0;
    double basm = 1.0;
    for (int d = 0; d < mdim; ++d) {
        int hd = mdim - 1 - d;                  /* Horner from the last dim down */
        col = nodes[hd] * col + ib[hd];
        int kind = 1;
        if (ib[d] > 1) {
            kind = 2;
            if (ib[d] >= nodes[d] - 2) kind = 3;
        }
        double xb = xmin[d] + (double)ib[d] * dx[d];
        double s = 1.0 / dx[d];
        basm *= basis_1d(kind, nderiv[d], x[d], xb, s);
    }
    *icol = col + 1;
    return basm;
}

/* ------------------------------------------------------------------------- */
/* Row-streaming least squares (suprls :1375-1695)                            */
/* ------------------------------------------------------------------------- */

typedef struct {
    int n, np1;          /* columns, n+1 (row length incl. rhs)                 */
    long nn;             /* scratch length the caller offered (:1384)           */
    long kdone;          /* rows reduced so far (me%k; NOT capped at n)         */
    int nbuf;            /* buffered (not yet reduced) rows (me%l - me%k)       */
    long cap;            /* buffer capacity in rows for the current batch       */
    long entered;        /* rows accepted so far (me%isav)                      */
    long iold;           /* last accepted row index (me%iold)                   */
    double errsum;
    double *tri;         /* packed triangle: row j (0-based) holds cols j..n    */
    double *buf;         /* buffered rows, each np1 long                        */
    int quiet;
} rls_t;

static long tri_len(int np1, int rows)
{   /* packed entries of the first `rows` triangle rows: sum_{j=1..rows} (np1-j+1) */
    return (long)rows * np1 - (long)rows * (rows - 1) / 2;
}

static double *tri_row(rls_t *s, int j)
{   /* pointer to packed row j (0-based); element c (c>=j) at [c-j] */
    return s->tri + tri_len(s->np1, j);
}

static void rls_msg(const rls_t *s, int ier, const char *m)
{   /* cfaerr :399-407 */
    if (s->quiet) return;
    if (ier) printf(" IERR=%5d\n", ier);
    printf("%s\n", m);
}

/* Apply the batch reduction to everything currently buffered (:1481-1643). */
static void rls_reduce(rls_t *s)
{
    const int n = s->n, np1 = s->np1;
    const double tol = 1.0e-18;                 /* :1423 */
    const int c = s->nbuf;                      /* new rows, me%l - me%k */
    const long l = s->kdone + c;                /* total rows present, me%l */
    const int k = (int)(s->kdone < n ? s->kdone : n);   /* triangle rows, min(me%k,n) */
    double *B = s->buf;

    if (s->kdone != 0) {
        if (c == 1) {
            /* Givens rotations against the single new row (:1488-1515) */
            for (int j = 0; j < k; ++j) {
                double *t = tri_row(s, j);       /* t[0] is the diagonal */
                double sq;
                if (fabs(B[j]) <= tol)        sq = sqrt(t[0] * t[0]);
                else if (fabs(t[0]) < tol)    sq = sqrt(B[j] * B[j]);
                else                          sq = sqrt(t[0] * t[0] + B[j] * B[j]);
                if (sq == 0.0) continue;
                double tmp = t[0];
                t[0] = sq;
                sq = 1.0 / sq;
                double cn = tmp * sq, sn = B[j] * sq;
                for (int q = j + 1; q < np1; ++q) {
                    double tv = t[q - j];
                    t[q - j] = cn * tv + sn * B[q];
                    B[q] = -sn * tv + cn * B[q];
                }
            }
        } else {
            /* Householder spanning the diagonal and all new rows (:1516-1549) */
            for (int j = 0; j < k; ++j) {
                double *t = tri_row(s, j);
                double sq = t[0] * t[0];
                for (int r = 0; r < c; ++r) sq += B[(long)r * np1 + j] * B[(long)r * np1 + j];
                if (sq == 0.0) continue;
                double tmp = t[0];
                t[0] = sqrt(sq);
                if (tmp > 0.0) t[0] = -t[0];
                tmp -= t[0];
                double tmp1 = 1.0 / (tmp * t[0]);
                for (int q = j + 1; q < np1; ++q) {
                    double acc = tmp * t[q - j];
                    for (int r = 0; r < c; ++r) acc += B[(long)r * np1 + j] * B[(long)r * np1 + q];
                    acc *= tmp1;
                    t[q - j] += acc * tmp;
                    for (int r = 0; r < c; ++r) B[(long)r * np1 + q] += acc * B[(long)r * np1 + j];
                }
            }
        }
        if (s->kdone >= n) {
            /* triangle already complete: what is left of the new rows is residual
             * (:1551-1566); they are then discarded */
            for (int r = 0; r < c; ++r) {
                double v = B[(long)r * np1 + n];
                s->errsum += v * v;
            }
            s->nbuf = 0;
            s->kdone = l;
            s->cap = (s->nn - tri_len(np1, n)) / np1;
            return;
        }
    }

    /* Extend the triangle with the new rows (:1569-1619): buffer row r becomes
     * triangle row k+r. */
    const int knew = (int)(l < n ? l : n);      /* me%k1 = min(l,n) */
    if (c != 1) {
        int jend = knew - 1;                    /* k1m1 = k1-1 ... */
        if (l > n) jend = n;                    /* ... or n when there are spare rows */
        for (int j = k; j < jend; ++j) {
            const int r0 = j - k;               /* pivot row inside the buffer */
            double sq = 0.0;
            for (int r = r0; r < c; ++r) sq += B[(long)r * np1 + j] * B[(long)r * np1 + j];
            if (sq == 0.0) continue;
            double *p = B + (long)r0 * np1;
            double tmp = p[j];
            p[j] = sqrt(sq);
            if (tmp > 0.0) p[j] = -p[j];
            tmp -= p[j];
            double tmp1 = 1.0 / (tmp * p[j]);
            for (int q = j + 1; q < np1; ++q) {
                double acc = tmp * p[q];
                for (int r = r0 + 1; r < c; ++r) acc += B[(long)r * np1 + j] * B[(long)r * np1 + q];
                acc *= tmp1;
                p[q] += acc * tmp;
                for (int r = r0 + 1; r < c; ++r) B[(long)r * np1 + q] += acc * B[(long)r * np1 + j];
            }
        }
        if (l > n) {
            /* rows below the completed triangle are residual (:1610-1618) */
            for (int r = n - k; r < c; ++r) {
                double v = B[(long)r * np1 + n];
                s->errsum += v * v;
            }
        }
    }
    /* "squeeze": keep only columns j..n of new triangle row j (:1620-1635) */
    for (int j = k; j < knew; ++j) {
        double *t = tri_row(s, j);
        const double *p = B + (long)(j - k) * np1;
        for (int q = j; q < np1; ++q) t[q - j] = p[q];
    }
    s->nbuf = 0;
    s->kdone = l;
    s->cap = (s->nn - tri_len(np1, knew)) / np1;     /* :1640 */
}

static int rls_init(rls_t *s, int n, long nn, int quiet)
{
    memset(s, 0, sizeof(*s));
    s->n = n; s->np1 = n + 1; s->nn = nn; s->quiet = quiet;
    s->cap = nn / s->np1;                       /* :1437 */
    /* the reference works in place in a(nn); the restatement keeps the triangle and
     * the row buffer apart but sizes every batch exactly as the in-place layout allows */
    s->tri = (double *)calloc((size_t)tri_len(s->np1, n) + 1, sizeof(double));
    long caprows = s->cap > 0 ? s->cap : 1;
    s->buf = (double *)calloc((size_t)caprows * s->np1, sizeof(double));
    return (s->tri && s->buf) ? 0 : -1;
}

static void rls_free(rls_t *s) { free(s->tri); free(s->buf); s->tri = s->buf = NULL; }

/* One call of suprls with i >= 1 (:1428-1479, then the reduction when the batch is full). */
static int rls_push(rls_t *s, long i, const double *row, double rhs)
{
    const int n = s->n, np1 = s->np1;
    if (i <= 1) {
        s->iold = 0; s->kdone = 0; s->nbuf = 0; s->errsum = 0.0; s->entered = 0;
        s->cap = s->nn / np1;                    /* :1437 */
        long nreq = ((long)(n + 5) * n + 2) / 2; /* :1443 */
        if (s->nn < nreq) {
            if (!s->quiet) { printf(" nn   =  %ld\n nreq =  %ld\n", s->nn, nreq); }
            rls_msg(s, 32, " suprls - insufficient scratch storage provided. "
                           "at least ((N+5)*N+2)/2 locations needed");
            return 32;
        }
    }
    if (i - s->iold != 1) {                      /* :1459-1465 */
        if (!s->quiet) { printf(" i    = %ld\n me%%iold = %ld\n", i, s->iold); }
        rls_msg(s, 35, " suprls - values of I not in sequence");
        return 35;
    }
    s->iold = i;
    double *dst = s->buf + (long)s->nbuf * np1;
    memcpy(dst, row, (size_t)n * sizeof(double));
    dst[n] = rhs;
    s->nbuf++;
    s->entered = i;
    if (s->nbuf < s->cap) return 0;              /* :1477, i < me%l */
    rls_reduce(s);
    return 0;
}

/* Final call (i = 0): finish the reduction and back-substitute (:1645-1693). */
static int rls_solve(rls_t *s, double *soln, double *err)
{
    const int n = s->n;
    if (s->entered < n) {                        /* :1650-1654 */
        rls_msg(s, 33, " suprls - array has too few rows.");
        return 33;
    }
    if (s->nbuf > 0) rls_reduce(s);          /* :1657, k /= isav */
    double *t = tri_row(s, n - 1);               /* last row: [diag, rhs] */
    if (t[0] == 0.0) {                           /* :1662-1667 */
        rls_msg(s, 34, " suprls - system is singular.");
        return 34;
    }
    soln[n - 1] = t[1] / t[0];
    for (int j = n - 2; j >= 0; --j) {           /* :1671-1690 */
        t = tri_row(s, j);
        double acc = t[n - j];                   /* rhs entry of row j */
        for (int q = n - 1; q > j; --q) acc -= t[q - j] * soln[q];
        if (t[0] == 0.0) {
            rls_msg(s, 34, " suprls - system is singular.");
            return 34;
        }
        soln[j] = acc / t[0];
    }
    *err = sqrt(s->errsum);                      /* :1693 */
    return 0;
}

/* ------------------------------------------------------------------------- */
/* splcw (:512-1060); splcc is wdata = [-1] (:440)                            */
/* ------------------------------------------------------------------------- */

static void say(int quiet, int ierr, const char *m)
{
    if (quiet) return;
    if (ierr) printf(" IERR=%5d\n", ierr);
    printf("%s\n", m);
}

#define MAXD 16

/* the residual norm suprls returns (:1693) and splcw discards (:690): kept so that the
 * GPU path's diagnostic can be checked against it */
static double g_last_reserr = 0.0;
double oracle_last_reserr(void) { return g_last_reserr; }

int oracle_splcw(int ndim, const double *xdata, int l1xdat, const double *ydata,
                 const double *wdata, int ndata, const double *xmin, const double *xmax,
                 const int *nodes, double xtrap, double *coef, int ncf,
                 double *work, long nwrk, int quiet)
{
    const char *fail107 = " splcc or splcw - suprls failure "
                          "(this usually indicates insufficient input data)";
    int ierror = 0;
    double dx[MAXD], dxin[MAXD], x[MAXD];
    int nderiv[MAXD], ib[MAXD], ibmn[MAXD], ibmx[MAXD], in[MAXD], inmx[MAXD];

    if (ndim < 1) { say(quiet, 101, " splcc or splcw - NDIM is less than 1"); return 101; }
    if (ndim > MAXD) return 101;
    long ncol = 1;
    for (int d = 0; d < ndim; ++d) {             /* :726-750 */
        int nod = nodes[d];
        if (nod < 4) { say(quiet, 102, " splcc or splcw - NODES(IDIM) is less than 4 for some IDIM"); return 102; }
        ncol *= nod;
        double xrng = xmax[d] - xmin[d];
        if (xrng == 0.0) { say(quiet, 103, " splcc or splcw - XMIN(IDIM) equals XMAX(IDIM) for some IDIM"); return 103; }
        dx[d] = xrng / (double)(nod - 1);
        dxin[d] = 1.0 / dx[d];
        nderiv[d] = 0;
    }
    if (ncol > ncf) { say(quiet, 104, " splcc or splcw - NCF (size of COEF) is too small"); return 104; }
    if (ndata < 1) { say(quiet, 105, " splcc or splcw - Ndata Is less than 1"); return 105; }
    const double swght = xtrap;
    long nwrk1 = 1;
    if (swght != 0.0) nwrk1 = ncol + 1;          /* :772 */
    long nwlft = nwrk - nwrk1 + 1;               /* :775 */
    if (nwlft < 1) { say(quiet, 106, " splcc or splcw - NWRK (size of WORK) is too small"); return 106; }

    const int n = (int)ncol;
    const int weighted = wdata[0] >= 0.0;        /* :796 */
    rls_t S;
    if (rls_init(&S, n, nwlft, quiet)) return -1;
    long irow = 0;
    double rowwt = 1.0;

    /* ---- data rows (:788-855) ---- */
    for (int idata = 0; idata < ndata; ++idata) {
        if (weighted) {
            rowwt = wdata[idata];
            if (rowwt == 0.0) continue;
        }
        ++irow;
        double rhs = rowwt * ydata[idata];
        for (int d = 0; d < ndim; ++d) x[d] = xdata[(long)idata * l1xdat + d];
        memset(coef, 0, (size_t)n * sizeof(double));
        for (int d = 0; d < ndim; ++d) {         /* window rule :821-827 */
            int nod = nodes[d];
            int it = (int)(dxin[d] * (x[d] - xmin[d]));      /* truncation toward zero */
            int lo = it - 1; if (lo < 0) lo = 0; if (lo > nod - 2) lo = nod - 2;
            int hi = it + 2; if (hi > nod - 1) hi = nod - 1; if (hi < 1) hi = 1;
            ibmn[d] = lo; ib[d] = lo; ibmx[d] = hi;
        }
        for (;;) {                               /* odometer, dim 1 fastest :829-846 */
            int icol;
            double basm = oracle_bascmp(ndim, x, nderiv, ib, xmin, dx, nodes, &icol);
            coef[icol - 1] = rowwt * basm;
            int d = 0;
            for (; d < ndim; ++d) {
                if (++ib[d] <= ibmx[d]) break;
                ib[d] = ibmn[d];
            }
            if (d == ndim) break;
        }
        if (rls_push(&S, irow, coef, rhs)) { ierror = 107; say(quiet, 107, fail107); }
    }

    /* ---- data-sparse areas (:862-1048) ---- */
    if (swght != 0.0) {
        long nrect = 1;
        for (int d = 0; d < ndim; ++d) { in[d] = 0; inmx[d] = nodes[d] - 1; nrect *= inmx[d]; }
        for (long i = 0; i < ncol; ++i) work[i] = 0.0;
        double totlwt = 0.0;
        for (int idata = 0; idata < ndata; ++idata) {        /* histogram :886-907 */
            double bump = 1.0;
            if (weighted) bump = wdata[idata];
            if (bump == 0.0) continue;
            long iin = 0;
            for (int dc = 0; dc < ndim; ++dc) {
                int d = ndim - 1 - dc;
                int inidim = (int)(dxin[d] * (xdata[(long)idata * l1xdat + d] - xmin[d]) + 0.5);
                /* out-of-range coordinate: that dimension is skipped in the Horner
                 * address but the point is still counted (:899) */
                if (inidim < 0 || inidim > inmx[d]) continue;
                iin = (long)(inmx[d] + 1) * iin + inidim;
            }
            work[iin] += bump;
            totlwt += bump;
        }
        const double wtprrc = totlwt / (double)nrect;        /* :910 */
        const double spcrit = 0.75;                           /* :696 */
        long iin = 0;
        for (;;) {                                            /* node odometer :921-1046 */
            double expect = wtprrc;
            for (int d = 0; d < ndim; ++d)
                if (in[d] == 0 || in[d] == inmx[d]) expect *= 0.5;
            if (work[iin] < spcrit * expect) {
                double dcwght = expect - work[iin];
                for (int d = 0; d < ndim; ++d) {
                    x[d] = xmin[d] + (double)in[d] * dx[d];
                    ibmn[d] = in[d] - 1; ibmx[d] = in[d] + 1;
                    if (in[d] == 0) ibmn[d] = 0;
                    if (in[d] == inmx[d]) ibmx[d] = inmx[d];
                    ib[d] = ibmn[d];
                }
                dcwght *= swght;
                memset(coef, 0, (size_t)n * sizeof(double));
                for (int idm = 0; idm < ndim; ++idm) {
                    for (int jdm = idm; jdm < ndim; ++jdm) {
                        for (int d = 0; d < ndim; ++d) nderiv[d] = 0;
                        int boundary = 1;
                        rowwt = 2.0 * dcwght;                 /* off-diagonal appears twice */
                        if (jdm == idm) {
                            rowwt = dcwght;
                            nderiv[jdm] = 2;
                            if (in[idm] != 0 && in[idm] != inmx[idm]) boundary = 0;
                        }
                        if (boundary) { nderiv[idm] = 1; nderiv[jdm] = 1; }
                        ++irow;
                        for (;;) {
                            int icol;
                            double basm = oracle_bascmp(ndim, x, nderiv, ib, xmin, dx, nodes, &icol);
                            coef[icol - 1] = rowwt * basm;
                            int d = 0;
                            for (; d < ndim; ++d) {
                                if (++ib[d] <= ibmx[d]) break;
                                ib[d] = ibmn[d];
                            }
                            if (d == ndim) break;
                        }
                        if (rls_push(&S, irow, coef, 0.0)) { ierror = 107; say(quiet, 107, fail107); }
                    }
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
    }

    /* ---- solve (:1051-1058) ---- */
    double reserr = 0.0;
    if (rls_solve(&S, coef, &reserr)) { ierror = 107; say(quiet, 107, fail107); }
    g_last_reserr = reserr;
    rls_free(&S);
    return ierror;
}

/* ------------------------------------------------------------------------- */
/* splde (:1089-1240); nderiv == NULL gives splfe (:1258-1275)                */
/* ------------------------------------------------------------------------- */

double oracle_splde(int ndim, const double *x, const int *nderiv_in, const double *coef,
                    const double *xmin, const double *xmax, const int *nodes, int *ierror)
{
    double dx[MAXD];
    int ib[MAXD], ibmn[MAXD], ibmx[MAXD], nderiv[MAXD];
    *ierror = 0;
    if (ndim < 1 || ndim > MAXD) { *ierror = 101; return 0.0; }
    long total = 1;
    for (int d = 0; d < ndim; ++d) {
        int nod = nodes[d];
        if (nod < 4) { *ierror = 102; return 0.0; }
        double xrng = xmax[d] - xmin[d];
        if (xrng == 0.0) { *ierror = 103; return 0.0; }
        nderiv[d] = nderiv_in ? nderiv_in[d] : 0;
        if (nderiv[d] < 0 || nderiv[d] > 2) *ierror = 104;   /* does not return, :1190-1194 */
        dx[d] = xrng / (double)(nod - 1);
        double dxin = 1.0 / dx[d];
        int it = (int)(dxin * (x[d] - xmin[d]));
        int lo = it - 1; if (lo < 0) lo = 0; if (lo > nod - 2) lo = nod - 2;
        int hi = it + 2; if (hi > nod - 1) hi = nod - 1; if (hi < 1) hi = 1;
        ibmn[d] = lo; ibmx[d] = hi; ib[d] = lo;
        total *= (hi - lo + 1);
    }
    if (*ierror == 104) {
        /* the reference computes on with an out-of-range nderiv (undefined select-case
         * fall-through); the restatement clamps so the call stays defined */
        for (int d = 0; d < ndim; ++d) { if (nderiv[d] < 0) nderiv[d] = 0; if (nderiv[d] > 2) nderiv[d] = 2; }
    }
    double sum = 0.0;
    for (long t = 0; t < total; ++t) {
        int icof;
        double basm = oracle_bascmp(ndim, x, nderiv, ib, xmin, dx, nodes, &icof);
        sum += coef[icof - 1] * basm;
        for (int d = 0; d < ndim; ++d) {
            if (++ib[d] <= ibmx[d]) break;
            ib[d] = ibmn[d];
        }
    }
    return sum;
}

int oracle_splde_many(int ndim, long nq, const double *xq, int ldxq, const int *nderiv,
                      const double *coef, const double *xmin, const double *xmax,
                      const int *nodes, double *out)
{
    int last = 0;
    for (long i = 0; i < nq; ++i) {
        int ie;
        out[i] = oracle_splde(ndim, xq + i * ldxq, nderiv, coef, xmin, xmax, nodes, &ie);
        if (ie) last = ie;
    }
    return last;
}
