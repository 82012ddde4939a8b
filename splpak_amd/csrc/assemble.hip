// Assembly of the least-squares problem on the GPU.
//
// The reference builds one dense row per data point and streams it through a
// dense Householder solver (splcw :788-855 -> suprls :1375-1695), O(ncol^2) work
// per row.  Each row has at most 4^ndim non-zeros, all inside one 4-wide window
// of nodes per dimension, so here the points are binned by window ("cell"), and
// every cell contributes ONE dense 4^ndim x 4^ndim Gram block B^T W^2 B to the
// normal equations N = A^T A, r = A^T b, stored as a half stencil
// nst[ncol][(7^ndim+1)/2] (row i, columns i+offset with offset tuple in
// [-3,3]^ndim, lower triangle only).
//
// Kernels (all HBM-bound streaming passes; algorithmic bytes 8*(ndim+1+[weighted])
// per point and pass):
//   keys_kernel        window key per point, per-cell counts, nearest-node
//                      sparse-area histogram (:886-907) and total weight
//   scan_kernel        exclusive scan of the counts
//   scatter_kernel     counting-sort scatter into cell-ordered SoA copies
//   gram_kernel        per-cell Gram block: point tables staged in LDS, register
//                      tiled rank-1 updates, f64 atomics into the half stencil
//   residual_kernel    rho += A^T W (W y - W A x) for iterative refinement
//   constraint_kernel  derivative-constraint rows of data-sparse nodes (:921-1046)
//   expand_kernel      half stencil -> band storage of the Cholesky factorisation
#include "basis.hpp"
#include "kernels.hpp"

namespace splpak {

namespace {

__device__ inline double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------
template <int D>
__global__ void __launch_bounds__(256)
keys_kernel(Grid g, long long m, const double *__restrict__ x, int ldx,
            const double *__restrict__ w, int *__restrict__ key, int *__restrict__ count,
            double *__restrict__ hist, double *__restrict__ scal)
{
    double lw = 0.0, lrows = 0.0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
        const double wv = w ? w[i] : 1.0;
        int k = g.ncell;                       // zero weight: ignored (:799, :891)
        if (wv != 0.0) {
            double xv[D];
            k = 0;
#pragma unroll
            for (int d = 0; d < D; ++d) xv[d] = x[i * ldx + d];        // caller's dimension order
#pragma unroll
            for (int d = 0; d < D; ++d) {
                int lo, hi;
                k += window_start(g, d, xv[g.perm[d]], lo, hi) * g.cellstride[d];
            }
            lrows += 1.0;
            if (hist) {
                atomicAdd(&hist[nearest_node_address(g, xv)], wv);   // :905
                lw += wv;                                            // :906
            }
        }
        key[i] = k;
        atomicAdd(&count[k], 1);
    }
    lw = wave_sum(lw);
    lrows = wave_sum(lrows);
    __shared__ double red[2][4];
    const int lane = threadIdx.x & 63, wv_id = threadIdx.x >> 6;
    if (lane == 0) { red[0][wv_id] = lw; red[1][wv_id] = lrows; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double a = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        const double b = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        if (a != 0.0) atomicAdd(&scal[SC_TOTLWT], a);
        if (b != 0.0) atomicAdd(&scal[SC_NROWS_DATA], b);
    }
}

// exclusive scan of count[0..n) into offset[0..n], single workgroup
__global__ void __launch_bounds__(1024)
scan_kernel(const int *__restrict__ count, int *__restrict__ offset, int n)
{
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int chunk = (n + 1023) / 1024;
    const int lo = t * chunk;
    const int hi = (lo + chunk < n) ? lo + chunk : n;
    int s = 0;
    for (int i = lo; i < hi; ++i) s += count[i];
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {        // Hillis-Steele inclusive scan
        int v = (t >= o) ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;                      // exclusive prefix of this chunk
    for (int i = lo; i < hi; ++i) { offset[i] = run; run += count[i]; }
    if (t == 1023) offset[n] = part[1023];
}

template <int D>
__global__ void __launch_bounds__(256)
scatter_kernel(Grid g, long long m, const double *__restrict__ x, int ldx,
               const double *__restrict__ y, const double *__restrict__ w,
               const int *__restrict__ key, const int *__restrict__ offset,
               int *__restrict__ cursor, double *__restrict__ xs, double *__restrict__ ys,
               double *__restrict__ ws, long long cap)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
        const int k = key[i];
        if (k >= g.ncell) continue;
        const long long pos = (long long)offset[k] + atomicAdd(&cursor[k], 1);
#pragma unroll
        for (int d = 0; d < D; ++d) xs[(long long)d * cap + pos] = x[i * ldx + g.perm[d]];
        ys[pos] = y[i];
        ws[pos] = w ? w[i] : 1.0;
    }
}

// ---------------------------------------------------------------------------
// Per-cell staging shared by the Gram and the residual kernels: for `np` points
// starting at sorted position `p0`, fill bw[p*LDB + c] = w * ((b0*b1)*b2...) --
// the row of the weighted least-squares matrix restricted to the cell's window
// (:833-837) -- and wy[p] = w*y (:806).
template <int D, int NB, int LDB, int NT>
__device__ inline void stage_points(const Grid &g, const double *__restrict__ xs,
                                    const double *__restrict__ ys,
                                    const double *__restrict__ ws, long long cap, long long p0,
                                    int np, double *tab /*[PCH][D][4]*/, double *bw, double *wy,
                                    double *wt)
{
    const int tid = threadIdx.x;
    for (int p = tid; p < np; p += NT) {
        const double wv = ws[p0 + p];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            double b[4];
            window_table(g, d, xs[(long long)d * cap + p0 + p], 0, b);
#pragma unroll
            for (int k = 0; k < 4; ++k) tab[(p * D + d) * 4 + k] = b[k];
        }
        wy[p] = wv * ys[p0 + p];
        wt[p] = wv;
    }
    __syncthreads();
    for (int idx = tid; idx < np * NB; idx += NT) {
        const int p = idx / NB, c = idx % NB;
        double prod = tab[(p * D + 0) * 4 + (c & 3)];
#pragma unroll
        for (int d = 1; d < D; ++d) prod *= tab[(p * D + d) * 4 + ((c >> (2 * d)) & 3)];
        bw[p * LDB + c] = wt[p] * prod;
    }
    __syncthreads();
}

// column of local basis index c (base-4 digits, dim 0 fastest) in a window whose
// first node has column `colbase`
template <int D>
__device__ inline int local_col(const Grid &g, int colbase, int c)
{
    int col = colbase;
#pragma unroll
    for (int d = 0; d < D; ++d) col += ((c >> (2 * d)) & 3) * g.colstride[d];
    return col;
}

template <int D>
struct GramCfg;
template <> struct GramCfg<1> { static constexpr int NB = 4,   TR = 1, TC = 1, NTY = 4,  NTX = 4,  NT = 64,   PCH = 64; };
template <> struct GramCfg<2> { static constexpr int NB = 16,  TR = 1, TC = 1, NTY = 16, NTX = 16, NT = 256,  PCH = 128; };
template <> struct GramCfg<3> { static constexpr int NB = 64,  TR = 4, TC = 4, NTY = 16, NTX = 16, NT = 256,  PCH = 64; };
template <> struct GramCfg<4> { static constexpr int NB = 256, TR = 4, TC = 4, NTY = 64, NTX = 16, NT = 1024, PCH = 16; };

template <int D>
__global__ void __launch_bounds__(GramCfg<D>::NT)
gram_kernel(Grid g, const int *__restrict__ offset, const double *__restrict__ xs,
            const double *__restrict__ ys, const double *__restrict__ ws, long long cap,
            double *__restrict__ nst, double *__restrict__ rhs)
{
    using C = GramCfg<D>;
    constexpr int NB = C::NB, TR = C::TR, TC = C::TC, NT = C::NT, PCH = C::PCH;
    constexpr int CW = C::NTX * TC;            // columns handled by this workgroup
    const int cell = blockIdx.x;
    const int cpass = blockIdx.y;
    const long long beg = offset[cell], end = offset[cell + 1];
    if (beg == end) return;

    __shared__ double tab[PCH * D * 4];
    __shared__ double bw[PCH * NB];
    __shared__ double wy[PCH];
    __shared__ double wt[PCH];

    const int tid = threadIdx.x;
    const int ty = tid / C::NTX, tx = tid % C::NTX;
    const int r0 = ty * TR, c0 = cpass * CW + tx * TC;
    const bool tile_on = (tid < C::NTY * C::NTX) && (r0 + TR - 1 >= c0);
    const bool rhs_on = (cpass == 0) && (tid < NB);

    double acc[TR][TC];
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
        for (int j = 0; j < TC; ++j) acc[i][j] = 0.0;
    double racc = 0.0;

    for (long long p0 = beg; p0 < end; p0 += PCH) {
        const int np = (int)((end - p0 < PCH) ? (end - p0) : PCH);
        stage_points<D, NB, NB, NT>(g, xs, ys, ws, cap, p0, np, tab, bw, wy, wt);
        if (tile_on) {
            for (int p = 0; p < np; ++p) {
                double a[TR], b[TC];
#pragma unroll
                for (int i = 0; i < TR; ++i) a[i] = bw[p * NB + r0 + i];
#pragma unroll
                for (int j = 0; j < TC; ++j) b[j] = bw[p * NB + c0 + j];
#pragma unroll
                for (int i = 0; i < TR; ++i)
#pragma unroll
                    for (int j = 0; j < TC; ++j) acc[i][j] += a[i] * b[j];
            }
        }
        if (rhs_on)
            for (int p = 0; p < np; ++p) racc += bw[p * NB + tid] * wy[p];
        __syncthreads();
    }

    // window -> first node column
    int colbase = 0;
#pragma unroll
    for (int d = 0; d < D; ++d) colbase += ((cell / g.cellstride[d]) % g.cells[d]) * g.colstride[d];

    if (tile_on) {
#pragma unroll
        for (int i = 0; i < TR; ++i) {
            const int r = r0 + i;
            const int rowcol = local_col<D>(g, colbase, r);
#pragma unroll
            for (int j = 0; j < TC; ++j) {
                const int c = c0 + j;
                if (c > r) continue;           // lower triangle only
                int o[D];
#pragma unroll
                for (int d = 0; d < D; ++d) o[d] = ((c >> (2 * d)) & 3) - ((r >> (2 * d)) & 3);
                const int code = stencil_code(o, D);
                atomicAdd(&nst[(long long)rowcol * g.hstencil + code], acc[i][j]);
            }
        }
    }
    if (rhs_on) atomicAdd(&rhs[local_col<D>(g, colbase, tid)], racc);
}

template <int D>
struct ResCfg;
template <> struct ResCfg<1> { static constexpr int NB = 4,   NT = 64,  PCH = 64; };
template <> struct ResCfg<2> { static constexpr int NB = 16,  NT = 256, PCH = 128; };
template <> struct ResCfg<3> { static constexpr int NB = 64,  NT = 256, PCH = 64; };
template <> struct ResCfg<4> { static constexpr int NB = 256, NT = 256, PCH = 16; };

template <int D>
__global__ void __launch_bounds__(ResCfg<D>::NT)
residual_kernel(Grid g, const int *__restrict__ offset, const double *__restrict__ xs,
                const double *__restrict__ ys, const double *__restrict__ ws, long long cap,
                const double *__restrict__ xvec, double *__restrict__ rho, double *__restrict__ ssq)
{
    using C = ResCfg<D>;
    constexpr int NB = C::NB, NT = C::NT, PCH = C::PCH, LDB = NB + 1;
    const int cell = blockIdx.x;
    const long long beg = offset[cell], end = offset[cell + 1];
    if (beg == end) return;

    __shared__ double tab[PCH * D * 4];
    __shared__ double bw[PCH * LDB];
    __shared__ double wy[PCH];
    __shared__ double wt[PCH];
    __shared__ double xloc[NB];

    const int tid = threadIdx.x;
    int colbase = 0;
#pragma unroll
    for (int d = 0; d < D; ++d) colbase += ((cell / g.cellstride[d]) % g.cells[d]) * g.colstride[d];
    int mycol = 0;
    if (tid < NB) {
        mycol = local_col<D>(g, colbase, tid);
        xloc[tid] = xvec[mycol];
    }
    double racc = 0.0, e2 = 0.0;
    for (long long p0 = beg; p0 < end; p0 += PCH) {
        const int np = (int)((end - p0 < PCH) ? (end - p0) : PCH);
        stage_points<D, NB, LDB, NT>(g, xs, ys, ws, cap, p0, np, tab, bw, wy, wt);
        // e_p = w y - (w b) . x   (row residual)
        for (int p = tid; p < np; p += NT) {
            double dot = 0.0;
            for (int c = 0; c < NB; ++c) dot += bw[p * LDB + c] * xloc[c];
            wy[p] = wy[p] - dot;
            e2 += wy[p] * wy[p];
        }
        __syncthreads();
        if (tid < NB)
            for (int p = 0; p < np; ++p) racc += bw[p * LDB + tid] * wy[p];
        __syncthreads();
    }
    if (tid < NB) atomicAdd(&rho[mycol], racc);
    if (ssq) {                                   // sum of squared row residuals (the reference's errsum)
        e2 = wave_sum(e2);
        if ((tid & 63) == 0 && e2 != 0.0) atomicAdd(ssq, e2);
    }
}

// ---------------------------------------------------------------------------
// Derivative-constraint rows (:921-1046): one wave per node.
template <int D>
__global__ void __launch_bounds__(256)
constraint_kernel(Grid g, const double *__restrict__ hist, const double *__restrict__ scal,
                  double xtrap, double *__restrict__ nst, const double *__restrict__ xvec,
                  double *__restrict__ rho, double *__restrict__ scal_out, double *__restrict__ ssq)
{
#pragma clang fp contract(off)
    constexpr int NE = (D == 1) ? 3 : (D == 2) ? 9 : (D == 3) ? 27 : 81;
    __shared__ double cv_s[4][NE];
    __shared__ int col_s[4][NE];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int node = blockIdx.x * 4 + wave;
    const bool in_range = node < g.ncol;
    double *cv = cv_s[wave];
    int *cl = col_s[wave];

    int in[D];
    double xnode[D];
    bool sparse = false;
    double dcwght = 0.0;
    if (in_range) {
        long long nrect = 1;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            in[d] = (node / g.colstride[d]) % g.nodes[d];
            xnode[d] = g.xmin[d] + (double)in[d] * g.dx[d];          // :943
            nrect *= (g.nodes[d] - 1);
        }
        const double wtprrc = scal[SC_TOTLWT] / (double)nrect;        // :910
        double expect = wtprrc;
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (in[d] == 0 || in[d] == g.nodes[d] - 1) expect = 0.5 * expect;   // :928
        int refnode = 0;                                              // the histogram is in the caller's order
#pragma unroll
        for (int d = 0; d < D; ++d) refnode += in[d] * g.refstride[d];
        const double have = hist[refnode];
        sparse = have < 0.75 * expect;                                // spcrit, :696, :936
        dcwght = expect - have;                                       // :938
        dcwght = xtrap * dcwght;                                      // :960
    } else {
#pragma unroll
        for (int d = 0; d < D; ++d) { in[d] = 0; xnode[d] = 0.0; }
    }
    if (sparse && lane == 0 && nst)
        atomicAdd(&scal_out[SC_NROWS_CONS], (double)(D * (D + 1) / 2));

    for (int idm = 0; idm < D; ++idm) {
        for (int jdm = idm; jdm < D; ++jdm) {
            int nder[D];
#pragma unroll
            for (int d = 0; d < D; ++d) nder[d] = 0;
            bool boundary = true;
            double rowwt = 2.0 * dcwght;                              // :983
            if (jdm == idm) {
                rowwt = dcwght;
                nder[jdm] = 2;
                if (in[idm] != 0 && in[idm] != g.nodes[idm] - 1) boundary = false;
            }
            if (boundary) { nder[idm] = 1; nder[jdm] = 1; }            // :998-999
            __syncthreads();
            for (int e = lane; e < NE; e += 64) {
                int ee = e, col = 0;
                bool ok = sparse;
                double basm = 1.0;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const int ib = in[d] - 1 + (ee % 3);
                    ee /= 3;
                    if (ib < 0 || ib > g.nodes[d] - 1) { ok = false; continue; }
                    col += ib * g.colstride[d];
                    const double xb = g.xmin[d] + (double)ib * g.dx[d];
                    basm *= basis_1d(basis_kind(ib, g.nodes[d]), nder[d], xnode[d], xb, g.dxin[d]);
                }
                cv[e] = ok ? rowwt * basm : 0.0;                      // :1011
                cl[e] = ok ? col : 0;
            }
            __syncthreads();
            if (sparse) {
                if (nst) {
                    for (int q = lane; q < NE * NE; q += 64) {
                        const int e1 = q / NE, e2 = q % NE;
                        if (e2 > e1) continue;
                        const double v = cv[e1] * cv[e2];
                        if (v == 0.0) continue;
                        int o[D], a = e1, b = e2;
#pragma unroll
                        for (int d = 0; d < D; ++d) { o[d] = (b % 3) - (a % 3); a /= 3; b /= 3; }
                        atomicAdd(&nst[(long long)cl[e1] * g.hstencil + stencil_code(o, D)], v);
                    }
                }
                if (xvec) {
                    double t = 0.0;
                    for (int e = lane; e < NE; e += 64) t += cv[e] * xvec[cl[e]];
                    t = wave_sum(t);
                    for (int e = lane; e < NE; e += 64)
                        if (cv[e] != 0.0) atomicAdd(&rho[cl[e]], -cv[e] * t);
                    if (ssq && lane == 0) atomicAdd(ssq, t * t);     // constraint rows have rhs 0
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
template <int D>
__global__ void __launch_bounds__(256)
expand_kernel(Grid g, const double *__restrict__ nst, double *__restrict__ ab, long long lda)
{
    const long long total = (long long)g.ncol * g.hstencil;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int i = (int)(t / g.hstencil);
        int code = (int)(t % g.hstencil);
        int j = i;
        bool ok = true;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int o = (code % 7) - 3;
            code /= 7;
            const int id = (i / g.colstride[d]) % g.nodes[d];
            const int jd = id + o;
            if (jd < 0 || jd > g.nodes[d] - 1) ok = false;
            j += o * g.colstride[d];
        }
        if (!ok) continue;
        ab[(long long)i + (long long)j * lda] = nst[t];
    }
}

__global__ void pad_diag_kernel(double *ab, long long lda, int n, int npad)
{
    const int i = n + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npad) ab[(long long)i + (long long)i * lda] = 1.0;
}

inline unsigned grid_for(long long n, int threads, long long maxblocks = 256LL * 16)
{
    long long b = (n + threads - 1) / threads;
    if (b < 1) b = 1;
    if (b > maxblocks) b = maxblocks;
    return (unsigned)b;
}

}  // namespace

#define DISPATCH_D(ndim, CALL)        \
    switch (ndim) {                   \
    case 1: { constexpr int D = 1; CALL; } break; \
    case 2: { constexpr int D = 2; CALL; } break; \
    case 3: { constexpr int D = 3; CALL; } break; \
    default: { constexpr int D = 4; CALL; } break; \
    }

__global__ void __launch_bounds__(256)
to_reference_order_kernel(Grid g, const double *__restrict__ xvec, double *__restrict__ coef)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= g.ncol) return;
    int ref = 0;
    for (int d = 0; d < g.ndim; ++d) ref += ((col / g.colstride[d]) % g.nodes[d]) * g.refstride[d];
    coef[ref] = xvec[col];
}

hipError_t launch_to_reference_order(const Grid &g, const double *xvec, double *coef, hipStream_t st)
{
    bool identity = true;
    for (int d = 0; d < g.ndim; ++d) identity = identity && g.perm[d] == d;
    if (identity) return hipMemcpyAsync(coef, xvec, sizeof(double) * (size_t)g.ncol, hipMemcpyDeviceToDevice, st);
    hipLaunchKernelGGL(to_reference_order_kernel, dim3((g.ncol + 255) / 256), dim3(256), 0, st, g, xvec, coef);
    return hipGetLastError();
}

hipError_t launch_keys(const Grid &g, long long m, const double *x, int ldx, const double *w,
                       const SortScratch &s, double *hist, double *scal, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(s.count, 0, sizeof(int) * (size_t)(g.ncell + 2), st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(s.cursor, 0, sizeof(int) * (size_t)(g.ncell + 1), st);
    if (e != hipSuccess) return e;
    if (m <= 0) return hipSuccess;
    dim3 gr(grid_for(m, 256)), bl(256);
    DISPATCH_D(g.ndim, hipLaunchKernelGGL(keys_kernel<D>, gr, bl, 0, st, g, m, x, ldx, w, s.key,
                                          s.count, hist, scal));
    return hipGetLastError();
}

hipError_t launch_scan_scatter(const Grid &g, long long m, const double *x, int ldx,
                               const double *y, const double *w, const SortScratch &s,
                               hipStream_t st)
{
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, s.count, s.offset, g.ncell + 1);
    if (m > 0) {
        dim3 gr(grid_for(m, 256)), bl(256);
        DISPATCH_D(g.ndim, hipLaunchKernelGGL(scatter_kernel<D>, gr, bl, 0, st, g, m, x, ldx, y, w,
                                              s.key, s.offset, s.cursor, s.xs, s.ys, s.ws, s.cap));
    }
    return hipGetLastError();
}

hipError_t launch_gram(const Grid &g, const SortScratch &s, double *nst, double *rhs, hipStream_t st)
{
    DISPATCH_D(g.ndim, {
        using C = GramCfg<D>;
        dim3 gr((unsigned)g.ncell, (unsigned)(C::NB / (C::NTX * C::TC)));
        hipLaunchKernelGGL(gram_kernel<D>, gr, dim3(C::NT), 0, st, g, s.offset, s.xs, s.ys, s.ws,
                           s.cap, nst, rhs);
    });
    return hipGetLastError();
}

hipError_t launch_residual(const Grid &g, const SortScratch &s, const double *xvec, double *rho,
                           double *ssq, hipStream_t st)
{
    DISPATCH_D(g.ndim, {
        using C = ResCfg<D>;
        hipLaunchKernelGGL(residual_kernel<D>, dim3((unsigned)g.ncell), dim3(C::NT), 0, st, g,
                           s.offset, s.xs, s.ys, s.ws, s.cap, xvec, rho, ssq);
    });
    return hipGetLastError();
}

hipError_t launch_constraints(const Grid &g, const double *hist, const double *scal, double xtrap,
                              double *nst, const double *xvec, double *rho, double *scal_out,
                              double *ssq, hipStream_t st)
{
    dim3 gr((unsigned)((g.ncol + 3) / 4)), bl(256);
    DISPATCH_D(g.ndim, hipLaunchKernelGGL(constraint_kernel<D>, gr, bl, 0, st, g, hist, scal, xtrap,
                                          nst, xvec, rho, scal_out, ssq));
    return hipGetLastError();
}

hipError_t launch_expand(const Grid &g, const double *nst, const Band &b, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(b.ab, 0, b.bytes, st);
    if (e != hipSuccess) return e;
    const long long total = (long long)g.ncol * g.hstencil;
    dim3 gr(grid_for(total, 256, 256LL * 64)), bl(256);
    DISPATCH_D(g.ndim, hipLaunchKernelGGL(expand_kernel<D>, gr, bl, 0, st, g, nst, b.ab, b.lda));
    if (b.npad > b.n)
        hipLaunchKernelGGL(pad_diag_kernel, dim3((b.npad - b.n + 255) / 256), dim3(256), 0, st, b.ab,
                           b.lda, b.n, b.npad);
    return hipGetLastError();
}

}  // namespace splpak
