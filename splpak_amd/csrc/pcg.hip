// Iterative solve of the fit's least-squares problem: preconditioned conjugate gradients on the normal equations, with the
// operator applied MATRIX-FREE FROM THE ROWS and a separable ("fast diagonalisation") preconditioner (round 6).
//
// Why: the direct factorisations need O(n^1.5) .. O(n^2) reals -- BASELINE config 5's grid (4-D, 32^4 = 1 048 576 columns) takes
// 476 GB of nested-dissection panels (band: 851 GB) against 309 GB of HBM, while its rows are 0.4 GB of points.  The reference
// accepts any grid (src/splpak.F90:512-534); SURVEY section 7.2-H3 names PCG beside the two factorisations.
//
//   operator      q = N p = A^T W^2 A p + C^T C p from the rows themselves: the refinement's residual pass (assemble.hip:
//                 launch_residual) run with y = 0.  At 32^4 / 1e7 points the rows are 0.4 GB of sorted points and the per-cell
//                 shares 1.45 GB, the assembled half stencil would be 10 GB per product.  The residual the iteration works on is
//                 therefore the one the fit is judged by (rows, not the rounded N).
//   preconditioner  E[N] for uniformly scattered points and uniformly scattered data-sparse nodes is separable:
//                     E[N] = rho (x)_k M_k  +  lambda [ sum_i K2_i (x)_{k != i} K0_k  +  4 sum_{i<j} K1_i K1_j (x)_{k != i,j} K0_k ],
//                 M_k the 1-D mass matrix of the basis, K0/K1/K2 the Gram matrices of its values / first / second derivatives at
//                 the nodes (the constraint rows' factors, :921-1046; boundary nodes use the first derivative, :998).  Per
//                 dimension the generalised eigenvectors V_k of (K2_k, K0_k) diagonalise two of the four matrices exactly and the
//                 other two nearly (all four are Toeplitz away from the ends); M^-1 = V diag^-1 V^T with V = (x)_k V_k costs
//                 2 d mode products of nodes_k x nodes_k matrices (0.5 GFLOP at 32^4) -- nothing beside one pass over the rows.
//   outer loop    the fit's own refinement against the rows (plan.hip) with this solve in place of the triangular solves.
//
// Where it works and where it does not (measured): the preconditioner knows the DENSITY of data-sparse nodes,
// not where they are.  4-D at config 5's density (26 % of the nodes data sparse, 10 rows each: the constraint rows alone have
// full column rank) converges in a few hundred iterations, slowly growing with the grid.  3-D grids (6 rows per sparse node) at
// 8 .. 25 % leave half of the spectrum to the data rows, 10^8 below the constraint rows: > 3 000 iterations at 24^3, the direct
// factorisation wins by far.  Hence: automatic only when no factorisation fits the device; a stagnation rule hands over to the
// factorisation when the plan has one, and reports 107 when it has not.
#include "plan.hpp"
#include "basis.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace splpak {

namespace {

constexpr int DOT_BLOCKS = 512;

// ---- host: small dense symmetric eigenproblems (nodes_k x nodes_k, once per plan) ------------------------------------------
// cyclic Jacobi: A (n x n, symmetric, row-major) -> eigenvalues in w, eigenvectors in the COLUMNS of U
void jacobi_eig(int n, std::vector<double> &A, std::vector<double> &w, std::vector<double> &U)
{
    U.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) U[(size_t)i * n + i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, dg = 0.0;
        for (int i = 0; i < n; ++i) {
            dg += A[(size_t)i * n + i] * A[(size_t)i * n + i];
            for (int j = i + 1; j < n; ++j) off += A[(size_t)i * n + j] * A[(size_t)i * n + j];
        }
        if (off <= 1e-30 * dg || off == 0.0) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[(size_t)p * n + q];
                if (apq == 0.0) continue;
                const double app = A[(size_t)p * n + p], aqq = A[(size_t)q * n + q];
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {           // columns p, q
                    const double akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
                    A[(size_t)k * n + p] = c * akp - s * akq;
                    A[(size_t)k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {           // rows p, q
                    const double apk = A[(size_t)p * n + k], aqk = A[(size_t)q * n + k];
                    A[(size_t)p * n + k] = c * apk - s * aqk;
                    A[(size_t)q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double ukp = U[(size_t)k * n + p], ukq = U[(size_t)k * n + q];
                    U[(size_t)k * n + p] = c * ukp - s * ukq;
                    U[(size_t)k * n + q] = s * ukp + c * ukq;
                }
            }
    }
    w.resize((size_t)n);
    for (int i = 0; i < n; ++i) w[(size_t)i] = A[(size_t)i * n + i];
}

// generalised problem K v = w B v, B positive definite: V^T B V = I, V^T K V = diag(w); V row-major, eigenvectors in columns
bool gen_eig(int n, const std::vector<double> &K, const std::vector<double> &B, std::vector<double> &w, std::vector<double> &V)
{
    std::vector<double> L(B);                           // B = L L^T
    for (int j = 0; j < n; ++j) {
        double d = L[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
        if (!(d > 0.0)) return false;
        d = std::sqrt(d);
        L[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = L[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) s -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
            L[(size_t)i * n + j] = s / d;
        }
        for (int i = 0; i < j; ++i) L[(size_t)i * n + j] = 0.0;
    }
    // C = L^-1 K L^-T
    std::vector<double> C(K);
    for (int c = 0; c < n; ++c)                          // L^-1 K (column by column: forward substitution)
        for (int i = 0; i < n; ++i) {
            double s = C[(size_t)i * n + c];
            for (int k = 0; k < i; ++k) s -= L[(size_t)i * n + k] * C[(size_t)k * n + c];
            C[(size_t)i * n + c] = s / L[(size_t)i * n + i];
        }
    for (int r = 0; r < n; ++r)                          // (.) L^-T (row by row)
        for (int i = 0; i < n; ++i) {
            double s = C[(size_t)r * n + i];
            for (int k = 0; k < i; ++k) s -= L[(size_t)i * n + k] * C[(size_t)r * n + k];
            C[(size_t)r * n + i] = s / L[(size_t)i * n + i];
        }
    for (int i = 0; i < n; ++i)                          // symmetrise the rounding
        for (int j = i + 1; j < n; ++j) C[(size_t)i * n + j] = C[(size_t)j * n + i] = 0.5 * (C[(size_t)i * n + j] + C[(size_t)j * n + i]);
    std::vector<double> U;
    jacobi_eig(n, C, w, U);
    // V = L^-T U (back substitution per column)
    V.assign((size_t)n * n, 0.0);
    for (int c = 0; c < n; ++c)
        for (int i = n - 1; i >= 0; --i) {
            double s = U[(size_t)i * n + c];
            for (int k = i + 1; k < n; ++k) s -= L[(size_t)k * n + i] * V[(size_t)k * n + c];
            V[(size_t)i * n + c] = s / L[(size_t)i * n + i];
        }
    return true;
}

// diag(V^T X V)
void congruence_diag(int n, const std::vector<double> &V, const std::vector<double> &X, std::vector<double> &out)
{
    out.assign((size_t)n, 0.0);
    std::vector<double> t((size_t)n);
    for (int j = 0; j < n; ++j) {
        for (int i = 0; i < n; ++i) {
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += X[(size_t)i * n + k] * V[(size_t)k * n + j];
            t[(size_t)i] = s;
        }
        double d = 0.0;
        for (int i = 0; i < n; ++i) d += V[(size_t)i * n + j] * t[(size_t)i];
        out[(size_t)j] = d;
    }
}

// ---- kernels ----------------------------------------------------------------------------------------------------------------
// Y[o][j][in] = sum_i MT[i * nk + j] X[o][i][in] (mode product along one dimension), optionally scaled elementwise
__global__ void __launch_bounds__(256)
mode_product_kernel(int nk, long long inner, long long total, const double *__restrict__ MT, const double *__restrict__ X,
                    double *__restrict__ Y, const double *__restrict__ scale)
{
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const long long in = e % inner, t = e / inner;
    const int j = (int)(t % nk);
    const long long o = t / nk;
    const double *__restrict__ xb = X + (o * nk) * inner + in;
    double acc = 0.0;
    for (int i = 0; i < nk; ++i) acc = fma(MT[(long long)i * nk + j], xb[(long long)i * inner], acc);
    if (scale) acc *= scale[e];
    Y[e] = acc;
}

// Two adjacent modes k, k + 1 in ONE pass: the workgroup's tile [n_{k+1}][n_k][ci] (ci consecutive entries of the dimensions below k)
// goes through LDS, mode k into a second image, mode k + 1 out of it to global memory (optionally scaled).  Halves the launches and
// the passes over the vector of the separable preconditioner (8 launches of 34 us at 32^4 -> 4 of ~20).
__global__ void __launch_bounds__(256)
mode_pair_kernel(int na, int nb, long long inner, int ci, const double *__restrict__ MTa, const double *__restrict__ MTb,
                 const double *__restrict__ X, double *__restrict__ Y, const double *__restrict__ scale)
{
    extern __shared__ double smem[];
    const int tile = na * nb * ci;
    double *__restrict__ T0 = smem, *__restrict__ T1 = smem + tile;
    const long long nchunk = inner / ci;
    const long long o = blockIdx.x / nchunk, ch = blockIdx.x % nchunk;
    const long long base = o * ((long long)na * nb * inner) + ch * ci;
    for (int e = threadIdx.x; e < tile; e += 256) {
        const int c = e % ci, r = e / ci;            // r = ja + na * jb
        T0[e] = X[base + c + inner * r];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < tile; e += 256) {
        const int c = e % ci, r = e / ci, ja = r % na, jb = r / na;
        const double *__restrict__ t = T0 + c + (long long)ci * na * jb;
        double acc = 0.0;
        for (int ia = 0; ia < na; ++ia) acc = fma(MTa[ia * na + ja], t[ci * ia], acc);
        T1[e] = acc;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < tile; e += 256) {
        const int c = e % ci, r = e / ci, ja = r % na, jb = r / na;
        const double *__restrict__ t = T1 + c + ci * ja;
        double acc = 0.0;
        for (int ib = 0; ib < nb; ++ib) acc = fma(MTb[ib * nb + jb], t[(long long)ci * na * ib], acc);
        const long long gi = base + c + inner * r;
        if (scale) acc *= scale[gi];
        Y[gi] = acc;
    }
}

// The same pass on the matrix pipe.  Each mode product of a tile is a small matrix product -- mode a: out[ja][(c, jb)] = sum_ia
// MTa[ia][ja] T0[ia][(c, jb)], mode b: out[jb][(c, ja)] = sum_ib MTb[ib][jb] T1[ib][(c, ja)] -- in 16 x 16 output tiles dealt to the
// four waves, K in steps of four (v_mfma_f64_16x16x4_f64; operand and result lane maps as in rowsop.hip).  The vector form above
// issues one LDS read and one cached global read per multiply-add (43 us per launch at 32^4, 3 TFLOP/s); this one reads two LDS
// operands per 1 024 of them.  LDS: T0 [nb][kpa][ci] and T1 [kpb][na][ci] with zero rows up to the next multiple of four (no
// guards in the K loop), the two matrices zero-padded to [kp][mp].
typedef double pcg_d4_t __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256)
mode_pair_mfma_kernel(int na, int nb, long long inner, int ci, const double *__restrict__ MTa, const double *__restrict__ MTb,
                      const double *__restrict__ X, double *__restrict__ Y, const double *__restrict__ scale)
{
    extern __shared__ double smem[];
    const int kpa = (na + 3) & ~3, mpa = (na + 15) & ~15, kpb = (nb + 3) & ~3, mpb = (nb + 15) & ~15;
    double *__restrict__ T0 = smem;
    double *__restrict__ T1 = T0 + nb * kpa * ci;
    double *__restrict__ SA = T1 + kpb * na * ci;
    double *__restrict__ SB = SA + kpa * mpa;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
    const long long nchunk = inner / ci;
    const long long o = blockIdx.x / nchunk, ch = blockIdx.x % nchunk;
    const long long base = o * ((long long)na * nb * inner) + ch * ci;
    for (int e = tid; e < nb * kpa * ci; e += 256) {
        const int c = e % ci, r = e / ci, ia = r % kpa, jb = r / kpa;
        T0[e] = ia < na ? X[base + c + inner * (ia + (long long)na * jb)] : 0.0;
    }
    for (int e = tid; e < kpa * mpa; e += 256) {
        const int ja = e % mpa, ia = e / mpa;
        SA[e] = (ia < na && ja < na) ? MTa[ia * na + ja] : 0.0;
    }
    for (int e = tid; e < kpb * mpb; e += 256) {
        const int jb = e % mpb, ib = e / mpb;
        SB[e] = (ib < nb && jb < nb) ? MTb[ib * nb + jb] : 0.0;
    }
    for (int e = tid; e < (kpb - nb) * na * ci; e += 256) T1[nb * na * ci + e] = 0.0;
    __syncthreads();
    {   // mode a
        const int ncol = ci * nb, nct = (ncol + 15) >> 4, nmt = mpa >> 4;
        for (int t = wave; t < nmt * nct; t += 4) {
            const int mt = t % nmt, ct = t / nmt;
            const int col = 16 * ct + l15;
            const bool okc = col < ncol;
            const int cc = okc ? col : 0, c = cc % ci, jb = cc / ci;
            const double *__restrict__ bp = T0 + c + ci * (kpa * jb + g4);
            const double *__restrict__ ap = SA + g4 * mpa + 16 * mt + l15;
            pcg_d4_t acc = {0.0, 0.0, 0.0, 0.0};
            for (int s4 = 0; s4 < kpa; s4 += 4) {
                const double bv = bp[ci * s4];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[mpa * s4], okc ? bv : 0.0, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ja = 16 * mt + g4 + 4 * r;
                if (okc && ja < na) T1[c + ci * (ja + na * jb)] = acc[r];
            }
        }
    }
    __syncthreads();
    {   // mode b
        const int ncol = ci * na, nct = (ncol + 15) >> 4, nmt = mpb >> 4;
        for (int t = wave; t < nmt * nct; t += 4) {
            const int mt = t % nmt, ct = t / nmt;
            const int col = 16 * ct + l15;
            const bool okc = col < ncol;
            const int cc = okc ? col : 0, c = cc % ci, ja = cc / ci;
            const double *__restrict__ bp = T1 + cc + ci * na * g4;
            const double *__restrict__ ap = SB + g4 * mpb + 16 * mt + l15;
            pcg_d4_t acc = {0.0, 0.0, 0.0, 0.0};
            for (int s4 = 0; s4 < kpb; s4 += 4) {
                const double bv = bp[ci * na * s4];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[mpb * s4], okc ? bv : 0.0, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int jb = 16 * mt + g4 + 4 * r;
                if (okc && jb < nb) {
                    const long long gi = base + c + inner * (ja + (long long)na * jb);
                    double v = acc[r];
                    if (scale) v *= scale[gi];
                    Y[gi] = v;
                }
            }
        }
    }
}

struct EvTabs {             // per-dimension diagonals in the eigenbasis (device arrays of nodes_k doubles each)
    const double *mu[MAXD], *k0[MAXD], *d1[MAXD], *l2[MAXD];
};

// 1 / (rho prod mu + lambda (sum_i l2_i prod_{k != i} k0_k + 4 sum_{i<j} d1_i d1_j prod_{k != i,j} k0_k))
__global__ void __launch_bounds__(256)
fd_diag_kernel(Grid g, EvTabs tb, double rho, double lambda, double *__restrict__ dinv)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= g.ncol) return;
    double mu[MAXD], k0[MAXD], d1[MAXD], l2[MAXD];
    for (int k = 0; k < g.ndim; ++k) {
        const int j = (e / g.colstride[k]) % g.nodes[k];
        mu[k] = tb.mu[k][j]; k0[k] = tb.k0[k][j]; d1[k] = tb.d1[k][j]; l2[k] = tb.l2[k][j];
    }
    double dm = rho, pen = 0.0;
    for (int k = 0; k < g.ndim; ++k) dm *= mu[k];
    for (int i = 0; i < g.ndim; ++i) {
        double t = l2[i];
        for (int k = 0; k < g.ndim; ++k) if (k != i) t *= k0[k];
        pen += t;
        for (int j = i + 1; j < g.ndim; ++j) {
            double u = 4.0 * d1[i] * d1[j];
            for (int k = 0; k < g.ndim; ++k) if (k != i && k != j) u *= k0[k];
            pen += u;
        }
    }
    dinv[e] = 1.0 / (dm + lambda * pen);
}

// partial[b] = sum over block b's contiguous chunk of a[i] * b[i] (fixed order: reproducible)
__global__ void __launch_bounds__(256)
dot_partial_kernel(long long n, const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ partial)
{
    __shared__ double red[256];
    const long long chunk = (n + gridDim.x - 1) / gridDim.x;
    const long long beg = (long long)blockIdx.x * chunk, end = beg + chunk < n ? beg + chunk : n;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    long long i = beg + threadIdx.x;
    for (; i + 768 < end; i += 1024) {
        s0 = fma(a[i], b[i], s0);
        s1 = fma(a[i + 256], b[i + 256], s1);
        s2 = fma(a[i + 512], b[i + 512], s2);
        s3 = fma(a[i + 768], b[i + 768], s3);
    }
    for (; i < end; i += 256) s0 = fma(a[i], b[i], s0);
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// sc[slot] = sign * sum(partial); hist != NULL: also hist[0] = that value
__global__ void __launch_bounds__(DOT_BLOCKS)
dot_finish_kernel(const double *__restrict__ partial, double sign, double *__restrict__ sc, int slot, double *__restrict__ hist)
{
    __shared__ double red[DOT_BLOCKS];
    red[threadIdx.x] = partial[threadIdx.x];
    __syncthreads();
    for (int o = DOT_BLOCKS / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double v = sign * red[0];
        sc[slot] = v;
        if (hist) hist[0] = v;
    }
}

enum { S_RZ = 0, S_PQ = 1, S_RZNEW = 2, S_COUNT = 8 };

// alpha = rz / pq;  x += alpha p;  r += alpha nq   (nq = -N p, the residual pass's sign)
__global__ void __launch_bounds__(256)
update_xr_kernel(long long n, const double *__restrict__ sc, double *__restrict__ x, double *__restrict__ r,
                 const double *__restrict__ pv, const double *__restrict__ nq)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double pq = sc[S_PQ];
    const double alpha = pq > 0.0 ? sc[S_RZ] / pq : 0.0;
    x[i] = fma(alpha, pv[i], x[i]);
    r[i] = fma(alpha, nq[i], r[i]);
}

// beta = rz_new / rz;  p = z + beta p
__global__ void __launch_bounds__(256)
update_p_kernel(long long n, const double *__restrict__ sc, double *__restrict__ pv, const double *__restrict__ z)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double rz = sc[S_RZ];
    const double beta = rz > 0.0 ? sc[S_RZNEW] / rz : 0.0;
    pv[i] = fma(beta, pv[i], z[i]);
}

__global__ void advance_kernel(double *__restrict__ sc) { sc[S_RZ] = sc[S_RZNEW]; }

// out[0] = sum of w^2 over the sorted points [0, offset[ncell]) -- the zero-weight points sit behind them (fixed order)
__global__ void __launch_bounds__(1024)
fd_sumw2_kernel(const double *__restrict__ ws, const int *__restrict__ offset, int ncell, double *__restrict__ out)
{
    __shared__ double red[1024];
    const int t = threadIdx.x;
    const long long m = offset[ncell];
    double s0 = 0.0, s1 = 0.0;
    long long i = t;
    for (; i + 1024 < m; i += 2048) { s0 = fma(ws[i], ws[i], s0); s1 = fma(ws[i + 1024], ws[i + 1024], s1); }
    if (i < m) s0 = fma(ws[i], ws[i], s0);
    red[t] = s0 + s1;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (t < o) red[t] += red[t + o];
        __syncthreads();
    }
    if (t == 0) out[0] = red[0];
}

// out[0] = sum over the nodes of (data sparse ? dcw^2 : 0)  (fixed order)
__global__ void __launch_bounds__(1024)
fd_lambda_kernel(const double *__restrict__ dcw, const unsigned char *__restrict__ spf, int ncol, double *__restrict__ out)
{
    __shared__ double red[1024];
    const int t = threadIdx.x;
    double s = 0.0;
    for (int i = t; i < ncol; i += 1024) if (spf[i]) s = fma(dcw[i], dcw[i], s);
    red[t] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (t < o) red[t] += red[t + o];
        __syncthreads();
    }
    if (t == 0) out[0] = red[0];
}

// ---- block-Jacobi component: the diagonal blocks of N over aligned boxes of nodes -------------------------------------------
struct BjGeom {
    int ndim, edge[MAXD], nbd[MAXD], nb;        // box edge and boxes per dimension, boxes in all
};

// node of local index l of box b (dimension 0 fastest in both), or -1 beyond the grid / the box
__device__ inline int bj_node(const Grid &g, const BjGeom &bg, int b, int l, int *coord)
{
    int node = 0;
    for (int d = 0; d < bg.ndim; ++d) {
        const int bd = b % bg.nbd[d], ld = l % bg.edge[d];
        b /= bg.nbd[d];
        l /= bg.edge[d];
        const int c = bd * bg.edge[d] + ld;
        if (c >= g.nodes[d]) return -1;
        if (coord) coord[d] = c;
        node += c * g.colstride[d];
    }
    return l == 0 ? node : -1;
}

// blocks[b] (column-major 256 x 256, lower triangle) = N restricted to the box's nodes, from the half stencil; identity on the padding
__global__ void __launch_bounds__(256)
bj_extract_kernel(Grid g, BjGeom bg, const double *__restrict__ nst, double *__restrict__ blocks)
{
    __shared__ int snode[256];
    __shared__ short sloc[256][MAXD];              // coordinates inside the box
    const int b = blockIdx.x, r = threadIdx.x;
    {
        snode[r] = bj_node(g, bg, b, r, nullptr);
        int l = r;
        for (int d = 0; d < MAXD; ++d) {
            sloc[r][d] = (short)(d < bg.ndim ? l % bg.edge[d] : 0);
            if (d < bg.ndim) l /= bg.edge[d];
        }
    }
    __syncthreads();
    double *__restrict__ A = blocks + (size_t)b * 65536;
    const int nr = snode[r];
    for (int c = 0; c < 256; ++c) {
        double v = 0.0;
        if (c <= r) {
            const int nc = snode[c];
            if (nr < 0 || nc < 0) v = (c == r) ? 1.0 : 0.0;
            else {
                int code = 0, m7 = 1;
                bool in = true;
                for (int d = 0; d < g.ndim; ++d) {
                    const int o = (int)sloc[c][d] - (int)sloc[r][d];
                    in = in && o >= -3 && o <= 3;
                    code += (o + 3) * m7;
                    m7 *= 7;
                }
                if (in) v = nst[(long long)nr * g.hstencil + code];       // row = the later node (local order = global order inside a box)
            }
        }
        A[r + (size_t)c * 256] = v;
    }
}

// The boxes WITHOUT assembled normal equations (iteration-only 4-D plans): box = s M_box + P_box, with P_box the EXACT constraint
// part (sum over the data-sparse nodes within one node of both entries, their rows in the reference's order, :921-1046) and the
// data part replaced by the mass matrix's sub-block scaled to the box's own trace of A^T W^2 A (ddiag, from the rows).  Measured
// (tools/pcg/exp8.py, 12^4): with constraint rows present the data part of a box does not matter -- exact, rho M, trace-scaled M
// and even diag(D) give the same 97 / 246 iterations at 2.8 / 2.0 rows per column; without any (xtrap = 0) 184 against 97 exact.
struct MassBands { const double *m[MAXD]; };      // per dimension [nodes][7]: M(i, i + o), o = -3 .. 3
__global__ void __launch_bounds__(256)
bj_build_kernel(Grid g, BjGeom bg, MassBands mb, const double *__restrict__ ctab, const double *__restrict__ dcw,      // (4-D grids only)
                const unsigned char *__restrict__ spf, const double *__restrict__ ddiag, double *__restrict__ blocks)
{
    constexpr int HMAX = 1296;                     // nodes of a box and the nodes within one of it: (4 + 2)^4 (3-D: 8^3, 2-D: 18^2, 1-D: 258)
    __shared__ int snode[256];
    __shared__ short scrd[256][MAXD];              // grid coordinates of the box's nodes
    __shared__ double red[2][256];
    __shared__ double sw2[HMAX];                   // squared constraint weight of the halo's nodes (0: not data sparse / outside the grid)
    __shared__ double sf[MAXD][18][3][3];          // sf[d][halo node][offset + 1][derivative]: the rows' factors (ctab) of the halo's nodes
    const int b = blockIdx.x, r = threadIdx.x;
    int h0[MAXD] = {0, 0, 0, 0}, hext[MAXD] = {1, 1, 1, 1};       // first halo coordinate and extent per dimension
    {
        int bb = b;
        for (int d = 0; d < g.ndim; ++d) {
            const int bd = bb % bg.nbd[d];
            bb /= bg.nbd[d];
            const int lo = bd * bg.edge[d] - 1, hi = bd * bg.edge[d] + bg.edge[d];
            h0[d] = lo < 0 ? 0 : lo;
            hext[d] = (hi > g.nodes[d] - 1 ? g.nodes[d] - 1 : hi) - h0[d] + 1;
        }
    }
    int hs[MAXD] = {1, hext[0], hext[0] * hext[1], hext[0] * hext[1] * hext[2]};
    const int htot = hs[3] * hext[3];
    int cbase[MAXD] = {0, 0, 0, 0};
    for (int d = 1; d < g.ndim; ++d) cbase[d] = cbase[d - 1] + 9 * g.nodes[d - 1];
    {
        int c[MAXD] = {0, 0, 0, 0};
        snode[r] = bj_node(g, bg, b, r, c);
        for (int d = 0; d < MAXD; ++d) scrd[r][d] = (short)c[d];
    }
    for (int e = r; e < htot; e += 256) {
        int t = e, nn = 0;
        for (int d = 0; d < g.ndim; ++d) { nn += (h0[d] + t % hext[d]) * g.colstride[d]; t /= hext[d]; }
        double w2 = 0.0;
        if (spf && spf[nn]) w2 = dcw[nn] * dcw[nn];
        sw2[e] = w2;
    }
    for (int e = r; e < g.ndim * 18 * 9; e += 256) {
        const int d = e / (18 * 9), hl = (e / 9) % 18, k = e % 9;
        double v = 0.0;
        if (hl < hext[d]) v = ctab[cbase[d] + (h0[d] + hl) * 9 + k];
        sf[d][hl][k / 3][k % 3] = v;
    }
    __syncthreads();
    const int nr = snode[r];
    {   // s = (trace of the data rows' Gram matrix over the box) / (trace of the mass sub-block)
        double td = 0.0, tm = 0.0;
        if (nr >= 0) {
            td = ddiag[nr];
            tm = 1.0;
            for (int d = 0; d < g.ndim; ++d) tm *= mb.m[d][scrd[r][d] * 7 + 3];
        }
        red[0][r] = td;
        red[1][r] = tm;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (r < o) { red[0][r] += red[0][r + o]; red[1][r] += red[1][r + o]; }
            __syncthreads();
        }
    }
    const double sbox = red[1][0] > 0.0 ? red[0][0] / red[1][0] : 0.0;
    double *__restrict__ A = blocks + (size_t)b * 65536;
    // the lower triangle, balanced: a thread takes the rows q and 255 - q (257 entries together) and of those the columns of its parity
    const int tq = threadIdx.x & 127, th = threadIdx.x >> 7;
    for (int pass = 0; pass < 2; ++pass)
    for (int c = th; c < 256; c += 2) {
        const int r = pass == 0 ? tq : 255 - tq;
        const int nr = snode[r];
        double v = 0.0;
        if (c > r) { A[r + (size_t)c * 256] = 0.0; continue; }
        {
            const int nc = snode[c];
            if (nr < 0 || nc < 0) v = (c == r) ? 1.0 : 0.0;
            else {
                int lo[MAXD] = {0, 0, 0, 0}, hi[MAXD] = {0, 0, 0, 0};
                bool near = true;
                double mass = sbox;
                for (int d = 0; d < g.ndim; ++d) {
                    const int i = scrd[r][d], j = scrd[c][d];
                    const int o = j - i;
                    if (o < -3 || o > 3) { mass = 0.0; near = false; break; }
                    mass *= mb.m[d][i * 7 + o + 3];
                    lo[d] = (i > j ? i : j) - 1;
                    hi[d] = (i < j ? i : j) + 1;
                    if (lo[d] < 0) lo[d] = 0;
                    if (hi[d] > g.nodes[d] - 1) hi[d] = g.nodes[d] - 1;
                    near = near && lo[d] <= hi[d];
                }
                v = mass;
                if (near && spf) {
                    // the data-sparse nodes n within one node of both entries: lo <= n <= hi per dimension, dimension 0 fastest (fixed order).
                    // Per node the ten rows' products in closed form: with G_d^a = f_d^a(i) f_d^a(j) the factor of dimension d at
                    // derivative order a (order 2 -> 1 at a boundary node, :998),
                    //   sum_rows = sum_i G_i^2 prod_{k != i} G_k^0 + 4 sum_{i<j} G_i^1 G_j^1 prod_{k != i,j} G_k^0     (weights dcw, 2 dcw: :983)
                    double acc = 0.0;
                    const int oi0 = scrd[r][0], oi1 = scrd[r][1], oi2 = scrd[r][2], oi3 = scrd[r][3];
                    const int oj0 = scrd[c][0], oj1 = scrd[c][1], oj2 = scrd[c][2], oj3 = scrd[c][3];
                    for (int n3 = lo[3]; n3 <= hi[3]; ++n3)
                        for (int n2 = lo[2]; n2 <= hi[2]; ++n2)
                            for (int n1 = lo[1]; n1 <= hi[1]; ++n1)
                                for (int n0 = lo[0]; n0 <= hi[0]; ++n0) {
                                    const int l0 = n0 - h0[0], l1 = n1 - h0[1], l2 = n2 - h0[2], l3 = n3 - h0[3];
                                    const double w2 = sw2[l0 + hs[1] * l1 + hs[2] * l2 + hs[3] * l3];
                                    if (w2 == 0.0) continue;
                                    double G0[4], G1[4], G2[4];
#define SPLPAK_G(d, nl, nn, oi, oj)                                                              \
                                    {                                                            \
                                        const double(*f)[3] = sf[d][nl];                         \
                                        const int a = (oi) - (nn) + 1, bq = (oj) - (nn) + 1;     \
                                        G0[d] = f[a][0] * f[bq][0];                              \
                                        G1[d] = f[a][1] * f[bq][1];                              \
                                        G2[d] = ((nn) == 0 || (nn) == g.nodes[d] - 1) ? G1[d] : f[a][2] * f[bq][2]; \
                                    }
                                    SPLPAK_G(0, l0, n0, oi0, oj0)
                                    SPLPAK_G(1, l1, n1, oi1, oj1)
                                    SPLPAK_G(2, l2, n2, oi2, oj2)
                                    SPLPAK_G(3, l3, n3, oi3, oj3)
#undef SPLPAK_G
                                    const double p01 = G0[0] * G0[1], p23 = G0[2] * G0[3];
                                    double sum = (G2[0] * G0[1] + G0[0] * G2[1]) * p23 + p01 * (G2[2] * G0[3] + G0[2] * G2[3]);
                                    sum += 4.0 * (G1[0] * G1[1] * p23 + p01 * G1[2] * G1[3] +
                                                  (G1[0] * G0[1] + G0[0] * G1[1]) * 0.0);
                                    // mixed pairs across the two halves: (0,2), (0,3), (1,2), (1,3)
                                    sum += 4.0 * ((G1[0] * G0[1]) * (G1[2] * G0[3] + G0[2] * G1[3]) + (G0[0] * G1[1]) * (G1[2] * G0[3] + G0[2] * G1[3]));
                                    acc = fma(w2, sum, acc);
                                }
                    v += acc;
                }
            }
        }
        A[r + (size_t)c * 256] = v;
    }
}

// z += (L L^T)^-1 v on every box: u = Linv v (dinvt[j][i] = Linv(i, j)), then Linv^T u (dinv[j][i] = Linv(j, i)); consecutive threads,
// consecutive words in both
template <typename T>
__global__ void __launch_bounds__(256)
bj_apply_kernel(Grid g, BjGeom bg, const T *__restrict__ dinv, const T *__restrict__ dinvt, const double *__restrict__ v,
                double *__restrict__ z)
{
    __shared__ double vb[256], ub[256];
    const int b = blockIdx.x, i = threadIdx.x, wave = i >> 6;
    const int node = bj_node(g, bg, b, i, nullptr);
    vb[i] = node >= 0 ? v[node] : 0.0;
    __syncthreads();
    const T *__restrict__ Mt = dinvt + (size_t)b * 65536, *__restrict__ M = dinv + (size_t)b * 65536;
    double a0 = 0.0, a1 = 0.0;
    const int jmax = 64 * wave + 63;
    for (int j = 0; j + 1 <= jmax; j += 2) {
        a0 = fma((double)Mt[(size_t)j * 256 + i], vb[j], a0);      // (entries above the diagonal of Linv are stored zeros)
        a1 = fma((double)Mt[(size_t)(j + 1) * 256 + i], vb[j + 1], a1);
    }
    ub[i] = a0 + a1;
    __syncthreads();
    a0 = a1 = 0.0;
    for (int j = 64 * wave; j + 1 < 256; j += 2) {
        a0 = fma((double)M[(size_t)j * 256 + i], ub[j], a0);
        a1 = fma((double)M[(size_t)(j + 1) * 256 + i], ub[j + 1], a1);
    }
    if (node >= 0) z[node] += a0 + a1;
}

// The apply from ONE single-precision copy of L^-1 (round 6, second form): the ten 64 x 64 blocks of its lower triangle, packed block
// row by block row.  Block row I (its I + 1 blocks) goes through LDS once and serves both products -- u_I = sum_J B_IJ v_J, complete
// after the row, then z_J += B_IJ^T u_I for the same blocks: 164 KB per box instead of the 328 KB the two triangular matrix-vector
// products read from the straight and the transposed copy (the pass is bound by these bytes).  Sums in a fixed order.
constexpr int BJ_LD = 65;          // floats per LDS row of a block (odd: the row-wise reads of the first product spread over the banks)
__global__ void __launch_bounds__(256)
bj_pack_kernel(long long nblk, const double *__restrict__ dinv, float *__restrict__ pk)
{   // pk[box][blk(I, J) = I (I + 1) / 2 + J][i][j] = Linv[64 I + i][64 J + j]   (dinv: row-major Linv)
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= nblk * 40960) return;
    const long long b = e / 40960;
    const int r = (int)(e % 40960), blk = r >> 12, i = (r >> 6) & 63, j = r & 63;
    const int I = blk < 1 ? 0 : (blk < 3 ? 1 : (blk < 6 ? 2 : 3)), J = blk - I * (I + 1) / 2;
    pk[e] = (float)dinv[b * 65536 + (long long)(64 * I + i) * 256 + 64 * J + j];
}

__global__ void __launch_bounds__(256)
bj_apply_packed_kernel(Grid g, BjGeom bg, const float *__restrict__ pk, const double *__restrict__ v, double *__restrict__ z)
{
    __shared__ float sblk[4 * 64 * BJ_LD];
    __shared__ double vb[256], ub[64], red[256];
    const int b = blockIdx.x, tid = threadIdx.x, l = tid & 63, part = tid >> 6;
    const int node = bj_node(g, bg, b, tid, nullptr);
    vb[tid] = node >= 0 ? v[node] : 0.0;
    const float *__restrict__ src = pk + (size_t)b * 40960;
    double zacc[4] = {0.0, 0.0, 0.0, 0.0};           // this thread's share (rows 16 part .. 16 part + 15 of every block) of z[64 J + l]
    for (int I = 0; I < 4; ++I) {
        const int nbk = I + 1;
        const float4 *__restrict__ s4 = reinterpret_cast<const float4 *>(src + (size_t)(I * (I + 1) / 2) * 4096);
        __syncthreads();                             // (the previous row's blocks are used up; first trip: vb is written)
        for (int e = tid; e < nbk * 1024; e += 256) {
            const float4 q = s4[e];
            const int J = e >> 10, i = (e >> 4) & 63, j = (e & 15) * 4;
            float *__restrict__ d = sblk + (J * 64 + i) * BJ_LD + j;
            d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w;
        }
        __syncthreads();
        {   // u_I[l]: thread (l, part) sums the columns 16 part .. 16 part + 15 of every block of the row
            double a = 0.0;
            for (int J = 0; J < nbk; ++J) {
                const float *__restrict__ row = sblk + (J * 64 + l) * BJ_LD + 16 * part;
                const double *__restrict__ vv = vb + 64 * J + 16 * part;
#pragma unroll
                for (int j = 0; j < 16; ++j) a = fma((double)row[j], vv[j], a);
            }
            red[tid] = a;
        }
        __syncthreads();
        if (tid < 64) ub[tid] = ((red[tid] + red[64 + tid]) + red[128 + tid]) + red[192 + tid];
        __syncthreads();
        for (int J = 0; J < nbk; ++J) {              // z_J[l] += sum_i B_IJ[i][l] u_I[i], i in this thread's sixteen rows
            const float *__restrict__ col = sblk + (J * 64 + 16 * part) * BJ_LD + l;
            double a = zacc[J];
#pragma unroll
            for (int i = 0; i < 16; ++i) a = fma((double)col[i * BJ_LD], ub[16 * part + i], a);
            zacc[J] = a;
        }
    }
    __syncthreads();
    double *__restrict__ zs = vb;                    // (vb is used up) z[64 J + l] = the four parts in order
    for (int J = 0; J < 4; ++J) {
        red[tid] = zacc[J];
        __syncthreads();
        if (tid < 64) zs[64 * J + tid] = ((red[tid] + red[64 + tid]) + red[128 + tid]) + red[192 + tid];
        __syncthreads();
    }
    if (node >= 0) z[node] += zs[tid];
}

// single-precision copies of the box inverses for the apply (a preconditioner: its rounding changes nothing the iteration converges
// to; half the bytes of a pass that is bound by reading them -- 0.48 -> 0.25 ms per iteration at 32^4)
__global__ void __launch_bounds__(256)
to_f32_kernel(long long n, const double *__restrict__ a, const double *__restrict__ b, float *__restrict__ fa, float *__restrict__ fb)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { fa[i] = (float)a[i]; fb[i] = (float)b[i]; }
}

// out[0] = sum of squares of v[0 .. n) with n read from the device (n = offset[ncell]): partial sums by block, then one block adds them
__global__ void __launch_bounds__(256)
sumsq_partial_kernel(const double *__restrict__ v, const int *__restrict__ nptr, double *__restrict__ partial)
{
    __shared__ double red[256];
    const long long n = nptr[0];
    const long long chunk = (n + gridDim.x - 1) / gridDim.x;
    const long long beg = (long long)blockIdx.x * chunk, end = beg + chunk < n ? beg + chunk : n;
    double s0 = 0.0;
    for (long long i = beg + threadIdx.x; i < end; i += 256) s0 = fma(v[i], v[i], s0);
    red[threadIdx.x] = s0;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

}  // namespace

struct PcgState {
    int device = 0;
    Grid g{};
    double *V[MAXD] = {nullptr, nullptr, nullptr, nullptr};      // V_k row-major (component i of eigenvector j at [i * n + j])
    double *VT[MAXD] = {nullptr, nullptr, nullptr, nullptr};     // its transpose
    double *tabs = nullptr;                                      // mu | k0 | d1 | l2 per dimension
    EvTabs ev{};
    double *dinv = nullptr;
    double *bvec = nullptr, *x = nullptr, *r = nullptr, *z = nullptr, *pv = nullptr, *t1 = nullptr, *t2 = nullptr;
    double *partial = nullptr, *sc = nullptr, *hist = nullptr, *mom = nullptr;
    // block-Jacobi component (NULL / bj_ready false: the separable preconditioner alone)
    BjGeom bg{};
    double *bj_blocks = nullptr, *bj_inv16 = nullptr, *bj_dinv = nullptr, *bj_dinvt = nullptr, *bj_scal = nullptr;
    float *bj_dinv32 = nullptr, *bj_dinvt32 = nullptr;     // single-precision copies for the apply (NULL: the f64 ones)
    float *bj_pk32 = nullptr;                              // the packed single-precision lower triangle of L^-1 (the default apply)
    void *bj_jobs = nullptr;
    int *bj_info = nullptr;
    bool bj_have = false, bj_ready = false;
    bool singular = false;                                 // this fit: a box of the assembled N failed the factorisations' pivot test
    double *mband = nullptr;                               // per dimension [nodes][7]: the 1-D mass matrix's bands (boxes without assembled N)
    MassBands mbands{};
    double *ddiag = nullptr;                               // diagonal of the data rows' Gram matrix (the same)
    bool no_pairs = false;                                 // A/B: the modes of the separable part one by one
    bool pairs_valu = false;                               // A/B: the pairs on the vector unit (round 6's first form)
    std::vector<double> hhist;
    int maxit = 4000;
    std::vector<void *> owned;
    size_t bytes = 0;
    // statistics of the last fit
    int total_iters = 0, solves = 0, last_iters = 0;
    double last_rel = 0.0, rho = 0.0, lambda = 0.0;
    bool failed = false;
};

static bool pcg_alloc(PcgState *s, double **q, size_t count)
{
    void *v = nullptr;
    if (count == 0) count = 1;
    if (hipMalloc(&v, count * sizeof(double)) != hipSuccess) {
        (void)hipGetLastError();
        set_error("pcg: device allocation failed");
        return false;
    }
    s->owned.push_back(v);
    s->bytes += count * sizeof(double);
    *q = static_cast<double *>(v);
    return true;
}

void pcg_destroy(PcgState *s)
{
    if (!s) return;
    for (void *q : s->owned) (void)hipFree(q);
    delete s;
}

size_t pcg_bytes(const PcgState *s) { return s ? s->bytes : 0; }

double *pcg_scratch(PcgState *s, int which) { return which == 0 ? s->t1 : s->t2; }
bool pcg_singular(const PcgState *s) { return s->singular; }
bool pcg_boxes_from_rows(const PcgState *s) { return s->bj_have && s->ddiag != nullptr; }

void pcg_stats(const PcgState *s, double *out6)
{
    if (!s || !out6) return;
    out6[0] = s->total_iters; out6[1] = s->solves; out6[2] = s->last_iters; out6[3] = s->last_rel; out6[4] = s->rho; out6[5] = s->lambda;
}

// Builds the separable preconditioner's per-dimension matrices for the plan's grid (internal dimension order).
// 0, or an SPLPAK_E_* code.
int pcg_attach(splpak_plan *p, PcgState **out)
{
    *out = nullptr;
    const Grid &g = p->g;
    for (int k = 0; k < g.ndim; ++k)
        if (g.nodes[k] > 512) { set_error("pcg: more than 512 nodes in one dimension (the per-dimension eigenproblems are dense)"); return SPLPAK_E_UNSUPPORTED; }
    PcgState *s = new PcgState();
    (void)hipGetDevice(&s->device);
    s->g = g;
    if (!p->band.ab && !p->fn_user) s->maxit = 12000;        // (nothing but the iteration: see the stagnation rule)
    if (const char *e = splpak::opt_get("SPLPAK_PCG_MAXIT")) s->maxit = std::max(1, atoi(e));
    s->no_pairs = splpak::opt_get("SPLPAK_PCG_NO_PAIRS") != nullptr;
    s->pairs_valu = splpak::opt_get("SPLPAK_PCG_PAIRS_VALU") != nullptr;
    bool ok = true;
    long long ntab = 0;
    for (int k = 0; k < g.ndim; ++k) ntab += 4LL * g.nodes[k];
    ok = ok && pcg_alloc(s, &s->tabs, (size_t)ntab);
    std::vector<double> htabs((size_t)ntab);
    long long toff = 0;
    long long mbtot = 0, mboff = 0, mboffs[MAXD] = {0, 0, 0, 0};
    for (int k = 0; k < g.ndim; ++k) mbtot += 7LL * g.nodes[k];
    std::vector<double> hmband((size_t)mbtot);
    const double qb = 0.5;          // boundary nodes: half the expected weight (:928) -> a quarter of dcw^2, about twice as often sparse
    for (int k = 0; k < g.ndim && ok; ++k) {
        const int n = g.nodes[k];
        const double dx = g.dx[k], s1 = g.dxin[k], x0 = g.xmin[k];
        std::vector<double> T0((size_t)n * n, 0.0), T1(T0), T2(T0), M(T0), K0(T0), K1(T0), K2(T0);
        for (int nd = 0; nd < n; ++nd)
            for (int ib = std::max(0, nd - 1); ib <= std::min(n - 1, nd + 1); ++ib) {
                const double xn = x0 + (double)nd * dx, xb = x0 + (double)ib * dx;
                const int kind = basis_kind(ib, n);
                T0[(size_t)nd * n + ib] = basis_1d(kind, 0, xn, xb, s1);
                T1[(size_t)nd * n + ib] = basis_1d(kind, 1, xn, xb, s1);
                T2[(size_t)nd * n + ib] = basis_1d(kind, (nd == 0 || nd == n - 1) ? 1 : 2, xn, xb, s1);      // :998
            }
        // mass matrix: 6-point Gauss-Legendre per interval (the integrand is a polynomial of degree 6)
        static const double gp[6] = {-0.932469514203152, -0.661209386466265, -0.238619186083197, 0.238619186083197, 0.661209386466265, 0.932469514203152};
        static const double gw[6] = {0.171324492379170, 0.360761573048139, 0.467913934572691, 0.467913934572691, 0.360761573048139, 0.171324492379170};
        for (int c = 0; c < n - 1; ++c)
            for (int q = 0; q < 6; ++q) {
                const double xq = x0 + ((double)c + 0.5 * (gp[q] + 1.0)) * dx, wq = 0.5 * gw[q] * dx;
                const int lo = std::max(0, c - 1), hi = std::min(n - 1, c + 2);
                double bv[4];
                for (int ib = lo; ib <= hi; ++ib) bv[ib - lo] = basis_1d(basis_kind(ib, n), 0, xq, x0 + (double)ib * dx, s1);
                for (int a = lo; a <= hi; ++a)
                    for (int b = lo; b <= hi; ++b) M[(size_t)a * n + b] += wq * bv[a - lo] * bv[b - lo];
            }
        auto gram = [&](const std::vector<double> &T, std::vector<double> &K) {
            for (int a = 0; a < n; ++a)
                for (int b = 0; b < n; ++b) {
                    double sum = 0.0;
                    for (int nd = std::max(0, std::max(a, b) - 1); nd <= std::min(n - 1, std::min(a, b) + 1); ++nd)
                        sum += ((nd == 0 || nd == n - 1) ? qb : 1.0) * T[(size_t)nd * n + a] * T[(size_t)nd * n + b];
                    K[(size_t)a * n + b] = sum;
                }
        };
        gram(T0, K0); gram(T1, K1); gram(T2, K2);
        for (int i = 0; i < n; ++i)
            for (int o = -3; o <= 3; ++o)
                hmband[(size_t)mboff + (size_t)i * 7 + (size_t)(o + 3)] = (i + o >= 0 && i + o < n) ? M[(size_t)i * n + (i + o)] : 0.0;
        mboffs[k] = mboff;
        mboff += 7LL * n;
        std::vector<double> w, V, mu, k0, d1, l2;
        if (!gen_eig(n, K2, K0, w, V)) { set_error("pcg: the nodal Gram matrix of a dimension is not positive definite"); ok = false; break; }
        congruence_diag(n, V, M, mu);
        congruence_diag(n, V, K0, k0);
        congruence_diag(n, V, K1, d1);
        congruence_diag(n, V, K2, l2);
        std::vector<double> VTh((size_t)n * n);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) VTh[(size_t)j * n + i] = V[(size_t)i * n + j];
        ok = ok && pcg_alloc(s, &s->V[k], (size_t)n * n) && pcg_alloc(s, &s->VT[k], (size_t)n * n);
        if (!ok) break;
        ok = ok && hip_ok(hipMemcpy(s->V[k], V.data(), sizeof(double) * (size_t)n * n, hipMemcpyHostToDevice), "pcg: upload")
                && hip_ok(hipMemcpy(s->VT[k], VTh.data(), sizeof(double) * (size_t)n * n, hipMemcpyHostToDevice), "pcg: upload");
        const std::vector<double> *arr[4] = {&mu, &k0, &d1, &l2};
        const double **dst[4] = {&s->ev.mu[k], &s->ev.k0[k], &s->ev.d1[k], &s->ev.l2[k]};
        for (int a = 0; a < 4; ++a) {
            std::memcpy(htabs.data() + toff, arr[a]->data(), sizeof(double) * (size_t)n);
            *dst[a] = s->tabs + toff;
            toff += n;
        }
    }
    for (int k = g.ndim; k < MAXD; ++k) { s->ev.mu[k] = s->ev.k0[k] = s->ev.d1[k] = s->ev.l2[k] = s->tabs; }
    ok = ok && hip_ok(hipMemcpy(s->tabs, htabs.data(), sizeof(double) * (size_t)ntab, hipMemcpyHostToDevice), "pcg: upload");
    const size_t n = (size_t)g.ncol;
    for (double **q : {&s->dinv, &s->bvec, &s->x, &s->r, &s->z, &s->pv, &s->t1, &s->t2}) ok = ok && pcg_alloc(s, q, n);
    ok = ok && pcg_alloc(s, &s->partial, DOT_BLOCKS) && pcg_alloc(s, &s->sc, S_COUNT) && pcg_alloc(s, &s->hist, (size_t)s->maxit + 8) &&
         pcg_alloc(s, &s->mom, 2);
    ok = ok && pcg_alloc(s, &s->mband, (size_t)mbtot) && hip_ok(hipMemcpy(s->mband, hmband.data(), sizeof(double) * (size_t)mbtot, hipMemcpyHostToDevice), "pcg: upload");
    for (int k = 0; k < MAXD; ++k) s->mbands.m[k] = s->mband + (k < g.ndim ? mboffs[k] : 0);
    // (also for the plans that have a factorisation behind the iteration: their fits assemble nothing until it is needed, plan.hip "lazy")
    if (ok && g.ndim == 4 && p->rowsop && p->ctab && (p->rows_only || p->solver_mode == 3)) {
        if (p->rows_only) ok = pcg_alloc(s, &s->ddiag, n);
        else if (!pcg_alloc(s, &s->ddiag, n)) { (void)hipGetLastError(); s->ddiag = nullptr; }
    }
    // Block-Jacobi component (round 6): the diagonal blocks of the assembled N over aligned boxes of nodes (4-D: 4^4, 3-D: 6^3,
    // 2-D: 16^2, 1-D: 256 nodes -- all at most 256, the block size of the diagonal-block kernels of the fronts), factored per fit
    // and ADDED to the separable preconditioner: M^-1 = V diag^-1 V^T + sum_boxes R^T (N_box)^-1 R.  The separable part knows the
    // global, smooth structure and the densities; the boxes know WHERE the data-sparse nodes are.  Measured at 12^4 (tools/pcg/exp7.py):
    // config 5's density 259 -> 97 iterations; 17 % data-sparse nodes (where the separable part alone stagnates) 505 iterations.
    // Needs the assembled normal equations (not the rows-only plans) and 3 x 0.5 MB per box.
    if (ok && !splpak::opt_get("SPLPAK_PCG_NO_BLOCKS")) {
        static const int edge_of[MAXD + 1] = {0, 256, 16, 6, 4};
        s->bg.ndim = g.ndim;
        s->bg.nb = 1;
        for (int d = 0; d < MAXD; ++d) {
            s->bg.edge[d] = d < g.ndim ? edge_of[g.ndim] : 1;
            s->bg.nbd[d] = d < g.ndim ? (g.nodes[d] + s->bg.edge[d] - 1) / s->bg.edge[d] : 1;
            s->bg.nb *= s->bg.nbd[d];
        }
        if (s->bg.nb <= 65535) {
            const size_t nb = (size_t)s->bg.nb;
            bool okb = pcg_alloc(s, &s->bj_blocks, nb * 65536) && pcg_alloc(s, &s->bj_dinv, nb * 65536) && pcg_alloc(s, &s->bj_dinvt, nb * 65536) &&
                       pcg_alloc(s, &s->bj_inv16, nb * 4096) && pcg_alloc(s, &s->bj_scal, 8);
            double *jb = nullptr, *ib = nullptr;
            okb = okb && pcg_alloc(s, &jb, (block_chol_job_bytes(s->bg.nb) + 7) / 8) && pcg_alloc(s, &ib, 2);
            if (okb) {
                s->bj_jobs = jb;
                s->bj_info = reinterpret_cast<int *>(ib);
                // real columns of every box (the rest of its 256 is identity padding)
                std::vector<int> ncols(nb);
                for (int b = 0; b < s->bg.nb; ++b) {
                    int bb = b, cnt = 1;
                    for (int d = 0; d < g.ndim; ++d) {
                        const int bd = bb % s->bg.nbd[d];
                        bb /= s->bg.nbd[d];
                        cnt *= std::min(s->bg.edge[d], g.nodes[d] - bd * s->bg.edge[d]);
                    }
                    (void)cnt;
                    ncols[(size_t)b] = 256;      // (padding is interleaved with the box's nodes unless the box is full: factor all 256)
                }
                okb = hip_ok(block_chol_prepare(s->bj_jobs, s->bg.nb, s->bj_blocks, s->bj_inv16, s->bj_dinv, s->bj_dinvt, ncols.data()), "pcg: block jobs");
            }
            if (okb && !splpak::opt_get("SPLPAK_PCG_BLOCKS_F64") && !splpak::opt_get("SPLPAK_PCG_BLOCKS_UNPACKED")) {
                double *f = nullptr;
                if (pcg_alloc(s, &f, nb * 20480)) s->bj_pk32 = reinterpret_cast<float *>(f);         // nb * 40 960 floats
                else (void)hipGetLastError();
            }
            if (okb && !s->bj_pk32 && !splpak::opt_get("SPLPAK_PCG_BLOCKS_F64")) {
                double *f = nullptr;
                if (pcg_alloc(s, &f, nb * 65536)) {            // two float arrays of nb * 65536 = one double array of that length
                    s->bj_dinv32 = reinterpret_cast<float *>(f);
                    s->bj_dinvt32 = s->bj_dinv32 + nb * 65536;
                } else (void)hipGetLastError();
            }
            s->bj_have = okb;
            if (!okb) (void)hipGetLastError();      // (out of memory for the boxes: the separable preconditioner alone)
        }
    }
    if (!ok) { pcg_destroy(s); return SPLPAK_E_NOMEM; }
    *out = s;
    return 0;
}

// After the binning: sum of w^2 of this rank's points into the histogram window's scalars (it travels through the sharded fit's first
// all-reduce, so that every rank builds the same preconditioner).
hipError_t pcg_sum_w2(splpak_plan *p, hipStream_t st)
{
    // (512 blocks, then one: the single-workgroup form took 2.4 ms for 1e7 weights)
    PcgState *s = p->pcg;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(DOT_BLOCKS), dim3(256), 0, st, (const double *)p->s.ws, (const int *)(p->s.offset + p->g.ncell), s->partial);
    hipLaunchKernelGGL(dot_finish_kernel, dim3(1), dim3(DOT_BLOCKS), 0, st, (const double *)s->partial, 1.0, p->scalH, (int)SC_SUMW2, (double *)nullptr);
    return hipGetLastError();
}

// Per fit, after the assembly: the two moments of the preconditioner (density of w^2: sumw2 as reduced over the ranks; mean squared
// constraint weight: from dcw / spf, which every rank of a fit that iterates computes) and its diagonal.
hipError_t pcg_prepare(splpak_plan *p, PcgState *s, double sumw2, bool smooth, bool from_rows, hipStream_t st)
{   // from_rows: the normal equations of this fit are not assembled (boxes = scaled mass + exact constraint part, bj_build_kernel)
    const Grid &g = p->g;
    double lam = 0.0;
    if (smooth) {
        hipLaunchKernelGGL(fd_lambda_kernel, dim3(1), dim3(1024), 0, st, (const double *)p->dcw, (const unsigned char *)p->spf, g.ncol, s->mom);
        hipError_t e = hipMemcpyAsync(&lam, s->mom, sizeof(double), hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) return e;
        e = hipStreamSynchronize(st);
        if (e != hipSuccess) return e;
    }
    double vol = 1.0;
    for (int k = 0; k < g.ndim; ++k) vol *= (double)(g.nodes[k] - 1) * g.dx[k];
    s->rho = sumw2 / std::fabs(vol);
    s->lambda = lam / (double)g.ncol;
    if (!(s->rho > 0.0)) s->rho = 1.0;
    hipLaunchKernelGGL(fd_diag_kernel, dim3((unsigned)((g.ncol + 255) / 256)), dim3(256), 0, st, g, s->ev, s->rho, s->lambda, s->dinv);
    s->total_iters = 0;
    s->solves = 0;
    s->failed = false;
    s->bj_ready = false;
    s->singular = false;
    if (s->bj_have && ((p->nst && !from_rows) || (from_rows && s->ddiag))) {
        const double inf = 1.0e300;
        hipError_t e = hipMemsetAsync(s->bj_info, 0, 2 * sizeof(int), st);
        if (e == hipSuccess) e = hipMemcpyAsync(s->bj_scal, &inf, sizeof(double), hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return e;
        if (!from_rows)
            hipLaunchKernelGGL(bj_extract_kernel, dim3((unsigned)s->bg.nb), dim3(256), 0, st, g, s->bg, (const double *)p->nst, s->bj_blocks);
        else {
            // no assembled N: the diagonal of the data rows' Gram matrix from the rows (this rank's points; summed over the ranks through the
            // residual window), then scaled mass + exact constraint part
            e = rowsop_data_diagonal(g, p->rowsop, p->s, s->t1, s->ddiag, st);
            if (e != hipSuccess) return e;
            if (p->world > 1) {
                e = hipMemcpyAsync(p->rho, s->ddiag, sizeof(double) * (size_t)g.ncol, hipMemcpyDeviceToDevice, st);
                if (e != hipSuccess) return e;
                e = hipMemsetAsync(p->rho + g.ncol, 0, sizeof(double) * (size_t)(p->lenR - g.ncol), st);
                if (e != hipSuccess) return e;
                if (plan_allreduce(p, p->rho, p->lenR, st) != 0) return hipErrorUnknown;
                e = hipMemcpyAsync(s->ddiag, p->rho, sizeof(double) * (size_t)g.ncol, hipMemcpyDeviceToDevice, st);
                if (e != hipSuccess) return e;
            }
            hipLaunchKernelGGL(bj_build_kernel, dim3((unsigned)s->bg.nb), dim3(256), 0, st, g, s->bg, s->mbands, (const double *)p->ctab,
                               (const double *)p->dcw, smooth ? (const unsigned char *)p->spf : (const unsigned char *)nullptr, (const double *)s->ddiag, s->bj_blocks);
        }
        e = block_chol_run(s->bj_jobs, s->bg.nb, s->bj_info, s->bj_scal, st);
        if (e != hipSuccess) return e;
        if (s->bj_pk32) {
            const long long nn = (long long)s->bg.nb * 40960;
            hipLaunchKernelGGL(bj_pack_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, (long long)s->bg.nb, (const double *)s->bj_dinv, s->bj_pk32);
        }
        if (s->bj_dinv32) {
            const long long nn = (long long)s->bg.nb * 65536;
            hipLaunchKernelGGL(to_f32_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, nn, (const double *)s->bj_dinv, (const double *)s->bj_dinvt,
                               s->bj_dinv32, s->bj_dinvt32);
        }
        int hinfo = 0;
        e = hipMemcpyAsync(&hinfo, s->bj_info, sizeof(int), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) return e;
        s->bj_ready = hinfo == 0;            // (a box that is not positive definite: the separable preconditioner alone)
        s->singular = hinfo != 0 && !from_rows;      // (an exact principal submatrix of N: N is not positive definite either)
        if (splpak::opt_get("SPLPAK_DEBUG")) fprintf(stderr, "[splpak pcg] %d boxes of N factored%s\n", s->bg.nb, s->bj_ready ? "" : ": one is not positive definite, dropped");
    }
    return hipGetLastError();
}

// z = M^-1 r.  (The boxes' term on a stream of its own beside the separable part's mode products was tried -- the way the constraint passes
// run beside the rows' tile kernel, rowsop.hip -- and bought nothing: 0.281 s per fit at 32^4 either way; both parts live on the memory system.)
static hipError_t pcg_precondition(PcgState *s, const double *r, double *z, hipStream_t st)
{
    const Grid &g = s->g;
    const long long total = g.ncol;
    const dim3 gr((unsigned)((total + 255) / 256)), bl(256);
    const double *src = r;
    double *bufs[2] = {s->t1, s->t2};
    int which = 0;
    // one direction of the transform: modes in pairs through LDS where the tile fits (both node counts <= 64), singly otherwise
    auto sweep = [&](double *const *mats, const double *scale_last, double *final_dst) {
        int k = 0;
        while (k < g.ndim) {
            const bool pair = !s->no_pairs && k + 1 < g.ndim && (long long)g.nodes[k] * g.nodes[k + 1] <= 4096;
            const int kend = pair ? k + 1 : k;
            double *dst = (kend == g.ndim - 1 && final_dst) ? final_dst : bufs[which];
            const double *sc = kend == g.ndim - 1 ? scale_last : nullptr;
            if (pair) {
                const int na = g.nodes[k], nb = g.nodes[k + 1];
                const long long inner = g.colstride[k];
                int ci = 1;
                const int kpa = (na + 3) & ~3, mpa = (na + 15) & ~15, kpb = (nb + 3) & ~3, mpb = (nb + 15) & ~15;
                auto lds_doubles = [&](int c) { return (long long)nb * kpa * c + (long long)kpb * na * c + (long long)kpa * mpa + (long long)kpb * mpb; };
                if (s->pairs_valu || lds_doubles(1) > 8192) {           // (the matrix-pipe form's images do not fit 64 KB: node counts near 64)
                    for (int c = 8; c >= 1; --c)
                        if (inner % c == 0 && (long long)na * nb * c <= 4096) { ci = c; break; }
                    const long long blocks = total / ((long long)na * nb * ci);
                    hipLaunchKernelGGL(mode_pair_kernel, dim3((unsigned)blocks), bl, sizeof(double) * 2 * (size_t)na * nb * ci, st, na, nb, inner, ci,
                                       (const double *)mats[k], (const double *)mats[k + 1], src, dst, sc);
                } else {
                    // (64 KB of LDS: the two images of the tile with their zero rows + the two padded matrices)
                    for (int c = 8; c >= 1; --c)
                        if (inner % c == 0 && lds_doubles(c) <= 8192) { ci = c; break; }
                    const long long blocks = total / ((long long)na * nb * ci);
                    hipLaunchKernelGGL(mode_pair_mfma_kernel, dim3((unsigned)blocks), bl, sizeof(double) * (size_t)lds_doubles(ci), st, na, nb, inner, ci,
                                       (const double *)mats[k], (const double *)mats[k + 1], src, dst, sc);
                }
            } else
                hipLaunchKernelGGL(mode_product_kernel, gr, bl, 0, st, g.nodes[k], (long long)g.colstride[k], total, (const double *)mats[k], src, dst, sc);
            src = dst;
            which ^= 1;
            k = kend + 1;
        }
    };
    sweep(s->V, s->dinv, nullptr);           // V_k^T along every dimension, the last one scaled by 1 / diag
    sweep(s->VT, nullptr, z);                // V_k along every dimension
    if (s->bj_ready && s->bj_pk32)
        hipLaunchKernelGGL(bj_apply_packed_kernel, dim3((unsigned)s->bg.nb), dim3(256), 0, st, g, s->bg, (const float *)s->bj_pk32, r, z);
    else if (s->bj_ready && s->bj_dinv32)
        hipLaunchKernelGGL(bj_apply_kernel<float>, dim3((unsigned)s->bg.nb), dim3(256), 0, st, g, s->bg, (const float *)s->bj_dinv32, (const float *)s->bj_dinvt32, r, z);
    else if (s->bj_ready)
        hipLaunchKernelGGL(bj_apply_kernel<double>, dim3((unsigned)s->bg.nb), dim3(256), 0, st, g, s->bg, (const double *)s->bj_dinv, (const double *)s->bj_dinvt, r, z);
    return hipGetLastError();
}

static hipError_t pcg_dot(PcgState *s, const double *a, const double *b, double sign, int slot, double *hist, hipStream_t st)
{
    hipLaunchKernelGGL(dot_partial_kernel, dim3(DOT_BLOCKS), dim3(256), 0, st, (long long)s->g.ncol, a, b, s->partial);
    hipLaunchKernelGGL(dot_finish_kernel, dim3(1), dim3(DOT_BLOCKS), 0, st, (const double *)s->partial, sign, s->sc, slot, hist);
    return hipGetLastError();
}

// v <- N^-1 v (approximately): conjugate gradients from a zero start until the preconditioned residual norm has fallen by `tol`.
// Returns 0 (converged), 1 (stagnated / iteration limit / breakdown: v holds the best iterate reached), or an SPLPAK_E_* code.
int pcg_solve(splpak_plan *p, PcgState *s, double *v, double tol, bool smooth, hipStream_t st)
{
    const Grid &g = p->g;
    const long long n = g.ncol;
    const dim3 gr((unsigned)((n + 255) / 256)), bl(256);
    const size_t nb = sizeof(double) * (size_t)n;
    const bool debug = splpak::opt_get("SPLPAK_DEBUG") != nullptr;
    SortScratch rows = p->s;
    rows.ys = nullptr;                                   // the residual pass as operator: rho = -N x
    SPLPAK_HIP_TRY(hipMemcpyAsync(s->r, v, nb, hipMemcpyDeviceToDevice, st), SPLPAK_E_NODEVICE);
    SPLPAK_HIP_TRY(hipMemsetAsync(s->x, 0, nb, st), SPLPAK_E_NODEVICE);
    SPLPAK_HIP_TRY(pcg_precondition(s, s->r, s->z, st), SPLPAK_E_NODEVICE);
    SPLPAK_HIP_TRY(hipMemcpyAsync(s->pv, s->z, nb, hipMemcpyDeviceToDevice, st), SPLPAK_E_NODEVICE);
    SPLPAK_HIP_TRY(pcg_dot(s, s->r, s->z, 1.0, S_RZ, s->hist, st), SPLPAK_E_NODEVICE);
    const int check = 5;
    s->hhist.assign((size_t)s->maxit + 8, 0.0);
    double rz0 = 0.0, best = 1.0;
    int it = 0, best_it = 0, status = 1, read_from = 0;
    double rel = 1.0;
    int last_gain_it = 0;                                // stagnation: the iteration at which the last factor of ten was gained
    double gain_level = 1.0;
    while (it < s->maxit) {
        const int upto = std::min(s->maxit, it + check);
        for (; it < upto; ++it) {
            SPLPAK_HIP_TRY(hipMemsetAsync(p->rho, 0, sizeof(double) * (size_t)(p->band.npad + SC_COUNT), st), SPLPAK_E_NODEVICE);
            SPLPAK_HIP_TRY(plan_rows_residual(p, rows, s->pv, smooth && p->rank == 0, p->rho, st), SPLPAK_E_NODEVICE);
            if (int rc = plan_allreduce(p, p->rho, p->lenR, st)) return rc;
            SPLPAK_HIP_TRY(pcg_dot(s, s->pv, p->rho, -1.0, S_PQ, nullptr, st), SPLPAK_E_NODEVICE);
            hipLaunchKernelGGL(update_xr_kernel, gr, bl, 0, st, n, (const double *)s->sc, s->x, s->r, (const double *)s->pv, (const double *)p->rho);
            SPLPAK_HIP_TRY(pcg_precondition(s, s->r, s->z, st), SPLPAK_E_NODEVICE);
            SPLPAK_HIP_TRY(pcg_dot(s, s->r, s->z, 1.0, S_RZNEW, s->hist + it + 1, st), SPLPAK_E_NODEVICE);
            hipLaunchKernelGGL(update_p_kernel, gr, bl, 0, st, n, (const double *)s->sc, s->pv, (const double *)s->z);
            hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(1), 0, st, s->sc);
        }
        const int first = read_from;
        read_from = it + 1;
        SPLPAK_HIP_TRY(hipMemcpyAsync(s->hhist.data() + first, s->hist + first, sizeof(double) * (size_t)(it - first + 1), hipMemcpyDeviceToHost, st),
                       SPLPAK_E_NODEVICE);
        SPLPAK_HIP_TRY(hipStreamSynchronize(st), SPLPAK_E_NODEVICE);
        if (rz0 == 0.0) {
            rz0 = s->hhist[0];
            if (rz0 == 0.0) { status = 0; rel = 0.0; break; }            // zero right-hand side
            if (!(rz0 > 0.0)) { status = 1; break; }
        }
        bool bad = false;
        for (int k = std::max(1, first); k <= it; ++k) {
            const double v2 = s->hhist[(size_t)k];
            if (!(v2 >= 0.0) || !std::isfinite(v2)) { bad = true; break; }
            rel = std::sqrt(v2 / rz0);
            if (rel < best) { best = rel; best_it = k; }
            if (rel < 0.1 * gain_level) { gain_level = rel; last_gain_it = k; }
        }
        if (debug) fprintf(stderr, "[splpak pcg] iteration %d: preconditioned residual %.3e (best %.3e at %d)\n", it, rel, best, best_it);
        if (bad) { status = 1; break; }
        if (rel <= tol) { status = 0; break; }
        // stagnation: no factor of ten gained in 400 iterations (a converging 4-D fit gains one in 10 .. 60) where a factorisation
        // stands behind the iteration; a plan that has nothing else is patient: 1 500 iterations per factor of ten (between 1.2 and
        // 1.5 constraint rows per column the iteration crawls at ~200 .. 600 per factor of ten, but it arrives)
        // (patience: 400 iterations without a factor of ten where a cheap factorisation stands behind the iteration, 1 500 where none
        //  does or it would take seconds)
        if (it - last_gain_it > ((p->solver_mode == 2 || p->factor_flop / 45.0e12 >= 1.0) ? 1500 : 400)) { status = 1; break; }
    }
    s->last_iters = it;
    s->last_rel = rel;
    s->total_iters += it;
    s->solves += 1;
    SPLPAK_HIP_TRY(hipMemcpyAsync(v, s->x, nb, hipMemcpyDeviceToDevice, st), SPLPAK_E_NODEVICE);
    if (status != 0) s->failed = true;
    return status;
}

}  // namespace splpak
