// Blocked band Cholesky  N = L L^T  and band triangular solves on gfx950.
//
// Replaces the triangularisation + back-substitution of the reference's dense
// row-streaming Householder solver suprls (src/splpak.F90:1375-1695) by a
// factorisation of the (banded) normal equations; iterative refinement against
// the rows (plan.hip) restores the accuracy of the orthogonal method.
//
// Storage: LAPACK-style lower band, column j holds A(j..j+ld-1, j) contiguously,
// so that A(i,j) = ab[i + j*lda] with lda = ld-1: every sub-block of the band is
// an ordinary column-major matrix with leading dimension lda.  ld is chosen so
// that lda is a multiple of 16 doubles (128-B lines) but not of a large power of
// two (HBM channel spread).
//
// Right-looking blocked algorithm with NBLK = 256 columns per step:
//   potrf_block_kernel  256x256 diagonal block, one workgroup, LDS-resident
//                       64-column panels
//   trsm_kernel         panel below the diagonal block, one thread per row,
//                       exact forward substitution (L_kk read through the scalar
//                       cache)
//   syrk_kernel         trailing update C -= P P^T on the f64 matrix cores
//                       (v_mfma_f64_16x16x4_f64), 128x128 tile per workgroup,
//                       LDS double-buffered K-chunks of the panel.  This kernel
//                       carries ~n*p^2 of the flops (4.1e13 at 64^3 nodes) and
//                       is the MFMA-bound roofline kernel of the fit.
// Solves use explicit inverses of the 256x256 diagonal blocks of L (trtri_kernel,
// computed once after the factorisation, off the critical path) so that every
// step of the forward / backward sweep is one short, fully parallel kernel.
#include "kernels.hpp"

namespace splpak {

namespace {

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));

constexpr int PNL = 64;            // potrf inner panel width
constexpr int PLD = NBLK + 1;      // LDS leading dimension of the potrf panel

// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
potrf_block_kernel(double *__restrict__ ab, long long lda, int k0, int *__restrict__ info,
                   double *__restrict__ minpiv)
{
    __shared__ double P[PNL * PLD];
    double *A = ab + (long long)k0 + (long long)k0 * lda;   // A(r,c) = A[r + c*lda], r >= c
    const int tid = threadIdx.x;

    for (int c0 = 0; c0 < NBLK; c0 += PNL) {
        for (int idx = tid; idx < PNL * NBLK; idx += 1024) {
            const int cc = idx / NBLK, r = idx % NBLK;
            double v = 0.0;
            if (r >= c0 + cc) v = A[r + (long long)(c0 + cc) * lda];
            P[cc * PLD + r] = v;
        }
        __syncthreads();
        for (int j = 0; j < PNL; ++j) {
            const double d = P[j * PLD + c0 + j];
            if (tid == 0) {
                if (!(d > 0.0)) atomicCAS(info, 0, k0 + c0 + j + 1);
                if (d < *minpiv) *minpiv = d;
            }
            const double sd = sqrt(d);
            __syncthreads();
            if (tid < NBLK) {
                const int r = tid;
                if (r > c0 + j) P[j * PLD + r] /= sd;
                else if (r == c0 + j) P[j * PLD + r] = sd;
            }
            __syncthreads();
            const int ncols = PNL - 1 - j;
            for (int idx = tid; idx < ncols * NBLK; idx += 1024) {
                const int cc = j + 1 + idx / NBLK, r = idx % NBLK;
                if (r >= c0 + cc) P[cc * PLD + r] -= P[j * PLD + r] * P[j * PLD + c0 + cc];
            }
            __syncthreads();
        }
        for (int idx = tid; idx < PNL * NBLK; idx += 1024) {
            const int cc = idx / NBLK, r = idx % NBLK;
            if (r >= c0 + cc) A[r + (long long)(c0 + cc) * lda] = P[cc * PLD + r];
        }
        const int nrem = NBLK - c0 - PNL;       // columns right of the panel
        for (int idx = tid; idx < nrem * NBLK; idx += 1024) {
            const int c = c0 + PNL + idx / NBLK, r = idx % NBLK;
            if (r >= c) {
                double s = 0.0;
#pragma unroll 8
                for (int kk = 0; kk < PNL; ++kk) s += P[kk * PLD + r] * P[kk * PLD + c];
                A[r + (long long)c * lda] -= s;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// X = A * L^{-T} for the rows below the diagonal block: one thread per row.
constexpr int TCB = 32;
__global__ void __launch_bounds__(64)
trsm_kernel(const double *__restrict__ L, double *__restrict__ Xbase, long long lda, int nrows)
{
    // L = diagonal block (read-only here, wave-uniform addresses -> scalar loads),
    // Xbase = first row below it; both views of the band with column stride lda
    const int rloc = blockIdx.x * 64 + threadIdx.x;
    if (rloc >= nrows) return;
    double *__restrict__ X = Xbase + rloc;   // X[c*lda]

    for (int cb = 0; cb < NBLK / TCB; ++cb) {
        double acc[TCB];
#pragma unroll
        for (int c = 0; c < TCB; ++c) acc[c] = X[(long long)(cb * TCB + c) * lda];
        for (int kb = 0; kb < cb; ++kb) {
            double xk[TCB];
#pragma unroll
            for (int k = 0; k < TCB; ++k) xk[k] = X[(long long)(kb * TCB + k) * lda];
#pragma unroll
            for (int k = 0; k < TCB; ++k) {
                const double *__restrict__ Lc = L + (cb * TCB) + (long long)(kb * TCB + k) * lda;
#pragma unroll
                for (int c = 0; c < TCB; ++c) acc[c] -= xk[k] * Lc[c];
            }
        }
        const double *__restrict__ Ld = L + (cb * TCB) + (long long)(cb * TCB) * lda;
#pragma unroll
        for (int c = 0; c < TCB; ++c) {
#pragma unroll
            for (int k = 0; k < c; ++k) acc[c] -= acc[k] * Ld[c + (long long)k * lda];
            acc[c] /= Ld[c + (long long)c * lda];
        }
#pragma unroll
        for (int c = 0; c < TCB; ++c) X[(long long)(cb * TCB + c) * lda] = acc[c];
    }
}

// ---------------------------------------------------------------------------
// Trailing update on the f64 matrix cores.
constexpr int TS = 128;            // C tile edge per workgroup
constexpr int KC = 16;             // K chunk staged per LDS buffer
constexpr int LDT = TS + 16;       // padded LDS row: k-rows land on alternating bank halves

__global__ void __launch_bounds__(256, 2)
syrk_kernel(double *__restrict__ ab, long long lda, int k0, int row0, int nt, int tj_begin)
{
    __shared__ double sI[2][KC * LDT];   // panel rows of the C-row block  (MFMA B operand)
    __shared__ double sJ[2][KC * LDT];   // panel rows of the C-col block  (MFMA A operand)

    // block -> lower-triangular tile (ti >= tj)
    int b = blockIdx.x, tj = tj_begin;
    while (b >= nt - tj) { b -= nt - tj; ++tj; }
    const int ti = tj + b;
    const bool diag = (ti == tj);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave & 1, wj = wave >> 1;
    const int l15 = lane & 15, kq = lane >> 4;
    const bool wave_on = !(diag && wi < wj);     // strictly-upper quarter of a diagonal tile

    const double *__restrict__ panI = ab + (long long)(row0 + ti * TS) + (long long)k0 * lda;
    const double *__restrict__ panJ = ab + (long long)(row0 + tj * TS) + (long long)k0 * lda;

    d4_t acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = (d4_t){0.0, 0.0, 0.0, 0.0};

    // staging: 4 x 16-byte loads per thread and operand per chunk; one
    // wave-instruction reads 1 KiB contiguous (one k column, 128 rows)
    const int skk = tid >> 6, srp = tid & 63;
    d2_t rI[4], rJ[4];
    auto gload = [&](int kc) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long off = (long long)(kc + u * 4 + skk) * lda + 2 * srp;
            rI[u] = *reinterpret_cast<const d2_t *>(panI + off);
            rJ[u] = *reinterpret_cast<const d2_t *>(panJ + off);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int o = (u * 4 + skk) * LDT + 2 * srp;
            *reinterpret_cast<d2_t *>(&sI[buf][o]) = rI[u];
            *reinterpret_cast<d2_t *>(&sJ[buf][o]) = rJ[u];
        }
    };

    gload(0);
    lstore(0);
    __syncthreads();
    int buf = 0;
    for (int kc = 0; kc < NBLK; kc += KC) {
        const bool more = (kc + KC < NBLK);
        if (more) gload(kc + KC);
        if (wave_on) {
#pragma unroll
            for (int ks = 0; ks < KC / 4; ++ks) {
                double a[4], bb[4];
                const int krow = (ks * 4 + kq) * LDT + l15;
#pragma unroll
                for (int m = 0; m < 4; ++m) a[m] = sJ[buf][krow + wj * 64 + m * 16];
#pragma unroll
                for (int n = 0; n < 4; ++n) bb[n] = sI[buf][krow + wi * 64 + n * 16];
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], bb[n], acc[m][n], 0, 0, 0);
            }
        }
        if (more) lstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    if (!wave_on) return;
    // D[i][j]: lane holds j = lane&15 (C row), i = (lane>>4) + 4*v (C column)
    double *__restrict__ C = ab + (long long)(row0 + ti * TS) + (long long)(row0 + tj * TS) * lda;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int r = wi * 64 + n * 16 + l15;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int c = wj * 64 + m * 16 + kq + 4 * v;
                if (!diag || r >= c) C[r + (long long)c * lda] -= acc[m][n][v];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Inverse of every 256x256 diagonal block of L, row-major: T[r*256 + c] = Linv(r,c).
// One workgroup per block, one thread per column of the inverse.
__global__ void __launch_bounds__(256)
trtri_kernel(const double *__restrict__ ab, long long lda, double *__restrict__ dinv)
{
    const int k0 = blockIdx.x * NBLK;
    const int c = threadIdx.x;
    const double *__restrict__ L = ab + ((long long)k0 + (long long)k0 * lda);
    double *__restrict__ T = dinv + (long long)blockIdx.x * NBLK * NBLK;

    for (int rb = 0; rb < NBLK / TCB; ++rb) {
        double acc[TCB];
#pragma unroll
        for (int r = 0; r < TCB; ++r) acc[r] = (rb * TCB + r == c) ? 1.0 : 0.0;
        for (int kb = 0; kb < rb; ++kb) {
            double xk[TCB];
#pragma unroll
            for (int k = 0; k < TCB; ++k) xk[k] = T[(kb * TCB + k) * NBLK + c];
#pragma unroll
            for (int k = 0; k < TCB; ++k) {
                const double *__restrict__ Lc = L + (rb * TCB) + (long long)(kb * TCB + k) * lda;
#pragma unroll
                for (int r = 0; r < TCB; ++r) acc[r] -= xk[k] * Lc[r];
            }
        }
        const double *__restrict__ Ld = L + (rb * TCB) + (long long)(rb * TCB) * lda;
#pragma unroll
        for (int r = 0; r < TCB; ++r) {
#pragma unroll
            for (int k = 0; k < r; ++k) acc[r] -= acc[k] * Ld[r + (long long)k * lda];
            acc[r] /= Ld[r + (long long)r * lda];
        }
#pragma unroll
        for (int r = 0; r < TCB; ++r) T[(rb * TCB + r) * NBLK + c] = acc[r];
    }
}

// ---------------------------------------------------------------------------
__device__ inline double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// forward step k:  y_k = Linv_k v_k  (-> yout),  v[rows below] -= L[rows, k] y_k
// grid: max(1, nrows/64) workgroups of 256 threads; every workgroup recomputes y_k.
__global__ void __launch_bounds__(256)
fwd_step_kernel(const double *__restrict__ ab, long long lda, const double *__restrict__ dinv,
                int k, int nrows, double *__restrict__ v, double *__restrict__ yout)
{
    __shared__ double sv[NBLK];
    __shared__ double sy[NBLK];
    __shared__ double part[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k0 = k * NBLK;
    sv[tid] = v[k0 + tid];
    __syncthreads();
    const double *__restrict__ T = dinv + (long long)k * NBLK * NBLK;
    for (int r = wave; r < NBLK; r += 4) {
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = lane + 64 * u;
            if (c <= r) s += T[r * NBLK + c] * sv[c];
        }
        s = wave_sum(s);
        if (lane == 0) sy[r] = s;
    }
    __syncthreads();
    if (blockIdx.x == 0) yout[k0 + tid] = sy[tid];
    if (nrows <= 0) return;
    // rows [blockIdx.x*64, +64) below the block: 4 column quarters per row
    const int r = blockIdx.x * 64 + lane;
    double s = 0.0;
    if (r < nrows) {
        const double *__restrict__ Lr = ab + (long long)(k0 + NBLK + r) + (long long)(k0 + wave * 64) * lda;
#pragma unroll 8
        for (int c = 0; c < 64; ++c) s += Lr[(long long)c * lda] * sy[wave * 64 + c];
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && r < nrows)
        v[k0 + NBLK + r] -= (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

// backward step k:  x_k = Linv_k^T y_k (-> xout),  y[cols left] -= L[k rows, cols]^T x_k
// grid: max(1, ncols/64) workgroups; columns jbeg .. jbeg+ncols-1 (ending at k0-1).
__global__ void __launch_bounds__(256)
bwd_step_kernel(const double *__restrict__ ab, long long lda, const double *__restrict__ dinv,
                int k, int jbeg, int ncols, double *__restrict__ y, double *__restrict__ xout)
{
    __shared__ double sy[NBLK];
    __shared__ double sx[NBLK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k0 = k * NBLK;
    sy[tid] = y[k0 + tid];
    __syncthreads();
    const double *__restrict__ T = dinv + (long long)k * NBLK * NBLK;
    {
        double s = 0.0;                      // x_c = sum_{r >= c} Linv(r,c) y_r
#pragma unroll 8
        for (int r = tid; r < NBLK; ++r) s += T[r * NBLK + tid] * sy[r];
        sx[tid] = s;
    }
    __syncthreads();
    if (blockIdx.x == 0) xout[k0 + tid] = sx[tid];
    if (ncols <= 0) return;
    // 64 columns per workgroup, 16 per wave; a wave dots one 256-row column segment
    for (int q = 0; q < 16; ++q) {
        const int jl = blockIdx.x * 64 + wave * 16 + q;
        if (jl >= ncols) break;
        const int j = jbeg + jl;
        const double *__restrict__ Lc = ab + (long long)k0 + (long long)j * lda;
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) s += Lc[lane + 64 * u] * sx[lane + 64 * u];
        s = wave_sum(s);
        if (lane == 0) y[j] -= s;
    }
}

__global__ void __launch_bounds__(256)
axpy_absmax_kernel(int n, double *__restrict__ x, const double *__restrict__ dx,
                   unsigned long long *__restrict__ absmax2)
{
    double mdx = 0.0, mx = 0.0;
    const int stride = gridDim.x * blockDim.x;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double d = dx[i];
        const double xv = x[i] + d;
        x[i] = xv;
        mdx = fmax(mdx, fabs(d));
        mx = fmax(mx, fabs(xv));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mdx = fmax(mdx, __shfl_xor(mdx, o, 64));
        mx = fmax(mx, __shfl_xor(mx, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        // non-negative doubles order like their bit patterns; NaN (sign clear) sorts above inf
        atomicMax(&absmax2[0], (unsigned long long)__double_as_longlong(mdx));
        atomicMax(&absmax2[1], (unsigned long long)__double_as_longlong(mx));
    }
}

}  // namespace

// ---------------------------------------------------------------------------
size_t band_bytes(int n, int halfbw, Band *d)
{
    Band b{};
    b.n = n;
    b.npad = ((n + NBLK - 1) / NBLK) * NBLK;
    b.nblk = b.npad / NBLK;
    b.bw = (halfbw + NBLK - 1) / NBLK;
    if (b.bw < 1) b.bw = 1;
    if (b.bw > b.nblk - 1) b.bw = (b.nblk - 1 > 0) ? b.nblk - 1 : 0;
    // rows touched in a column: up to (bw+1)*NBLK; lda multiple of 16, odd multiple of 16
    b.lda = (long long)(b.bw + 1) * NBLK + 16;
    const long long ld = b.lda + 1;
    b.bytes = (size_t)ld * (size_t)b.npad * sizeof(double) + 4096;
    if (d) *d = b;
    return b.bytes;
}

hipError_t band_cholesky(const Band &b, int *info_dev, double *minpiv_dev, hipStream_t st,
                         CholStats *stats)
{
    hipEvent_t e0 = nullptr, e1 = nullptr, f0 = nullptr, f1 = nullptr;
    const bool timing = stats && stats->enabled;
    if (timing) {
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&f0); (void)hipEventCreate(&f1);
        stats->syrk_launches = stats->syrk_ms = stats->syrk_flop = stats->factor_ms = 0;
        (void)hipEventRecord(f0, st);
    }
    // per-launch event pairs are collected and read after the loop
    std::vector<hipEvent_t> evs;
    for (int k = 0; k < b.nblk; ++k) {
        const int k0 = k * NBLK;
        hipLaunchKernelGGL(potrf_block_kernel, dim3(1), dim3(1024), 0, st, b.ab, b.lda, k0, info_dev,
                           minpiv_dev);
        int tb = b.nblk - 1 - k;
        if (tb > b.bw) tb = b.bw;
        if (tb <= 0) continue;
        const int nrows = tb * NBLK;
        hipLaunchKernelGGL(trsm_kernel, dim3(nrows / 64), dim3(64), 0, st,
                           (const double *)(b.ab + (long long)k0 + (long long)k0 * b.lda),
                           b.ab + (long long)(k0 + NBLK) + (long long)k0 * b.lda, b.lda, nrows);
        const int nt = nrows / TS;
        const int ntiles = nt * (nt + 1) / 2;
        if (timing) {
            hipEvent_t a, c;
            (void)hipEventCreate(&a); (void)hipEventCreate(&c);
            (void)hipEventRecord(a, st);
            hipLaunchKernelGGL(syrk_kernel, dim3(ntiles), dim3(256), 0, st, b.ab, b.lda, k0, k0 + NBLK, nt, 0);
            (void)hipEventRecord(c, st);
            evs.push_back(a); evs.push_back(c);
            stats->syrk_launches += 1;
            stats->syrk_flop += 2.0 * (double)ntiles * TS * TS * NBLK;
        } else {
            hipLaunchKernelGGL(syrk_kernel, dim3(ntiles), dim3(256), 0, st, b.ab, b.lda, k0, k0 + NBLK, nt, 0);
        }
    }
    hipLaunchKernelGGL(trtri_kernel, dim3(b.nblk), dim3(256), 0, st, b.ab, b.lda, b.dinv);
    hipError_t err = hipGetLastError();
    if (timing) {
        (void)hipEventRecord(f1, st);
        (void)hipEventSynchronize(f1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, f0, f1);
        stats->factor_ms = ms;
        for (size_t i = 0; i + 1 < evs.size(); i += 2) {
            (void)hipEventElapsedTime(&ms, evs[i], evs[i + 1]);
            stats->syrk_ms += ms;
            (void)hipEventDestroy(evs[i]); (void)hipEventDestroy(evs[i + 1]);
        }
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(f0); (void)hipEventDestroy(f1);
    }
    return err;
}

hipError_t band_solve(const Band &b, double *x, double *tmp, hipStream_t st)
{
    // forward: x -> tmp ; backward: tmp -> x
    for (int k = 0; k < b.nblk; ++k) {
        int tb = b.nblk - 1 - k;
        if (tb > b.bw) tb = b.bw;
        const int nrows = tb * NBLK;
        const int wg = nrows > 0 ? nrows / 64 : 1;
        hipLaunchKernelGGL(fwd_step_kernel, dim3(wg), dim3(256), 0, st, b.ab, b.lda, b.dinv, k, nrows, x, tmp);
    }
    for (int k = b.nblk - 1; k >= 0; --k) {
        int tb = k;
        if (tb > b.bw) tb = b.bw;
        const int ncols = tb * NBLK;
        const int jbeg = k * NBLK - ncols;
        const int wg = ncols > 0 ? ncols / 64 : 1;
        hipLaunchKernelGGL(bwd_step_kernel, dim3(wg), dim3(256), 0, st, b.ab, b.lda, b.dinv, k, jbeg, ncols, tmp, x);
    }
    return hipGetLastError();
}

hipError_t launch_axpy_absmax(int n, double *x, const double *dx, double *absmax2, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(absmax2, 0, 2 * sizeof(double), st);
    if (e != hipSuccess) return e;
    int blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(axpy_absmax_kernel, dim3(blocks), dim3(256), 0, st, n, x, dx,
                       reinterpret_cast<unsigned long long *>(absmax2));
    return hipGetLastError();
}

}  // namespace splpak
