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
// Right-looking blocked algorithm with NBLK = 256 columns per step (band_cholesky):
//   potrf_strip_kernel  256x256 diagonal block: one workgroup, 64-column strips in LDS, leaf in
//                       registers (v_readlane), in-strip update on the matrix cores; also
//                       writes the inverses of the 16x16 diagonal leaves (chol_device.hpp)
//   trsm_kernel         panel below the diagonal block: one wave per 16 rows, 16-column
//                       blocks on the matrix cores against the leaf inverses
//   syrk64_kernel       trailing update C -= P P^T on the f64 matrix cores
//                       (v_mfma_f64_16x16x4_f64): one wave per 64x64 item, operands streamed
//                       from L2 into a register queue, no LDS.  This kernel carries ~n*p^2
//                       of the flops (4.1e13 at 64^3 nodes) and is the roofline kernel of the
//                       band fit.  The earlier LDS-tiled 128x128 form and the
//                       ablation variants live in tools/ablation_kernels.hpp, outside the library.
// The steps are pipelined over four HIP streams with one block column of look-ahead.
// Solves (band_solve) use explicit inverses of the 256x256 diagonal blocks of L
// (trinv_kernel) and the coupling blocks of sweepmat_kernel, computed once after the
// factorisation, so that every step of the forward / backward sweep is ONE short launch.
#include "kernels.hpp"
#include "chol_device.hpp"
#include <chrono>
#include <hip/hip_ext.h>
#include <cstdlib>

namespace splpak {

namespace {



__global__ void __launch_bounds__(64)
trsm_kernel(const double *__restrict__ L, double *__restrict__ Xbase, long long lda,
            const double *__restrict__ inv16, int nrows)
{
    __shared__ double xs[(NBLK - 16) * 16];               // xs[col*16 + row] = -X(row, col), columns 0..239
    const int r0 = blockIdx.x * 16;
    if (r0 >= nrows) return;
    __builtin_amdgcn_s_setprio(3);
    trsm_rows<false>(L, Xbase, lda, lda, inv16, nullptr, r0, xs);
}

// Inverses of the 256x256 diagonal blocks of L as the panel solve of the identity, X = I L^{-T}: wave (y, x)
// produces rows 16x.. of X for the y-th block, so that  dinvt[r*256 + c] = X(r, c) = Linv(c, r)  (row-major
// L^{-T}) and dinv = its transpose (row-major Linv; the zero triangles are stored, the sweeps read full rows).
// Replaces a thread-per-column substitution that took 0.5 ms per block (a fixed cost of every fit, 1.6 ms at
// 64^3); this one runs on the matrix cores against the leaf inverses potrf leaves behind.
__global__ void __launch_bounds__(64)
trinv_kernel(const double *__restrict__ ab, long long lda, const double *__restrict__ inv16, double *__restrict__ dinv,
             double *__restrict__ dinvt, DistMap dm, const int *__restrict__ blocks)
{
    __shared__ double xs[(NBLK - 16) * 16];
    // block column handled by this workgroup: the blockIdx.y-th OWNED one when the band is distributed
    const int J = blocks ? blocks[blockIdx.y] : (int)blockIdx.y;
    const long long k0 = (long long)J * NBLK;
    const double *__restrict__ L = ab + dm_shift(dm, J) + (k0 + k0 * lda);
    // column-major X with leading dimension 256 is row-major Linv: X(r, c) at [r + 256 c] = Linv(c, r)
    trsm_rows<true>(L, dinv + (long long)blockIdx.y * NBLK * NBLK, lda, NBLK, inv16 + (long long)blockIdx.y * 4 * 64 * 64,
                    dinvt + (long long)blockIdx.y * NBLK * NBLK, blockIdx.x * 16, xs);
}

// Cholesky of one 256x256 diagonal block, strip form (chol_device.hpp), one workgroup.
__global__ void __launch_bounds__(256)
potrf_strip_kernel(double *__restrict__ ab, long long lda, int k0, int *__restrict__ info,
                   double *__restrict__ minpiv, double *__restrict__ inv16)
{
    potrf_strip_body(ab + (long long)k0 + (long long)k0 * lda, lda, k0, info, minpiv, inv16);
}

// ---------------------------------------------------------------------------
// Trailing update, register-streaming form: one wave = one 64x64 piece of C, no LDS, no
// barriers.  The MFMA operands are loaded straight from the panel (L2 / Infinity-Cache
// resident) in fragment shape -- lane (l15, q) reads P[row0 + 16m + l15][k + q], 16 rows
// = one 128-B line per k -- SD k-steps ahead into a rotating register queue; every loaded
// operand feeds 4 MFMAs, so the load path carries only 16 B/clk/CU.  v_mfma_f64_16x16x4
// occupies the pipe for 64 cycles, which leaves ample time for 8 loads per 16 MFMAs.  The
// accumulators START as the C tile and the products are subtracted, so the tile is read at
// the very beginning -- together with the first operands, one exposed latency -- and the
// epilogue is stores only.  (The LDS-tiled 128x128 form, the variants without operand refills /
// without the C read-modify-write, K = 512 / 1024 passes, XCD-blocked item orders and four-wave
// workgroups that were measured against this kernel live in tools/ablation_kernels.hpp.)
#ifndef SYRK_SD
#define SYRK_SD 4
#endif
__device__ inline unsigned my_cu_id()
{   // (XCC id, shader engine, CU) of the CU this wave runs on
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    return ((xcc & 0xfu) << 16) | (hw & 0xff00u);
}

__global__ void whoami_kernel(unsigned *out) { if (threadIdx.x == 0 && out) out[0] = my_cu_id(); }

// SD = k-steps of look-ahead in the operand queue; WPS = waves per SIMD the register budget allows.
// NEGEND = false (the bulk launch, two waves per SIMD hide each other's waits): the A operand is negated at load
// time, the accumulators collect C - P P^T directly.
// NEGEND = true (the small launches of the panel chain, which have a SIMD per wave): accumulators start as -C and
// collect +P P^T, the epilogue stores their negatives (bit for bit C - P P^T), and the refills are pinned SD k-steps
// ahead of their use with scheduling barriers.  With the operand negated at load time the compiler waits for every
// refill right behind its issue (s_waitcnt vmcnt(2) after the loads, then the v_xor), which a second wave on the SIMD
// hides in the bulk launch (measured there: 0.695 ms this way, 0.730 ms pinned with two waves and SD 4) but a lone
// wave pays as one memory round trip per k-step.
// queue[0]: next item, queue[1]: waves that stepped aside.  A wave that finds itself on the CU reserved for the panel
// factorisation (`reserved`, ~0u = none) steps aside without taking an item -- the grid carries `margin` spare waves
// for that -- unless the margin is used up.
template <int SD, int WPS, bool NEGEND>
__global__ void __launch_bounds__(64, WPS)
syrk64_kernel(double *__restrict__ ab, long long lda, int k0, int row0, int cb, int ce, int rb, int re,
              int nitems, int margin, unsigned reserved, int *__restrict__ queue)
{
    int it = (int)blockIdx.x;
    if (queue) {
        if (reserved != ~0u && my_cu_id() == reserved) {
            int e = 0;
            if (threadIdx.x == 0) e = atomicAdd(&queue[1], 1);
            e = __builtin_amdgcn_readfirstlane(e);
            if (e < margin) {
                __builtin_amdgcn_s_sleep(127);       // ~8k cycles: do not drain the grid through this CU
                __builtin_amdgcn_s_sleep(127);
                return;
            }
        }
        if (threadIdx.x == 0) it = atomicAdd(&queue[0], 1);
        it = __builtin_amdgcn_readfirstlane(it);
        if (it >= nitems) return;
    }
    // item -> (tj, ti) in 64-row units: columns [cb, ce), rows [max(tj, rb), re)
    int tj = cb, ti;
    for (;;) {
        const int lo = tj > rb ? tj : rb;
        const int cnt = re - lo;
        if (it < cnt) { it += lo; break; }
        it -= cnt;
        ++tj;
    }
    ti = it;
    const bool diag = (ti == tj);
    const int lane = threadIdx.x & 63, l15 = lane & 15, q = lane >> 4;

    // per-lane operand streams: element (m, step) at base + 16*m + step*4*lda
    const double *__restrict__ pJ = ab + (long long)(row0 + tj * 64 + l15) + (long long)(k0 + q) * lda;
    const double *__restrict__ pI = ab + (long long)(row0 + ti * 64 + l15) + (long long)(k0 + q) * lda;
    double *__restrict__ C = ab + (long long)(row0 + ti * 64) + (long long)(row0 + tj * 64) * lda;
    d4_t acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const double c0v = __builtin_nontemporal_load(&C[(n * 16 + l15) + (long long)(m * 16 + q + 4 * v) * lda]);
                acc[m][n][v] = NEGEND ? -c0v : c0v;
            }

    double qa[SD][4], qb[SD][4];
    auto fetch = [&](int slot, int step) {
        const long long off = (long long)(4 * step) * lda;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            qa[slot][m] = NEGEND ? pJ[off + 16 * m] : -pJ[off + 16 * m];
            qb[slot][m] = pI[off + 16 * m];
        }
    };
#pragma unroll
    for (int d = 0; d < SD; ++d) fetch(d, d);
    constexpr int NSTEP = NBLK / 4;
    static_assert(NSTEP % SD == 0, "queue depth must divide the k-steps");
    for (int ks = 0; ks < NSTEP; ks += SD) {
#pragma unroll
        for (int d = 0; d < SD; ++d) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[d][m], qb[d][n], acc[m][n], 0, 0, 0);
            // NEGEND: the refill of slot d is issued HERE, SD k-steps ahead of its use, and stays here (without the
            // scheduling barriers the compiler sinks the side-effect free loads down to the MFMA that consumes them)
            if (NEGEND) __builtin_amdgcn_sched_barrier(0);
            if (ks + d + SD < NSTEP) fetch(d, ks + d + SD);
            if (NEGEND) __builtin_amdgcn_sched_barrier(0);
        }
    }
    // C(r, c) <- acc: lane holds r = l15 (+16n), c = q + 4v (+16m)
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int r = n * 16 + l15;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int c = m * 16 + q + 4 * v;
                if (!diag || r >= c) __builtin_nontemporal_store(NEGEND ? -acc[m][n][v] : acc[m][n][v], &C[r + (long long)c * lda]);
            }
        }
}


// out[0..255] = M v for a row-major 256x256 block M (a diagonal-block inverse or its
// transpose; the zero triangle is stored).  grid 16 x 256 threads: a wave dots 4 rows.
__global__ void __launch_bounds__(256)
blockmv_kernel(const double *__restrict__ M, const double *__restrict__ v, double *__restrict__ out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * 16 + wave * 4;
    double vv[4], s[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) vv[u] = v[lane + 64 * u];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s[i] = 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) s[i] += M[(r0 + i) * NBLK + lane + 64 * u] * vv[u];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i] = wave_sum(s[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) out[r0 + i] = s[i];
    }
}

// forward sweep, panel part:  v[rows below block k] -= L[rows, block k] y_k.
// workgroup = 64 rows x 256 columns, 512 threads = 32 row pairs x 16 column groups.
__global__ void __launch_bounds__(512)
fwd_update_kernel(const double *__restrict__ Lpanel, long long lda, const double *__restrict__ yk,
                  double *__restrict__ vbelow, int nrows)
{
    __shared__ double sy[NBLK];
    __shared__ double part[16][64];
    const int tid = threadIdx.x;
    if (tid < NBLK) sy[tid] = yk[tid];
    __syncthreads();
    const int rp = tid & 31, cg = tid >> 5;
    const int r = blockIdx.x * 64 + 2 * rp;
    const double *__restrict__ Lr = Lpanel + r + (long long)(cg * 16) * lda;
    d2_t l[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) l[c] = *reinterpret_cast<const d2_t *>(Lr + (long long)c * lda);
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const double yv = sy[cg * 16 + c];
        s0 += l[c][0] * yv;
        s1 += l[c][1] * yv;
    }
    part[cg][2 * rp] = s0;
    part[cg][2 * rp + 1] = s1;
    __syncthreads();
    if (tid < 64) {
        double s = 0.0;
#pragma unroll
        for (int g = 0; g < 16; ++g) s += part[g][tid];
        vbelow[blockIdx.x * 64 + tid] -= s;
    }
}

// backward sweep, panel part:  y[j] -= L[block k rows, j]^T x_k for the columns left of block k.
// workgroup = 64 columns; a wave takes 16 of them, 4 at a time: lane = (column, 16-row segment).
__global__ void __launch_bounds__(256)
bwd_update_kernel(const double *__restrict__ Lrows, long long lda, const double *__restrict__ xk,
                  double *__restrict__ yleft, int ncols)
{
    __shared__ double sx[NBLK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    sx[tid] = xk[tid];
    __syncthreads();
    const int seg = lane & 15, cs = lane >> 4;
    double xs[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) xs[i] = sx[seg * 16 + i];
    d2_t l[4][8];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
        const int j = blockIdx.x * 64 + wave * 16 + cc * 4 + cs;
        const double *__restrict__ Lc = Lrows + (long long)j * lda + seg * 16;
#pragma unroll
        for (int u = 0; u < 8; ++u) l[cc][u] = *reinterpret_cast<const d2_t *>(Lc + 2 * u);
    }
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
        const int j = blockIdx.x * 64 + wave * 16 + cc * 4 + cs;
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) s += l[cc][u][0] * xs[2 * u] + l[cc][u][1] * xs[2 * u + 1];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (seg == 0 && j < ncols) yleft[j] -= s;
    }
}

// ---------------------------------------------------------------------------
// Coupling blocks of the one-kernel-per-step band sweeps.  With
//   M_k = Linv_{k+1} L_{k+1,k}          (forward)      N_k = Linv_k^T L_{k+1,k}^T     (backward)
// the next solved block is  y_{k+1} = Linv_{k+1} v'_{k+1} - M_k y_k  (v' = right-hand side updated by
// the steps before k only), so a step no longer waits for the panel update of the previous one:
// the 256x512 product [Linv | -M] [v'; y_k] and the panel update by y_k run in the same launch.
// grid (2*(nblk-1), 16): x = block pair and direction, y = 64x64 tile; a wave owns 16 rows.
__global__ void __launch_bounds__(256)
sweepmat_kernel(const double *__restrict__ ab, long long lda, const double *__restrict__ dinv,
                const double *__restrict__ dinvt, double *__restrict__ mfwd, double *__restrict__ mbwd,
                int npairs)
{
    const bool back = (int)blockIdx.x >= npairs;
    const int k = back ? blockIdx.x - npairs : blockIdx.x;          // couples blocks k and k+1
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, q = lane >> 4;
    const int r0 = (blockIdx.y >> 2) * 64 + wave * 16, c0 = (blockIdx.y & 3) * 64;
    // Out[r][c] = sum_j A[r][j] * B(j, c);  forward: A = Linv_{k+1}, B(j,c) = L(j,c);
    // backward: A = Linv_k^T (dinvt), B(j,c) = L(c,j)   with L = block (k+1,k)
    const double *__restrict__ A = (back ? dinvt + (long long)k * NBLK * NBLK
                                         : dinv + (long long)(k + 1) * NBLK * NBLK) + (long long)(r0 + l15) * NBLK + q;
    const double *__restrict__ L = ab + (long long)(k + 1) * NBLK + (long long)k * NBLK * lda;
    const long long bj = back ? lda : 1, bc = back ? 1 : lda;      // address of B(j,c) = L + j*bj + c*bc
    const double *__restrict__ B = L + (long long)q * bj + (long long)(c0 + l15) * bc;
    d4_t acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = (d4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int j0 = 0; j0 < NBLK; j0 += 4) {
        const double a = A[j0];
        double bv[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) bv[n] = B[(long long)j0 * bj + (long long)(16 * n) * bc];
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv[n], acc[n], 0, 0, 0);
    }
    double *__restrict__ O = (back ? mbwd : mfwd) + (long long)k * NBLK * NBLK;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int v = 0; v < 4; ++v) O[(long long)(r0 + q + 4 * v) * NBLK + c0 + 16 * n + l15] = acc[n][v];
}

// rows [r0, r0+R) of  out = T v - M y  for row-major 256x256 blocks T, M; one wave, R rows
template <int R>
__device__ inline void coupled_rows(const double *__restrict__ T, const double *__restrict__ M,
                                    const double *__restrict__ v, const double *__restrict__ y,
                                    double *__restrict__ out, int r0, int lane)
{
    double vv[4], yy[4], s[R];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        vv[u] = v[lane + 64 * u];
        yy[u] = y[lane + 64 * u];
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a += T[(r0 + i) * NBLK + lane + 64 * u] * vv[u];
            b += M[(r0 + i) * NBLK + lane + 64 * u] * yy[u];
        }
        s[i] = a - b;
    }
#pragma unroll
    for (int i = 0; i < R; ++i) s[i] = wave_sum(s[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < R; ++i) out[r0 + i] = s[i];
    }
}

// Forward step k: workgroups 0..15 solve block k+1 (y1 = Linv1 v1 - Mk yk), the others apply
// the panel update of y_k to the rows from block k+2 on (as fwd_update_kernel).
__global__ void __launch_bounds__(512)
fwd_step_kernel(const double *__restrict__ Linv1, const double *__restrict__ Mk,
                const double *__restrict__ Lpanel2, long long lda, const double *__restrict__ yk,
                const double *__restrict__ v1, double *__restrict__ y1, double *__restrict__ vbelow2)
{
    __shared__ double sy[NBLK];
    __shared__ double part[16][64];
    const int tid = threadIdx.x;
    if (blockIdx.x < 16) {
        coupled_rows<2>(Linv1, Mk, v1, yk, y1, blockIdx.x * 16 + (tid >> 6) * 2, tid & 63);
        return;
    }
    const int wg = blockIdx.x - 16;
    if (tid < NBLK) sy[tid] = yk[tid];
    __syncthreads();
    const int rp = tid & 31, cg = tid >> 5;
    const int r = wg * 64 + 2 * rp;
    const double *__restrict__ Lr = Lpanel2 + r + (long long)(cg * 16) * lda;
    d2_t l[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) l[c] = *reinterpret_cast<const d2_t *>(Lr + (long long)c * lda);
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const double yv = sy[cg * 16 + c];
        s0 += l[c][0] * yv;
        s1 += l[c][1] * yv;
    }
    part[cg][2 * rp] = s0;
    part[cg][2 * rp + 1] = s1;
    __syncthreads();
    if (tid < 64) {
        double s = 0.0;
#pragma unroll
        for (int g = 0; g < 16; ++g) s += part[g][tid];
        vbelow2[wg * 64 + tid] -= s;
    }
}

// Backward step k: workgroups 0..15 solve block k-1 (x0 = Linv0^T y0 - N x_k), the others apply
// L[block k rows, j]^T x_k to the columns left of block k-1 (as bwd_update_kernel).
__global__ void __launch_bounds__(256)
bwd_step_kernel(const double *__restrict__ Linvt0, const double *__restrict__ Nk,
                const double *__restrict__ Lrows, long long lda, const double *__restrict__ xk,
                const double *__restrict__ y0, double *__restrict__ x0, double *__restrict__ yleft)
{
    __shared__ double sx[NBLK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (blockIdx.x < 16) {
        coupled_rows<4>(Linvt0, Nk, y0, xk, x0, blockIdx.x * 16 + wave * 4, lane);
        return;
    }
    const int wg = blockIdx.x - 16;
    sx[tid] = xk[tid];
    __syncthreads();
    // lane = (column cs, row pair seg): rows 2 seg + 32 u (+1), so that the 16 lanes of a column read
    // 256 contiguous bytes per load
    const int seg = lane & 15, cs = lane >> 4;
    double xs[16];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        xs[2 * u] = sx[2 * seg + 32 * u];
        xs[2 * u + 1] = sx[2 * seg + 32 * u + 1];
    }
    d2_t l[4][8];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
        const int j = wg * 64 + wave * 16 + cc * 4 + cs;
        const double *__restrict__ Lc = Lrows + (long long)j * lda + 2 * seg;
#pragma unroll
        for (int u = 0; u < 8; ++u) l[cc][u] = *reinterpret_cast<const d2_t *>(Lc + 32 * u);
    }
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
        const int j = wg * 64 + wave * 16 + cc * 4 + cs;
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) s += l[cc][u][0] * xs[2 * u] + l[cc][u][1] * xs[2 * u + 1];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (seg == 0) yleft[j] -= s;
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
    // one pair of atomics per workgroup (a pair per wave of 1 024 workgroups queued 8 192 atomics on two words: 96 us for a
    // 2 MB vector, twice per fit; round 3)
    __shared__ double red[2][4];
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = mdx;
        red[1][threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        mdx = fmax(fmax(red[0][0], red[0][1]), fmax(red[0][2], red[0][3]));
        mx = fmax(fmax(red[1][0], red[1][1]), fmax(red[1][2], red[1][3]));
        // non-negative doubles order like their bit patterns; NaN (sign clear) sorts above inf
        atomicMax(&absmax2[0], (unsigned long long)__double_as_longlong(mdx));
        atomicMax(&absmax2[1], (unsigned long long)__double_as_longlong(mx));
    }
}


// ---------------------------------------------------------------------------
// Distributed band: block columns dealt to the ranks (GPUs) in chunks of `c` (DistMap, kernels.hpp).
// A rank stores only its own block columns, packed: block column J lives in local slot
// Jl = ((J / c) / R) c + J mod c.  With the dense-view addressing A(i,j) = ab[i + j lda] (lda = ld - 1)
// every block column is the same view shifted by dm_shift(J) = 256 ld (Jl - J) doubles, so the
// single-GPU kernels work on a rank's storage when they are handed ab + dm_shift(J).
//
// Trailing update of the block columns a rank owns by a panel that arrived in a buffer of its own
// (P, leading dimension ldp, first row = global row `row0`): same register-streaming form as
// syrk64_kernel (one wave = one 64x64 piece of C, accumulators initialised with the C tile, operands
// streamed in MFMA fragment shape).
__global__ void __launch_bounds__(64, 2)
syrk64d_kernel(double *__restrict__ abl, long long lda, DistMap dm, const double *__restrict__ P, long long ldp,
               int row0, int jb, int je, int re, int nitems)
{
    constexpr int SD = SYRK_SD;
    int it = (int)blockIdx.x;
    if (it >= nitems) return;
    // item -> (J, tj, ti): owned block columns J in [jb, je), their four 64-wide tile columns, rows tj..re-1
    const int J0 = row0 / NBLK;
    int J = jb, tj = 0, ti = 0;
    bool found = false;
    for (; J < je && !found; ++J) {
        if (!dm_owned(dm, J)) continue;
#pragma unroll 1
        for (int t = 0; t < 4; ++t) {
            const int tjj = 4 * (J - J0) + t;
            const int cnt = re - tjj;
            if (cnt <= 0) continue;
            if (it < cnt) { tj = tjj; ti = tjj + it; found = true; break; }
            it -= cnt;
        }
        if (found) break;
    }
    if (!found) return;
    const bool diag = (ti == tj);
    const int lane = threadIdx.x & 63, l15 = lane & 15, q = lane >> 4;
    const double *__restrict__ pJ = P + (long long)(tj * 64 + l15) + (long long)q * ldp;
    const double *__restrict__ pI = P + (long long)(ti * 64 + l15) + (long long)q * ldp;
    double *__restrict__ C = abl + dm_shift(dm, J) + (long long)(row0 + ti * 64) + (long long)(row0 + tj * 64) * lda;
    d4_t acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int v = 0; v < 4; ++v)       // -C + P P^T, stored negated (see syrk64_kernel)
                acc[m][n][v] = -__builtin_nontemporal_load(&C[(n * 16 + l15) + (long long)(m * 16 + q + 4 * v) * lda]);
    double qa[SD][4], qb[SD][4];
    auto fetch = [&](int slot, int step) {
        const long long off = (long long)(4 * step) * ldp;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            qa[slot][m] = pJ[off + 16 * m];
            qb[slot][m] = pI[off + 16 * m];
        }
    };
#pragma unroll
    for (int d = 0; d < SD; ++d) fetch(d, d);
    constexpr int NSTEP = NBLK / 4;
#pragma unroll 1
    for (int ks = 0; ks < NSTEP; ks += SD) {
#pragma unroll
        for (int d = 0; d < SD; ++d) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[d][m], qb[d][n], acc[m][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);          // the refill stays SD k-steps ahead of its use (see syrk64_kernel)
            if (ks + d + SD < NSTEP) fetch(d, ks + d + SD);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int r = n * 16 + l15;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int c = m * 16 + q + 4 * v;
                if (!diag || r >= c) __builtin_nontemporal_store(-acc[m][n][v], &C[r + (long long)c * lda]);
            }
        }
}

#ifndef SYRK32_SD
#define SYRK32_SD 8
#endif
// The update of ONE 256x256 diagonal block by the panel (the "topA" launch on the chain of the look-ahead
// pipeline) in 32x32 pieces: 36 waves of a quarter of the work each instead of 10 waves of 64x64 -- the
// launch is on the critical path of chain-bound (narrow-band) factorisations, where it ran 47-85 us for
// 21 MFlop.  Same operand streaming as syrk64_kernel, 2x2 accumulator tiles.
__global__ void __launch_bounds__(64)
syrk32_kernel(double *__restrict__ ab, long long lda, int k0, int row0, int nt)
{
    constexpr int SD = SYRK32_SD;            // 8 k-steps (4 MFMAs each) in flight: one wave per SIMD has to cover a memory round trip
    int it = (int)blockIdx.x, tj = 0;
    while (it >= nt - tj) { it -= nt - tj; ++tj; }
    const int ti = tj + it;
    if (tj >= nt) return;
    const bool diag = (ti == tj);
    const int lane = threadIdx.x & 63, l15 = lane & 15, q = lane >> 4;
    const double *__restrict__ pJ = ab + (long long)(row0 + tj * 32 + l15) + (long long)(k0 + q) * lda;
    const double *__restrict__ pI = ab + (long long)(row0 + ti * 32 + l15) + (long long)(k0 + q) * lda;
    double *__restrict__ C = ab + (long long)(row0 + ti * 32) + (long long)(row0 + tj * 32) * lda;
    d4_t acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[m][n][v] = -C[(n * 16 + l15) + (long long)(m * 16 + q + 4 * v) * lda];
    double qa[SD][2], qb[SD][2];
    auto fetch = [&](int slot, int step) {
        const long long off = (long long)(4 * step) * lda;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            qa[slot][m] = pJ[off + 16 * m];
            qb[slot][m] = pI[off + 16 * m];
        }
    };
#pragma unroll
    for (int d = 0; d < SD; ++d) fetch(d, d);
    constexpr int NSTEP = NBLK / 4;
#pragma unroll 1
    for (int ks = 0; ks < NSTEP; ks += SD) {
#pragma unroll
        for (int d = 0; d < SD; ++d) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[d][m], qb[d][n], acc[m][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + d + SD < NSTEP) fetch(d, ks + d + SD);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int r = n * 16 + l15;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int c = m * 16 + q + 4 * v;
                if (!diag || r >= c) C[r + (long long)c * lda] = -acc[m][n][v];
            }
        }
}

// dst[r + c nrows] = src[r + c lda]: the solved panel, packed for the transfer to the other ranks
__global__ void __launch_bounds__(256)
pack_panel_kernel(const double *__restrict__ src, long long lda, double *__restrict__ dst, int nrows)
{
    const int c = blockIdx.y;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < nrows; r += gridDim.x * 256)
        dst[(long long)c * nrows + r] = src[r + (long long)c * lda];
}

// backward sweep of the distributed band, panel part:  part[split][c] = sum_r L[r, c] x[r]  over the rows of
// this split (rows below the diagonal block of column block k; L = the local panel in place).
// grid (4, nsplit): 64 columns per workgroup; lane = (column, 16-row segment) as in bwd_update_kernel.
__global__ void __launch_bounds__(256)
panel_tdot_kernel(const double *__restrict__ Lp, long long lda, const double *__restrict__ x, int nrows,
                  int rows_per_split, double *__restrict__ part)
{
    __shared__ double sx[NBLK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int seg = lane & 15, cs = lane >> 4;
    const int rbeg = blockIdx.y * rows_per_split;
    const int rend = (rbeg + rows_per_split < nrows) ? rbeg + rows_per_split : nrows;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int r0 = rbeg; r0 < rend; r0 += NBLK) {       // nrows and rows_per_split are multiples of 256
        __syncthreads();
        sx[tid] = x[r0 + tid];
        __syncthreads();
        double xs[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) xs[i] = sx[seg * 16 + i];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const int j = blockIdx.x * 64 + wave * 16 + cc * 4 + cs;
            const double *__restrict__ Lc = Lp + (long long)j * lda + r0 + seg * 16;
            double t = 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const d2_t l = *reinterpret_cast<const d2_t *>(Lc + 2 * u);
                t += l[0] * xs[2 * u] + l[1] * xs[2 * u + 1];
            }
            s[cc] += t;
        }
    }
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
        double t = s[cc];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        const int j = blockIdx.x * 64 + wave * 16 + cc * 4 + cs;
        if (seg == 0) part[(long long)blockIdx.y * NBLK + j] = t;
    }
}

// x_k = Linv_k^T (y_k - sum_split part[split])   (dinvt row-major = Linv^T); grid 16 x 256 threads
__global__ void __launch_bounds__(256)
bwd_block_kernel(const double *__restrict__ Mt, const double *__restrict__ yk, const double *__restrict__ part,
                 int nsplit, double *__restrict__ xk)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * 16 + wave * 4;
    double vv[4], s[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        double v = yk[lane + 64 * u];
        for (int sp = 0; sp < nsplit; ++sp) v -= part[(long long)sp * NBLK + lane + 64 * u];
        vv[u] = v;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s[i] = 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) s[i] += Mt[(r0 + i) * NBLK + lane + 64 * u] * vv[u];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i] = wave_sum(s[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) xk[r0 + i] = s[i];
    }
}

// dst[i] += src[i]   /   dst[i] = owned(block of i) ? dst[i] : 0
__global__ void __launch_bounds__(256)
vec_add_kernel(long long n, double *__restrict__ dst, const double *__restrict__ src)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] += src[i];
}
__global__ void __launch_bounds__(256)
mask_owned_kernel(int n, DistMap dm, double *__restrict__ x)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !dm_owned(dm, i / NBLK)) x[i] = 0.0;
}

// absmax2[0] = max |a[i]|, absmax2[1] = max |b[i]|
__global__ void __launch_bounds__(256)
absmax2_kernel(int n, const double *__restrict__ a, const double *__restrict__ b,
               unsigned long long *__restrict__ absmax2)
{
    double ma = 0.0, mb = 0.0;
    const int stride = gridDim.x * blockDim.x;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        ma = fmax(ma, fabs(a[i]));
        mb = fmax(mb, fabs(b[i]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ma = fmax(ma, __shfl_xor(ma, o, 64));
        mb = fmax(mb, __shfl_xor(mb, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&absmax2[0], (unsigned long long)__double_as_longlong(ma));
        atomicMax(&absmax2[1], (unsigned long long)__double_as_longlong(mb));
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

// Streams/events of the look-ahead pipeline (created once per process and device).
namespace {
struct Pipeline {
    hipStream_t panel = nullptr, col = nullptr, res = nullptr, upd = nullptr;
    std::vector<hipEvent_t> evP, evU, evC, evI, evT;
    hipEvent_t evR[2] = {nullptr, nullptr};
    unsigned reserved = ~0u;      // CU id (my_cu_id) left to the panel factorisation, ~0u = none
    int *queues = nullptr;        // [4*nblk+8][2] item queues of the trailing-update launches
    int nqueues = 0;
    int dev = -1;
    hipEvent_t done = nullptr;    // end of the previous factorisation that used this pipeline
    bool used = false;
    std::vector<hipEvent_t> evA;  // start events of the timed bulk launches (kernel timing only)
    hipEvent_t f0 = nullptr, f1 = nullptr;
    bool res_tried = false;       // the CU-masked stream is created on the first factorisation that pins potrf
};
void pipeline_release(Pipeline &p);
// wide = the four-stream form (column and bulk streams, optionally the CU-masked one); a narrow-band chain only
// uses `panel` beside the caller's stream.  Every stream is a hardware queue, and two event-coupled pipelines
// of four streams each do not run side by side (measured un-profiled at 32^3: 28.6 ms against 21.3 ms for two
// two-stream chains and 35.6 ms for one chain), so nothing is created unused.
Pipeline &pipeline(void *&slot, int nblk, bool wide, bool want_res)
{
    if (!slot) slot = new Pipeline();
    Pipeline &p = *static_cast<Pipeline *>(slot);
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (p.panel == nullptr || p.dev != dev) {
        if (p.panel != nullptr) pipeline_release(p);          // the owner moved to another device
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);      // hi = numerically lowest = highest priority
        (void)hipStreamCreateWithPriority(&p.panel, hipStreamNonBlocking, hi);
        p.reserved = ~0u;
        p.res_tried = false;
        p.dev = dev;
        p.evP.clear();
        p.evU.clear();
        p.evC.clear();
        p.evI.clear();
        p.evT.clear();
        (void)hipEventCreateWithFlags(&p.done, hipEventDisableTiming);
        p.used = false;
        (void)hipEventCreateWithFlags(&p.evR[0], hipEventDisableTiming);
        (void)hipEventCreateWithFlags(&p.evR[1], hipEventDisableTiming);
    }
    if (p.nqueues < 4 * nblk + 8) {        // 4 queued launches per step: topA, topB, colU, bulk
        if (p.queues) (void)hipFree(p.queues);
        p.nqueues = 4 * nblk + 8;
        (void)hipMalloc(&p.queues, sizeof(int) * 2 * (size_t)p.nqueues);
    }
    while ((int)p.evP.size() < nblk + 1) {
        hipEvent_t e;
        (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
        p.evP.push_back(e);
        (void)hipEventCreate(&e);                 // evU[k] is also the stop event of bulk launch k (timing statistics)
        p.evU.push_back(e);
        (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
        p.evC.push_back(e);
        (void)hipEventCreate(&e);                 // evI[k], evT[k]: completion (stop event) of potrf(k) / the top panel solve
        p.evI.push_back(e);
        (void)hipEventCreate(&e);
        p.evT.push_back(e);
    }
    if (wide && !p.col) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        (void)hipStreamCreateWithPriority(&p.col, hipStreamNonBlocking, hi);
        // the bulk runs on a private non-blocking stream: the caller's stream may be the legacy
        // NULL stream, which would serialise against the (blocking) CU-masked stream below
        (void)hipStreamCreateWithFlags(&p.upd, hipStreamNonBlocking);
    }
    if (want_res && !p.res_tried) {
        p.res_tried = true;
        // One CU is left to the diagonal-block factorisation: v_mfma_f64 runs on the same f64
        // pipes as f64 VALU code, so the latency-bound potrf workgroup ran 3x slower beside
        // trailing-update waves.  potrf is pinned to that CU through a CU-masked stream; the
        // trailing-update waves are NOT masked (masked queues cost them ~8 %): a wave that finds
        // itself on the reserved CU steps aside (syrk64_kernel).  The CU's id is read back once.
        // (Only created for bands wide enough to use it: every stream is a hardware queue, and the
        // fewer of them a two-ended factorisation holds, the better its two chains overlap.)
        if (!splpak::opt_get("SPLPAK_NO_PANEL_CU")) {
            hipDeviceProp_t prop;
            (void)hipGetDeviceProperties(&prop, dev);
            const int ncu = prop.multiProcessorCount;
            std::vector<uint32_t> only((size_t)(ncu + 31) / 32, 0u);
            only[0] = 1u;
            unsigned *d = nullptr, h = ~0u;
            if (hipExtStreamCreateWithCUMask(&p.res, (uint32_t)only.size(), only.data()) == hipSuccess &&
                hipMalloc(&d, sizeof(unsigned)) == hipSuccess) {
                hipLaunchKernelGGL(whoami_kernel, dim3(1), dim3(64), 0, p.res, d);
                if (hipStreamSynchronize(p.res) == hipSuccess &&
                    hipMemcpy(&h, d, sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess)
                    p.reserved = h;
            }
            if (d) (void)hipFree(d);
            if (p.reserved == ~0u) { (void)hipGetLastError(); p.res = nullptr; }
        }
    }
    return p;
}
}  // namespace

namespace {
void pipeline_release(Pipeline &p)
{
    for (auto *v : {&p.evP, &p.evU, &p.evC, &p.evI, &p.evT, &p.evA}) {
        for (hipEvent_t e : *v) (void)hipEventDestroy(e);
        v->clear();
    }
    for (hipEvent_t &e : p.evR) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    for (hipEvent_t *e : {&p.done, &p.f0, &p.f1}) { if (*e) (void)hipEventDestroy(*e); *e = nullptr; }
    p.used = false;
    for (hipStream_t *s : {&p.panel, &p.col, &p.res, &p.upd}) { if (*s) (void)hipStreamDestroy(*s); *s = nullptr; }
    if (p.queues) (void)hipFree(p.queues);
    p.queues = nullptr;
    p.nqueues = 0;
    p.dev = -1;
}
}  // namespace

// The look-ahead pipeline (streams, events, item queues) belongs to whoever owns the Band (a plan):
// created on the first factorisation, released with it.
void band_pipeline_destroy(void *slot)
{
    if (!slot) return;
    Pipeline *p = static_cast<Pipeline *>(slot);
    pipeline_release(*p);
    delete p;
}

// Right-looking factorisation with one block column of look-ahead:
//   panel stream : [update of block column k+1 by panel k] -> potrf(k+1) -> trsm(k+1)
//   update stream: bulk trailing update by panel k (block columns >= k+2)
// so the latency-bound panel work runs beside the MFMA-bound bulk update.
// Block columns [kbeg, kend) are eliminated (default: all of them): afterwards the columns from kend on carry
// every update of the eliminated ones but are not factored -- a later call continues there, or the caller adds
// another Schur complement first (the two-ended factorisation of twoend.hip).  nfinish = number of leading block
// columns whose inverses / sweep coupling blocks are produced at the end (default: all; 0 = none yet).
hipError_t band_cholesky(const Band &b, int *info_dev, double *minpiv_dev, hipStream_t st,
                         CholStats *stats, int kbeg, int kend, int nfinish)
{
    if (kend < 0 || kend > b.nblk) kend = b.nblk;
    if (nfinish < 0 || nfinish > b.nblk) nfinish = b.nblk;
    if (kbeg < 0 || kbeg >= kend) return hipErrorInvalidValue;
    const bool partial = kend < b.nblk;
    {   // narrow bands: the two-stream form below (SPLPAK_NO_NARROW keeps the four-stream pipeline, for comparison)
        if (b.bw < narrow_band_limit() && !splpak::opt_get("SPLPAK_NO_NARROW")) {
            if (stats) *stats = CholStats{stats->enabled};
            return band_cholesky_narrow(b, info_dev, minpiv_dev, st, kbeg, kend, nfinish);
        }
    }
    const bool timing = stats && stats->enabled;
    const auto t_enq = std::chrono::steady_clock::now();
    // potrf is pinned to the reserved CU (own stream, two event hops per step) when the trailing update is
    // heavy enough to starve it; with a narrow band the step is bound by the chain itself and the hops cost more
    const int pin_bw = splpak::opt_get("SPLPAK_PIN_BW") ? atoi(splpak::opt_get("SPLPAK_PIN_BW")) : 24;
    Pipeline &pl = pipeline(b.pipe, b.nblk, true, b.bw >= pin_bw);
    hipStream_t sP = pl.panel, sC = pl.col, sU = pl.upd;
    hipStream_t sR = (pl.res && b.bw >= pin_bw) ? pl.res : pl.panel;
    if (!sP || !sC || !sU || splpak::opt_get("SPLPAK_NO_LOOKAHEAD")) sP = sC = sU = sR = st;   // no streams / diagnostics: no overlap
    // the item queues and events belong to the pipeline, not to the caller's stream: a factorisation
    // enqueued from another stream must not clear them while the previous one is still running
    if (pl.used && pl.done) (void)hipStreamWaitEvent(st, pl.done, 0);
    (void)hipMemsetAsync(pl.queues, 0, sizeof(int) * 2 * (size_t)pl.nqueues, st);
    std::vector<hipEvent_t> evs;
    if (timing) {
        // the timing events live in the pipeline: nothing is created or destroyed inside a timed fit
        if (!pl.f0) { (void)hipEventCreate(&pl.f0); (void)hipEventCreate(&pl.f1); }
        while ((int)pl.evA.size() < b.nblk + 1) {
            hipEvent_t e;
            (void)hipEventCreate(&e);
            pl.evA.push_back(e);
        }
        stats->syrk_launches = stats->syrk_ms = stats->syrk_flop = stats->factor_ms = 0;
        stats->bulk_launches = stats->bulk_flop = 0;
        (void)hipEventRecord(pl.f0, st);
    }
    if (stats) stats->total_flop = stats->bulk_launches = stats->bulk_flop = 0;
    auto tb_of = [&](int k) { int t = b.nblk - 1 - k; return t > b.bw ? b.bw : t; };
    int qnext = 0;
    // trailing update of the 64-row-unit range cols [cb,ce) x rows [max(col,rb), re) by panel k
    // bulk = true: the launch that carries ~92 % of the flops; it is a separate template
    // instantiation (ABL bit 8, no functional difference) so that profilers list it under its own
    // name, and it alone feeds the roofline statistics
    const bool bulk_stop_event = splpak::opt_get("SPLPAK_NO_STOPEV") == nullptr;
    auto syrk = [&](hipStream_t s, int k, int cb, int ce, int rb, int re, bool bulk = false) {
        const int k0 = k * NBLK;
        long long items = 0;
        for (int c = cb; c < ce; ++c) items += (re - (c > rb ? c : rb)) > 0 ? re - (c > rb ? c : rb) : 0;
        if (items <= 0) return;
        hipEvent_t a = nullptr, c = nullptr;
        // Timed bulk launches carry their HIP events in the dispatch itself (hipExtLaunchKernelGGL:
        // start/stop are taken from the kernel's own dispatch packet), so timing adds no packet to
        // the stream; events recorded around the launch cost ~4 us each between two launches.
        const bool timed = timing && bulk && bulk_stop_event;
        if (timed) a = pl.evA[k];
        // the bulk launch's completion IS evU[k]: no separate event record behind it in the stream
        if (bulk && bulk_stop_event) c = pl.evU[k];
        const bool queued = pl.reserved != ~0u && qnext < pl.nqueues;
        const int margin = queued ? 512 : 0;
        int *queue = queued ? pl.queues + 2 * (qnext++) : nullptr;
        if (bulk)
            hipExtLaunchKernelGGL((syrk64_kernel<16, 1, false>), dim3((unsigned)items + margin), dim3(64), 0, s, a, c, 0,
                                  b.ab, b.lda, k0, k0 + NBLK, cb, ce, rb, re, (int)items, margin, pl.reserved, queue);
        else
            hipLaunchKernelGGL((syrk64_kernel<SYRK_SD, 2, true>), dim3((unsigned)items + margin), dim3(64), 0, s, b.ab,
                               b.lda, k0, k0 + NBLK, cb, ce, rb, re, (int)items, margin, pl.reserved, queue);
        if (timed) {
            evs.push_back(a);
            evs.push_back(c);
            stats->syrk_launches += 1;
            stats->syrk_flop += 2.0 * (double)items * 64 * 64 * NBLK;
        }
        if (stats) stats->total_flop += 2.0 * (double)items * 64 * 64 * NBLK;
        if (stats && bulk) {
            stats->bulk_launches += 1;
            stats->bulk_flop += 2.0 * (double)items * 64 * 64 * NBLK;
        }
    };
    const bool top32 = splpak::opt_get("SPLPAK_TOPA64") == nullptr;          // topA in 32x32 pieces (36 waves) unless asked otherwise
    auto potrf = [&](int k) {        // potrf(k) (+ the 16x16 leaf inverses) pinned to the reserved CU
        const int k0 = k * NBLK;
        if (sR != sP) {
            (void)hipEventRecord(pl.evR[0], sP);
            (void)hipStreamWaitEvent(sR, pl.evR[0], 0);
        }
        // the kernel's completion is evI[k] (stop event of the dispatch): no separate record packets
        hipExtLaunchKernelGGL(potrf_strip_kernel, dim3(1), dim3(256), 0, sR, nullptr, pl.evI[k], 0, b.ab, b.lda, k0,
                              info_dev, minpiv_dev, b.inv64 + (long long)k * 4 * 64 * 64);
        if (sR != sP) (void)hipStreamWaitEvent(sP, pl.evI[k], 0);
    };
    // panel solve of rows [r0, r1) below the diagonal block k
    auto trsm = [&](hipStream_t s, int k, int r0, int r1, hipEvent_t done = nullptr) {
        const int k0 = k * NBLK;
        if (r1 <= r0) {
            if (done) (void)hipEventRecord(done, s);
            return;
        }
        const double *Lk = b.ab + (long long)k0 + (long long)k0 * b.lda;
        double *Xk = b.ab + (long long)(k0 + NBLK + r0) + (long long)k0 * b.lda;
        const double *ik = b.inv64 + (long long)k * 4 * 64 * 64;
        hipExtLaunchKernelGGL(trsm_kernel, dim3((r1 - r0) / 16), dim3(64), 0, s, nullptr, done, 0, Lk, Xk, b.lda, ik, r1 - r0);
    };

    // Dependency structure per step k (X_k = solved panel k; block (I,J) = 256x256 block):
    //   chain  (sP/sR): topA(k)  update of block (k+1,k+1) by X_k          [needs X_k row k+1 = evT[k], bulk(k-1)]
    //                   potrf(k+1) (with its 16x16 leaf inverses)
    //                   trsm_top(k+1): block (k+2,k+1) -> X_{k+1} row k+2  [needs topB(k)]      -> evT[k+1]
    //   column (sC)   : topB(k)  update of block (k+2,k+1) by X_k          [needs all of X_k = evP[k], bulk(k-1)]
    //                   colU(k)  update of blocks (>=k+3, k+1)
    //                   trsm_rest(k+1): rows below block row k+2           [needs potrf(k+1)]   -> evP[k+1]
    //   bulk   (sU)   : block columns >= k+2 by X_k                        [needs evP[k]]       -> evU[k]
    // The only cycle is the chain (~topA + potrf + trsm of 256 rows); everything that
    // needs the whole panel hangs off it with a step of slack.
    (void)hipEventRecord(pl.evU[b.nblk], st);      // start after everything queued on the caller's stream
    (void)hipStreamWaitEvent(sP, pl.evU[b.nblk], 0);
    (void)hipStreamWaitEvent(sC, pl.evU[b.nblk], 0);
    if (sR != sP) (void)hipStreamWaitEvent(sR, pl.evU[b.nblk], 0);
    (void)hipStreamWaitEvent(sU, pl.evU[b.nblk], 0);
    potrf(kbeg);
    trsm(sP, kbeg, 0, tb_of(kbeg) * NBLK, pl.evT[kbeg]);
    (void)hipEventRecord(pl.evP[kbeg], sP);
    for (int k = kbeg; k < kend; ++k) {
        const int tb = tb_of(k);
        if (tb <= 0) continue;
        const bool next = k + 1 < kend;                 // block column k+1 is factored in this call
        const int n64 = tb * NBLK / 64;                 // trailing rows / columns in 64-row units
        const int nrows1 = tb_of(k + 1) * NBLK;         // rows below the diagonal block of panel k+1
        if (k > kbeg) {
            (void)hipStreamWaitEvent(sP, pl.evU[k - 1], 0);
            (void)hipStreamWaitEvent(sC, pl.evU[k - 1], 0);
        }
        if (top32)                                      // topA: block (k+1,k+1), in 36 32x32 pieces
            hipLaunchKernelGGL(syrk32_kernel, dim3(36), dim3(64), 0, sP, b.ab, b.lda, k * NBLK, k * NBLK + NBLK, 8);
        else syrk(sP, k, 0, 4, 0, 4);
        if (next) potrf(k + 1);
        (void)hipStreamWaitEvent(sC, pl.evP[k], 0);
        syrk(sC, k, 0, 4, 4, n64 < 8 ? n64 : 8);        // topB: block (k+2,k+1)
        (void)hipEventRecord(pl.evC[k], sC);
        syrk(sC, k, 0, 4, 8, n64);                      // colU: blocks (>=k+3, k+1)
        if (next) {
            (void)hipStreamWaitEvent(sP, pl.evC[k], 0);
            trsm(sP, k + 1, 0, nrows1 < NBLK ? nrows1 : NBLK, pl.evT[k + 1]);
            (void)hipStreamWaitEvent(sC, pl.evI[k + 1], 0);
            trsm(sC, k + 1, NBLK, nrows1);
            (void)hipStreamWaitEvent(sC, pl.evT[k + 1], 0);
            (void)hipEventRecord(pl.evP[k + 1], sC);
        }
        (void)hipStreamWaitEvent(sU, pl.evP[k], 0);
        syrk(sU, k, 4, n64, 0, n64, true);              // bulk: block columns >= k+2
        if (!bulk_stop_event) (void)hipEventRecord(pl.evU[k], sU);
    }
    if (partial) {                                 // the last step's chain and column pieces are not followed by a panel
        (void)hipEventRecord(pl.evT[kend], sP);
        (void)hipEventRecord(pl.evP[kend], sC);
        (void)hipStreamWaitEvent(sU, pl.evT[kend], 0);
        (void)hipStreamWaitEvent(sU, pl.evP[kend], 0);
    } else {
        (void)hipStreamWaitEvent(sU, pl.evP[b.nblk - 1], 0);
    }
    (void)hipEventRecord(pl.evC[b.nblk], sU);      // join: the caller's stream continues after the pipeline
    (void)hipStreamWaitEvent(st, pl.evC[b.nblk], 0);
    if (splpak::opt_get("SPLPAK_DEBUG"))
        std::fprintf(stderr, "[splpak] band_cholesky: host enqueue of %d steps took %.1f ms\n", b.nblk,
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enq).count());
    // (computing these block by block on a side stream beside a narrow-band chain was tried: the 256-thread
    // inversions slow the chain's own workgroups by as much as the two launches cost here - no gain at 2-D 64^2 / 32^3)
    if (nfinish > 0) {
        hipLaunchKernelGGL(trinv_kernel, dim3(NBLK / 16, nfinish), dim3(64), 0, st, (const double *)b.ab, b.lda,
                           (const double *)b.inv64, b.dinv, b.dinvt, DistMap{1, 0, 1, b.lda + 1}, (const int *)nullptr);
        // pair k couples blocks k and k+1; with nfinish < nblk the last pair has only its backward block valid
        // (N = Linv_k^T L_{k+1,k}^T needs block k alone), which is the one a sweep that is GIVEN block nfinish uses
        const int npairs = nfinish < b.nblk ? nfinish : b.nblk - 1;
        if (b.mfwd && b.mbwd && b.bw > 0 && npairs > 0)
            hipLaunchKernelGGL(sweepmat_kernel, dim3(2 * npairs, 16), dim3(256), 0, st, (const double *)b.ab, b.lda,
                               (const double *)b.dinv, (const double *)b.dinvt, b.mfwd, b.mbwd, npairs);
    }
    if (pl.done) {
        (void)hipEventRecord(pl.done, st);
        pl.used = true;
    }
    hipError_t err = hipGetLastError();
    if (timing) {
        (void)hipEventRecord(pl.f1, st);
        (void)hipEventSynchronize(pl.f1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, pl.f0, pl.f1);
        stats->factor_ms = ms;
        for (size_t i = 0; i + 1 < evs.size(); i += 2) {
            (void)hipEventElapsedTime(&ms, evs[i], evs[i + 1]);
            stats->syrk_ms += ms;
        }
    }
    return err;
}

// The same factorisation for NARROW bands (chain-bound: the trailing update of a step is a fraction of the
// potrf -> panel solve -> diagonal update chain), on two streams instead of four:
//   chain (own high-priority stream): topA(k) -> potrf(k+1) -> panel solve of block (k+2, k+1)
//   rest  (the caller's stream)     : ONE launch for every other tile panel k updates, then the panel solve of
//                                     the rows below block row k+2
// One launch instead of three for the updates: with the chip idle all of its items run at once (one item time,
// ~60 us), where topB / colU / bulk each cost that much one after the other.  Two streams instead of four:
// two such factorisations run side by side (twoend.hip) on four hardware queues in all, which is what it took
// for both chains to run at full speed (see pipeline()).  Same range arguments as band_cholesky.
hipError_t band_cholesky_narrow(const Band &b, int *info_dev, double *minpiv_dev, hipStream_t st, int kbeg, int kend,
                                int nfinish)
{
    if (kend < 0 || kend > b.nblk) kend = b.nblk;
    if (nfinish < 0 || nfinish > b.nblk) nfinish = b.nblk;
    if (kbeg < 0 || kbeg >= kend) return hipErrorInvalidValue;
    Pipeline &pl = pipeline(b.pipe, b.nblk, false, false);
    hipStream_t sP = pl.panel;
    if (!sP || splpak::opt_get("SPLPAK_NO_LOOKAHEAD")) sP = st;
    if (pl.used && pl.done) (void)hipStreamWaitEvent(st, pl.done, 0);
    auto tb_of = [&](int k) { int t = b.nblk - 1 - k; return t > b.bw ? b.bw : t; };
    auto potrf = [&](int k) {
        hipExtLaunchKernelGGL(potrf_strip_kernel, dim3(1), dim3(256), 0, sP, nullptr, pl.evI[k], 0, b.ab, b.lda, k * NBLK,
                              info_dev, minpiv_dev, b.inv64 + (long long)k * 4 * 64 * 64);
    };
    auto trsm = [&](hipStream_t s, int k, int r0, int r1, hipEvent_t done) {
        if (r1 <= r0) {
            if (done) (void)hipEventRecord(done, s);
            return;
        }
        const int k0 = k * NBLK;
        hipExtLaunchKernelGGL(trsm_kernel, dim3((r1 - r0) / 16), dim3(64), 0, s, nullptr, done, 0,
                              (const double *)(b.ab + (long long)k0 + (long long)k0 * b.lda),
                              b.ab + (long long)(k0 + NBLK + r0) + (long long)k0 * b.lda, b.lda,
                              (const double *)(b.inv64 + (long long)k * 4 * 64 * 64), r1 - r0);
    };
    (void)hipEventRecord(pl.evU[b.nblk], st);              // the chain starts after everything queued on the caller's stream
    if (sP != st) (void)hipStreamWaitEvent(sP, pl.evU[b.nblk], 0);
    if (b.bw <= 1) {
        // block tridiagonal (2-D grids up to 84 nodes in the fast dimension, every 1-D grid): the trailing window of a
        // step IS block (k+1,k+1), so the chain is all there is -- three plain launches per step in stream order,
        // no events and no hops to the other stream on the critical path
        auto potrf_plain = [&](int k) {
            hipLaunchKernelGGL(potrf_strip_kernel, dim3(1), dim3(256), 0, sP, b.ab, b.lda, k * NBLK, info_dev, minpiv_dev,
                               b.inv64 + (long long)k * 4 * 64 * 64);
        };
        auto trsm_plain = [&](int k) {
            const int k0 = k * NBLK, rows = tb_of(k) * NBLK;
            if (rows > 0)
                hipLaunchKernelGGL(trsm_kernel, dim3(rows / 16), dim3(64), 0, sP,
                                   (const double *)(b.ab + (long long)k0 + (long long)k0 * b.lda),
                                   b.ab + (long long)(k0 + NBLK) + (long long)k0 * b.lda, b.lda,
                                   (const double *)(b.inv64 + (long long)k * 4 * 64 * 64), rows);
        };
        potrf_plain(kbeg);
        trsm_plain(kbeg);
        for (int k = kbeg; k < kend; ++k) {
            if (tb_of(k) <= 0) continue;
            hipLaunchKernelGGL(syrk32_kernel, dim3(36), dim3(64), 0, sP, b.ab, b.lda, k * NBLK, k * NBLK + NBLK, 8);
            if (k + 1 < kend) {
                potrf_plain(k + 1);
                trsm_plain(k + 1);
            }
        }
    } else {
    potrf(kbeg);
    trsm(sP, kbeg, 0, tb_of(kbeg) * NBLK, pl.evT[kbeg]);    // the first panel is solved whole on the chain
    (void)hipEventRecord(pl.evP[kbeg], sP);
    for (int k = kbeg; k < kend; ++k) {
        const int tb = tb_of(k);
        if (tb <= 0) continue;
        const bool next = k + 1 < kend;
        const int n64 = tb * NBLK / 64;
        const int nrows1 = tb_of(k + 1) * NBLK;
        // chain: block (k+1,k+1) has the updates of the steps before k (evU[k-1], waited for ahead of the last panel solve)
        hipLaunchKernelGGL(syrk32_kernel, dim3(36), dim3(64), 0, sP, b.ab, b.lda, k * NBLK, k * NBLK + NBLK, 8);
        if (next) potrf(k + 1);
        // rest: every tile of the trailing window except block (k+1,k+1): columns [0, n64) x rows [max(col, 4), n64)
        if (sP != st) (void)hipStreamWaitEvent(st, pl.evP[k], 0);
        long long items = 0;
        for (int c = 0; c < n64; ++c) items += n64 - (c > 4 ? c : 4);
        if (items > 0)
            hipExtLaunchKernelGGL((syrk64_kernel<SYRK_SD, 2, true>), dim3((unsigned)items), dim3(64), 0, st, nullptr,
                                  pl.evU[k], 0, b.ab, b.lda, k * NBLK, k * NBLK + NBLK, 0, n64, 4, n64, (int)items, 0, ~0u,
                                  (int *)nullptr);
        else (void)hipEventRecord(pl.evU[k], st);
        if (sP != st) (void)hipStreamWaitEvent(sP, pl.evU[k], 0);
        if (next) {
            trsm(sP, k + 1, 0, nrows1 < NBLK ? nrows1 : NBLK, pl.evT[k + 1]);
            if (sP != st) (void)hipStreamWaitEvent(st, pl.evI[k + 1], 0);
            trsm(st, k + 1, NBLK, nrows1, nullptr);
            if (sP != st) (void)hipStreamWaitEvent(st, pl.evT[k + 1], 0);
            (void)hipEventRecord(pl.evP[k + 1], st);
        }
    }
    }
    if (sP != st) {                                         // join (the last step's diagonal update, when no panel followed)
        (void)hipEventRecord(pl.evR[0], sP);
        (void)hipStreamWaitEvent(st, pl.evR[0], 0);
    }
    if (nfinish > 0) {
        hipLaunchKernelGGL(trinv_kernel, dim3(NBLK / 16, nfinish), dim3(64), 0, st, (const double *)b.ab, b.lda,
                           (const double *)b.inv64, b.dinv, b.dinvt, DistMap{1, 0, 1, b.lda + 1}, (const int *)nullptr);
        const int npairs = nfinish < b.nblk ? nfinish : b.nblk - 1;
        if (b.mfwd && b.mbwd && b.bw > 0 && npairs > 0)
            hipLaunchKernelGGL(sweepmat_kernel, dim3(2 * npairs, 16), dim3(256), 0, st, (const double *)b.ab, b.lda,
                               (const double *)b.dinv, (const double *)b.dinvt, b.mfwd, b.mbwd, npairs);
    }
    if (pl.done) {
        (void)hipEventRecord(pl.done, st);
        pl.used = true;
    }
    return hipGetLastError();
}

hipError_t band_forward(const Band &b, double *x, double *tmp, int kb, int ke, hipStream_t st)
{
    if (ke > b.nblk) ke = b.nblk;
    if (kb < 0 || kb >= ke) return hipSuccess;
    const long long nb2 = (long long)NBLK * NBLK;
    auto tb_of = [&](int k) { int t = b.nblk - 1 - k; return t > b.bw ? b.bw : t; };
    const bool fused = b.mfwd && b.mbwd && b.bw > 0 && b.nblk > 1;
    if (fused) {
        // one launch per block step (coupling blocks M_k from sweepmat_kernel)
        hipLaunchKernelGGL(blockmv_kernel, dim3(16), dim3(256), 0, st, (const double *)(b.dinv + kb * nb2),
                           (const double *)(x + (long long)kb * NBLK), tmp + (long long)kb * NBLK);
        for (int k = kb; k + 1 < ke; ++k) {
            const int k0 = k * NBLK;
            const int nrows2 = (tb_of(k) - 1) * NBLK;
            hipLaunchKernelGGL(fwd_step_kernel, dim3(16 + nrows2 / 64), dim3(512), 0, st,
                               (const double *)(b.dinv + (k + 1) * nb2), (const double *)(b.mfwd + k * nb2),
                               (const double *)(b.ab + (long long)(k0 + 2 * NBLK) + (long long)k0 * b.lda), b.lda,
                               (const double *)(tmp + k0), (const double *)(x + k0 + NBLK), tmp + k0 + NBLK,
                               x + k0 + 2 * NBLK);
        }
        // the panel of the last solved block has not been applied to the rows below it (a fused step does that
        // together with the solve of the next block)
        const int k = ke - 1, k0 = k * NBLK, nrows = tb_of(k) * NBLK;
        if (ke < b.nblk && nrows > 0)
            hipLaunchKernelGGL(fwd_update_kernel, dim3(nrows / 64), dim3(512), 0, st,
                               (const double *)(b.ab + (long long)(k0 + NBLK) + (long long)k0 * b.lda), b.lda,
                               (const double *)(tmp + k0), x + k0 + NBLK, nrows);
        return hipGetLastError();
    }
    for (int k = kb; k < ke; ++k) {
        const int k0 = k * NBLK;
        const int nrows = tb_of(k) * NBLK;
        hipLaunchKernelGGL(blockmv_kernel, dim3(16), dim3(256), 0, st,
                           (const double *)(b.dinv + (long long)k * NBLK * NBLK), (const double *)(x + k0), tmp + k0);
        if (nrows > 0)
            hipLaunchKernelGGL(fwd_update_kernel, dim3(nrows / 64), dim3(512), 0, st,
                               (const double *)(b.ab + (long long)(k0 + NBLK) + (long long)k0 * b.lda), b.lda,
                               (const double *)(tmp + k0), x + k0 + NBLK, nrows);
    }
    return hipGetLastError();
}

hipError_t band_backward(const Band &b, double *x, double *tmp, int kgiven, hipStream_t st, int kstop, bool resume)
{
    const long long nb2 = (long long)NBLK * NBLK;
    const bool fused = b.mfwd && b.mbwd && b.bw > 0 && b.nblk > 1;
    if (kgiven > b.nblk) kgiven = b.nblk;
    if (kstop < 0) kstop = 0;
    // a solved block pushes its contributions L[block k rows, j]^T x_k into the right-hand sides to its left
    auto push = [&](int k) {
        const int k0 = k * NBLK;
        const int tb = k > b.bw ? b.bw : k;
        const int ncols = tb * NBLK, jbeg = k0 - ncols;
        if (ncols > 0)
            hipLaunchKernelGGL(bwd_update_kernel, dim3(ncols / 64), dim3(256), 0, st,
                               (const double *)(b.ab + (long long)k0 + (long long)jbeg * b.lda), b.lda,
                               (const double *)(x + k0), tmp + jbeg, ncols);
    };
    if (fused) {
        // step k solves block k-1 (coupled to x_k through N_{k-1}) and pushes x_k to the left of block k-1
        int kstart = kgiven;
        if (!resume) {
            if (kgiven >= b.nblk) {
                const int last = (b.nblk - 1) * NBLK;
                hipLaunchKernelGGL(blockmv_kernel, dim3(16), dim3(256), 0, st, (const double *)(b.dinvt + (b.nblk - 1) * nb2),
                                   (const double *)(tmp + last), x + last);
                kstart = b.nblk - 1;
            } else {
                for (int k = b.nblk - 1; k > kgiven; --k) push(k);     // the lowest given block is pushed by its step
            }
        }
        for (int k = kstart; k >= 1 && k - 1 >= kstop; --k) {
            const int k0 = k * NBLK;
            const int tb = k > b.bw ? b.bw : k;
            const int ncols2 = (tb - 1) * NBLK;
            const int jbeg = k0 - tb * NBLK;
            hipLaunchKernelGGL(bwd_step_kernel, dim3(16 + ncols2 / 64), dim3(256), 0, st,
                               (const double *)(b.dinvt + (k - 1) * nb2), (const double *)(b.mbwd + (k - 1) * nb2),
                               (const double *)(b.ab + (long long)k0 + (long long)jbeg * b.lda), b.lda,
                               (const double *)(x + k0), (const double *)(tmp + k0 - NBLK), x + k0 - NBLK, tmp + jbeg);
        }
        return hipGetLastError();
    }
    for (int k = resume ? kgiven - 1 : b.nblk - 1; k >= kstop; --k) {
        const int k0 = k * NBLK;
        if (k < kgiven)
            hipLaunchKernelGGL(blockmv_kernel, dim3(16), dim3(256), 0, st,
                               (const double *)(b.dinvt + (long long)k * NBLK * NBLK), (const double *)(tmp + k0), x + k0);
        push(k);
    }
    return hipGetLastError();
}

hipError_t band_solve(const Band &b, double *x, double *tmp, hipStream_t st)
{
    // forward: x -> tmp (y), the not-yet-solved part of x is updated in place;
    // backward: tmp -> x
    hipError_t e = band_forward(b, x, tmp, 0, b.nblk, st);
    if (e != hipSuccess) return e;
    return band_backward(b, x, tmp, b.nblk, st, 0, false);
}

hipError_t launch_axpy_absmax(int n, double *x, const double *dx, double *absmax2, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(absmax2, 0, 2 * sizeof(double), st);
    if (e != hipSuccess) return e;
    int blocks = (n + 255) / 256;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(axpy_absmax_kernel, dim3(blocks), dim3(256), 0, st, n, x, dx,
                       reinterpret_cast<unsigned long long *>(absmax2));
    return hipGetLastError();
}

// ---- launchers of the distributed-band pieces (dist.hip drives them) ----------------------------------
hipError_t launch_potrf_block(double *abJ, long long lda, int k0, int *info, double *minpiv, double *inv16, hipStream_t st)
{
    hipLaunchKernelGGL(potrf_strip_kernel, dim3(1), dim3(256), 0, st, abJ, lda, k0, info, minpiv, inv16);
    return hipGetLastError();
}

hipError_t launch_trsm_panel(const double *Lkk, double *X, long long lda, const double *inv16, int nrows, hipStream_t st)
{
    if (nrows <= 0) return hipSuccess;
    hipLaunchKernelGGL(trsm_kernel, dim3(nrows / 16), dim3(64), 0, st, Lkk, X, lda, inv16, nrows);
    return hipGetLastError();
}

hipError_t launch_pack_panel(const double *src, long long lda, double *dst, int nrows, hipStream_t st)
{
    if (nrows <= 0) return hipSuccess;
    int gx = (nrows + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(pack_panel_kernel, dim3(gx, NBLK), dim3(256), 0, st, src, lda, dst, nrows);
    return hipGetLastError();
}

long long syrk64d_items(const DistMap &dm, int row0, int jb, int je, int re)
{
    long long items = 0;
    const int J0 = row0 / NBLK;
    for (int J = jb; J < je; ++J) {
        if (!dm_owned(dm, J)) continue;
        for (int t = 0; t < 4; ++t) {
            const int cnt = re - (4 * (J - J0) + t);
            if (cnt > 0) items += cnt;
        }
    }
    return items;
}

hipError_t launch_syrk64d(double *abl, long long lda, const DistMap &dm, const double *P, long long ldp, int row0,
                          int jb, int je, int re, hipStream_t st)
{
    const long long items = syrk64d_items(dm, row0, jb, je, re);
    if (items <= 0) return hipSuccess;
    hipLaunchKernelGGL(syrk64d_kernel, dim3((unsigned)items), dim3(64), 0, st, abl, lda, dm, P, ldp, row0, jb, je, re, (int)items);
    return hipGetLastError();
}

hipError_t launch_trtri_owned(const double *abl, long long lda, const DistMap &dm, const int *blocks_dev, int nown,
                              const double *inv16, double *dinv, double *dinvt, hipStream_t st)
{
    if (nown <= 0) return hipSuccess;
    hipLaunchKernelGGL(trinv_kernel, dim3(NBLK / 16, nown), dim3(64), 0, st, abl, lda, inv16, dinv, dinvt, dm, blocks_dev);
    return hipGetLastError();
}

hipError_t launch_blockmv(const double *M, const double *v, double *out, hipStream_t st)
{
    hipLaunchKernelGGL(blockmv_kernel, dim3(16), dim3(256), 0, st, M, v, out);
    return hipGetLastError();
}

hipError_t launch_fwd_update(const double *Lpanel, long long lda, const double *yk, double *vbelow, int nrows, hipStream_t st)
{
    if (nrows <= 0) return hipSuccess;
    hipLaunchKernelGGL(fwd_update_kernel, dim3(nrows / 64), dim3(512), 0, st, Lpanel, lda, yk, vbelow, nrows);
    return hipGetLastError();
}

hipError_t launch_bwd_column(const double *Lpanel, long long lda, const double *xbelow, int nrows, const double *dinvt_k,
                             const double *yk, double *part, double *xk, hipStream_t st)
{
    int nsplit = 0;
    if (nrows > 0) {
        const int rows_per_split = 4 * NBLK;
        nsplit = (nrows + rows_per_split - 1) / rows_per_split;
        hipLaunchKernelGGL(panel_tdot_kernel, dim3(4, nsplit), dim3(256), 0, st, Lpanel, lda, xbelow, nrows, rows_per_split, part);
    }
    hipLaunchKernelGGL(bwd_block_kernel, dim3(16), dim3(256), 0, st, dinvt_k, yk, (const double *)part, nsplit, xk);
    return hipGetLastError();
}

hipError_t launch_vec_add(long long n, double *dst, const double *src, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(vec_add_kernel, dim3((unsigned)blocks), dim3(256), 0, st, n, dst, src);
    return hipGetLastError();
}

hipError_t launch_mask_owned(int n, const DistMap &dm, double *x, hipStream_t st)
{
    hipLaunchKernelGGL(mask_owned_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, dm, x);
    return hipGetLastError();
}

hipError_t launch_absmax2(int n, const double *a, const double *b, double *absmax2, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(absmax2, 0, 2 * sizeof(double), st);
    if (e != hipSuccess) return e;
    int blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(absmax2_kernel, dim3(blocks), dim3(256), 0, st, n, a, b,
                       reinterpret_cast<unsigned long long *>(absmax2));
    return hipGetLastError();
}

}  // namespace splpak
