// The rows of a 4-D fit applied to a vector, tile by tile (round 6): rho = A^T W (W y - W A x) - C^T C x, the pass that the
// refinement runs a few times per fit and the iterative solve (pcg.hip) runs once per iteration with y = 0.
//
// The cell-by-cell form of rounds 2-5 (assemble.hip: residual_cell4_kernel, constraint_dots_kernel, rho_gather_kernel) writes a
// 256-double share per cell (1.45 GB at 32^4) that a thread per node gathers back 8 bytes at a time from 256 different cells
// (every 64-byte sector fetched for one double: 5.0 ms), evaluates the constraint rows entry by entry (1.8 ms + most of the
// gather), and re-reads the cell's 256 coefficients from global memory for every cell: 10.3 ms per pass at 32^4 / 1e7 points
// (profiles/r06_c5_pcg_first_kernel_stats.csv).  Here:
//
//   data rows (src/splpak.F90:788-855)   a workgroup owns a TILE of 3 x 3 x 3 x 2 cells: the 6 x 6 x 6 x 5 coefficients it touches sit in LDS once,
//       each of its four waves takes every fourth cell (points of a cell in their sorted order, sixteen per trip: lanes = (point,
//       dimension) for the window tables, then both products of the trip -- the window of coefficients times the points' factor
//       tables and its transpose -- as small matrix products on the f64 matrix pipe), adds the cell's 256 shares into ITS OWN LDS
//       image of the tile's nodes, and the four images are added in a fixed order into the tile's partial sums: 104 MB instead of
//       1.45 GB, coalesced, no atomics -- the bits do not depend on the schedule.
//   constraint rows (:921-1046)   every row is a tensor product of tridiagonal node matrices (values / first / second derivative of
//       the three basis functions around a node; boundary nodes take the first derivative, :998) times a node weight: C x and
//       C^T (C x) are d + d passes of tridiagonal mode products over arrays that share their prefixes (3, 6, 10, 10 arrays
//       forward, 10, 6, 3, 1 back in 4-D) -- 0.5 GB of cache traffic instead of 81 entries x 10 rows x 4 table look-ups per node.
//       In a plan without a factorisation these passes run beside the tile kernel, on a stream of the operator's own.
//   gather   node i adds the <= 16 tile partials that hold it (tile order) and subtracts the constraint term.
#include "plan.hpp"
#include "basis.hpp"

#include <algorithm>
#include <cstring>
#include <vector>

namespace splpak {

namespace {

#ifndef SPLPAK_TILE_SHAPE
#define SPLPAK_TILE_SHAPE 3, 3, 3, 2         // measured at 32^4, 1e7 points (tile kernel + gather, us): 2,2,2,2 1484; 3,3,2,2 1402; 3,3,3,2 1362; 2,3,3,3 1409; 3,3,3,3 1572; 4,2,2,2 1878
#endif
constexpr int TCS[4] = {SPLPAK_TILE_SHAPE};                        // cells per tile, by dimension
constexpr int TBS[4] = {TCS[0] + 3, TCS[1] + 3, TCS[2] + 3, TCS[3] + 3};      // nodes per tile, by dimension
constexpr int TST[4] = {1, TBS[0], TBS[0] * TBS[1], TBS[0] * TBS[1] * TBS[2]};  // strides of the tile's node image
constexpr int TB4 = TBS[0] * TBS[1] * TBS[2] * TBS[3];              // nodes of a tile
constexpr int NCT = TCS[0] * TCS[1] * TCS[2] * TCS[3];              // cells of a tile
constexpr int PCHUNK = 16;               // points per trip of a wave
constexpr int TLD = 17;                  // 16 table values per point + 1 (bank spread)
typedef double d4_t __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------------------
// data rows
__global__ void __launch_bounds__(256)
rows4_tile_kernel(Grid g, int nt0, int nt1, int nt2, const int *__restrict__ offset, const double *__restrict__ xs,
                  const double *__restrict__ ys, const double *__restrict__ ws, long long cap, const double *__restrict__ xvec,
                  double *__restrict__ partial, int squared, double *__restrict__ esq)
{   // esq != NULL: also esq[tile] = the sum of the squared row residuals of the tile's points (the fit's `reserr`), in a fixed order
    // squared != 0: the DIAGONAL of A^T W^2 A instead -- sum over the points of w^2 b_c^2: the tables hold the squares, every point's
    // "residual" is w (the boxes of the iterative solve scale their data part by it, pcg.hip)
    __shared__ double pt[TB4];
    __shared__ double acc[4][TB4];
    __shared__ double tab[4][PCHUNK * TLD];
    __shared__ double swe[4][PCHUNK];
    __shared__ int cbeg[NCT], cend[NCT], clb[NCT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (known to be uniform: the walk over the cells below stays in scalar registers)
    int t = blockIdx.x;
    int cb[4];
    cb[0] = (t % nt0) * TCS[0]; t /= nt0;
    cb[1] = (t % nt1) * TCS[1]; t /= nt1;
    cb[2] = (t % nt2) * TCS[2]; t /= nt2;
    cb[3] = t * TCS[3];
    for (int idx = tid; idx < TB4; idx += 256) {
        int r = idx, col = 0;
        bool ok = true;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int nd = cb[d] + r % TBS[d];
            r /= TBS[d];
            ok = ok && nd < g.nodes[d];
            col += nd * g.colstride[d];
        }
        pt[idx] = ok ? xvec[col] : 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) acc[w][idx] = 0.0;
    }
    // the point ranges of the tile's cells, looked up once (a wave that asked for them cell by cell waited out two dependent global
    // round trips per cell -- with two workgroups per CU nothing hid them: 1.85 ms per pass at 32^4, twice the LDS-bound estimate)
    if (tid < NCT) {
        int r = tid, cell = 0, lb = 0, mul = 1;
        bool ok = true;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int a = r % TCS[d];
            r /= TCS[d];
            ok = ok && cb[d] + a < g.cells[d];
            cell += (cb[d] + a) * g.cellstride[d];
            lb += a * mul;
            mul *= TBS[d];
        }
        cbeg[tid] = ok ? offset[cell] : 0;
        cend[tid] = ok ? offset[cell + 1] : 0;
        clb[tid] = lb;
    }
    __syncthreads();
    double *__restrict__ mytab = tab[wave];
    double *__restrict__ mywe = swe[wave];
    double *__restrict__ myacc = acc[wave];
    const int pi = lane & 15, sl = lane >> 4;                              // staging: (point, dimension)
    const int l15 = lane & 15, g4 = lane >> 4;                             // the matrix products' lane coordinates
    // (the cells' ranges come out of LDS at uniform addresses: into scalar registers, so that the walk needs no lane masks)
    auto sld = [&](const int *q) { return __builtin_amdgcn_readfirstlane(*q); };
    auto next_cell = [&](int lc) { while (lc < NCT && sld(&cbeg[lc]) == sld(&cend[lc])) lc += 4; return lc; };
    // (Tried: the points' window tables kept from one pass to the next -- they depend on the points alone, 128 B per point read instead
    //  of ~half of the kernel's vector instructions -- : 0.274 -> 0.268 s per fit at 32^4 for 1.3 GB; not kept.  The kernel waits on the
    //  dependent matrix-pipe and LDS steps of a trip with three waves per SIMD, not on the issue of the tables alone.)
    // this lane's dimension of the grid, as dimension 0 of a copy: the table code below then reads registers, not the argument block
    Grid gl = g;
    gl.nodes[0] = g.nodes[sl]; gl.xmin[0] = g.xmin[sl]; gl.dx[0] = g.dx[sl]; gl.dxin[0] = g.dxin[sl];
    // the chunk after the current one is loaded while the current one is worked on  (two trips ahead, with the shorter trips of the
    // matrix-pipe form: measured again, no gain -- 0.276 against 0.272 s per fit)
    double xpre = 0.0, wpre = 0.0, ypre = 0.0;
    auto issue = [&](int lc, int p0) {
        if (lc >= NCT) return;
        if (p0 + pi < sld(&cend[lc])) {
            xpre = xs[(long long)sl * cap + p0 + pi];
            if (sl == 0) {
                wpre = ws[p0 + pi];
                ypre = ys ? ys[p0 + pi] : 0.0;
            }
        }
    };
    int lc = next_cell(wave);
    int p0 = lc < NCT ? sld(&cbeg[lc]) : 0;
    issue(lc, p0);
    d4_t racc = {0.0, 0.0, 0.0, 0.0};
    double esum = 0.0;
    double cw[4] = {0.0, 0.0, 0.0, 0.0};
    bool cell_new = true;
    while (lc < NCT) {
        const int end = sld(&cend[lc]), lbase = sld(&clb[lc]);
        const int np = end - p0 < PCHUNK ? end - p0 : PCHUNK;
        const double xcur = xpre, wcur = wpre, ycur = ypre;
        int lcn = lc, p0n = p0 + PCHUNK;
        if (p0n >= end) {
            lcn = next_cell(lc + 4);
            p0n = lcn < NCT ? sld(&cbeg[lcn]) : 0;
        }
        issue(lcn, p0n);
        {   // window tables: lane = (point pi, dimension sl)
            double b[4] = {0.0, 0.0, 0.0, 0.0};
            // (the closed forms of the evaluation kernels -- interior window / next to an end / general, basis.hpp -- : the general form
            //  alone was 360 of the ~650 instructions of a trip, and the trip is issue bound; same values to rounding)
            int form;
            if (pi < np) window_table_selected(gl, 0, xcur, b, form);
            if (squared) {
#pragma unroll
                for (int k = 0; k < 4; ++k) b[k] *= b[k];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) mytab[pi * TLD + 4 * sl + k] = b[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // Both products of a trip are small matrix products over the cell's 16 x 16 window of coefficients, W[m][n] with
        // m = k2 + 4 k3 and n = k0 + 4 k1 -- a point's row is v (x) u with u[n] = b0[k0] b1[k1], v[m] = b2[k2] b3[k3] -- and run on the
        // matrix pipe (v_mfma_f64_16x16x4_f64: A[row = lane & 15][k = lane >> 4], B[k = lane >> 4][col = lane & 15], result rows
        // (lane >> 4) + 4 r in register r).  The pipe has the vector unit's f64 rate on this chip; what it saves is the issue slots
        // around the multiply-adds: the vector form read two LDS operands and spent twelve further instructions per point on the
        // transposed product alone (650 wave instructions per trip of 16 points, the kernel was issue bound).
        if (cell_new) {                   // the window's entries this lane feeds the forward product with: W[m = l15][n = 4 s + g4]
            const double *__restrict__ px = pt + lbase + g4 + TST[2] * (l15 & 3) + TST[3] * (l15 >> 2);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) cw[s4] = px[TST[1] * s4];
        }
        {   // forward: T = W U, U[n][q] = u_q[n] (K = n: four steps), then s_q = sum_m v_q[m] T[m][q]
            const double *__restrict__ tb = mytab + l15 * TLD;
            const double ub = tb[g4];
            d4_t T = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) T = __builtin_amdgcn_mfma_f64_16x16x4f64(cw[s4], ub * tb[4 + s4], T, 0, 0, 0);
            // this lane: rows m = g4 + 4 r of point l15, i.e. k2 = g4, k3 = r
            double r3 = T[0] * tb[12];
            r3 = fma(T[1], tb[13], r3);
            r3 = fma(T[2], tb[14], r3);
            r3 = fma(T[3], tb[15], r3);
            const double part = tb[8 + g4] * r3;
            const double q0 = __shfl(part, l15, 64), q1 = __shfl(part, l15 + 16, 64), q2 = __shfl(part, l15 + 32, 64), q3 = __shfl(part, l15 + 48, 64);
            const double tsum = ((q0 + q1) + q2) + q3;
            if (g4 == 0) {
                double we = 0.0;
                if (l15 < np) {
                    const double e = squared ? wcur : wcur * ycur - wcur * tsum;      // row residual w y - (w b) . x  (y = 0: the rows as an operator)
                    we = wcur * e;
                    esum = fma(e, e, esum);
                }
                mywe[l15] = we;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        {   // transposed: G[m][n] += sum_q (we_q v_q[m]) u_q[n]  (K = the points, four per step, in their order)
            const int nsteps = (np + 3) >> 2;
            const double *__restrict__ tq0 = mytab + g4 * TLD;
            const double *__restrict__ we0 = mywe + g4;
            const int ia = 8 + (l15 & 3), ja = 12 + (l15 >> 2), ib = l15 & 3, jb = 4 + (l15 >> 2);
#pragma unroll
            for (int s4 = 0; s4 < PCHUNK / 4; ++s4) {      // (unrolled: the steps' LDS addresses differ by constants)
                if (s4 < nsteps) {
                    const double *__restrict__ tq = tq0 + 4 * s4 * TLD;
                    const double a = (we0[4 * s4] * tq[ia]) * tq[ja];
                    const double b = tq[ib] * tq[jb];
                    racc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, racc, 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        cell_new = lcn != lc;
        if (cell_new) {                   // the cell is done: its 256 shares (k0 = l15 & 3, k1 = l15 >> 2, k2 = g4, k3 = register) into this wave's image of the tile
            const int li = lbase + (l15 & 3) + TST[1] * (l15 >> 2) + TST[2] * g4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                myacc[li + TST[3] * j] += racc[j];
                racc[j] = 0.0;
            }
        }
        lc = lcn;
        p0 = p0n;
    }
    if (esq) {                            // lanes 0 .. 15 of every wave hold shares: a fixed tree over them, then the waves in order
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) esum += __shfl_down(esum, o, 64);
        if (lane == 0) swe[wave][0] = esum;
    }
    __syncthreads();
    if (esq && tid == 0) esq[blockIdx.x] = ((swe[0][0] + swe[1][0]) + swe[2][0]) + swe[3][0];
    double *__restrict__ out = partial + (long long)blockIdx.x * TB4;
    for (int idx = tid; idx < TB4; idx += 256) out[idx] = ((acc[0][idx] + acc[1][idx]) + acc[2][idx]) + acc[3][idx];
}

// rho[i] = sum of the tile partials that hold node i (tile order, dimension 0 fastest) - cterm[i]
template <bool REFADD>
__global__ void __launch_bounds__(256)
rows4_gather_kernel(Grid g, int nt0, int nt1, int nt2, int nt3, const double *__restrict__ partial, const double *__restrict__ cterm,
                    double *__restrict__ rho)
{
    const int node = blockIdx.x * 256 + threadIdx.x;
    if (node >= g.ncol) return;
    const int nt[4] = {nt0, nt1, nt2, nt3};
    int tlo[4], tcnt[4], in[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        in[d] = (node / g.colstride[d]) % g.nodes[d];
        // tiles t with TCS t <= in <= TCS t + TBS - 1
        int lo = in[d] - (TBS[d] - 1);
        lo = lo <= 0 ? 0 : (lo + TCS[d] - 1) / TCS[d];
        int hi = in[d] / TCS[d];
        if (hi > nt[d] - 1) hi = nt[d] - 1;
        tlo[d] = lo;
        tcnt[d] = hi - lo + 1;
    }
    double acc = 0.0;
    for (int e3 = 0; e3 < tcnt[3]; ++e3)
        for (int e2 = 0; e2 < tcnt[2]; ++e2)
            for (int e1 = 0; e1 < tcnt[1]; ++e1)
                for (int e0 = 0; e0 < tcnt[0]; ++e0) {
                    const int t0 = tlo[0] + e0, t1 = tlo[1] + e1, t2 = tlo[2] + e2, t3 = tlo[3] + e3;
                    const long long tile = ((long long)(t3 * nt2 + t2) * nt1 + t1) * nt0 + t0;
                    const int li = (in[0] - TCS[0] * t0) + TST[1] * (in[1] - TCS[1] * t1) + TST[2] * (in[2] - TCS[2] * t2) + TST[3] * (in[3] - TCS[3] * t3);
                    acc += partial[tile * TB4 + li];
                }
    if constexpr (REFADD) {                      // the histogram: kept in the caller's dimension order, on top of what is there
        int refnode = 0;
#pragma unroll
        for (int d = 0; d < 4; ++d) refnode += in[d] * g.refstride[d];
        rho[refnode] += acc;
    } else {
        rho[node] = cterm ? acc - cterm[node] : acc;
    }
}

// The nearest-node histogram of the sparse-area test (:886-907), tile by tile like the rows: a wave takes every fourth cell of
// the tile, lane = point computes the point's slot in its window (nearest_slot's arithmetic: the reference's per dimension), and
// the weights are added into the wave's image of the tile in the points' order.  A point so far outside the grid that its
// address is not a node of its window (the :899 quirk) is added to the histogram directly -- the one floating-point atomic of the
// assembly, as in the Gram kernels.
__global__ void __launch_bounds__(256)
rows4_hist_kernel(Grid g, int nt0, int nt1, int nt2, const int *__restrict__ offset, const double *__restrict__ xs,
                  const double *__restrict__ ws, long long cap, double *__restrict__ partial, double *__restrict__ hist)
{
    __shared__ double acc[4][TB4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int t = blockIdx.x;
    int cb[4];
    cb[0] = (t % nt0) * TCS[0]; t /= nt0;
    cb[1] = (t % nt1) * TCS[1]; t /= nt1;
    cb[2] = (t % nt2) * TCS[2]; t /= nt2;
    cb[3] = t * TCS[3];
    for (int idx = tid; idx < TB4; idx += 256)
#pragma unroll
        for (int w = 0; w < 4; ++w) acc[w][idx] = 0.0;
    __syncthreads();
    double *__restrict__ myacc = acc[wave];
    for (int lc = wave; lc < NCT; lc += 4) {
        int r = lc, cell = 0, lbase = 0, mul = 1;
        bool ok = true;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int a = r % TCS[d];
            r /= TCS[d];
            ok = ok && cb[d] + a < g.cells[d];
            cell += (cb[d] + a) * g.cellstride[d];
            lbase += a * mul;
            mul *= TBS[d];
        }
        if (!ok) continue;
        const int beg = offset[cell], end = offset[cell + 1];
        for (int p0 = beg; p0 < end; p0 += 64) {
            const int np = end - p0 < 64 ? end - p0 : 64;
            int li = -1;
            double wv = 0.0;
            if (lane < np) {
                double xv[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) xv[d] = xs[(long long)d * cap + p0 + lane];
                wv = ws[p0 + lane];
                // nearest_slot (assemble.hip): the node's place in the cell's window, or outside it
                bool inwin = true;
                int loc = 0, m6 = 1;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
#pragma clang fp contract(off)
                    const double tt = g.dxin[d] * (xv[d] - g.xmin[d]) + 0.5;
                    const int inidim = (tt >= 2.0e9) ? 2000000000 : (tt <= -2.0e9 ? -2000000000 : (int)tt);
                    int lo, hi;
                    const int l = inidim - window_start(g, d, xv[d], lo, hi);
                    inwin = inwin && inidim >= 0 && inidim <= g.nodes[d] - 1 && l >= 0 && l <= 3;
                    loc += l * m6;
                    m6 *= TBS[d];
                }
                if (inwin) li = lbase + loc;
                else {
                    double xr[MAXD] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int d = 0; d < 4; ++d) xr[g.perm[d]] = xv[d];
                    atomicAdd(&hist[nearest_node_address(g, xr)], wv);       // :905
                }
            }
            for (int q = 0; q < np; ++q) {                 // in the points' order: one owner, fixed order
                const int lq = __shfl(li, q, 64);
                const double wq = __shfl(wv, q, 64);
                if (lane == 0 && lq >= 0) myacc[lq] += wq;
            }
        }
    }
    __syncthreads();
    double *__restrict__ out = partial + (long long)blockIdx.x * TB4;
    for (int idx = tid; idx < TB4; idx += 256) out[idx] = ((acc[0][idx] + acc[1][idx]) + acc[2][idx]) + acc[3][idx];
}

__global__ void __launch_bounds__(256)
abs_kernel(long long n, const double *__restrict__ x, double *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = fabs(x[i]);
}

// den = |rhs| - nabs   (nabs = -(|A|^T W^2 |A| |x| + |C|^T |C| |x|): the residual pass's sign)
__global__ void __launch_bounds__(256)
den_kernel(long long n, const double *__restrict__ rhs, const double *__restrict__ nabs, double *__restrict__ den)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) den[i] = fabs(rhs[i]) - nabs[i];
}

// ---------------------------------------------------------------------------------------------------------------------------
// constraint rows as tridiagonal mode products
struct PassJob {            // one output array of a pass
    int nsrc;               // forward: 1
    int src[3];             // index of the source array in the input pool
    int ord[3];             // derivative order of the factor along the pass's dimension (2 = second derivative, first at the ends)
    int wsel;               // forward, last pass: 0 none, 1 weight dcw^2, 2 weight (2 dcw)^2 on the data-sparse nodes (0 elsewhere)
};
constexpr int MAXJOBS = 10;
struct PassTable {
    PassJob job[MAXJOBS];
};

// factor of basis function n + o at node n along dimension d, derivative order `ord` (2: the first derivative at the two ends, :998)
__device__ inline double node_factor(const Grid &g, const double *__restrict__ ctab, int base, int d, int n, int o, int ord)
{
    const int nder = (ord == 2 && (n == 0 || n == g.nodes[d] - 1)) ? 1 : ord;
    return ctab[base + (n * 3 + o + 1) * 3 + nder];
}

// forward: out_j[n] = w(n) sum_o F(n_k, o) in_src(j)[n + o e_k];  transposed: out_j[m] = sum_{sources} sum_o F(m_k + o, -o) in_s[m + o e_k]
template <bool TRANSPOSED, bool ABS>
__global__ void __launch_bounds__(256)
tri_pass_kernel(Grid g, int k, int ctbase, PassTable tabl, const double *__restrict__ ctab, const double *__restrict__ inpool,
                double *__restrict__ outpool, const double *__restrict__ dcw, const unsigned char *__restrict__ spf)
{
    const int node = blockIdx.x * 256 + threadIdx.x;
    if (node >= g.ncol) return;
    const PassJob &jb = tabl.job[blockIdx.y];
    const int nk = g.nodes[k], stride = g.colstride[k];
    const int ik = (node / stride) % nk;
    const long long ncol = g.ncol;
    double acc = 0.0;
    for (int s = 0; s < jb.nsrc; ++s) {
        const double *__restrict__ in = inpool + (long long)jb.src[s] * ncol;
        const int ord = jb.ord[s];
#pragma unroll
        for (int o = -1; o <= 1; ++o) {
            const int jk = ik + o;
            if (jk < 0 || jk > nk - 1) continue;
            double f = TRANSPOSED ? node_factor(g, ctab, ctbase, k, jk, -o, ord) : node_factor(g, ctab, ctbase, k, ik, o, ord);
            if (ABS) f = fabs(f);
            acc = fma(f, in[node + o * stride], acc);
        }
    }
    if (!TRANSPOSED && jb.wsel != 0) {
        double wgt = 0.0;
        if (spf[node]) {
            const double dc = jb.wsel == 1 ? dcw[node] : 2.0 * dcw[node];      // :983
            wgt = dc * dc;
        }
        acc *= wgt;
    }
    outpool[(long long)blockIdx.y * ncol + node] = acc;
}

// Two forward passes in one (dimensions k, k + 1): output j of the second pass straight from the input of the first --
//   out_j[n] = w(n) sum_{ob} F_{k+1}(n, ob) sum_{oa} F_k(n, oa) in[n + oa e_k + ob e_{k+1}]
// (nine cached reads per output instead of writing and re-reading the first pass's arrays: 8 launches per product -> 4)
template <bool ABS>
__global__ void __launch_bounds__(256)
tri_fwd2_kernel(Grid g, int k, int ctb0, int ctb1, PassTable ta, PassTable tb, const double *__restrict__ ctab, const double *__restrict__ inpool,
                double *__restrict__ outpool, const double *__restrict__ dcw, const unsigned char *__restrict__ spf)
{
    const int node = blockIdx.x * 256 + threadIdx.x;
    if (node >= g.ncol) return;
    const PassJob &jb = tb.job[blockIdx.y];
    const PassJob &ja = ta.job[jb.src[0]];
    const int na = g.nodes[k], sa = g.colstride[k], nb = g.nodes[k + 1], sb = g.colstride[k + 1];
    const int ia = (node / sa) % na, ib = (node / sb) % nb;
    const double *__restrict__ in = inpool + (long long)ja.src[0] * g.ncol;
    double fa[3], acc = 0.0;
#pragma unroll
    for (int o = -1; o <= 1; ++o) {
        const bool ok = ia + o >= 0 && ia + o <= na - 1;
        double f = ok ? node_factor(g, ctab, ctb0, k, ia, o, ja.ord[0]) : 0.0;
        fa[o + 1] = ABS ? fabs(f) : f;
    }
#pragma unroll
    for (int ob = -1; ob <= 1; ++ob) {
        if (ib + ob < 0 || ib + ob > nb - 1) continue;
        double f = node_factor(g, ctab, ctb1, k + 1, ib, ob, jb.ord[0]);
        if (ABS) f = fabs(f);
        const double *__restrict__ row = in + node + ob * sb;
        double u = 0.0;
#pragma unroll
        for (int oa = -1; oa <= 1; ++oa)
            if (fa[oa + 1] != 0.0) u = fma(fa[oa + 1], row[oa * sa], u);
        acc = fma(f, u, acc);
    }
    if (jb.wsel != 0) {
        double wgt = 0.0;
        if (spf[node]) {
            const double dc = jb.wsel == 1 ? dcw[node] : 2.0 * dcw[node];      // :983
            wgt = dc * dc;
        }
        acc *= wgt;
    }
    outpool[(long long)blockIdx.y * g.ncol + node] = acc;
}

// Two transposed passes in one (dimension k + 1, then k): output j of the pass over dimension k from the inputs of the pass over k + 1
template <bool ABS>
__global__ void __launch_bounds__(256)
tri_bwd2_kernel(Grid g, int k, int ctb0, int ctb1, PassTable t0, PassTable t1, const double *__restrict__ ctab, const double *__restrict__ inpool,
                double *__restrict__ outpool)
{
    const int node = blockIdx.x * 256 + threadIdx.x;
    if (node >= g.ncol) return;
    const PassJob &j0 = t0.job[blockIdx.y];
    const int n0 = g.nodes[k], s0 = g.colstride[k], n1 = g.nodes[k + 1], s1 = g.colstride[k + 1];
    const int i0 = (node / s0) % n0, i1 = (node / s1) % n1;
    const long long ncol = g.ncol;
    double acc = 0.0;
    for (int a = 0; a < j0.nsrc; ++a) {
        const PassJob &j1 = t1.job[j0.src[a]];
#pragma unroll
        for (int o0 = -1; o0 <= 1; ++o0) {
            const int m0 = i0 + o0;
            if (m0 < 0 || m0 > n0 - 1) continue;
            double f0 = node_factor(g, ctab, ctb0, k, m0, -o0, j0.ord[a]);
            if (ABS) f0 = fabs(f0);
            double inter = 0.0;             // the pass over dimension k + 1 at the node m + o0 e_k (same coordinate along k + 1)
            for (int c = 0; c < j1.nsrc; ++c) {
                const double *__restrict__ in = inpool + (long long)j1.src[c] * ncol + node + o0 * s0;
#pragma unroll
                for (int o1 = -1; o1 <= 1; ++o1) {
                    const int m1 = i1 + o1;
                    if (m1 < 0 || m1 > n1 - 1) continue;
                    double f1 = node_factor(g, ctab, ctb1, k + 1, m1, -o1, j1.ord[c]);
                    if (ABS) f1 = fabs(f1);
                    inter = fma(f1, in[o1 * s1], inter);
                }
            }
            acc = fma(f0, inter, acc);
        }
    }
    outpool[(long long)blockIdx.y * ncol + node] = acc;
}

}  // namespace

// share[node] = sum over the node's constraint rows of (row weight x row . x)^2 from the last forward pass's arrays, which hold
// (row weight)^2 (row . x): the constraint rows' part of the sum of squared residuals (`reserr`), node by node for a fixed-order sum
struct WeightSel { int w[MAXJOBS]; };
__global__ void __launch_bounds__(256)
cons_sq_kernel(long long ncol, int narr, WeightSel ws, const double *__restrict__ pool, const double *__restrict__ dcw,
               const unsigned char *__restrict__ spf, double *__restrict__ share)
{
    const long long node = (long long)blockIdx.x * 256 + threadIdx.x;
    if (node >= ncol) return;
    double acc = 0.0;
    if (spf[node]) {
        const double dc = dcw[node];
        for (int j = 0; j < narr; ++j) {
            const double rw = ws.w[j] == 1 ? dc : 2.0 * dc;
            const double v = rw != 0.0 ? pool[(long long)j * ncol + node] / rw : 0.0;
            acc = fma(v, v, acc);
        }
    }
    share[node] = acc;
}

struct RowsOp {
    int nt[4] = {1, 1, 1, 1};
    long long ntiles = 0;
    double *partial = nullptr;        // [ntiles][TB4]
    double *poolA = nullptr, *poolB = nullptr;      // [10][ncol] each
    int ctbase[MAXD] = {0, 0, 0, 0};
    // passes: forward k = 0 .. D-1, transposed k = D-1 .. 0
    PassTable fwd[MAXD], bwd[MAXD];
    int nfwd[MAXD] = {0, 0, 0, 0}, nbwd[MAXD] = {0, 0, 0, 0};
    size_t bytes = 0;
    bool no_pairs = false;            // A/B: the passes of the constraint rows one dimension at a time
    // the constraint rows' passes run BESIDE the data rows' tile kernel, on a stream of their own: the tile kernel is bound by the
    // issue of vector / matrix instructions and leaves wave slots and the memory system to them (0.26 of an iteration's 1.29 ms at
    // 32^4 were these passes behind it).  NULL: one stream (SPLPAK_ROWS_ONE_STREAM, or the stream could not be had)
    hipStream_t side = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    std::vector<void *> owned;
};

void rowsop_destroy(RowsOp *r)
{
    if (!r) return;
    if (r->side) { (void)hipStreamSynchronize(r->side); (void)hipStreamDestroy(r->side); }
    if (r->ev_in) (void)hipEventDestroy(r->ev_in);
    if (r->ev_out) (void)hipEventDestroy(r->ev_out);
    for (void *q : r->owned) (void)hipFree(q);
    delete r;
}

size_t rowsop_bytes(const RowsOp *r) { return r ? r->bytes : 0; }

// 4-D grids only; NULL for the others.  (A 3-D form of the tile kernel -- 64 window functions, a lane each in the transposed product --
// was built and measured in round 6: 22.4 ms of solves + refinement per fit at config 3 against 22.5 ms with the wave-per-cell pass of
// assemble.hip, and 59.6 against 47.9 ms at 1e8 points (440 points per cell: the wave-per-cell pass takes 64 points per trip, the
// tile kernel 16).  Removed again.)
int rowsop_create(const Grid &g, bool side_stream, RowsOp **out)
{   // side_stream: the plan has no factorisation.  (Beside a nested-dissection plan's streams -- one of them bound to eight reserved CUs --
    //  the operator's own stream made the iteration SLOWER: 181 against 112 ms per fit at 24^4; the runtime maps streams onto a few
    //  hardware queues, and a queue carries its CU mask.)
    *out = nullptr;
    const char *sw = splpak::opt_get("SPLPAK_ROWS_TILES");           // A/B switch: 0 = the cell-by-cell passes
    if (g.ndim != 4 || (sw && atoi(sw) == 0)) return 0;
    RowsOp *r = new RowsOp();
    // (the passes in pairs -- tri_fwd2 / tri_bwd2 -- were measured at 32^4: 144 + 157 us per product against 142 + 126 us one dimension at
    //  a time: nine cached reads per output cost what the saved arrays gain.  Kept behind a switch.)
    r->no_pairs = splpak::opt_get("SPLPAK_PCG_TRI_PAIRS") == nullptr;
    r->ntiles = 1;
    for (int d = 0; d < 4; ++d) {
        r->nt[d] = (g.cells[d] + TCS[d] - 1) / TCS[d];
        r->ntiles *= r->nt[d];
    }
    auto alloc = [&](double **q, size_t count) {
        void *v = nullptr;
        if (hipMalloc(&v, count * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); return false; }
        r->owned.push_back(v);
        r->bytes += count * sizeof(double);
        *q = static_cast<double *>(v);
        return true;
    };
    if (!alloc(&r->partial, (size_t)r->ntiles * TB4) || !alloc(&r->poolA, (size_t)MAXJOBS * g.ncol) || !alloc(&r->poolB, (size_t)MAXJOBS * g.ncol)) {
        rowsop_destroy(r);
        set_error("rows operator: device allocation failed");
        return SPLPAK_E_NOMEM;
    }
    if (side_stream && !splpak::opt_get("SPLPAK_ROWS_ONE_STREAM")) {
        if (hipStreamCreateWithFlags(&r->side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&r->ev_in, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&r->ev_out, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            if (r->side) (void)hipStreamDestroy(r->side);
            r->side = nullptr;
        }
    }
    int base = 0;
    for (int d = 0; d < g.ndim; ++d) { r->ctbase[d] = base; base += 9 * g.nodes[d]; }
    // The rows' patterns (derivative order per dimension): (i, i) -> 2 e_i, (i < j) -> e_i + e_j, in the reference's row order.
    const int D = g.ndim;
    std::vector<std::vector<int>> fin;
    std::vector<int> wsel;
    for (int i = 0; i < D; ++i)
        for (int j = i; j < D; ++j) {
            std::vector<int> a((size_t)D, 0);
            if (i == j) a[(size_t)i] = 2; else { a[(size_t)i] = 1; a[(size_t)j] = 1; }
            fin.push_back(a);
            wsel.push_back(i == j ? 1 : 2);
        }
    // prefixes[k] = distinct prefixes of length k + 1 (in order of first appearance); prefixes[D-1] = fin
    std::vector<std::vector<std::vector<int>>> pre((size_t)D);
    for (int k = 0; k < D; ++k)
        for (const auto &a : fin) {
            std::vector<int> p(a.begin(), a.begin() + k + 1);
            if (std::find(pre[(size_t)k].begin(), pre[(size_t)k].end(), p) == pre[(size_t)k].end()) pre[(size_t)k].push_back(p);
        }
    auto index_of = [&](int k, const std::vector<int> &p) {
        return (int)(std::find(pre[(size_t)k].begin(), pre[(size_t)k].end(), p) - pre[(size_t)k].begin());
    };
    for (int k = 0; k < D; ++k) {
        // forward pass k: output prefix (a_0 .. a_k) from its prefix (a_0 .. a_{k-1}) (k = 0: from x itself, source 0)
        r->nfwd[k] = (int)pre[(size_t)k].size();
        for (int j = 0; j < r->nfwd[k]; ++j) {
            const auto &p = pre[(size_t)k][(size_t)j];
            PassJob jb{};
            jb.nsrc = 1;
            jb.src[0] = k == 0 ? 0 : index_of(k - 1, std::vector<int>(p.begin(), p.end() - 1));
            jb.ord[0] = p.back();
            jb.wsel = 0;
            if (k == D - 1) jb.wsel = wsel[(size_t)j];       // (pre[D-1] is `fin` in its own order)
            r->fwd[k].job[j] = jb;
        }
        // transposed pass k: output prefix of length k (k = 0: the single result) from the arrays of length k + 1 that extend it
        r->nbwd[k] = k == 0 ? 1 : (int)pre[(size_t)k - 1].size();
        for (int j = 0; j < r->nbwd[k]; ++j) {
            PassJob jb{};
            jb.nsrc = 0;
            for (int s = 0; s < (int)pre[(size_t)k].size(); ++s) {
                const auto &p = pre[(size_t)k][(size_t)s];
                const bool ext = k == 0 || std::equal(p.begin(), p.end() - 1, pre[(size_t)k - 1][(size_t)j].begin());
                if (!ext) continue;
                jb.src[jb.nsrc] = s;
                jb.ord[jb.nsrc] = p.back();
                ++jb.nsrc;
            }
            r->bwd[k].job[j] = jb;
        }
    }
    *out = r;
    return 0;
}

// rho = A^T W (W y - W A x) [- C^T C x]   (rows.ys == NULL: y = 0)
static hipError_t rowsop_apply_t(const Grid &g, RowsOp *r, const SortScratch &rows, const double *xvec, const double *dcw, const unsigned char *spf,
                                 const double *ctab, bool constraints, bool abs_factors, double *rho, hipStream_t st,
                                 double *e2tiles = nullptr, double *e2nodes = nullptr)
{   // e2tiles [ntiles], e2nodes [ncol] (both or none): the shares of the sum of squared row residuals, data rows by tile, constraint rows by node
    hipStream_t st_rows = st;
    if (constraints && r->side) {             // (the vector is ready on `st`; the passes' arrays were last read by the previous gather, on `st` too)
        (void)hipEventRecord(r->ev_in, st_rows);
        (void)hipStreamWaitEvent(r->side, r->ev_in, 0);
    }
    hipLaunchKernelGGL(rows4_tile_kernel, dim3((unsigned)r->ntiles), dim3(256), 0, st_rows, g, r->nt[0], r->nt[1], r->nt[2], (const int *)rows.offset,
                       (const double *)rows.xs, (const double *)rows.ys, (const double *)rows.ws, rows.cap, xvec, r->partial, 0, e2tiles);
    const double *cterm = nullptr;
    if (constraints && r->side) st = r->side;
    if (constraints) {
        const int D = g.ndim;
        const dim3 bl(256);
        const unsigned gx = (unsigned)((g.ncol + 255) / 256);
        const double *in = xvec;
        double *pools[2] = {r->poolA, r->poolB};
        int which = 0;
        const bool pairs = D == 4 && !r->no_pairs;
        if (pairs) {
            for (int k = 0; k < D; k += 2) {
                if (abs_factors) hipLaunchKernelGGL(tri_fwd2_kernel<true>, dim3(gx, (unsigned)r->nfwd[k + 1]), bl, 0, st, g, k, r->ctbase[k], r->ctbase[k + 1], r->fwd[k], r->fwd[k + 1], ctab, in, pools[which], dcw, spf);
                else hipLaunchKernelGGL(tri_fwd2_kernel<false>, dim3(gx, (unsigned)r->nfwd[k + 1]), bl, 0, st, g, k, r->ctbase[k], r->ctbase[k + 1], r->fwd[k], r->fwd[k + 1], ctab, in, pools[which], dcw, spf);
                in = pools[which];
                which ^= 1;
            }
            for (int k = D - 2; k >= 0; k -= 2) {
                if (abs_factors) hipLaunchKernelGGL(tri_bwd2_kernel<true>, dim3(gx, (unsigned)r->nbwd[k]), bl, 0, st, g, k, r->ctbase[k], r->ctbase[k + 1], r->bwd[k], r->bwd[k + 1], ctab, in, pools[which]);
                else hipLaunchKernelGGL(tri_bwd2_kernel<false>, dim3(gx, (unsigned)r->nbwd[k]), bl, 0, st, g, k, r->ctbase[k], r->ctbase[k + 1], r->bwd[k], r->bwd[k + 1], ctab, in, pools[which]);
                in = pools[which];
                which ^= 1;
            }
        }
        for (int k = 0; k < D && !pairs; ++k) {
            if (abs_factors) hipLaunchKernelGGL((tri_pass_kernel<false, true>), dim3(gx, (unsigned)r->nfwd[k]), bl, 0, st, g, k, r->ctbase[k], r->fwd[k], ctab, in, pools[which], dcw, spf);
            else hipLaunchKernelGGL((tri_pass_kernel<false, false>), dim3(gx, (unsigned)r->nfwd[k]), bl, 0, st, g, k, r->ctbase[k], r->fwd[k], ctab, in, pools[which], dcw, spf);
            in = pools[which];
            which ^= 1;
        }
        for (int k = D - 1; k >= 0 && !pairs; --k) {
            if (k == D - 1 && e2nodes) {             // (between the forward and the transposed passes: `in` = the weighted row products)
                WeightSel ws{};
                for (int j = 0; j < r->nfwd[D - 1]; ++j) ws.w[j] = r->fwd[D - 1].job[j].wsel;
                hipLaunchKernelGGL(cons_sq_kernel, dim3(gx), bl, 0, st, (long long)g.ncol, r->nfwd[D - 1], ws, in, dcw, spf, e2nodes);
            }
            if (abs_factors) hipLaunchKernelGGL((tri_pass_kernel<true, true>), dim3(gx, (unsigned)r->nbwd[k]), bl, 0, st, g, k, r->ctbase[k], r->bwd[k], ctab, in, pools[which], dcw, spf);
            else hipLaunchKernelGGL((tri_pass_kernel<true, false>), dim3(gx, (unsigned)r->nbwd[k]), bl, 0, st, g, k, r->ctbase[k], r->bwd[k], ctab, in, pools[which], dcw, spf);
            in = pools[which];
            which ^= 1;
        }
        cterm = in;
    }
    if (st != st_rows) {
        (void)hipEventRecord(r->ev_out, st);
        st = st_rows;
        (void)hipStreamWaitEvent(st, r->ev_out, 0);
    }
    hipLaunchKernelGGL(rows4_gather_kernel<false>, dim3((unsigned)((g.ncol + 255) / 256)), dim3(256), 0, st, g, r->nt[0], r->nt[1], r->nt[2], r->nt[3],
                       (const double *)r->partial, cterm, rho);
    return hipGetLastError();
}

hipError_t rowsop_apply(const Grid &g, RowsOp *r, const SortScratch &rows, const double *xvec, const double *dcw, const unsigned char *spf,
                        const double *ctab, bool constraints, double *rho, hipStream_t st)
{
    return rowsop_apply_t(g, r, rows, xvec, dcw, spf, ctab, constraints, false, rho, st);
}

// The same pass with the sum of the squared row residuals (data and constraint rows: the reference's `reserr`^2, suprls :1693) into
// ssq[0]: e2buf [ncell + ncol] takes the shares -- tiles first, nodes from ncell on -- and one workgroup adds them in a fixed order.
// (Tiled pairs of passes, the A/B form, leave no array of row products: they take the single passes here.)
hipError_t rowsop_residual(const Grid &g, RowsOp *r, const SortScratch &rows, const double *xvec, const double *dcw, const unsigned char *spf,
                           const double *ctab, bool constraints, double *rho, double *ssq, double *e2buf, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(e2buf, 0, sizeof(double) * ((size_t)g.ncell + (size_t)g.ncol), st);
    if (e != hipSuccess) return e;
    const bool keep = r->no_pairs;
    r->no_pairs = true;
    e = rowsop_apply_t(g, r, rows, xvec, dcw, spf, ctab, constraints, false, rho, st, e2buf, constraints ? e2buf + g.ncell : nullptr);
    r->no_pairs = keep;
    if (e != hipSuccess) return e;
    return launch_sum_fixed(e2buf, (long long)g.ncell + g.ncol, ssq, st);
}

// diag[i] = sum over the points of w^2 b_i(x)^2: the diagonal of the data rows' Gram matrix (xvec: any vector of ncol doubles, unused)
hipError_t rowsop_data_diagonal(const Grid &g, RowsOp *r, const SortScratch &rows, const double *xvec, double *diag, hipStream_t st)
{
    hipLaunchKernelGGL(rows4_tile_kernel, dim3((unsigned)r->ntiles), dim3(256), 0, st, g, r->nt[0], r->nt[1], r->nt[2], (const int *)rows.offset,
                       (const double *)rows.xs, (const double *)nullptr, (const double *)rows.ws, rows.cap, xvec, r->partial, 1, (double *)nullptr);
    hipLaunchKernelGGL(rows4_gather_kernel<false>, dim3((unsigned)((g.ncol + 255) / 256)), dim3(256), 0, st, g, r->nt[0], r->nt[1], r->nt[2], r->nt[3],
                       (const double *)r->partial, (const double *)nullptr, diag);
    return hipGetLastError();
}

// hist (caller's dimension order; zero or holding other ranks' nothing yet) += the nearest-node histogram of the binned points
hipError_t rowsop_histogram(const Grid &g, RowsOp *r, const SortScratch &rows, double *hist, hipStream_t st)
{
    hipLaunchKernelGGL(rows4_hist_kernel, dim3((unsigned)r->ntiles), dim3(256), 0, st, g, r->nt[0], r->nt[1], r->nt[2], (const int *)rows.offset,
                       (const double *)rows.xs, (const double *)rows.ws, rows.cap, r->partial, hist);
    hipLaunchKernelGGL(rows4_gather_kernel<true>, dim3((unsigned)((g.ncol + 255) / 256)), dim3(256), 0, st, g, r->nt[0], r->nt[1], r->nt[2], r->nt[3],
                       (const double *)r->partial, (const double *)nullptr, hist);
    return hipGetLastError();
}

// den[i] = (|A|^T W^2 |A| |x|)_i + (|C|^T |C| |x|)_i + |rhs_i|: the size of the terms whose sum the refinement residual is, from the
// rows themselves (the basis functions are non-negative: |A| = A).  An upper bound of the half stencil's (|N| |x|)_i + |rhs_i| of
// launch_backward_denominators (|sum c_i c_j| <= sum |c_i| |c_j|); absx, tmp: scratch of ncol doubles.
hipError_t rowsop_backward_denominators(const Grid &g, RowsOp *r, const SortScratch &rows, const double *xvec, const double *rhs, const double *dcw,
                                        const unsigned char *spf, const double *ctab, bool constraints, double *absx, double *tmp, double *den,
                                        hipStream_t st)
{
    const unsigned gx = (unsigned)((g.ncol + 255) / 256);
    hipLaunchKernelGGL(abs_kernel, dim3(gx), dim3(256), 0, st, (long long)g.ncol, xvec, absx);
    SortScratch op = rows;
    op.ys = nullptr;
    hipError_t e = rowsop_apply_t(g, r, op, absx, dcw, spf, ctab, constraints, true, tmp, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(den_kernel, dim3(gx), dim3(256), 0, st, (long long)g.ncol, rhs, (const double *)tmp, den);
    return hipGetLastError();
}

}  // namespace splpak
