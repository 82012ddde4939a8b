// Nested-dissection multifrontal Cholesky of the normal equations on gfx950.
//
// Replaces, for large 3-D / 4-D grids, the band factorisation of bandchol.hip -- and through it the dense
// row-streaming Householder triangularisation + back-substitution of suprls (src/splpak.F90:1375-1695, called
// from splcw :849, :1025, :1052) -- by a factorisation in the nested-dissection order of ndtree.hpp: 1.2e13
// flop and 14 GB of factor at 64^3 nodes instead of 4.1e13 / 26.9 GB, 2.7e14 flop at 24^4 instead of 6.2e14.
// The refinement against the rows (plan.hip) is unchanged, so the result is the same minimiser.
//
// Every front is a dense column-major PANEL (rows: own | border, columns: own, padded to 256 with identity)
// and, while it is being eliminated, a dense Schur buffer S (border x border):
//
//     for the 256-column blocks k of the panel:   potrf(k) -> panel solve of all rows below -> update of the
//         panel columns right of k (chain stream) ;  S -= L21_k L21_k^T (second stream, beside the chain of k+1)
//     then S is added into the parent's panel / Schur buffer through the monotone child -> parent row map.
//
// Fronts of one tree depth are processed TOGETHER: every launch is a batch over job tables built once per plan
// (potrf: a workgroup per front; panel solve: a wave per 16 rows; updates: a wave per 64x64x256 item on the f64
// matrix cores, the register-streaming form of bandchol.hip's trailing update).  The two children of a parent
// add their Schur complements in two launches (slot 0, then slot 1), so every sum has a fixed order: the
// factor is bitwise reproducible from run to run.
//
// Solves walk the tree with per-front local vectors: forward bottom-up (gather the right-hand side, add the
// children's border updates, y_k = Linv_k v_k and v_below -= L_below,k y_k per block), backward top-down (border
// values from the parent, x_k = Linv_k^T (y_k - L_below,k^T x_below)), with explicit inverses of the 256x256
// diagonal blocks as in bandchol.hip.
#include "plan.hpp"
#include "ndtree.hpp"
#include <functional>
#include "chol_device.hpp"
#include <hip/hip_ext.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <thread>

namespace splpak {

namespace {

// ---------------------------------------------------------------------------------------------------------
// job tables (device PODs)
struct PotrfJob { double *A; double *inv16; long long ld; int k0; int ncols; };      // ncols: real columns of the block (the rest is identity padding)
struct TrsmJob { const double *L; double *X; const double *inv16; long long ld; int nrows; int wg0; int ncb; int pad; };   // ncb: 16-column blocks that hold real columns
// C(ti, tj) -= P_ti P_tj^T for the 64-row tiles tj in [0, nc), ti in [tj, nr); K = 256 columns of P
struct SyrkJob {
    const double *P; double *C; long long ldp, ldc; int nc, nr; int item0; int kb; int ksl; int zinit;
    // final pass of a front fused with its extend-add (pm != NULL): the finished tile is ADDED into the parent's panel
    // (columns < wpp) / Schur buffer through the child -> parent row map instead of being stored back; zinit: the front has
    // no children, its Schur buffer is never materialised (the tile starts as zero)
    const int *pm; double *Pp; double *Sp; long long ldpp, ldsp; int wpp; int h;
};
// (a job of the root's look-ahead may be a RECTANGLE of tiles instead: zinit < 0 means tile rows start at rb = -zinit for every
//  one of its nc <= rb tile columns)   // kb = 256-column blocks of P per pass, ksl = k-steps (4 columns each, multiple of 4) of the LAST of them that hold real columns
struct ZeroJob { double *S; long long lds; int nt; int tile0; };
struct InitJob { int front; int col0; };         // the panel columns [col0, col0 + wp) of a stage's launch belong to `front`
struct TrinvJob { const double *L; const double *inv16; double *dinv; double *dinvt; long long ld; };
// child's Schur buffer -> parent's panel (columns < wpp) / Schur buffer
struct AddJob { const double *S; const int *pm; double *P; double *Sp; long long lds, ldp, ldsp; int h, nt, wpp, tile0; };
struct MvJob { const double *M; const double *v; double *out; };
struct FwdJob { const double *L; const double *y; double *v; long long ld; int nrows; int wg0; };
struct DotJob { const double *L; const double *x; double *part; long long ld; int nrows; int nsplit; int rps; int wg0; };
struct BwdJob { const double *Mt; const double *y; const double *part; double *x; int nsplit; int pad; };
struct MapJob { double *child; double *par; const int *pm; int h; int pad; };
struct FrontDev { long long panel_off, ld, bofs; int own0, w, wp, h; int top; int pad; };   // panel_off < 0: not stored on this rank; top >= 0: first entry of the (distributed) front in the TopColDev table
struct TopColDev { long long off, ld; };          // block column of a top front: doubles into the arena (-1: another rank's), leading dimension
// child's Schur complement -> the block columns of its (distributed) parent this rank owns: PULLED by the owner of the parent's
// block column from wherever the child's columns live (another GPU's memory, read through the peer mapping)
struct PullJob { const double *src; long long lds; const int *pm; double *dst; long long ldd; int c0, c1, h, row0, tile0, ntr, ntc, pad; };

constexpr int DOT_RPS = 1024;          // rows per split of the backward sweep's column dots

// job of the flat workgroup / item index `b`: first[j] <= b < first[j + 1] (first = the wg0 / item0 / tile0 field)
template <typename J, typename F>
__device__ __forceinline__ int find_job(const J *__restrict__ jobs, int njobs, int b, F &&first)
{
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (first(jobs[mid]) <= b) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

// Schur buffers come in two forms (round 5).  Square (lds > 0): element (r, c) at S[r + c lds].  PACKED (lds < 0, L = -lds =
// hp + 16): only the 64-column tile columns of the lower triangle, tile column c stored from its diagonal tile down with the
// leading dimension L - 64 c -- half the bytes; a tile is still an ordinary column-major 64 x 64 matrix.
// tile (ti, tj), ti >= tj -> its first element; ld: leading dimension inside its tile column
__device__ __forceinline__ double *schur_tile(double *S, long long lds, int ti, int tj, long long &ld)
{
    if (lds >= 0) {
        ld = lds;
        return S + (long long)(ti * 64) + (long long)(tj * 64) * lds;
    }
    const long long L = -lds;
    ld = L - 64 * tj;
    return S + 64 * L * tj - 2048LL * tj * (tj - 1) + (long long)(ti - tj) * 64;
}
// column c -> p with p[r] = element (r, c), r >= 64 (c / 64)
__device__ __forceinline__ double *schur_col(double *S, long long lds, int c)
{
    if (lds >= 0) return S + (long long)c * lds;
    const long long L = -lds;
    const int tj = c >> 6;
    return S + 64 * L * tj - 2048LL * tj * (tj - 1) + (long long)(c & 63) * (L - 64 * tj) - 64 * tj;
}

// item -> (tj, ti) of a trapezoid of 64-row tiles stored column by column: column tj holds ti = tj .. nr-1
__device__ __forceinline__ void trapezoid_decode(int it, int nr, int &tj, int &ti)
{
    const double b = 2.0 * nr + 1.0;
    int c = (int)((b - sqrt(b * b - 8.0 * (double)it)) * 0.5);
    if (c < 0) c = 0;
    while (c > 0 && (long long)c * nr - (long long)c * (c - 1) / 2 > it) --c;
    while ((long long)(c + 1) * nr - (long long)(c + 1) * c / 2 <= it) ++c;
    tj = c;
    ti = c + it - (int)((long long)c * nr - (long long)c * (c - 1) / 2);
}

// ---------------------------------------------------------------------------------------------------------
template <int NW>
__global__ void __launch_bounds__(64 * NW)
nd_potrf_kernel(const PotrfJob *__restrict__ jobs, int *__restrict__ info, double *__restrict__ minpiv)
{
    const PotrfJob j = jobs[blockIdx.x];
    potrf_strip_body<NW>(j.A, j.ld, j.k0, info, minpiv, j.inv16, j.ncols);
}

__global__ void __launch_bounds__(64)
nd_trsm_kernel(const TrsmJob *__restrict__ jobs, int njobs)
{
    constexpr int KREG = 6;                     // parked blocks in registers (chol_device.hpp: trsm_rows): 18 KB of LDS per wave, 8 waves per CU
    __shared__ double xs[(NBLK - 16 - 16 * KREG) * 16];
    const int b = blockIdx.x;
    const int ji = find_job(jobs, njobs, b, [](const TrsmJob &t) { return t.wg0; });
    const TrsmJob j = jobs[ji];
    const int r0 = (b - j.wg0) * 16;
    if (r0 >= j.nrows) return;
    __builtin_amdgcn_s_setprio(3);
    trsm_rows<false, KREG>(j.L, j.X, j.ld, j.ld, j.inv16, nullptr, r0, xs, j.ncb);
}

__global__ void __launch_bounds__(64)
nd_trinv_kernel(const TrinvJob *__restrict__ jobs)
{
    constexpr int KREG = 6;                     // (as in nd_trsm_kernel)
    __shared__ double xs[(NBLK - 16 - 16 * KREG) * 16];
    const TrinvJob j = jobs[blockIdx.y];
    trsm_rows<true, KREG>(j.L, j.dinv, j.ld, NBLK, j.inv16, j.dinvt, blockIdx.x * 16, xs);
}

// ints per item queue: [0] item counter, [1] waves that stepped aside, [2 .. 9] item counters of the eight XCD slices (xmode)
constexpr int ND_QSTRIDE = 16;

// Panel-major order of the n x n lower trapezoid of tiles: panels of four tile columns, row by row inside a panel -- the four
// consecutive items of a row share their row operand, and the four column operands of a panel (2 MB at K = 1024) stay in the
// L2 while the panel is walked.  u = index in that order -> (tj, ti).
__device__ __forceinline__ void panel_decode(int u, int n, int &tj, int &ti)
{
    int c0 = 0;
    for (;;) {
        const int m = n - c0, w = m < 4 ? m : 4;
        const int sz = w * (w + 1) / 2 + (m - w) * w;
        if (u < sz) break;
        u -= sz;
        c0 += 4;
    }
    const int m = n - c0, w = m < 4 ? m : 4, tri = w * (w + 1) / 2;
    if (u < tri) {
        int i = 0;
        while (u >= i + 1) { u -= i + 1; ++i; }
        ti = c0 + i;
        tj = c0 + u;
    } else {
        u -= tri;
        ti = c0 + w + u / w;
        tj = c0 + u % w;
    }
}

// (XCC, shader engine, CU) of the CU this wave runs on, as a 12-bit index
__device__ inline unsigned nd_cu_index()
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    return ((xcc & 0xfu) << 8) | ((hw >> 8) & 0xffu);
}

// bitmap of the CUs a CU-masked stream runs on (discovery: many short workgroups on that stream)
__global__ void __launch_bounds__(64)
nd_whoami_kernel(unsigned *__restrict__ map)
{
    if (threadIdx.x == 0) {
        const unsigned i = nd_cu_index();
        atomicOr(&map[i >> 5], 1u << (i & 31));
    }
    __builtin_amdgcn_s_sleep(64);
}

// The trailing update of the multifrontal factorisation: one wave = one 64x64 item of C -= P_i P_j^T, in the
// register-streaming form of bandchol.hip's syrk64_kernel (operands loaded in MFMA fragment shape ahead of
// their use, 16 independent v_mfma_f64_16x16x4_f64 accumulators that start as the C tile; no LDS, no barriers).
// P and C have their own base pointers and leading dimensions: P is a block column of a front's panel (kb
// consecutive 256-column blocks of it per pass: K = 256 kb, the C tile is read and written once per pass),
// C the panel right of it or the front's Schur buffer.  SCHUR only names the instantiation: the Schur-buffer
// passes (K up to 1024, ~85 % of the flops of a large fit, one launch at a time) are the roofline kernel of
// bench.py, the panel updates (K = 256, on the chain) are listed separately by the profilers.
// queue != NULL: items are taken from an atomic counter and a wave that finds itself on a CU reserved for the
// diagonal-block factorisations (resmap) steps aside -- the launch carries `margin` spare waves for that.
// SPLIT = 1: a wave computes a whole 64 x 64 item.  SPLIT = 4 / 16: four / sixteen waves share an item (one 16-column slice of
// it each, or one 16 x 16 tile each) -- for launches of a few hundred items, which otherwise leave most of the chip's 1 024
// SIMDs idle while one wave per item works through its 1 024 MFMAs of 64 cycles each (27 us per K = 256, measured 30-57 us
// per launch at BASELINE config 2).  Every element sees the same sequence of operations: bitwise the same result.
template <int SD, int WPS, bool SCHUR, int SPLIT = 1, int WGW = 1>
__global__ void __launch_bounds__(64 * WGW, WPS)
nd_syrk_kernel(const SyrkJob *__restrict__ jobs, int njobs, int nitems, int margin, const unsigned *__restrict__ resmap,
               int *__restrict__ queue, int full_diag, int xmode)
{
    // xmode (Schur passes, one wave per item): XCD-aware item map.  The launch's items are cut into eight contiguous slices,
    // one per XCD (workgroups are dealt to the XCDs round robin: blockIdx & 7; with an item queue the XCC id register and
    // one counter per slice, a drained XCD steals from the next), and a front's items are walked in PANEL-major order
    // (panel_decode): operands are then fetched into ONE L2 and reused there instead of streaming through all eight.
    // WGW = 4: four waves per workgroup take four CONSECUTIVE items -- items are stored tile column by tile column, so the
    // four share their column operand, which then comes from the CU's L1 three times out of four (less operand traffic
    // = less power = a higher clock in the long power-limited Schur launches; round 2 measured +7 % for the band's bulk
    // update in sustained runs)
    constexpr int M = SPLIT == 1 ? 4 : 1, N = SPLIT == 16 ? 1 : 4;
    int b = blockIdx.x;
    const bool xm = SCHUR && SPLIT == 1 && WGW == 1 && xmode != 0;
    if (xm && !queue) {
        const int chunk = (nitems + 7) >> 3, loc = b >> 3;
        b = (b & 7) * chunk + loc;
        if (loc >= chunk) return;
    }
    if (queue) {
        if constexpr (WGW == 1) {
            const unsigned ci = nd_cu_index();
            if (resmap[ci >> 5] & (1u << (ci & 31))) {
                int e = 0;
                if (threadIdx.x == 0) e = atomicAdd(&queue[1], 1);
                e = __builtin_amdgcn_readfirstlane(e);
                if (e < margin) {
                    __builtin_amdgcn_s_sleep(127);       // do not drain the grid through this CU
                    __builtin_amdgcn_s_sleep(127);
                    return;
                }
            }
            if (xm) {
                const int chunk = (nitems + 7) >> 3, x0 = (int)((ci >> 8) & 7u);
                int t = -1;
                if (threadIdx.x == 0) {
                    for (int k = 0; k < 8 && t < 0; ++k) {
                        const int y = (x0 + k) & 7, lim = nitems - y * chunk < chunk ? nitems - y * chunk : chunk;
                        if (lim <= 0) continue;
                        const int e = atomicAdd(&queue[2 + y], 1);
                        if (e < lim) t = y * chunk + e;
                    }
                }
                b = __builtin_amdgcn_readfirstlane(t);
                if (b < 0) return;
            } else {
                if (threadIdx.x == 0) b = atomicAdd(&queue[0], 1);
                b = __builtin_amdgcn_readfirstlane(b);
            }
        } else {
            __shared__ int s_b[2];
            if (threadIdx.x == 0) {
                int skip = 0;
                const unsigned ci = nd_cu_index();
                if (resmap[ci >> 5] & (1u << (ci & 31))) skip = atomicAdd(&queue[1], 1) < margin ? 1 : 0;
                s_b[1] = skip;
                s_b[0] = skip ? 0 : atomicAdd(&queue[0], 1);
            }
            __syncthreads();
            if (s_b[1]) {
                __builtin_amdgcn_s_sleep(127);
                __builtin_amdgcn_s_sleep(127);
                return;
            }
            b = s_b[0];
        }
    }
    if constexpr (WGW > 1) b = b * WGW + (int)(threadIdx.x >> 6);
    if (b >= nitems) return;
    const int sub = SPLIT == 1 ? 0 : b % SPLIT;
    if (SPLIT > 1) b /= SPLIT;
    const int m0 = SPLIT == 1 ? 0 : (SPLIT == 4 ? sub : sub >> 2), n0 = SPLIT == 16 ? (sub & 3) : 0;
    const int ji = find_job(jobs, njobs, b, [](const SyrkJob &t) { return t.item0; });
    const SyrkJob j = jobs[ji];
    int tj, ti;
    if (!SCHUR && j.zinit < 0) {                     // rectangle: tile rows rb .. nr-1 of the tile columns 0 .. nc-1
        const int rb = -j.zinit, per = j.nr - rb, it = b - j.item0;
        tj = it / per;
        ti = rb + it - tj * per;
    } else if (xm && j.nc == j.nr)
        panel_decode(b - j.item0, j.nr, tj, ti);
    else
        trapezoid_decode(b - j.item0, j.nr, tj, ti);
    if (tj >= j.nc || ti >= j.nr) return;
    const bool diag = ti == tj;
    if (SPLIT == 16 && diag && n0 < m0) return;      // a 16 x 16 tile above the diagonal
    const int lane = threadIdx.x & 63, l15 = lane & 15, q = lane >> 4;
    const double *__restrict__ pJ = j.P + (long long)(tj * 64 + 16 * m0 + l15) + (long long)q * j.ldp;
    const double *__restrict__ pI = j.P + (long long)(ti * 64 + 16 * n0 + l15) + (long long)q * j.ldp;
    long long ldc;
    double *__restrict__ C = schur_tile(j.C, j.ldc, ti, tj, ldc);
    const long long ldp = j.ldp;
    d4_t acc[M][N];
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
        for (int n = 0; n < N; ++n)
#pragma unroll
            for (int v = 0; v < 4; ++v)
                acc[m][n][v] = (SCHUR && j.zinit) ? 0.0 : __builtin_nontemporal_load(&C[((n0 + n) * 16 + l15) + (long long)((m0 + m) * 16 + q + 4 * v) * ldc]);
    const bool skipu = diag && !full_diag;
    double qa[SD][M], qb[SD][N];
    auto fetch = [&](int slot, int step) {
        const long long off = (long long)(4 * step) * ldp;
#pragma unroll
        for (int m = 0; m < M; ++m) qa[slot][m] = -pJ[off + 16 * m];
#pragma unroll
        for (int n = 0; n < N; ++n) qb[slot][n] = pI[off + 16 * n];
    };
#pragma unroll
    for (int d = 0; d < SD; ++d) fetch(d, d);
    constexpr int NSTEP = NBLK / 4;
    static_assert(NSTEP % SD == 0, "queue depth must divide the k-steps");
    const int last = j.kb * NSTEP - 1;               // last k-step of the pass
#pragma unroll 1
    for (int h = 0; h < j.kb; ++h) {                 // one 256-column block per trip: the unrolled body of K = 256
        const int base = h * NSTEP;
        const int kend = (h == j.kb - 1) ? j.ksl : NSTEP;      // the columns beyond are identity padding: zero in these rows
        for (int ks = 0; ks < NSTEP; ks += SD) {
            if (ks >= kend) break;
#pragma unroll
            for (int d = 0; d < SD; ++d) {
#pragma unroll
                for (int m = 0; m < M; ++m)
#pragma unroll
                    for (int n = 0; n < N; ++n) {
                        // (a diagonal item stores its lower triangle only: the 6 of its 16 tiles above the diagonal are skipped --
                        // 1 % of the items of the root, 12 % of those of a front of 16 tile rows; wave-uniform branch)
                        if (SPLIT == 1 && skipu && m > n) continue;
                        acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[d][m], qb[d][n], acc[m][n], 0, 0, 0);
                    }
                if (ks + d + SD < NSTEP) fetch(d, base + ks + d + SD);
                else {                               // the first steps of the next block (clamped: re-reads in the last one)
                    const int nx = base + ks + d + SD;
                    fetch(d, nx < last ? nx : last);
                }
            }
        }
    }
    if (SCHUR && j.pm) {
        // the front's last pass: its Schur complement goes straight into the parent (extend-add), every entry once --
        // the parent's entries of THIS child are touched by no other wave of the launch (the map is injective, the
        // launch holds children of one slot only), so plain read-modify-writes are safe and the order of the sums is
        // fixed: child of slot 0, then child of slot 1
        int prow[N];
#pragma unroll
        for (int n = 0; n < N; ++n) {
            const int r = ti * 64 + (n0 + n) * 16 + l15;
            prow[n] = r < j.h ? j.pm[r] : -1;
        }
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int cc = (m0 + m) * 16 + q + 4 * v, c = tj * 64 + cc;
                const int pcol = c < j.h ? j.pm[c] : -1;
                if (pcol < 0) continue;
                double *__restrict__ colp = pcol < j.wpp ? j.Pp + (long long)pcol * j.ldpp : schur_col(j.Sp, j.ldsp, pcol - j.wpp) - j.wpp;
#pragma unroll
                for (int n = 0; n < N; ++n) {
                    const int rr = (n0 + n) * 16 + l15;
                    if (prow[n] < 0 || (diag && rr < cc)) continue;
                    colp[prow[n]] += acc[m][n][v];
                }
            }
        return;
    }
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
        for (int n = 0; n < N; ++n) {
            const int r = (n0 + n) * 16 + l15;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int c = (m0 + m) * 16 + q + 4 * v;
                if (!diag || r >= c) __builtin_nontemporal_store(acc[m][n][v], &C[r + (long long)c * ldc]);
            }
        }
}

// half stencil -> panels: entry (i, j) of N, j <= i in the natural order, belongs to the front that owns the
// earlier eliminated of the two nodes, at the row of the other one (own row, or border row found by bisection
// of the front's ascending border positions)
template <int D>
__global__ void __launch_bounds__(256)
nd_assemble_kernel(Grid g, const double *__restrict__ nst, const int *__restrict__ pos, const int *__restrict__ front_of,
                   const FrontDev *__restrict__ fd, const int *__restrict__ bpos, double *__restrict__ factor,
                   const TopColDev *__restrict__ topcol)
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
        const int pi = pos[i], pj = pos[j];
        const int c = pi < pj ? i : j;
        const int pc = pi < pj ? pi : pj, pr = pi < pj ? pj : pi;
        const FrontDev f = fd[front_of[c]];
        const int col = pc - f.own0;
        int row;
        if (pr < f.own0 + f.w) row = pr - f.own0;
        else {
            const int *__restrict__ bp = bpos + f.bofs;
            int lo = 0, hi = f.h - 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (bp[mid] < pr) lo = mid + 1;
                else hi = mid;
            }
            row = f.wp + lo;
        }
        if (f.top >= 0) {                        // a front distributed by block columns: this rank's columns only
            const int J = col >> 8;
            const TopColDev tc = topcol[f.top + J];
            if (tc.off >= 0) factor[tc.off + (row - (J << 8)) + (long long)(col & 255) * tc.ld] = nst[t];
        } else if (f.panel_off >= 0)
            factor[f.panel_off + row + (long long)col * f.ld] = nst[t];
    }
}

// The panels of ONE STAGE from scratch (round 5): a workgroup per panel column writes the column's zeros and then its entries of
// N -- the FULL stencil of the column's node, those neighbours that are eliminated later (own rows below the diagonal, border
// rows by bisection, as above); a padding column gets its unit diagonal.  Replaces, per stage, the clearing of the whole factor
// arena (14 GB at 64^3: 2.2 ms of memset that either ran before the assembly or shared the memory system with it) and
// nd_assemble_kernel's pass over all of N: the stages whose panels are alive when the factorisation starts are written
// behind the assembly (2.4 GB at 64^3), every later stage when its buffers come alive -- on the update stream at the start of
// the first stage that adds into them, beside that stage's diagonal blocks and panel solves.
template <int D>
__global__ void __launch_bounds__(256)
nd_init_kernel(Grid g, const double *__restrict__ nst, const int *__restrict__ pos, const int *__restrict__ ipos,
               const FrontDev *__restrict__ fd, const int *__restrict__ bpos, double *__restrict__ factor,
               const InitJob *__restrict__ jobs, int njobs)
{
    constexpr int NE = (D == 1) ? 7 : (D == 2) ? 49 : (D == 3) ? 343 : 2401;
    const int b = blockIdx.x;
    const int ji = find_job(jobs, njobs, b, [](const InitJob &t) { return t.col0; });
    const InitJob jb = jobs[ji];
    const FrontDev f = fd[jb.front];
    const int col = b - jb.col0;
    if (col >= f.wp || f.panel_off < 0) return;
    double *__restrict__ cp = factor + f.panel_off + (long long)col * f.ld;
    {
        double *z = cp;
        long long n = f.ld;
        if (reinterpret_cast<unsigned long long>(z) & 8) {
            if (threadIdx.x == 0) z[0] = 0.0;
            ++z, --n;
        }
        d2_t *__restrict__ z2 = reinterpret_cast<d2_t *>(z);
        for (long long i = threadIdx.x; i < (n >> 1); i += 256) __builtin_nontemporal_store((d2_t){0.0, 0.0}, z2 + i);
        if ((n & 1) && threadIdx.x == 0) z[n - 1] = 0.0;
    }
    __syncthreads();                              // (the zeros of the other waves have arrived before an entry goes on top)
    if (col >= f.w) {
        if (threadIdx.x == 0) cp[col] = 1.0;
        return;
    }
    const int pc = f.own0 + col, c = ipos[pc];
    int cd[D];
#pragma unroll
    for (int d = 0; d < D; ++d) cd[d] = (c / g.colstride[d]) % g.nodes[d];
    for (int e = threadIdx.x; e < NE; e += 256) {
        int j = c, code = e, t = e;
        bool ok = true;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int o = (t % 7) - 3;
            t /= 7;
            const int jd = cd[d] + o;
            if (jd < 0 || jd > g.nodes[d] - 1) ok = false;
            j += o * g.colstride[d];
        }
        if (!ok) continue;
        const int pj = pos[j];
        if (pj < pc) continue;                    // that entry lives in the column of j
        // N(c, j): the half stencil keeps it in the row of the larger natural index, code of (smaller - larger)
        const double v = j <= c ? nst[(long long)c * g.hstencil + code] : nst[(long long)j * g.hstencil + (NE - 1 - code)];
        int row;
        if (pj < f.own0 + f.w) row = pj - f.own0;
        else {
            const int *__restrict__ bp = bpos + f.bofs;
            int lo = 0, hi = f.h - 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (bp[mid] < pj) lo = mid + 1;
                else hi = mid;
            }
            row = f.wp + lo;
        }
        cp[row] = v;
    }
}

// extend-add into a distributed front (see PullJob): workgroup = 64 x 64 tile of the child's columns [c0, c1), rows >= c0
__global__ void __launch_bounds__(256)
nd_pull_add_kernel(const PullJob *__restrict__ jobs, int njobs)
{
    __shared__ int pr[64], pc[64];
    const int b = blockIdx.x;
    const int ji = find_job(jobs, njobs, b, [](const PullJob &t) { return t.tile0; });
    const PullJob j = jobs[ji];
    const int lt = b - j.tile0;
    const int tj = lt / j.ntr, ti = lt - tj * j.ntr;
    if (tj >= j.ntc) return;
    const int rbase = (j.c0 >> 6) << 6;
    const int r0 = rbase + ti * 64, cc0 = j.c0 + tj * 64;
    if (r0 + 63 < cc0) return;                   // the tile lies above the diagonal
    const int tid = threadIdx.x;
    if (tid < 64) {
        const int r = r0 + tid;
        pr[tid] = r < j.h ? j.pm[r] : -1;
    } else if (tid < 128) {
        const int c = cc0 + tid - 64;
        pc[tid - 64] = c < j.c1 ? j.pm[c] : -1;
    }
    __syncthreads();
    const int rl = tid & 63;
    const int prow = pr[rl];
    if (prow < 0) return;
    const int r = r0 + rl;
#pragma unroll 4
    for (int u = 0; u < 16; ++u) {
        const int cl = (tid >> 6) + 4 * u;
        const int pcol = pc[cl];
        const int c = cc0 + cl;
        if (pcol < 0 || r < c) continue;
        const double v = j.src[(long long)r + (long long)c * j.lds];
        j.dst[(long long)(prow - j.row0) + (long long)(pcol - j.row0) * j.ldd] += v;
    }
}

__global__ void __launch_bounds__(256)
nd_pad_diag_kernel(const long long *__restrict__ where, int n, double *__restrict__ factor)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) factor[where[i]] = 1.0;
}

// Schur buffer of a child -> its parent: workgroup = one 64x64 tile of the child's lower triangle
__global__ void __launch_bounds__(256)
nd_extend_add_kernel(const AddJob *__restrict__ jobs, int njobs)
{
    __shared__ int pr[64], pc[64];
    const int b = blockIdx.x;
    const int ji = find_job(jobs, njobs, b, [](const AddJob &t) { return t.tile0; });
    const AddJob j = jobs[ji];
    int tj, ti;
    trapezoid_decode(b - j.tile0, j.nt, tj, ti);
    if (tj >= j.nt || ti >= j.nt) return;
    const int tid = threadIdx.x;
    if (tid < 64) {
        const int r = ti * 64 + tid;
        pr[tid] = r < j.h ? j.pm[r] : -1;
    } else if (tid < 128) {
        const int c = tj * 64 + tid - 64;
        pc[tid - 64] = c < j.h ? j.pm[c] : -1;
    }
    __syncthreads();
    const int r = tid & 63;
    const int prow = pr[r];
    if (prow < 0) return;
    long long lds;
    const double *__restrict__ S = schur_tile(const_cast<double *>(j.S), j.lds, ti, tj, lds) + r;
#pragma unroll 4
    for (int u = 0; u < 16; ++u) {
        const int c = (tid >> 6) + 4 * u;
        const int pcol = pc[c];
        if (pcol < 0 || (ti == tj && r < c)) continue;
        const double v = S[(long long)c * lds];
        double *dst = pcol < j.wpp ? j.P + prow + (long long)pcol * j.ldp : schur_col(j.Sp, j.ldsp, pcol - j.wpp) + (prow - j.wpp);
        *dst += v;
    }
}

// zeroes the lower-triangle 64x64 tiles of Schur buffers (what the updates and the extend-add read): half the bytes of a
// memset of the square buffers
__global__ void __launch_bounds__(256)
nd_zero_kernel(const ZeroJob *__restrict__ jobs, int njobs)
{
    const int b = blockIdx.x;
    const int ji = find_job(jobs, njobs, b, [](const ZeroJob &t) { return t.tile0; });
    const ZeroJob j = jobs[ji];
    int tj, ti;
    trapezoid_decode(b - j.tile0, j.nt, tj, ti);
    if (tj >= j.nt || ti >= j.nt) return;
    long long lds;
    double *__restrict__ S = schur_tile(j.S, j.lds, ti, tj, lds);
    const int r2 = (threadIdx.x & 31) * 2, c0 = threadIdx.x >> 5;
#pragma unroll
    for (int u = 0; u < 8; ++u) *reinterpret_cast<d2_t *>(S + r2 + (long long)(c0 + 8 * u) * lds) = (d2_t){0.0, 0.0};
}

// The early clear of the panels, beside the binning of the points.  Not a memset of the runtime: that one spreads its workgroups
// over every CU until it is done (2.2 ms for 12 GB), and the binning's scatter kernel -- one workgroup takes a whole CU: 144 KB
// of LDS, 16 waves of 128 registers -- then only starts where a CU has drained: round 5 saw its workgroups run on the even
// XCDs first and on the odd ones 450 us later (1.15 ms instead of 0.42 ms alone).  A few resident workgroups write as fast and
// leave the other CUs whole.
__global__ void __launch_bounds__(1024)
nd_clear_kernel(double *__restrict__ p, long long n)
{
    if (n > 0 && (reinterpret_cast<unsigned long long>(p) & 8)) {
        if (blockIdx.x == 0 && threadIdx.x == 0) p[0] = 0.0;
        ++p, --n;
    }
    d2_t *__restrict__ q = reinterpret_cast<d2_t *>(p);
    const long long n2 = n >> 1, step = (long long)gridDim.x * 1024 * 4;
    for (long long i = (long long)blockIdx.x * 4096 + threadIdx.x; i < n2; i += step) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + 1024 * u < n2) __builtin_nontemporal_store((d2_t){0.0, 0.0}, q + i + 1024 * u);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) p[n - 1] = 0.0;
}

// distributed factorisation: the lower-triangle tiles of a Schur buffer <-> a contiguous image (tile after tile, column-major
// inside a tile), so that the join sums half the bytes of the square buffer
template <bool PACK>
__global__ void __launch_bounds__(256)
nd_tripack_kernel(double *__restrict__ S, long long lds, int nt, double *__restrict__ img)
{
    int tj, ti;
    trapezoid_decode(blockIdx.x, nt, tj, ti);
    if (tj >= nt || ti >= nt) return;
    double *__restrict__ T = S + (long long)(ti * 64) + (long long)(tj * 64) * lds;
    double *__restrict__ I = img + (long long)blockIdx.x * 4096;
    const int r2 = (threadIdx.x & 31) * 2, c0 = threadIdx.x >> 5;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int c = c0 + 8 * u;
        if (PACK) *reinterpret_cast<d2_t *>(I + r2 + 64 * c) = *reinterpret_cast<const d2_t *>(T + r2 + (long long)c * lds);
        else *reinterpret_cast<d2_t *>(T + r2 + (long long)c * lds) = *reinterpret_cast<const d2_t *>(I + r2 + 64 * c);
    }
}

// distributed factorisation: the pivot status is made collective (a rank must not leave the fit alone)
__global__ void nd_flag_kernel(const int *__restrict__ info, double *__restrict__ flag, int phase)
{
    if (phase == 0) flag[0] = info[0] != 0 ? 1.0 : 0.0;
}
__global__ void nd_unflag_kernel(int *__restrict__ info, const double *__restrict__ flag)
{
    if (flag[0] != 0.0 && info[0] == 0) info[0] = 0x7fffffff;        // another rank's subtree failed
}

// ---- solves ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
nd_gather_kernel(long long n, const int *__restrict__ rowsrc, const double *__restrict__ b, double *__restrict__ V)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int s = rowsrc[i];
        V[i] = s >= 0 ? b[s] : 0.0;
    }
}

__global__ void __launch_bounds__(256)
nd_scatter_kernel(long long n, const int *__restrict__ rowsrc, const double *__restrict__ V, double *__restrict__ x)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int s = rowsrc[i];
        if (s >= 0) x[s] = V[i];
    }
}

// forward: parent rows += the child's border updates; backward: the child's border values = parent rows
template <bool TAKE>
__global__ void __launch_bounds__(256)
nd_map_kernel(const MapJob *__restrict__ jobs)
{
    const MapJob j = jobs[blockIdx.y];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < j.h; i += gridDim.x * blockDim.x) {
        const int r = j.pm[i];
        if (TAKE) j.child[i] = j.par[r];
        else j.par[r] += j.child[i];
    }
}

// out = M v for row-major 256x256 blocks; grid (16, jobs) x 256 threads: a wave dots 4 rows
__global__ void __launch_bounds__(256)
nd_mv_kernel(const MvJob *__restrict__ jobs)
{
    const MvJob j = jobs[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * 16 + wave * 4;
    double vv[4], s[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) vv[u] = j.v[lane + 64 * u];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s[i] = 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) s[i] += j.M[(r0 + i) * NBLK + lane + 64 * u] * vv[u];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i] = wave_sum(s[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) j.out[r0 + i] = s[i];
    }
}

// v[rows below] -= L[rows, block] y: workgroup = 64 rows x 256 columns, 512 threads = 32 row pairs x 16 column groups
__global__ void __launch_bounds__(512)
nd_fwd_kernel(const FwdJob *__restrict__ jobs, int njobs)
{
    __shared__ double sy[NBLK];
    __shared__ double part[16][64];
    const int b = blockIdx.x;
    const int ji = find_job(jobs, njobs, b, [](const FwdJob &t) { return t.wg0; });
    const FwdJob j = jobs[ji];
    const int wg = b - j.wg0;
    if (wg * 64 >= j.nrows) return;
    const int tid = threadIdx.x;
    if (tid < NBLK) sy[tid] = j.y[tid];
    __syncthreads();
    const int rp = tid & 31, cg = tid >> 5;
    const int r = wg * 64 + 2 * rp;
    const double *__restrict__ Lr = j.L + r + (long long)(cg * 16) * j.ld;
    d2_t l[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) l[c] = *reinterpret_cast<const d2_t *>(Lr + (long long)c * j.ld);
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
        j.v[wg * 64 + tid] -= s;
    }
}

// part[split][c] = sum over the rows of the split of L[r, c] x[r]; workgroup = (16 columns, split), a wave takes 4 columns
__global__ void __launch_bounds__(256)
nd_dot_kernel(const DotJob *__restrict__ jobs, int njobs)
{
    const int b = blockIdx.x;
    const int ji = find_job(jobs, njobs, b, [](const DotJob &t) { return t.wg0; });
    const DotJob j = jobs[ji];
    const int lw = b - j.wg0;
    const int cg = lw & 15, split = lw >> 4;
    if (split >= j.nsplit) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = cg * 16 + wave * 4;
    const int rbeg = split * j.rps;
    const int rend = rbeg + j.rps < j.nrows ? rbeg + j.rps : j.nrows;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const double *__restrict__ Lc = j.L + lane + (long long)c0 * j.ld;
    // four row groups (20 loads) in flight per round: the rolled loop paid one memory round trip per 64 rows, 16 in a row
    // for a split of 1 024 -- the launch sits on the chain of every backward step.  Same order of the sums.
    int r = rbeg;
    for (; r + 192 < rend; r += 256) {
        double xr[4], l[4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            xr[t] = j.x[r + 64 * t + lane];
#pragma unroll
            for (int c = 0; c < 4; ++c) l[t][c] = Lc[r + 64 * t + (long long)c * j.ld];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] += l[t][c] * xr[t];
    }
    for (; r < rend; r += 64) {
        const double xr = j.x[r + lane];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] += Lc[r + (long long)c * j.ld] * xr;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = wave_sum(acc[c]);
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) j.part[(long long)split * NBLK + c0 + c] = acc[c];
    }
}

// x_k = Linv_k^T (y_k - sum_split part[split]); grid (16, jobs) x 256 threads
__global__ void __launch_bounds__(256)
nd_bwd_kernel(const BwdJob *__restrict__ jobs)
{
    const BwdJob j = jobs[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * 16 + wave * 4;
    // the 16 matrix entries are in flight while the partial dots are summed; the four columns of a split are loaded together
    // (the sums keep their order: split after split) -- the launch sits on the chain of every backward step
    double vv[4], s[4], mt[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int u = 0; u < 4; ++u) mt[i][u] = j.Mt[(r0 + i) * NBLK + lane + 64 * u];
#pragma unroll
    for (int u = 0; u < 4; ++u) vv[u] = j.y[lane + 64 * u];
    int sp = 0;
    for (; sp + 1 < j.nsplit; sp += 2) {
        double p0[4], p1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            p0[u] = j.part[(long long)sp * NBLK + lane + 64 * u];
            p1[u] = j.part[(long long)(sp + 1) * NBLK + lane + 64 * u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) vv[u] = (vv[u] - p0[u]) - p1[u];
    }
    if (sp < j.nsplit) {
#pragma unroll
        for (int u = 0; u < 4; ++u) vv[u] -= j.part[(long long)sp * NBLK + lane + 64 * u];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s[i] = 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) s[i] += mt[i][u] * vv[u];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i] = wave_sum(s[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) j.x[r0 + i] = s[i];
    }
}

// ---------------------------------------------------------------------------------------------------------
// host side
struct Launch { int first = 0, count = 0; unsigned grid = 0; double flop = 0; };

template <typename J>
struct JobTable {
    std::vector<J> host;
    J *dev = nullptr;
};

struct NdState {
    NdTree t;
    int device = 0;
    // arenas
    double *factor = nullptr, *dinv = nullptr, *dinvt = nullptr, *inv16 = nullptr;
    // Elimination schedule (round 5, ndtree.hpp NdSchedule): the stages in execution order -- one per tree depth (cut = 0), or
    // the fronts above depth `cut` one by one in postorder and the subtrees below one after the other -- and ONE Schur arena
    // whose blocks are reused along the schedule; Schur buffers as packed lower triangles (not on a rank of a one-process
    // multi-GPU fit, whose subtree roots are read by the other GPUs' pull kernels in the square form).
    NdSchedule sc;
    double *sarena = nullptr;
    long long sarena_doubles = 0;
    std::vector<char> needs;                       // [front] its Schur buffer is materialised (a leaf whose only pass is fused with the extend-add has none)
    std::vector<std::vector<int>> starts;          // [stage] the stages whose buffers come alive (are zeroed) at its start
    int root_stage = -1;                           // stage of the root (single-GPU plans; -1 otherwise)
    int schur_kb = 4;                              // panel blocks per Schur pass (SPLPAK_ND_KB: 1 .. 4)
    std::string desc;                              // what splpak_plan_factorisation reports
    double *V = nullptr, *Y = nullptr, *part = nullptr;
    long long part_cap = 0;                        // doubles of the backward sweep's partial sums (one launch at a time)
    int *pos = nullptr, *front_of = nullptr, *bpos = nullptr, *pmap = nullptr, *rowsrc = nullptr;
    long long *padwhere = nullptr;
    int npad = 0;
    FrontDev *fdev = nullptr;
    // job tables; launches indexed [depth][step]
    JobTable<PotrfJob> potrf;
    JobTable<TrsmJob> trsm, trsmb;                 // trsmb: the rows beyond the next diagonal block, beside the chain (root look-ahead)
    JobTable<SyrkJob> upd, schur;
    JobTable<TrinvJob> trinv;
    JobTable<AddJob> add;
    JobTable<ZeroJob> zero;
    JobTable<InitJob> init;                        // [stage] the panel columns of its fronts (nd_init_kernel)
    std::vector<Launch> l_init;
    int *ipos = nullptr;                           // node at an elimination position (-1: none)
    bool staged_init = false;                      // the panels are written stage by stage (not the distributed forms)
    std::vector<std::vector<int>> istarts;         // [stage i] the stages whose panels are written at the start of stage i: the first
                                                   // stage that adds into them; a stage without children (nothing orders its diagonal
                                                   // blocks behind the update stream) one stage early, and waited for through evP
    std::vector<hipEvent_t> evP;
    JobTable<MvJob> mv;
    JobTable<FwdJob> fwd;
    JobTable<DotJob> dot;
    JobTable<BwdJob> bwd;
    JobTable<MapJob> map;
    std::vector<std::vector<Launch>> l_potrf, l_trsm, l_trsmb, l_upd, l_updr, l_updo, l_schur, l_mv, l_fwd, l_dot, l_bwd;
    std::vector<char> lookahead;                   // per depth: no Schur buffers (the root) -> the panel update is split: next block column on the chain, the rest beside it
    JobTable<SyrkJob> updr;
    std::vector<char> chain_la;                    // [stage] look-ahead inside the groups of the chain: the in-group panel update of a step
                                                   // is split into the next diagonal block (l_upd) and the rest (l_updr, same stream),
                                                   // and the next step's diagonal blocks are factored on the reserved CUs beside the rest
    JobTable<SyrkJob> updo;                        // outer panel passes: K = 1024 update of the panel columns right of a group of blocks
    JobTable<SyrkJob> fin[2];                      // final Schur passes fused with the extend-add, by child slot
    std::vector<std::vector<Launch>> l_fin[2];
    std::vector<Launch> l_add[2];                  // [stage] separate extend-add launches of its fronts, by child slot (SPLPAK_ND_NO_FUSE)
    std::vector<Launch> l_zero;                    // [stage] zero the lower-triangle tiles of its Schur buffers
    bool fused = true;                             // SPLPAK_ND_NO_FUSE (read when the plan is created): separate extend-add launches
    std::vector<hipEvent_t> evW;                   // rest of the panel update of step k done
    std::vector<Launch> l_mapslot[2], l_mapall;     // per depth (of the children)
    // streams / events
    hipStream_t sP = nullptr, sU = nullptr, sR = nullptr;   // chain, Schur updates (+ their memsets), CU-masked: diagonal blocks
    std::vector<hipEvent_t> evF;                   // [stage] its last Schur passes (fused with the extend-add) are done
    unsigned *resmap = nullptr;                    // bitmap (nd_cu_index) of the CUs of sR; nres of them
    int nres = 0;
    int potrf_waves = 8;                           // waves per diagonal-block workgroup (measured 4 / 8 / 16: C2 factor 0.813 / 0.789 / 0.839 ms, 32^3 11.53 / 11.22 / 11.67, C3 the same)
    // Sharded fit with the factorisation DISTRIBUTED by subtrees (one process per GPU; SPLPAK_ND_DIST=0 turns it off): the 2^dcut subtrees
    // below tree depth dcut = ceil(log2 ranks) are dealt to the ranks; a rank eliminates its own subtrees only, the Schur
    // complements they leave in the fronts of depth dcut - 1 are summed over the ranks through the plan's all-reduce hook, and
    // the top of the tree is factored by every rank.  The solves follow the same split.
    bool dist = false;
    int world = 1, rank = 0, dcut = 0;
    std::vector<char> mine;                        // [front] this rank eliminates it
    std::vector<int> rowsrc_host;                  // [vec_doubles] variable of every front row (-1: border / padding)
    double *join_scratch = nullptr;                // packed lower triangle of the largest Schur buffer the join sums
    long long join_scratch_doubles = 0;
    int *rowsrc_out = nullptr;                     // rowsrc restricted to the variables this rank reports (the rest arrive by all-reduce)
    int ntrinv = 0;
    // ---- per-rank storage (round 4).  A plan of the one-process multi-GPU fit (mdist) keeps only ITS subtrees' panels, Schur
    // buffers and block inverses, plus its block columns of the top fronts; everything is addressed through these tables
    // (single GPU: poff = the tree's panel_off, lblk = blk0).
    std::vector<long long> poff;                   // [front] doubles into this rank's arena (-1: not stored here)
    std::vector<int> lblk;                         // [front] local index of its first 256 x 256 diagonal block (-1)
    long long factor_doubles = 0;                  // this rank's arena: panels of its subtrees | its block columns of the top fronts
    int nblocks = 0;                               // diagonal blocks whose inverses this rank keeps
    bool mdist = false;                            // rank of a one-process multi-GPU fit (NdGroup)
    NdGroup *grp = nullptr;
    int mrank = 0;
    NdPartition pt;
    std::vector<long long> tbase;                  // [top index] first entry of the front in topcol
    std::vector<TopColDev> topcol;                 // [sum of block columns of the top fronts] this rank's view
    std::vector<int> toplblk;                      // [same] local diagonal-block index of an owned, eliminated block column (-1)
    TopColDev *topcol_dev = nullptr;
    double *pbuf[3] = {nullptr, nullptr, nullptr}; // receive buffers of the panels of the top steps
    double *stagev = nullptr;                      // staging of a vector pulled from another rank (solves)
    long long stagev_doubles = 0;
    hipStream_t sCopy = nullptr;
    JobTable<PotrfJob> tpotrf;
    JobTable<TrsmJob> ttrsm;
    JobTable<SyrkJob> tchain, tbulk;
    JobTable<PullJob> tpull;
    JobTable<TrinvJob> ttrinv;
    JobTable<MvJob> tmv;
    JobTable<FwdJob> tfwd;
    JobTable<DotJob> tdot;
    JobTable<BwdJob> tbwd;
    JobTable<MapJob> tmapf, tmapb, tmaps;           // forward: children -> (F, 0); backward: parent -> (F, last); parent -> my subtree roots
    std::vector<Launch> lt_potrf, lt_trsm, lt_chain, lt_bulk, lt_mv, lt_fwd, lt_dot, lt_bwd;     // [global top step]
    std::vector<Launch> lt_pull[2];                // [top index] by child slot
    std::vector<int> lt_mapf[2], lt_mapb, lt_maps; // job indices (-1: none): [top index] per slot; [top index]; [my subtree roots, in order]
    std::vector<int> subroots;                     // my fronts of depth dcut
    std::vector<int> rslot;                        // [global top step] receive buffer of the step's panel here (-1: own panel, in place)
    std::vector<hipEvent_t> evReady, evArr, evCol, evBulk, evSF, evSB, evAdd;
    hipEvent_t evSub = nullptr, evTop = nullptr;
    int fgen = 0, sgen = 0;                        // generation of the current factorisation / solve (progress flags of the group)
    int xmode = 1;                                 // XCD-aware item map of the Schur passes (SPLPAK_ND_XCD=0: off)
    int full_diag = 0;                             // (A/B: diagonal items compute all 16 tiles)
    bool small_queue = false;                      // (A/B: small launches take the item queue too)
    int pinned_split = 4;                          // most waves per item of a small launch that runs beside a bulk update
    int wg4 = 0;                                   // 1: Schur launches in 4-wave workgroups, 2: the panel updates too
    int small_grid = 1024;                         // update launches of at most this many items are split over 4 waves per item, a quarter of it: 16
    int *queues = nullptr;                         // [nqueues][2] item counters of the update launches of one factorisation
    int nqueues = 0;
    hipEvent_t evR0 = nullptr;
    std::vector<hipEvent_t> evI;                   // potrf of step k done (per step of the current depth)
    std::vector<hipEvent_t> evT;                   // panel of step k solved (per step of the current depth)
    std::vector<hipEvent_t> evE;                   // [stage] its separate extend-add launches are done (SPLPAK_ND_NO_FUSE)
    hipEvent_t ev0 = nullptr, evJ = nullptr, evU = nullptr, evZlast = nullptr, evDone = nullptr, evPre = nullptr, evTail = nullptr;
    bool zlast_valid = false, used = false;
    bool tail_pending = false;                     // nd_prefit is clearing factor[head_doubles ..) on sU (evTail)
    long long head_doubles = 0;
    bool s_clean = false;                          // the Schur buffers that are alive when the first stage starts are zero
    std::vector<hipEvent_t> evA, evB;              // start / stop of the timed update launches
    hipEvent_t f0 = nullptr, f1 = nullptr;
    std::vector<void *> owned;
    size_t owned_bytes = 0;
};

template <typename T>
bool nd_alloc(NdState *s, T **ptr, size_t count)
{
    void *q = nullptr;
    if (count == 0) count = 1;
    hipError_t e = hipMalloc(&q, count * sizeof(T));
    if (e != hipSuccess && release_cached_plan_for_memory()) {
        (void)hipGetLastError();
        e = hipMalloc(&q, count * sizeof(T));
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        char buf[160];
        snprintf(buf, sizeof buf, "nested dissection: hipMalloc of %.3f GB failed", (double)(count * sizeof(T)) / 1e9);
        set_error(buf);
        return false;
    }
    s->owned.push_back(q);
    s->owned_bytes += count * sizeof(T);
    *ptr = static_cast<T *>(q);
    return true;
}

size_t nd_bytes(void *user) { return user ? static_cast<NdState *>(user)->owned_bytes : 0; }

template <typename T>
bool nd_upload(NdState *s, T **dev, const std::vector<T> &host)
{
    if (!nd_alloc(s, dev, host.size())) return false;
    if (host.empty()) return true;
    return hip_ok(hipMemcpy(*dev, host.data(), sizeof(T) * host.size(), hipMemcpyHostToDevice), "nested dissection: table upload");
}

void nd_destroy(void *user)
{
    NdState *s = static_cast<NdState *>(user);
    if (!s) return;
    (void)hipDeviceSynchronize();
    (void)hipSetDevice(s->device);
    for (hipStream_t *q : {&s->sP, &s->sU, &s->sR, &s->sCopy}) if (*q) (void)hipStreamDestroy(*q);
    for (auto *v : {&s->evT, &s->evE, &s->evP, &s->evA, &s->evB, &s->evI, &s->evW, &s->evF, &s->evReady, &s->evArr,
                    &s->evCol, &s->evBulk, &s->evSF, &s->evSB, &s->evAdd})
        for (hipEvent_t e : *v) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : {s->ev0, s->evJ, s->evU, s->evZlast, s->evDone, s->evPre, s->evTail, s->f0, s->f1, s->evR0, s->evSub, s->evTop}) if (e) (void)hipEventDestroy(e);
    for (void *q : s->owned) (void)hipFree(q);
    delete s;
}

long long trapezoid_items(long long nc, long long nr) { return nc * nr - nc * (nc - 1) / 2; }

// Schur buffer of front `id` (NULL: none is materialised) and the leading-dimension argument the kernels take for it
// (negative: the packed form, see schur_tile / schur_col)
inline double *s_ptr(NdState *s, int id)
{
    const long long o = s->sc.soff[(size_t)id];
    return o >= 0 ? s->sarena + o : nullptr;
}
inline long long s_ld(NdState *s, int id) { return nd_schur_ld(s->t.fr[(size_t)id], s->sc.packed); }

// Job tables of the FACTORISATION: one set of launches per stage of the schedule and block step.
bool nd_build_factor_jobs(NdState *s)
{
    NdTree &t = s->t;
    const int nstage = (int)s->sc.st.size();
    for (auto *L : {&s->l_potrf, &s->l_trsm, &s->l_trsmb, &s->l_upd, &s->l_updr, &s->l_updo, &s->l_schur}) L->assign((size_t)nstage, {});
    const bool two_level = splpak::opt_get("SPLPAK_ND_NO_OUTER") == nullptr;
    s->lookahead.assign((size_t)nstage, 0);
    s->chain_la.assign((size_t)nstage, 0);
    for (int sl = 0; sl < 2; ++sl) { s->l_fin[sl].assign((size_t)nstage, {}); s->l_add[sl].assign((size_t)nstage, Launch()); }
    s->l_zero.assign((size_t)nstage, Launch());
    s->l_init.assign((size_t)nstage, Launch());
    // Schur buffer passes: groups of up to schur_kb panel blocks (K = 1024: the C tiles are read and written once per
    // group; measured at 64^3: 257.6 ms per factorisation against 262.4 with K = 512 and 270.9 with K = 256; groups that
    // ramp up 1, 2, 4, 4, .. so that the first pass of a depth starts earlier made no difference)
    const int schur_kb = s->schur_kb;
    // groups of panel blocks that share a Schur pass (and the outer panel pass): schur_kb blocks each, the same boundaries for every
    // front of a stage.  (Round 5 tried a RAMP of smaller first groups -- 2, then 4 blocks -- so that the first pass of a stage would
    // not wait for a chain of four block steps: -1.2 ms of 225 at 64^3, paid for with slower K = 512 passes; it made the group
    // boundaries differ between the fronts of one stage, which the look-ahead inside the groups did not allow for -- a front whose
    // group ended at step k - 1 could have its block k factored while the outer pass was still writing it (round-5 advice).  The
    // switch is gone.)
    auto group_of = [&](int k, int nsteps, int &g0, int &gend) {
        g0 = (k / schur_kb) * schur_kb;
        gend = std::min(g0 + schur_kb, nsteps) - 1;
    };
    const SyrkJob syrk_end{nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0, 0, nullptr, nullptr, nullptr, 0, 0, 0, 0};
    for (int stg = 0; stg < nstage; ++stg) {
        const std::vector<int> &ids = s->sc.st[(size_t)stg].ids;
        const int d = s->sc.st[(size_t)stg].depth;
        int steps = 0;
        for (int id : ids) steps = std::max(steps, t.fr[(size_t)id].nsteps);
        for (auto *L : {&s->l_potrf, &s->l_trsm, &s->l_trsmb, &s->l_upd, &s->l_updr, &s->l_updo, &s->l_schur}) (*L)[(size_t)stg].assign((size_t)steps, Launch());
        for (int sl = 0; sl < 2; ++sl) s->l_fin[sl][(size_t)stg].assign((size_t)steps, Launch());
        bool any_schur = false;
        for (int id : ids) any_schur = any_schur || t.fr[(size_t)id].hp > 0;
        const bool la = !any_schur && steps >= 4 && !splpak::opt_get("SPLPAK_ND_NO_ROOT_LOOKAHEAD");
        // 2: the WHOLE next block column is updated on the chain (its diagonal block and the rows below it), so that the next
        // panel solve runs beside the trailing pass of this step instead of behind it (round 5); 1: only the next diagonal block
        // (round 4: the panel solve of every step, 50 us, waited for the trailing pass and was waited for by the next one)
        const bool la2 = la && !(splpak::opt_get("SPLPAK_ND_ROOT_LA") && atoi(splpak::opt_get("SPLPAK_ND_ROOT_LA")) == 1);
        const int cla_blocks = splpak::opt_get("SPLPAK_ND_CHAIN_LA") ? atoi(splpak::opt_get("SPLPAK_ND_CHAIN_LA")) : 8;   // (64^3: 219.2 ms with 8, 219.9 with 16, 221.0 with 64 or 0)
        const bool cla = !la && two_level && steps >= 2 && (int)ids.size() <= cla_blocks && !s->mdist;
        s->chain_la[(size_t)stg] = cla ? 1 : 0;
        s->lookahead[(size_t)stg] = la ? (la2 ? 2 : 1) : 0;
        for (int k = 0; k < steps; ++k) {
            Launch lp, lt, ltb, lu, lur, luo, ls, lfin[2];
            long long tbwg = 0, twg = 0, ui = 0, uri = 0, uoi = 0, si = 0;
            double uoflop = 0.0;
            luo.first = (int)s->updo.host.size();
            long long fi[2] = {0, 0};
            double fflop[2] = {0.0, 0.0}, sflop = 0.0;
            ltb.first = (int)s->trsmb.host.size();
            for (int sl = 0; sl < 2; ++sl) lfin[sl].first = (int)s->fin[sl].host.size();
            lp.first = (int)s->potrf.host.size();
            lt.first = (int)s->trsm.host.size();
            lu.first = (int)s->upd.host.size();
            lur.first = (int)s->updr.host.size();
            ls.first = (int)s->schur.host.size();
            for (int id : ids) {
                const NdFront &f = t.fr[(size_t)id];
                if (k >= f.nsteps) continue;
                double *panel = s->factor + s->poff[(size_t)id];
                double *diag = panel + (long long)k * 256 + (long long)k * 256 * f.ld;
                double *below = diag + 256;
                const int nrows = f.fp - (k + 1) * 256;
                double *i16 = s->inv16 + (long long)(s->lblk[(size_t)id] + k) * 4096;
                const int ncols = std::max(1, std::min(256, f.w - k * 256));     // real columns of block k (w > 256 (nsteps - 1) by construction)
                s->potrf.host.push_back(PotrfJob{diag, i16, f.ld, f.own0 + k * 256, ncols});
                ++lp.count;
                const int nc = (f.wp - (k + 1) * 256) / 64, nr = nrows / 64;
                if (nrows > 0) {
                    // with look-ahead (the root) only the rows of the NEXT diagonal block are solved on the chain, the rest beside it
                    const int ntop = (la && !la2 && nc > 0) ? std::min(nrows, 256) : nrows;
                    s->trsm.host.push_back(TrsmJob{diag, below, i16, f.ld, ntop, (int)twg, (ncols + 15) / 16, 0});
                    twg += ntop / 16;
                    ++lt.count;
                    if (nrows > ntop) {
                        s->trsmb.host.push_back(TrsmJob{diag, below + ntop, i16, f.ld, nrows - ntop, (int)tbwg, (ncols + 15) / 16, 0});
                        tbwg += (nrows - ntop) / 16;
                        ++ltb.count;
                    }
                }
                if (nc > 0) {
                    // panel columns right of block k: rows and columns relative to row (k+1)*256.  With look-ahead only the next
                    // DIAGONAL BLOCK (4 x 4 tiles) is updated on the chain; the rows below it in that block column (a rectangle
                    // of tiles) and the columns beyond (a trapezoid) are one launch beside the chain
                    const SyrkJob proto{below, below + (long long)256 * f.ld, f.ld, f.ld, 0, 0, 0, 1, 64, 0, nullptr, nullptr, nullptr, 0, 0, 0, 0};
                    if (!la && two_level) {
                        // TWO-LEVEL blocking of the panel (round 3): block k updates only the columns of its own group of
                        // schur_kb blocks here (K = 256); the columns beyond the group receive all of the group's blocks in ONE
                        // pass of K = 256 kb when its last block is solved -- the same sums in the same order (the accumulators
                        // start as the tile and subtract block after block), a quarter of the read-modify-writes, and launches
                        // that run at the rate of the Schur passes instead of 34 TFLOP/s (rocprofv3, 64^3)
                        int pg0 = 0, pgend = 0;
                        group_of(k, f.nsteps, pg0, pgend);
                        const int nc_in = std::min(nc, (pgend - k) * 4);
                        if (nc_in > 0 && cla) {
                            // look-ahead inside the group: the next diagonal block first, the rest of the group's columns behind it
                            const int n4 = std::min(nc_in, 4);
                            SyrkJob a = proto;
                            a.nc = n4; a.nr = std::min(nr, 4); a.item0 = (int)ui;
                            s->upd.host.push_back(a);
                            ui += trapezoid_items(a.nc, a.nr);
                            ++lu.count;
                            if (nr > 4) {
                                SyrkJob r = proto;
                                r.nc = n4; r.nr = nr; r.item0 = (int)uri; r.zinit = -4;
                                s->updr.host.push_back(r);
                                uri += (long long)n4 * (nr - 4);
                                ++lur.count;
                            }
                            if (nc_in > 4) {
                                SyrkJob t2 = proto;
                                t2.P = below + 256;
                                t2.C = below + (long long)256 * f.ld + 256 + (long long)256 * f.ld;
                                t2.nc = nc_in - 4; t2.nr = nr - 4; t2.item0 = (int)uri;
                                s->updr.host.push_back(t2);
                                uri += trapezoid_items(nc_in - 4, nr - 4);
                                ++lur.count;
                            }
                        } else if (nc_in > 0) {
                            SyrkJob a = proto;
                            a.nc = nc_in; a.nr = nr; a.item0 = (int)ui;
                            s->upd.host.push_back(a);
                            ui += trapezoid_items(nc_in, nr);
                            ++lu.count;
                        }
                        if (k == pgend) {             // nc > 0: columns remain beyond the group
                            SyrkJob o = proto;
                            o.P = panel + (long long)(k + 1) * 256 + (long long)pg0 * 256 * f.ld;
                            o.kb = k - pg0 + 1;
                            o.nc = nc; o.nr = nr; o.item0 = (int)uoi;
                            s->updo.host.push_back(o);
                            uoi += trapezoid_items(nc, nr);
                            uoflop += 2.0 * 64 * 64 * 256.0 * o.kb * (double)trapezoid_items(nc, nr);
                            ++luo.count;
                        }
                    } else if (!la) {
                        SyrkJob a = proto;
                        a.nc = nc; a.nr = nr; a.item0 = (int)ui;
                        s->upd.host.push_back(a);
                        ui += trapezoid_items(nc, nr);
                        ++lu.count;
                    } else {
                        const int n4 = std::min(nc, 4);
                        SyrkJob a = proto;                                   // next diagonal block
                        a.nc = n4; a.nr = std::min(nr, 4); a.item0 = (int)ui;
                        s->upd.host.push_back(a);
                        ui += trapezoid_items(a.nc, a.nr);
                        ++lu.count;
                        if (nr > 4 && la2) {                                 // rows below it in the next block column: on the chain too
                            SyrkJob r = proto;
                            r.nc = n4; r.nr = nr; r.item0 = (int)ui; r.zinit = -4;
                            s->upd.host.push_back(r);
                            ui += (long long)n4 * (nr - 4);
                            ++lu.count;
                        } else if (nr > 4) {                                 // ... or beside it
                            SyrkJob r = proto;
                            r.nc = n4; r.nr = nr; r.item0 = (int)uri; r.zinit = -4;
                            s->updr.host.push_back(r);
                            uri += (long long)n4 * (nr - 4);
                            ++lur.count;
                        }
                        if (nc > 4) {                                        // the columns beyond
                            SyrkJob t2 = proto;
                            t2.P = below + 256;
                            t2.C = below + (long long)256 * f.ld + 256 + (long long)256 * f.ld;
                            t2.nc = nc - 4; t2.nr = nr - 4; t2.item0 = (int)uri;
                            s->updr.host.push_back(t2);
                            uri += trapezoid_items(nc - 4, nr - 4);
                            ++lur.count;
                        }
                    }
                }
                // Schur buffer: one pass per group of up to schur_kb panel blocks, launched when the group's last block is solved
                const int ns = f.hp / 64;
                int g0 = 0, gend = 0;
                group_of(k, f.nsteps, g0, gend);
                if (ns > 0 && k == gend) {
                    const int kb = k - g0 + 1;
                    // (the last block of a front holds ncols real columns: the k-loop stops behind them, in chunks of 16 columns)
                    const int ksl = (k == f.nsteps - 1) ? 4 * ((ncols + 15) / 16) : 64;
                    // (the ns diagonal items skip the 6 of their 16 tiles above the diagonal)
                    const double jitems = (double)trapezoid_items(ns, ns) - (s->full_diag ? 0.0 : 0.375 * ns);
                    const double jflop = 2.0 * 64 * 64 * (256.0 * (kb - 1) + 4.0 * ksl) * jitems;
                    // (a subtree root of a multi-GPU fit keeps its Schur complement: the owners of the parent's block columns pull it)
                    const bool boundary = s->mdist && f.depth == s->pt.dcut;
                    if (s->fused && k == f.nsteps - 1 && f.parent >= 0 && !boundary) {
                        // the front's last pass carries its Schur complement into the parent itself
                        const NdFront &pf = t.fr[(size_t)f.parent];
                        const int sl = f.slot;
                        const int leaf = s->needs[(size_t)id] ? 0 : 1;       // no children, one pass: the buffer is never materialised
                        s->fin[sl].host.push_back(SyrkJob{panel + f.wp + (long long)g0 * 256 * f.ld, s_ptr(s, id), f.ld,
                                                          s_ld(s, id), ns, ns, (int)fi[sl], kb, ksl, leaf, s->pmap + f.bofs,
                                                          s->factor + s->poff[(size_t)f.parent], s_ptr(s, f.parent), pf.ld, s_ld(s, f.parent), pf.wp, f.h});
                        fi[sl] += trapezoid_items(ns, ns);
                        fflop[sl] += jflop;
                        ++lfin[sl].count;
                    } else {
                        s->schur.host.push_back(SyrkJob{panel + f.wp + (long long)g0 * 256 * f.ld, s_ptr(s, id), f.ld,
                                                        s_ld(s, id), ns, ns, (int)si, kb, ksl, 0, nullptr, nullptr, nullptr, 0, 0, 0, 0});
                        si += trapezoid_items(ns, ns);
                        sflop += jflop;
                        ++ls.count;
                    }
                }
            }
            if (twg > 0x7fffffffLL || ui > 0x7fffffffLL || si > 0x7fffffffLL || uri > 0x7fffffffLL || uoi > 0x7fffffffLL) { set_error("nested dissection: launch too large"); return false; }
            lp.grid = (unsigned)lp.count;
            lt.grid = (unsigned)twg;
            ltb.grid = (unsigned)tbwg;
            lu.grid = (unsigned)ui;
            lu.flop = 2.0 * 64 * 64 * 256 * (double)ui;
            lur.grid = (unsigned)uri;
            lur.flop = 2.0 * 64 * 64 * 256 * (double)uri;
            luo.grid = (unsigned)uoi;
            luo.flop = uoflop;
            ls.grid = (unsigned)si;
            ls.flop = sflop;
            // sentinels for the job search (first field of the element after the last job)
            if (lt.count) s->trsm.host.push_back(TrsmJob{nullptr, nullptr, nullptr, 0, 0, (int)twg, 0, 0});
            if (ltb.count) s->trsmb.host.push_back(TrsmJob{nullptr, nullptr, nullptr, 0, 0, (int)tbwg, 0, 0});
            SyrkJob e = syrk_end;
            e.item0 = (int)ui;
            if (lu.count) s->upd.host.push_back(e);
            e.item0 = (int)uri;
            if (lur.count) s->updr.host.push_back(e);
            e.item0 = (int)uoi;
            if (luo.count) s->updo.host.push_back(e);
            e.item0 = (int)si;
            if (ls.count) s->schur.host.push_back(e);
            for (int sl = 0; sl < 2; ++sl) {
                if (fi[sl] > 0x7fffffffLL) { set_error("nested dissection: launch too large"); return false; }
                lfin[sl].grid = (unsigned)fi[sl];
                lfin[sl].flop = fflop[sl];
                e.item0 = (int)fi[sl];
                if (lfin[sl].count) s->fin[sl].host.push_back(e);
                s->l_fin[sl][(size_t)stg][(size_t)k] = lfin[sl];
            }
            s->l_potrf[(size_t)stg][(size_t)k] = lp;
            s->l_trsm[(size_t)stg][(size_t)k] = lt;
            s->l_trsmb[(size_t)stg][(size_t)k] = ltb;
            s->l_upd[(size_t)stg][(size_t)k] = lu;
            s->l_updr[(size_t)stg][(size_t)k] = lur;
            s->l_updo[(size_t)stg][(size_t)k] = luo;
            s->l_schur[(size_t)stg][(size_t)k] = ls;
        }
        // lower-triangle tiles of the stage's Schur buffers
        {
            Launch lz;
            lz.first = (int)s->zero.host.size();
            long long tiles = 0;
            for (int id : ids) {
                const NdFront &f = t.fr[(size_t)id];
                if (f.hp == 0 || !s->needs[(size_t)id]) continue;       // (a leaf's buffer is never materialised: fused last pass)
                const int nt = f.hp / 64;
                s->zero.host.push_back(ZeroJob{s_ptr(s, id), s_ld(s, id), nt, (int)tiles});
                tiles += trapezoid_items(nt, nt);
                ++lz.count;
            }
            if (tiles > 0x7fffffffLL) { set_error("nested dissection: launch too large"); return false; }
            lz.grid = (unsigned)tiles;
            if (lz.count) s->zero.host.push_back(ZeroJob{nullptr, 0, 0, (int)tiles});
            s->l_zero[(size_t)stg] = lz;
        }
        // panel columns of the stage's fronts
        {
            Launch li;
            li.first = (int)s->init.host.size();
            long long cols = 0;
            for (int id : ids) {
                if (s->poff[(size_t)id] < 0) continue;
                s->init.host.push_back(InitJob{id, (int)cols});
                cols += t.fr[(size_t)id].wp;
                ++li.count;
            }
            if (cols > 0x7fffffffLL) { set_error("nested dissection: launch too large"); return false; }
            li.grid = (unsigned)cols;
            if (li.count) s->init.host.push_back(InitJob{-1, (int)cols});
            s->l_init[(size_t)stg] = li;
        }
        // separate extend-add launches (SPLPAK_ND_NO_FUSE): children of this stage -> their parents
        if (d >= 1 && !s->fused) {
            for (int sl = 0; sl < 2; ++sl) {
                Launch la2;
                la2.first = (int)s->add.host.size();
                long long tiles = 0;
                for (int id : ids) {
                    const NdFront &f = t.fr[(size_t)id];
                    if (f.slot != sl || f.h == 0) continue;
                    const NdFront &p = t.fr[(size_t)f.parent];
                    const int nt = f.hp / 64;
                    s->add.host.push_back(AddJob{s_ptr(s, id), s->pmap + f.bofs, s->factor + s->poff[(size_t)f.parent], s_ptr(s, f.parent), s_ld(s, id), p.ld,
                                                 s_ld(s, f.parent), f.h, nt, p.wp, (int)tiles});
                    tiles += trapezoid_items(nt, nt);
                    ++la2.count;
                }
                if (tiles > 0x7fffffffLL) { set_error("nested dissection: launch too large"); return false; }
                la2.grid = (unsigned)tiles;
                if (la2.count) s->add.host.push_back(AddJob{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0, (int)tiles});
                s->l_add[sl][(size_t)stg] = la2;
            }
        }
    }
    for (size_t id = 0; id < t.fr.size(); ++id) {
        const NdFront &f = t.fr[id];
        if (!s->mine.empty() && !s->mine[id]) continue;
        for (int k = 0; k < f.nsteps; ++k) {
            const double *diag = s->factor + s->poff[id] + (long long)k * 256 + (long long)k * 256 * f.ld;
            const long long lb = s->lblk[id] + k;
            s->trinv.host.push_back(TrinvJob{diag, s->inv16 + lb * 4096, s->dinv + lb * 65536, s->dinvt + lb * 65536, f.ld});
        }
    }
    s->ntrinv = (int)s->trinv.host.size();
    return true;
}

// Job tables of the SOLVES: per tree depth (all fronts of the depth, whatever their pipeline) and block step.
bool nd_build_solve_jobs(NdState *s)
{
    NdTree &t = s->t;
    const int nd = t.maxdepth + 1;
    for (auto *L : {&s->l_mv, &s->l_fwd, &s->l_dot, &s->l_bwd}) L->assign((size_t)nd, {});
    for (int sl = 0; sl < 2; ++sl) s->l_mapslot[sl].assign((size_t)nd, Launch());
    s->l_mapall.assign((size_t)nd, Launch());
    long long part_max = 0;
    for (int d = 0; d < nd; ++d) {
        std::vector<int> ids;
        for (int id : t.by_depth[(size_t)d])
            if (s->mine.empty() || s->mine[(size_t)id]) ids.push_back(id);
        int steps = 0;
        for (int id : ids) steps = std::max(steps, t.fr[(size_t)id].nsteps);
        for (auto *L : {&s->l_mv, &s->l_fwd, &s->l_dot, &s->l_bwd}) (*L)[(size_t)d].assign((size_t)steps, Launch());
        for (int k = 0; k < steps; ++k) {
            Launch lm, lf, ld, lb;
            lm.first = (int)s->mv.host.size();
            lf.first = (int)s->fwd.host.size();
            ld.first = (int)s->dot.host.size();
            lb.first = (int)s->bwd.host.size();
            long long fwg = 0, dwg = 0, partofs = 0;
            for (int id : ids) {
                const NdFront &f = t.fr[(size_t)id];
                if (k >= f.nsteps) continue;
                const double *below = s->factor + s->poff[(size_t)id] + (long long)k * 256 + (long long)k * 256 * f.ld + 256;
                const int nrows = f.fp - (k + 1) * 256;
                double *Vf = s->V + f.vofs, *Yf = s->Y + f.vofs;
                s->mv.host.push_back(MvJob{s->dinv + (long long)(s->lblk[(size_t)id] + k) * 65536, Vf + k * 256, Yf + k * 256});
                ++lm.count;
                int nsplit = 0;
                double *partp = s->part + partofs;
                if (nrows > 0) {
                    s->fwd.host.push_back(FwdJob{below, Yf + k * 256, Vf + (k + 1) * 256, f.ld, nrows, (int)fwg});
                    fwg += nrows / 64;
                    ++lf.count;
                    nsplit = (nrows + DOT_RPS - 1) / DOT_RPS;
                    s->dot.host.push_back(DotJob{below, Vf + (k + 1) * 256, partp, f.ld, nrows, nsplit, DOT_RPS, (int)dwg});
                    dwg += 16 * nsplit;
                    partofs += (long long)nsplit * 256;
                    ++ld.count;
                }
                s->bwd.host.push_back(BwdJob{s->dinvt + (long long)(s->lblk[(size_t)id] + k) * 65536, Yf + k * 256, partp, Vf + k * 256, nsplit, 0});
                ++lb.count;
            }
            if (fwg > 0x7fffffffLL || dwg > 0x7fffffffLL) { set_error("nested dissection: launch too large"); return false; }
            part_max = std::max(part_max, partofs);
            lm.grid = (unsigned)lm.count;
            lf.grid = (unsigned)fwg;
            ld.grid = (unsigned)dwg;
            lb.grid = (unsigned)lb.count;
            if (lf.count) s->fwd.host.push_back(FwdJob{nullptr, nullptr, nullptr, 0, 0, (int)fwg});
            if (ld.count) s->dot.host.push_back(DotJob{nullptr, nullptr, nullptr, 0, 0, 0, 0, (int)dwg});
            s->l_mv[(size_t)d][(size_t)k] = lm;
            s->l_fwd[(size_t)d][(size_t)k] = lf;
            s->l_dot[(size_t)d][(size_t)k] = ld;
            s->l_bwd[(size_t)d][(size_t)k] = lb;
        }
        // children at depth d <-> parents at depth d - 1 (border values of the sweeps)
        if (d >= 1) {
            for (int sl = 0; sl < 2; ++sl) {
                Launch lmj;
                lmj.first = (int)s->map.host.size();
                for (int id : ids) {
                    const NdFront &f = t.fr[(size_t)id];
                    if (f.slot != sl || f.h == 0) continue;
                    const NdFront &p = t.fr[(size_t)f.parent];
                    s->map.host.push_back(MapJob{s->V + f.vofs + f.wp, s->V + p.vofs, s->pmap + f.bofs, f.h, 0});
                    ++lmj.count;
                }
                lmj.grid = (unsigned)lmj.count;
                s->l_mapslot[sl][(size_t)d] = lmj;
            }
            // backward: both slots at once = the two consecutive runs of map jobs
            Launch all;
            all.first = s->l_mapslot[0][(size_t)d].first;
            all.count = s->l_mapslot[0][(size_t)d].count + s->l_mapslot[1][(size_t)d].count;
            all.grid = (unsigned)all.count;
            s->l_mapall[(size_t)d] = all;
        }
    }
    return part_max <= s->part_cap;
}

bool nd_build_jobs(NdState *s)
{
    return nd_build_factor_jobs(s) && nd_build_solve_jobs(s);
}

// diagonal blocks of one step of a depth: a workgroup per front, 8 waves each (SPLPAK_ND_POTRF_WAVES = 4: the band path's form, 16)
void launch_potrf(NdState *s, const Launch &lp, hipStream_t st, int *info_dev, double *minpiv_dev)
{
    const PotrfJob *jobs = s->potrf.dev + lp.first;
    if (s->potrf_waves == 4) hipLaunchKernelGGL(nd_potrf_kernel<4>, dim3(lp.grid), dim3(256), 0, st, jobs, info_dev, minpiv_dev);
    else if (s->potrf_waves == 8) hipLaunchKernelGGL(nd_potrf_kernel<8>, dim3(lp.grid), dim3(512), 0, st, jobs, info_dev, minpiv_dev);
    else hipLaunchKernelGGL(nd_potrf_kernel<16>, dim3(lp.grid), dim3(1024), 0, st, jobs, info_dev, minpiv_dev);
}

// schur: the Schur-buffer passes (timed: the roofline kernel); otherwise the panel update of the chain.
// pinned: diagonal blocks are being factored on the reserved CUs -- the waves take their items from a queue and
// step aside there.
void launch_syrk(NdState *s, const JobTable<SyrkJob> &tab, const Launch &l, hipStream_t st, CholStats *stats, bool timing, bool schur,
                 bool pinned, int &qnext)
{
    if (l.count == 0 || l.grid == 0) return;
    hipEvent_t a = nullptr, b = nullptr;
    if (timing && schur) {
        const size_t i = (size_t)stats->syrk_launches;
        while (s->evA.size() <= i) {
            hipEvent_t e;
            (void)hipEventCreate(&e);
            s->evA.push_back(e);
            (void)hipEventCreate(&e);
            s->evB.push_back(e);
        }
        a = s->evA[i];
        b = s->evB[i];
        stats->syrk_launches += 1;
        stats->syrk_flop += l.flop;
    }
    if (stats) {
        stats->total_flop += l.flop;
        if (schur) {
            stats->bulk_launches += 1;
            stats->bulk_flop += l.flop;
        }
    }
    // A small launch (split over several waves per item) does without the item queue: its waves are gone in microseconds, so
    // they need not keep off the reserved CUs -- and the queue costs it dearly: 2 048 placeholder workgroups plus one atomic
    // per workgroup on ONE word made the 10-item update of the root's next diagonal block a 60-75 us launch, on the chain of
    // every one of the root's 48 steps (round 3, tools/last_fit_trace.py).
    const bool small_launch = (int)l.grid <= s->small_grid && !s->small_queue;
    int *queue = nullptr;
    int margin = 0;
    if (pinned && s->nres > 0 && qnext < s->nqueues && !small_launch) {
        queue = s->queues + ND_QSTRIDE * (qnext++);
        margin = 256 * s->nres;
    }
    const SyrkJob *jobs = tab.dev + l.first;
    // SD = 4 k-steps of operand look-ahead, two waves per SIMD (244 registers): measured at 64^3 against the 16-deep
    // queue / one wave per SIMD form the band's bulk update uses -- 257.6 against 282.2 ms per factorisation, because
    // the queue is carried across the block loop of a K = 1024 pass and then has to live in registers (256 + 180)
    // Launches of a few hundred items leave most SIMDs idle while one wave per item works through its MFMAs: they are split
    // over 4 / 16 waves per item (SPLIT above; same arithmetic order, bitwise the same result)
    // (beside a bulk update that fills every wave slot -- `pinned` -- the waves of this launch are placed as slots retire,
    // ~37 per us at 64^3: sixteen waves per item then wait longer than they save; four per item there)
    int split = (int)l.grid * 4 <= s->small_grid ? 16 : ((int)l.grid <= s->small_grid ? 4 : 1);
    if (pinned && split > s->pinned_split) split = s->pinned_split;
    const int nit = (int)l.grid * split;
    const bool wg4 = split == 1 && (s->wg4 >= 2 || (s->wg4 == 1 && schur));
    unsigned gx = wg4 ? (l.grid + 3) / 4 : l.grid * (unsigned)split;
    if (s->xmode && schur && split == 1 && !wg4 && !queue) gx = (gx + 7u) / 8u * 8u;      // eight equal slices
    const dim3 grid(gx + (unsigned)margin);
    // operand look-ahead in k-steps: a split wave issues 1 (4) MFMA per step, so 4 steps cover 256 (1 024) cycles -- less than
    // one memory round trip: 77 us per K = 256 launch of the root's look-ahead block.  32 (16) steps in flight instead.
#define ND_SD(SPL) ((SPL) == 16 ? 32 : ((SPL) == 4 ? 16 : 4))
#define ND_SYRK_GO(SCH, SPL, WW)                                                                                               \
    do {                                                                                                                       \
        if (SCH) hipExtLaunchKernelGGL((nd_syrk_kernel<ND_SD(SPL), 2, SCH, SPL, WW>), grid, dim3(64 * WW), 0, st, a, b, 0, jobs, l.count, nit, margin,  \
                                       (const unsigned *)s->resmap, queue, s->full_diag, s->xmode);                            \
        else hipLaunchKernelGGL((nd_syrk_kernel<ND_SD(SPL), 2, SCH, SPL, WW>), grid, dim3(64 * WW), 0, st, jobs, l.count, nit, margin,   \
                                (const unsigned *)s->resmap, queue, s->full_diag, s->xmode);                                  \
    } while (0)
    if (schur) {
        if (split == 16) ND_SYRK_GO(true, 16, 1); else if (split == 4) ND_SYRK_GO(true, 4, 1); else if (wg4) ND_SYRK_GO(true, 1, 4); else ND_SYRK_GO(true, 1, 1);
    } else {
        if (split == 16) ND_SYRK_GO(false, 16, 1); else if (split == 4) ND_SYRK_GO(false, 4, 1); else if (wg4) ND_SYRK_GO(false, 1, 4); else ND_SYRK_GO(false, 1, 1);
    }
#undef ND_SYRK_GO
#undef ND_SD
}

#include "ndtop.inc"

// panels of stage x from the half stencil (nd_init_kernel)
static void nd_init_stage(NdState *s, splpak_plan *p, int x, hipStream_t q)
{
    const Launch &li = s->l_init[(size_t)x];
    if (!li.count) return;
    const Grid &g = p->g;
#define ND_INIT_GO(DD) hipLaunchKernelGGL(nd_init_kernel<DD>, dim3(li.grid), dim3(256), 0, q, g, (const double *)p->nst, (const int *)s->pos, (const int *)s->ipos, \
                                          (const FrontDev *)s->fdev, (const int *)s->bpos, s->factor, (const InitJob *)(s->init.dev + li.first), li.count)
    switch (g.ndim) {
    case 1: ND_INIT_GO(1); break;
    case 2: ND_INIT_GO(2); break;
    case 3: ND_INIT_GO(3); break;
    default: ND_INIT_GO(4); break;
    }
#undef ND_INIT_GO
}

// The panels start from zero (14 GB at 64^3: 2.2 ms of memset).  Only the head of the arena is busy during the assembly -- the
// per-cell Gram blocks live there until the stencil gather has read them -- so the rest is cleared on the second stream
// while the points are binned and the blocks computed, and nd_assemble clears the head.
hipError_t nd_prefit(splpak_plan *p, hipStream_t st, void *user)
{
    NdState *s = static_cast<NdState *>(user);
    s->tail_pending = false;
    if (s->staged_init && !s->dist) return hipSuccess;         // (nothing to clear: nd_init_kernel writes every panel column whole)
    if (!s->sU || !s->evPre || splpak::opt_get("SPLPAK_ND_NO_EARLY_CLEAR")) return hipSuccess;
    long long head = 0;
    if (p->gscratch == s->factor) head = p->gscratch_doubles < s->factor_doubles ? p->gscratch_doubles : s->factor_doubles;
    else if (p->gscratch >= s->factor && p->gscratch < s->factor + s->factor_doubles) return hipSuccess;     // (not laid out that way)
    if (head >= s->factor_doubles) return hipSuccess;
    hipError_t e = hipEventRecord(s->evPre, st);                 // the previous fit's solves have read the factor by now
    if (e == hipSuccess) e = hipStreamWaitEvent(s->sU, s->evPre, 0);
    const int clear_wgs = splpak::opt_get("SPLPAK_ND_CLEAR_WGS") ? atoi(splpak::opt_get("SPLPAK_ND_CLEAR_WGS")) : 128;
    if (e == hipSuccess) {
        if (clear_wgs > 0) {
            hipLaunchKernelGGL(nd_clear_kernel, dim3((unsigned)clear_wgs), dim3(1024), 0, s->sU, s->factor + head, s->factor_doubles - head);
            e = hipGetLastError();
        } else
            e = hipMemsetAsync(s->factor + head, 0, sizeof(double) * (size_t)(s->factor_doubles - head), s->sU);
    }
    if (e == hipSuccess) e = hipEventRecord(s->evTail, s->sU);
    if (e != hipSuccess) return e;
    s->tail_pending = true;
    s->head_doubles = head;
    return hipSuccess;
}

hipError_t nd_assemble(splpak_plan *p, hipStream_t st, void *user)
{
    NdState *s = static_cast<NdState *>(user);
    const Grid &g = p->g;
    hipError_t e = hipSuccess;
    if (s->staged_init && !s->dist) {             // the stages whose panels are alive when the first stage starts; the others in nd_factor
        if (!s->istarts.empty())
            for (int x : s->istarts[0]) nd_init_stage(s, p, x, st);
        return hipGetLastError();
    }
    if (s->tail_pending) {
        if (s->head_doubles > 0) e = hipMemsetAsync(s->factor, 0, sizeof(double) * (size_t)s->head_doubles, st);
        if (e == hipSuccess) e = hipStreamWaitEvent(st, s->evTail, 0);
        s->tail_pending = false;
    } else
        e = hipMemsetAsync(s->factor, 0, sizeof(double) * (size_t)s->factor_doubles, st);
    if (e != hipSuccess) return e;
    const long long total = (long long)g.ncol * g.hstencil;
    long long blocks = (total + 255) / 256;
    if (blocks > 256LL * 64) blocks = 256LL * 64;
    switch (g.ndim) {
    case 1: hipLaunchKernelGGL(nd_assemble_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, g, (const double *)p->nst, (const int *)s->pos, (const int *)s->front_of, (const FrontDev *)s->fdev, (const int *)s->bpos, s->factor, (const TopColDev *)s->topcol_dev); break;
    case 2: hipLaunchKernelGGL(nd_assemble_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, st, g, (const double *)p->nst, (const int *)s->pos, (const int *)s->front_of, (const FrontDev *)s->fdev, (const int *)s->bpos, s->factor, (const TopColDev *)s->topcol_dev); break;
    case 3: hipLaunchKernelGGL(nd_assemble_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, st, g, (const double *)p->nst, (const int *)s->pos, (const int *)s->front_of, (const FrontDev *)s->fdev, (const int *)s->bpos, s->factor, (const TopColDev *)s->topcol_dev); break;
    default: hipLaunchKernelGGL(nd_assemble_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, g, (const double *)p->nst, (const int *)s->pos, (const int *)s->front_of, (const FrontDev *)s->fdev, (const int *)s->bpos, s->factor, (const TopColDev *)s->topcol_dev); break;
    }
    if (s->npad > 0)
        hipLaunchKernelGGL(nd_pad_diag_kernel, dim3((unsigned)((s->npad + 255) / 256)), dim3(256), 0, st, (const long long *)s->padwhere, s->npad, s->factor);
    return hipGetLastError();
}

hipError_t nd_factor(splpak_plan *p, int *info_dev, double *minpiv_dev, hipStream_t st, void *user)
{
    NdState *s = static_cast<NdState *>(user);
    NdTree &t = s->t;
    CholStats *stats = &p->stats;
    const bool timing = stats->enabled;
    const bool enabled = stats->enabled;
    *stats = CholStats{};
    stats->enabled = enabled;
    const int ns = (int)s->sc.st.size();
    const bool serial = splpak::opt_get("SPLPAK_NO_LOOKAHEAD") != nullptr;
    hipStream_t sP = s->sP, sU = s->sU, sR = s->sR;
    if (serial) sP = sU = st;
    if (serial || !sR || splpak::opt_get("SPLPAK_NO_PANEL_CU")) sR = nullptr;
    // potrf goes to the reserved CUs while a stage has at most this many diagonal blocks per step per reserved CU (round 5: 1 --
    // two rounds of 140 us on the reserved CUs lose against one round on the whole chip beside the pass: 219.3 against 220.7 ms at 64^3)
    const int pin_rounds = splpak::opt_get("SPLPAK_ND_PIN_ROUNDS") ? atoi(splpak::opt_get("SPLPAK_ND_PIN_ROUNDS")) : 1;
    if (timing) {
        if (!s->f0) { (void)hipEventCreate(&s->f0); (void)hipEventCreate(&s->f1); }
        (void)hipEventRecord(s->f0, st);
    }
    if (s->used && s->evDone) (void)hipStreamWaitEvent(st, s->evDone, 0);
    if (s->zlast_valid) (void)hipStreamWaitEvent(st, s->evZlast, 0);
    // the Schur buffers of stage x are zeroed (lower-triangle tiles) when they come alive: at the start of the first stage
    // that adds into them, on the update stream -- whatever used their place in the arena before was last touched there
    auto zero_block = [&](int x, hipStream_t q) {
        const Launch &lz = s->l_zero[(size_t)x];
        if (lz.count) hipLaunchKernelGGL(nd_zero_kernel, dim3(lz.grid), dim3(256), 0, q, (const ZeroJob *)(s->zero.dev + lz.first), lz.count);
    };
    if (!s->s_clean && ns > 0)  // first fit, or the previous one was abandoned (otherwise the previous fit left them zeroed: evZlast)
        for (int x : s->starts[0]) zero_block(x, st);
    s->s_clean = false;
    if (s->queues) (void)hipMemsetAsync(s->queues, 0, sizeof(int) * ND_QSTRIDE * (size_t)s->nqueues, st);
    if (s->dist && s->rank != 0)        // the fronts the subtrees' Schur complements are summed in: their entries of N come from rank 0 alone
        for (int id : t.by_depth[(size_t)(s->dcut - 1)]) {
            const NdFront &f = t.fr[(size_t)id];
            (void)hipMemsetAsync(s->factor + s->poff[(size_t)id], 0, sizeof(double) * (size_t)(f.ld * f.wp), st);
        }
    bool comm_failed = false;
    // every rank has eliminated its subtrees: sum what they left in the fronts of depth dcut - 1 (panel and Schur buffer)
    auto dist_join = [&]() {
        for (hipStream_t q : {sP, sU, sR})
            if (q) (void)hipStreamSynchronize(q);
        // (square Schur buffers: their lower-triangle tiles only, packed into a scratch image; the square buffer if that could
        //  not be had.  Buffers in the packed form are summed where they lie.)
        double *scratch = s->join_scratch;
        long long scap = s->join_scratch_doubles;
        if (!s->sc.packed) {   // every rank must sum windows of the same size: the scratch image only if ALL ranks have the scratch for it
            const double mine_missing = scratch ? 0.0 : 1.0;
            double any_missing = 0.0;
            (void)hipMemcpyAsync(s->part + 1, &mine_missing, sizeof(double), hipMemcpyHostToDevice, st);
            if (plan_allreduce(p, s->part + 1, 1, st) != 0) comm_failed = true;
            (void)hipMemcpyAsync(&any_missing, s->part + 1, sizeof(double), hipMemcpyDeviceToHost, st);
            (void)hipStreamSynchronize(st);
            if (any_missing != 0.0) scap = 0;
        }
        for (int id : t.by_depth[(size_t)(s->dcut - 1)]) {
            const NdFront &f = t.fr[(size_t)id];
            if (plan_allreduce(p, s->factor + s->poff[(size_t)id], f.ld * (long long)f.wp, st) != 0) comm_failed = true;
            if (f.hp == 0) continue;
            const int nt = f.hp / 64;
            const long long tiles = trapezoid_items(nt, nt);
            if (s->sc.packed) {
                if (plan_allreduce(p, s_ptr(s, id), nd_schur_doubles(f, true), st) != 0) comm_failed = true;
            } else if (tiles * 4096 <= scap && !splpak::opt_get("SPLPAK_ND_JOIN_SQUARE")) {
                hipLaunchKernelGGL(nd_tripack_kernel<true>, dim3((unsigned)tiles), dim3(256), 0, st, s_ptr(s, id), f.lds, nt, scratch);
                if (plan_allreduce(p, scratch, tiles * 4096, st) != 0) comm_failed = true;
                hipLaunchKernelGGL(nd_tripack_kernel<false>, dim3((unsigned)tiles), dim3(256), 0, st, s_ptr(s, id), f.lds, nt, scratch);
            } else if (plan_allreduce(p, s_ptr(s, id), f.lds * (long long)f.hp, st) != 0)
                comm_failed = true;
        }
        (void)hipStreamSynchronize(st);
    };
    int qnext = 0;
    (void)hipEventRecord(s->ev0, st);
    for (hipStream_t q : {sP, sU})
        if (q != st) (void)hipStreamWaitEvent(q, s->ev0, 0);
    if (sR) (void)hipStreamWaitEvent(sR, s->ev0, 0);
    auto ensure_events = [&](int steps) {
        for (auto *v : {&s->evT, &s->evI, &s->evW})
            while ((int)v->size() < steps) {
                hipEvent_t e;
                (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
                v->push_back(e);
            }
    };
    // one block step of a stage's chain: potrf (on the reserved CUs when pinned) -> panel solve -> panel update, on the
    // chain stream; the Schur passes that become ready go to the update stream
    // (pin_potrf: the diagonal blocks go to the reserved CUs -- only once a Schur pass of the stage is running beside the chain:
    //  before the first one the chip is idle, and 16 blocks on 8 reserved CUs are two rounds of 140 us where one would do)
    std::function<void()> after_solve;                   // (set for one call: launched on the update stream behind the step's panel solve)
    int la_evt = -1;                                     // step whose next-diagonal-block update is marked by evW (look-ahead inside a group)
    auto chain_step = [&](int stg, int k, bool pinned, bool pin_potrf) {
        hipStream_t sC = sP;
        const Launch &lp = s->l_potrf[(size_t)stg][(size_t)k], &lt = s->l_trsm[(size_t)stg][(size_t)k];
        const Launch &lu = s->l_upd[(size_t)stg][(size_t)k], &ls = s->l_schur[(size_t)stg][(size_t)k];
        const Launch &lf0 = s->l_fin[0][(size_t)stg][(size_t)k], &lf1 = s->l_fin[1][(size_t)stg][(size_t)k];
        const bool ahead = pinned && k > 0 && la_evt == k - 1;    // these diagonal blocks were updated before the rest of step k - 1
        if (pinned && (pin_potrf || ahead)) {           // two event hops: chain -> reserved CUs -> chain
            if (ahead) (void)hipStreamWaitEvent(sR, s->evW[(size_t)(k - 1)], 0);
            else {
                (void)hipEventRecord(s->evR0, sC);
                (void)hipStreamWaitEvent(sR, s->evR0, 0);
            }
            launch_potrf(s, lp, sR, info_dev, minpiv_dev);
            (void)hipEventRecord(s->evI[(size_t)k], sR);
            (void)hipStreamWaitEvent(sC, s->evI[(size_t)k], 0);
        } else
            launch_potrf(s, lp, sC, info_dev, minpiv_dev);
        if (lt.count)
            hipLaunchKernelGGL(nd_trsm_kernel, dim3(lt.grid), dim3(64), 0, sC, (const TrsmJob *)(s->trsm.dev + lt.first), lt.count);
        if (after_solve) {
            (void)hipEventRecord(s->evU, sC);
            (void)hipStreamWaitEvent(sU, s->evU, 0);
            after_solve();
            after_solve = nullptr;
        }
        if ((ls.count || lf0.count || lf1.count) && sU != sC) {
            (void)hipEventRecord(s->evT[(size_t)k], sC);
            (void)hipStreamWaitEvent(sU, s->evT[(size_t)k], 0);
        }
        launch_syrk(s, s->upd, lu, sC, stats, timing, false, pinned, qnext);
        if (s->chain_la[(size_t)stg] && s->l_updr[(size_t)stg][(size_t)k].count) {      // the rest of the in-group update, behind the next diagonal block
            if (pinned) {
                (void)hipEventRecord(s->evW[(size_t)k], sC);
                la_evt = k;
            }
            launch_syrk(s, s->updr, s->l_updr[(size_t)stg][(size_t)k], sC, stats, timing, false, pinned, qnext);
        }
        launch_syrk(s, s->updo, s->l_updo[(size_t)stg][(size_t)k], sC, stats, timing, false, pinned, qnext);     // the group's outer panel pass
        launch_syrk(s, s->schur, ls, sU, stats, timing, true, pinned, qnext);
        launch_syrk(s, s->fin[0], lf0, sU, stats, timing, true, pinned, qnext);      // final passes, fused with the extend-add:
        launch_syrk(s, s->fin[1], lf1, sU, stats, timing, true, pinned, qnext);      // children of slot 0, then of slot 1
    };
    // the root: no Schur buffer to hide its chain behind, hence the look-ahead split
    auto root_stage = [&](int stg) {
        const int steps = (int)s->l_potrf[(size_t)stg].size();
        ensure_events(steps);
        const bool pinned = sR != nullptr && s->nres > 0 && steps > 0 && (int)s->l_potrf[(size_t)stg][0].grid <= pin_rounds * s->nres;
        const bool la = s->lookahead[(size_t)stg] != 0, la2 = s->lookahead[(size_t)stg] == 2;
        for (int k = 0; k < steps; ++k) {
            if (!la) { chain_step(stg, k, pinned, true); continue; }
            const Launch &lp = s->l_potrf[(size_t)stg][(size_t)k], &lt = s->l_trsm[(size_t)stg][(size_t)k], &ltb = s->l_trsmb[(size_t)stg][(size_t)k];
            const Launch &lu = s->l_upd[(size_t)stg][(size_t)k], &lur = s->l_updr[(size_t)stg][(size_t)k];
            if (pinned) {
                (void)hipEventRecord(s->evR0, sP);
                (void)hipStreamWaitEvent(sR, s->evR0, 0);
                launch_potrf(s, lp, sR, info_dev, minpiv_dev);
                (void)hipEventRecord(s->evI[(size_t)k], sR);
                (void)hipStreamWaitEvent(sP, s->evI[(size_t)k], 0);
            } else
                launch_potrf(s, lp, sP, info_dev, minpiv_dev);
            if (ltb.count) {            // the panel rows beyond the next diagonal block are solved beside the chain
                if (sU != sP) {
                    if (!pinned) (void)hipEventRecord(s->evI[(size_t)k], sP);
                    (void)hipStreamWaitEvent(sU, s->evI[(size_t)k], 0);
                }
                hipLaunchKernelGGL(nd_trsm_kernel, dim3(ltb.grid), dim3(64), 0, sU, (const TrsmJob *)(s->trsmb.dev + ltb.first), ltb.count);
            }
            // everything of step k - 1 that is not the next diagonal block ran beside the chain; the panel rows this step
            // solves and the block it updates were last written there.  (la2: block column k was completed on the chain by step
            // k - 1 -- which waited for the trailing pass of step k - 2 --, so the whole panel is solved here, beside the trailing
            // pass of step k - 1; only the update of block column k + 1 below has to wait for that pass)
            if (!la2 && k > 0 && sU != sP) (void)hipStreamWaitEvent(sP, s->evW[(size_t)(k - 1)], 0);
            if (lt.count)
                hipLaunchKernelGGL(nd_trsm_kernel, dim3(lt.grid), dim3(64), 0, sP, (const TrsmJob *)(s->trsm.dev + lt.first), lt.count);
            if (lur.count && sU != sP) {
                (void)hipEventRecord(s->evT[(size_t)k], sP);
                (void)hipStreamWaitEvent(sU, s->evT[(size_t)k], 0);
            }
            if (la2 && k > 0 && sU != sP) (void)hipStreamWaitEvent(sP, s->evW[(size_t)(k - 1)], 0);
            launch_syrk(s, s->upd, lu, sP, stats, timing, false, pinned, qnext);
            launch_syrk(s, s->updr, lur, sU, stats, timing, false, pinned, qnext);
            if (sU != sP) (void)hipEventRecord(s->evW[(size_t)k], sU);
        }
        if (sU != sP) {
            (void)hipEventRecord(s->evU, sU);
            (void)hipStreamWaitEvent(sP, s->evU, 0);
        }
    };
    // ---- the stages in schedule order (children before parents).  A rank of a one-process multi-GPU fit eliminates its
    // subtrees here and the fronts above them in the top phase, together with the other ranks (ndtop.inc).
    bool joined = false;
    for (int i = 0; i < ns; ++i) {
        const NdStage &S = s->sc.st[(size_t)i];
        if (s->dist && !joined && S.depth <= s->dcut - 1) { dist_join(); joined = true; }
        // what comes alive with this stage: the Schur buffers are zeroed, the panels written (zeros + entries), on the update
        // stream.  2.6 .. 3.5 GB of stores per stage at 64^3: beside the stage's FIRST diagonal blocks they made those 0.93 ms
        // instead of 0.3 -- so they are launched behind the first block step of the chain (beside its panel update), when the
        // update stream has nothing to do before that anyway (round 5)
        auto stage_prep = [&]() {
            for (int x : s->starts[(size_t)i]) zero_block(x, sU);
            if (s->staged_init && !s->dist)
                for (int x : s->istarts[(size_t)i]) {
                    nd_init_stage(s, p, x, sU);
                    if (s->sc.st[(size_t)x].dep < 0 && sU != sP) (void)hipEventRecord(s->evP[(size_t)x], sU);
                }
        };
        const int nsteps_i = (int)s->l_potrf[(size_t)i].size();
        const bool prep_late = splpak::opt_get("SPLPAK_ND_PREP_EARLY") == nullptr;
        const bool defer = prep_late && i > 0 && i != s->root_stage && sU != sP && nsteps_i >= 2 && !s->l_schur[(size_t)i][0].count &&
                           !s->l_fin[0][(size_t)i][0].count && !s->l_fin[1][(size_t)i][0].count;
        if (i > 0) {
            if (!defer) stage_prep();
            if (s->staged_init && !s->dist && S.dep < 0 && sU != sP) (void)hipStreamWaitEvent(sP, s->evP[(size_t)i], 0);   // (its panels were written one stage ago)
        }
        // the fronts' children have added their Schur complements (their last passes run on the update stream)
        if (s->fused && S.dep >= 0 && sU != sP) (void)hipStreamWaitEvent(sP, s->evF[(size_t)S.dep], 0);
        if (i == s->root_stage) {
            la_evt = -1;
            root_stage(i);
            continue;
        }
        const int steps = (int)s->l_potrf[(size_t)i].size();
        ensure_events(steps);
        // (a stage with look-ahead inside its groups uses the reserved CUs whatever the number of rounds: its diagonal blocks are
        //  factored beside the rest of the previous step's update)
        const bool pinned = sR != nullptr && s->nres > 0 && steps > 0 &&
                            ((int)s->l_potrf[(size_t)i][0].grid <= pin_rounds * s->nres || s->chain_la[(size_t)i]);
        la_evt = -1;
        const bool unpin_first = splpak::opt_get("SPLPAK_ND_PIN_FIRST") == nullptr;
        bool pass_running = !unpin_first;
        for (int k = 0; k < steps; ++k) {
            if (k == 0 && defer) after_solve = stage_prep;        // (behind the first diagonal blocks and panel solve of the chain)
            chain_step(i, k, pinned, pass_running);
            if (s->l_schur[(size_t)i][(size_t)k].count || s->l_fin[0][(size_t)i][(size_t)k].count || s->l_fin[1][(size_t)i][(size_t)k].count) pass_running = true;
        }
        if (s->fused) (void)hipEventRecord(s->evF[(size_t)i], sU);     // (stream order: the stage's last passes are behind it)
        else {                  // separate extend-add launches: the stage's passes, then slot 0, then slot 1
            if (sU != sP) {
                (void)hipEventRecord(s->evU, sU);
                (void)hipStreamWaitEvent(sP, s->evU, 0);
            }
            for (int sl = 0; sl < 2; ++sl) {
                const Launch &la = s->l_add[sl][(size_t)i];
                if (la.count)
                    hipLaunchKernelGGL(nd_extend_add_kernel, dim3(la.grid), dim3(256), 0, sP, (const AddJob *)(s->add.dev + la.first), la.count);
            }
            (void)hipEventRecord(s->evE[(size_t)i], sP);       // (the arena blocks of this stage are reused on the update stream)
            if (sU != sP) (void)hipStreamWaitEvent(sU, s->evE[(size_t)i], 0);
        }
    }
    if (s->dist && !joined && s->dcut >= 1) dist_join();
    hipError_t top_err = hipSuccess;
    if (s->mdist) {
        ++s->fgen;
        top_err = nd_top_factor(s, st, info_dev, minpiv_dev, stats, timing);
    }
    // what is alive when the first stage starts is zeroed for the NEXT fit here, beside the tail of this one (waited for
    // through evZlast; multi-GPU: after the top phase, the other ranks have pulled the subtree roots' Schur complements)
    if (top_err == hipSuccess && ns > 0)
        for (int x : s->starts[0]) zero_block(x, sU);
    // inverses of all diagonal blocks (the solves' operands)
    if (s->ntrinv > 0)
        hipLaunchKernelGGL(nd_trinv_kernel, dim3(NBLK / 16, (unsigned)s->ntrinv), dim3(64), 0, sP, (const TrinvJob *)s->trinv.dev);
    (void)hipEventRecord(s->evJ, sP);
    if (sP != st) (void)hipStreamWaitEvent(st, s->evJ, 0);
    if (s->mdist && top_err == hipSuccess) top_err = nd_top_pivots(s, st, info_dev, minpiv_dev);
    if (s->dist) {      // (a failed pivot poisons the fronts above it with NaN, so every rank fails anyway; this makes it explicit)
        hipLaunchKernelGGL(nd_flag_kernel, dim3(1), dim3(1), 0, st, (const int *)info_dev, s->part, 0);
        if (plan_allreduce(p, s->part, 1, st) != 0) comm_failed = true;
        hipLaunchKernelGGL(nd_unflag_kernel, dim3(1), dim3(1), 0, st, info_dev, (const double *)s->part);
    }
    if (sU != st) {                                   // (the zero launches for the next fit may still be running: it waits for them)
        (void)hipEventRecord(s->evZlast, sU);
        s->zlast_valid = true;
    }
    s->s_clean = true;
    (void)hipEventRecord(s->evDone, st);
    s->used = true;
    hipError_t err = hipGetLastError();
    if (timing) {
        (void)hipEventRecord(s->f1, st);
        (void)hipEventSynchronize(s->f1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, s->f0, s->f1);
        stats->factor_ms = ms;
        for (size_t i = 0; i < (size_t)stats->syrk_launches; ++i) {
            if (hipEventElapsedTime(&ms, s->evA[i], s->evB[i]) == hipSuccess) stats->syrk_ms += ms;
            else (void)hipGetLastError();
        }
    }
    if (err != hipSuccess || top_err != hipSuccess) s->s_clean = false;
    if (comm_failed) return hipErrorUnknown;
    return top_err != hipSuccess ? top_err : err;
}

hipError_t nd_solve(splpak_plan *p, double *x, double *tmp, hipStream_t st, void *user)
{
    (void)tmp;
    NdState *s = static_cast<NdState *>(user);
    NdTree &t = s->t;
    const long long n = t.vec_doubles;
    const unsigned gb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(nd_gather_kernel, dim3(gb), dim3(256), 0, st, n, (const int *)s->rowsrc, (const double *)x, s->V);
    bool comm_failed = false;
    if (s->dist && s->rank != 0)        // the right-hand side of the fronts the subtrees report into comes from rank 0 alone
        for (int id : t.by_depth[(size_t)(s->dcut - 1)]) {
            const NdFront &f = t.fr[(size_t)id];
            (void)hipMemsetAsync(s->V + f.vofs, 0, sizeof(double) * (size_t)f.fp, st);
        }
    // (one-process multi-GPU fit: the subtrees here, the fronts above them step by step with the other ranks -- ndtop.inc)
    const int dlow = s->mdist ? s->pt.dcut : 0;
    if (s->mdist) ++s->sgen;
    // forward, bottom-up
    for (int d = t.maxdepth; d >= dlow; --d) {
        if (d < t.maxdepth)
            for (int sl = 0; sl < 2; ++sl) {
                const Launch &lm = s->l_mapslot[sl][(size_t)(d + 1)];
                if (lm.count) hipLaunchKernelGGL(nd_map_kernel<false>, dim3(8, lm.grid), dim3(256), 0, st, (const MapJob *)(s->map.dev + lm.first));
            }
        if (s->dist && d == s->dcut - 1)        // the subtrees' updates of these fronts' vectors, summed over the ranks
            for (int id : t.by_depth[(size_t)d]) {
                const NdFront &f = t.fr[(size_t)id];
                if (plan_allreduce(p, s->V + f.vofs, f.fp, st) != 0) comm_failed = true;
            }
        const int steps = (int)s->l_mv[(size_t)d].size();
        for (int k = 0; k < steps; ++k) {
            const Launch &lm = s->l_mv[(size_t)d][(size_t)k], &lf = s->l_fwd[(size_t)d][(size_t)k];
            hipLaunchKernelGGL(nd_mv_kernel, dim3(16, lm.grid), dim3(256), 0, st, (const MvJob *)(s->mv.dev + lm.first));
            if (lf.count) hipLaunchKernelGGL(nd_fwd_kernel, dim3(lf.grid), dim3(512), 0, st, (const FwdJob *)(s->fwd.dev + lf.first), lf.count);
        }
    }
    if (s->mdist) {
        hipError_t e = nd_top_forward(s, st);
        if (e == hipSuccess) e = nd_top_backward(s, st);
        if (e != hipSuccess) return e;
    }
    // backward, top-down
    for (int d = dlow; d <= t.maxdepth; ++d) {
        if (d >= 1 && d > dlow) {
            const Launch &lm = s->l_mapall[(size_t)d];
            if (lm.count) hipLaunchKernelGGL(nd_map_kernel<true>, dim3(8, lm.grid), dim3(256), 0, st, (const MapJob *)(s->map.dev + lm.first));
        }
        const int steps = (int)s->l_mv[(size_t)d].size();
        for (int k = steps - 1; k >= 0; --k) {
            const Launch &ld = s->l_dot[(size_t)d][(size_t)k], &lb = s->l_bwd[(size_t)d][(size_t)k];
            if (ld.count) hipLaunchKernelGGL(nd_dot_kernel, dim3(ld.grid), dim3(256), 0, st, (const DotJob *)(s->dot.dev + ld.first), ld.count);
            hipLaunchKernelGGL(nd_bwd_kernel, dim3(16, lb.grid), dim3(256), 0, st, (const BwdJob *)(s->bwd.dev + lb.first));
        }
    }
    if (s->dist || s->mdist) {
        // every rank reports the variables of its own subtrees (rank 0 also those of the top of the tree); the sum is the solution
        (void)hipMemsetAsync(x, 0, sizeof(double) * (size_t)p->g.ncol, st);
        hipLaunchKernelGGL(nd_scatter_kernel, dim3(gb), dim3(256), 0, st, n, (const int *)s->rowsrc_out, (const double *)s->V, x);
        if (plan_allreduce(p, x, p->g.ncol, st) != 0) comm_failed = true;
    } else
        hipLaunchKernelGGL(nd_scatter_kernel, dim3(gb), dim3(256), 0, st, n, (const int *)s->rowsrc, (const double *)s->V, x);
    if (comm_failed) return hipErrorUnknown;
    return hipGetLastError();
}

template <typename T>
void nd_free_dev(NdState *s, T **ptr)
{
    if (!*ptr) return;
    for (size_t i = 0; i < s->owned.size(); ++i)
        if (s->owned[i] == (void *)*ptr) { s->owned.erase(s->owned.begin() + (long)i); break; }
    (void)hipFree(*ptr);
    *ptr = nullptr;
}

// device copies of the job tables (before they are uploaded again for another set of ranks)
void nd_free_jobs(NdState *s)
{
    nd_free_dev(s, &s->potrf.dev); nd_free_dev(s, &s->trsm.dev); nd_free_dev(s, &s->trsmb.dev); nd_free_dev(s, &s->upd.dev);
    nd_free_dev(s, &s->updr.dev); nd_free_dev(s, &s->updo.dev); nd_free_dev(s, &s->fin[0].dev); nd_free_dev(s, &s->fin[1].dev);
    nd_free_dev(s, &s->schur.dev); nd_free_dev(s, &s->trinv.dev); nd_free_dev(s, &s->add.dev); nd_free_dev(s, &s->zero.dev);
    nd_free_dev(s, &s->init.dev);
    nd_free_dev(s, &s->mv.dev); nd_free_dev(s, &s->fwd.dev); nd_free_dev(s, &s->dot.dev); nd_free_dev(s, &s->bwd.dev);
    nd_free_dev(s, &s->map.dev); nd_free_dev(s, &s->rowsrc_out);
}

bool nd_upload_jobs(NdState *s)
{
    return nd_upload(s, &s->potrf.dev, s->potrf.host) && nd_upload(s, &s->trsm.dev, s->trsm.host) && nd_upload(s, &s->trsmb.dev, s->trsmb.host) &&
           nd_upload(s, &s->upd.dev, s->upd.host) && nd_upload(s, &s->updr.dev, s->updr.host) && nd_upload(s, &s->updo.dev, s->updo.host) &&
           nd_upload(s, &s->fin[0].dev, s->fin[0].host) && nd_upload(s, &s->fin[1].dev, s->fin[1].host) && nd_upload(s, &s->schur.dev, s->schur.host) &&
           nd_upload(s, &s->trinv.dev, s->trinv.host) && nd_upload(s, &s->add.dev, s->add.host) && nd_upload(s, &s->zero.dev, s->zero.host) &&
           nd_upload(s, &s->init.dev, s->init.host) &&
           nd_upload(s, &s->mv.dev, s->mv.host) && nd_upload(s, &s->fwd.dev, s->fwd.host) && nd_upload(s, &s->dot.dev, s->dot.host) &&
           nd_upload(s, &s->bwd.dev, s->bwd.host) && nd_upload(s, &s->map.dev, s->map.host);
}

// (Re)builds the elimination schedule for the fronts in s->mine and sizes the Schur arena for it.  cut < 0: chosen here --
// the level-by-level order (cut = 0: the largest batches) if its arena fits beside `other_bytes` of further allocations in the
// free device memory, otherwise the smallest cut that does (SPLPAK_ND_CUT overrides).
bool nd_make_schedule(NdState *s, int cut, size_t other_bytes)
{
    NdTree &t = s->t;
    const bool packed = !s->mdist && splpak::opt_get("SPLPAK_ND_SQUARE") == nullptr;
    const int dlow = s->mdist ? s->pt.dcut : 0;
    s->needs.assign(t.fr.size(), 0);
    for (size_t id = 0; id < t.fr.size(); ++id) {
        const NdFront &f = t.fr[id];
        const bool boundary = s->mdist && f.depth == s->pt.dcut;
        s->needs[id] = f.hp > 0 && !(s->fused && f.child[0] < 0 && f.nsteps <= s->schur_kb && !boundary) ? 1 : 0;
    }
    if (cut < 0) {
        cut = 0;
        if (const char *e = splpak::opt_get("SPLPAK_ND_CUT")) cut = std::max(0, std::min(t.maxdepth, atoi(e)));
        else if (!s->mdist) {
            size_t fr = 0, tot = 0;
            if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); fr = 0; }
            const double room = (double)fr - (double)other_bytes - std::max(1.0e9, 0.03 * (double)tot);
            long long best = -1;
            int best_cut = 0;
            for (int c = 0; c <= std::min(t.maxdepth, 6); ++c) {
                nd_schedule(t, c, packed, &s->mine, &s->needs, dlow, s->sc);
                if (best < 0 || s->sc.arena < best) { best = s->sc.arena; best_cut = c; }
                if (fr == 0 || 8.0 * (double)s->sc.arena <= room) { best_cut = c; break; }
            }
            cut = best_cut;         // (nothing fits: the smallest arena -- the allocation then fails with the byte counts in the message)
        }
    }
    // (the half-stages: single-GPU plans in the level-by-level order)
    const int halves = (cut == 0 && !s->mdist && !s->dist && splpak::opt_get("SPLPAK_ND_HALVES")) ? atoi(splpak::opt_get("SPLPAK_ND_HALVES")) : 0;
    nd_schedule(t, cut, packed, &s->mine, &s->needs, dlow, s->sc, halves);
    const int ns = (int)s->sc.st.size();
    s->starts.assign((size_t)std::max(ns, 1), {});
    s->root_stage = -1;
    for (int i = 0; i < ns; ++i) {
        const NdStage &S = s->sc.st[(size_t)i];
        s->starts[(size_t)S.first].push_back(i);
        if (S.ids.size() == 1 && t.fr[(size_t)S.ids[0]].parent < 0 && !s->mdist) s->root_stage = i;
    }
    if (splpak::opt_get("SPLPAK_ND_DEBUG_STAGES"))
        for (int i = 0; i < ns; ++i) {
            const NdStage &S = s->sc.st[(size_t)i];
            long long pd = 0, cols = 0;
            for (int id : S.ids) { const NdFront &f = t.fr[(size_t)id]; pd += f.ld * (long long)f.wp; cols += f.wp; }
            fprintf(stderr, "[nd stage %d] depth %d, %zu fronts, first %d, dep %d, panels %.3f GB in %lld columns, Schur %.3f GB\n", i, S.depth, S.ids.size(), S.first, S.dep,
                    8e-9 * (double)pd, cols, 8e-9 * (double)S.doubles);
        }
    s->istarts.assign((size_t)std::max(ns, 1), {});
    for (int i = 0; i < ns; ++i) {
        const NdStage &S = s->sc.st[(size_t)i];
        s->istarts[(size_t)((S.first == i && i > 0) ? i - 1 : S.first)].push_back(i);
    }
    for (auto *v : {&s->evF, &s->evE, &s->evP})
        while ((int)v->size() < ns + 1) {
            hipEvent_t e = nullptr;
            (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
            v->push_back(e);
        }
    if (s->sc.arena + 64 > s->sarena_doubles) {
        if (s->sarena) {
            (void)hipDeviceSynchronize();
            nd_free_dev(s, &s->sarena);
            s->owned_bytes -= sizeof(double) * (size_t)s->sarena_doubles;
        }
        s->sarena_doubles = 0;
        if (!nd_alloc(s, &s->sarena, (size_t)s->sc.arena + 64)) return false;
        s->sarena_doubles = s->sc.arena + 64;
    }
    return true;
}

// Ownership of the fronts for `world` ranks and the job tables that follow from it (see NdState::dist).  Called when the
// sharded fit's ranks become known (splpak_plan_set_allreduce comes after the plan); SPLPAK_ND_DIST=0 opts out.
int nd_set_ranks_impl(splpak_plan *p, int rank, int world)
{
    NdState *s = static_cast<NdState *>(p->fn_user);
    if (!s || s->mdist) return 0;
    NdTree &t = s->t;
    int dcut = 0;
    while ((1 << dcut) < world) ++dcut;
    const char *sw = splpak::opt_get("SPLPAK_ND_DIST");                  // 0 = every rank factors everything (round 2's form)
    // (the join sums front panels, Schur buffers and solve vectors that live outside the plan's communication buffer: only with a
    //  hook that declared it accepts any device pointer -- SPLPAK_AR_ANY_POINTER; round-3 advice)
    const bool want = world > 1 && p->ar != nullptr && (p->ar_flags & SPLPAK_AR_ANY_POINTER) != 0 && !(sw && atoi(sw) == 0) && dcut >= 1 &&
                      dcut <= t.maxdepth && (s->sc.cut == 0 || s->dist);
    if (!want && !s->dist) return 0;
    (void)hipDeviceSynchronize();
    s->dist = want;
    s->world = world;
    s->rank = rank;
    s->dcut = want ? dcut : 0;
    s->mine.assign(t.fr.size(), 1);
    if (want) {
        std::vector<int> slot_of(t.fr.size(), -1);          // index of the depth-dcut ancestor among the fronts of that depth
        const std::vector<int> &cut = t.by_depth[(size_t)dcut];
        for (size_t i = 0; i < cut.size(); ++i) slot_of[(size_t)cut[i]] = (int)i;
        for (int id = (int)t.fr.size() - 1; id >= 0; --id) {   // parents have larger ids than their children (postorder)
            const NdFront &f = t.fr[(size_t)id];
            if (f.depth > dcut) slot_of[(size_t)id] = slot_of[(size_t)f.parent];
            if (f.depth >= dcut) s->mine[(size_t)id] = (slot_of[(size_t)id] % world) == rank ? 1 : 0;
        }
    }
    if (!nd_make_schedule(s, want ? 0 : -1, 0)) return SPLPAK_E_NOMEM;
    for (auto *h : {&s->upd, &s->updr, &s->updo, &s->schur, &s->fin[0], &s->fin[1]}) h->host.clear();
    s->potrf.host.clear(); s->trsm.host.clear(); s->trsmb.host.clear(); s->trinv.host.clear(); s->add.host.clear(); s->zero.host.clear(); s->init.host.clear();
    s->mv.host.clear(); s->fwd.host.clear(); s->dot.host.clear(); s->bwd.host.clear(); s->map.host.clear();
    if (!nd_build_jobs(s)) { set_error("nested dissection: job tables (ranks)"); return SPLPAK_E_UNSUPPORTED; }
    nd_free_jobs(s);                                           // the superseded device tables
    if (!nd_upload_jobs(s)) return SPLPAK_E_NOMEM;
    std::vector<int> out(s->rowsrc_host);
    for (size_t id = 0; id < t.fr.size(); ++id) {
        const NdFront &f = t.fr[id];
        const bool report = want ? (f.depth >= dcut ? s->mine[id] != 0 : rank == 0) : true;
        if (!report)
            for (int r = 0; r < f.fp; ++r) out[(size_t)(f.vofs + r)] = -1;
    }
    if (!nd_upload(s, &s->rowsrc_out, out)) return SPLPAK_E_NOMEM;
    if (want && !s->sc.packed) {
        long long need = 0;
        for (int id : t.by_depth[(size_t)(dcut - 1)]) {
            const long long nt = t.fr[(size_t)id].hp / 64;
            need = std::max(need, trapezoid_items(nt, nt) * 4096);
        }
        if (need > s->join_scratch_doubles) {
            double *q = nullptr;
            if (hipMalloc(&q, sizeof(double) * (size_t)need) == hipSuccess) {      // (no room: the join sums the square buffers)
                s->owned.push_back(q);
                s->join_scratch = q;
                s->join_scratch_doubles = need;
            } else
                (void)hipGetLastError();
        }
    }
    s->s_clean = false;
    if (splpak::opt_get("SPLPAK_DEBUG")) {
        int nm = 0;
        for (char c : s->mine) nm += c;
        fprintf(stderr, "[splpak] nested dissection: rank %d of %d eliminates %d of %zu fronts (subtrees below depth %d)\n", rank, world, nm, t.fr.size(), dcut);
    }
    return 0;
}

}  // namespace

// SPLPAK_ND: 0 = never, 1 = always; otherwise 2-D / 3-D grids of at least 4 096 columns and 4-D grids of at least 20 000.
// Measured on MI355X (tools/nd_crossover.sh, fit time band -> nested dissection): 2-D 48^2 2.08 -> 2.27 ms (band stays),
// 64^2 (BASELINE config 2) 3.29 -> 2.46, 90^2 5.7 -> 4.3, 128^2 10.4 -> 5.1, 256^2 41.5 -> 13.1; 3-D 16^3 4.3 -> 3.7, 20^3 6.6 ->
// 5.9, 24^3 11.4 -> 9.0, 32^3 26.3 -> 17.8, 40^3 67.7 -> 42.5, 48^3 166 -> 85, 64^3 831 -> 280; 4-D 8^4 10.9 -> 11.2 and 10^4
// 19.5 -> 20.3 (band stays), 12^4 41.9 -> 41.0, 16^4 239 -> 184, 24^4 10.5 s -> 4.9 s.
int nd_set_ranks(splpak_plan *p, int rank, int world) { return (p && p->fn_code == 4) ? nd_set_ranks_impl(p, rank, world) : 0; }

bool nd_wanted_for(int ndim, const int *nodes, const double *xmin, const double *xmax)
{
    Grid g;
    if (build_grid(ndim, nodes, xmin, xmax, g, nullptr, splpak::opt_get("SPLPAK_NO_REORDER") == nullptr) != 0) return false;
    Band b{};
    return nd_wanted(g, b);
}

bool nd_wanted(const Grid &g, const Band &band)
{
    (void)band;
    if (const char *e = splpak::opt_get("SPLPAK_ND")) return atoi(e) != 0;
    if (g.ndim == 2 || g.ndim == 3) return g.ncol >= 4096;
    return g.ndim == 4 && g.ncol >= 20000;
}

// Installs the nested-dissection factorisation on a single-GPU plan: builds the tree, allocates the arenas,
// uploads the tables.  Returns 0, or an SPLPAK_E_* code (the plan is then unusable).  *factor_arena /
// *factor_doubles: the factor storage, idle until the half stencil is assembled (the Gram scratch may live there).
int nd_attach(splpak_plan *p, double **factor_arena, long long *factor_doubles, NdGroup *grp, int rank)
{
    NdState *s = new NdState();
    (void)hipGetDevice(&s->device);
    p->fn_user = s;
    p->fn_destroy = nd_destroy;
    p->fn_bytes = nd_bytes;
    if (!nd_build(p->g, s->t, nd_default_split_min(p->g.ndim))) { set_error("nested dissection: inconsistent tree"); return SPLPAK_E_BADARG; }
    NdTree &t = s->t;
    // one-process multi-GPU fit: this plan is rank `rank` of the group
    s->grp = grp;
    s->mrank = rank;
    s->mdist = grp != nullptr && grp->R > 1;
    nd_partition(t, s->mdist ? grp->R : 1, s->mdist ? grp->chunk : 1, s->pt);
    if (s->mdist && s->pt.dcut < 1) s->mdist = false;                 // (a tree of one front: nothing to distribute)
    if (grp) {
        if (rank < 0 || rank >= grp->R) { set_error("nested dissection: bad rank"); return SPLPAK_E_BADARG; }
        grp->st[(size_t)rank] = s;
    }
    if (grp && grp->R > 1 && !s->mdist) { set_error("nested dissection: the tree of this grid has a single front; use one GPU"); return SPLPAK_E_UNSUPPORTED; }
    const NdPartition &pt = s->pt;
    s->mine.assign(t.fr.size(), 1);
    if (s->mdist)
        for (size_t id = 0; id < t.fr.size(); ++id) s->mine[id] = pt.owner[id] == rank ? 1 : 0;
    // (round 3 also knew two PIPELINES -- the two subtrees below the root side by side on two chain streams, SPLPAK_ND_PIPES=2:
    //  235.0 against 234.9 ms per factorisation at 64^3; removed in round 5, the postorder schedule gives the same overlap)
    s->fused = splpak::opt_get("SPLPAK_ND_NO_FUSE") == nullptr;
    if (const char *e = splpak::opt_get("SPLPAK_ND_KB")) s->schur_kb = std::max(1, std::min(4, atoi(e)));
    // this rank's storage: panels and Schur buffers of the fronts it eliminates, then its block columns of the top fronts
    s->poff.assign(t.fr.size(), -1);
    s->lblk.assign(t.fr.size(), -1);
    for (size_t id = 0; id < t.fr.size(); ++id) {
        const NdFront &f = t.fr[id];
        if (!s->mine[id]) continue;
        s->poff[id] = s->factor_doubles;
        s->factor_doubles += f.ld * (long long)f.wp;
        s->lblk[id] = s->nblocks;
        s->nblocks += f.nsteps;
    }
    std::vector<long long> padwhere;
    long long max_fp = 0;
    if (s->mdist) {
        s->tbase.assign(pt.top.size(), 0);
        for (size_t ti = 0; ti < pt.top.size(); ++ti) {
            const NdFront &f = t.fr[(size_t)pt.top[ti]];
            s->tbase[ti] = (long long)s->topcol.size();
            max_fp = std::max(max_fp, (long long)f.fp);
            const int nb = top_nblocks(f);
            for (int J = 0; J < nb; ++J) {
                TopColDev tc{-1, top_block_ld(f, J)};
                int lb = -1;
                if (top_owner(pt, J) == rank) {
                    tc.off = s->factor_doubles;
                    s->factor_doubles += tc.ld * top_block_cols(f, J);
                    if (J < f.nsteps) lb = s->nblocks++;
                    for (int c = J * 256; c < J * 256 + top_block_cols(f, J); ++c)            // identity on the padding of the own columns
                        if (c >= f.w && c < f.wp) padwhere.push_back(tc.off + (long long)(c - J * 256) * (tc.ld + 1));
                }
                s->topcol.push_back(tc);
                s->toplblk.push_back(lb);
            }
        }
    }
    // Everything but the Schur arena first; the schedule is then chosen for the device memory that is left (the plan still
    // allocates its communication buffer -- half stencil, right-hand side, histogram, residual -- and two vectors after this)
    bool ok = nd_alloc(s, &s->factor, (size_t)s->factor_doubles + 64) && nd_alloc(s, &s->dinv, (size_t)s->nblocks * 65536) &&
              nd_alloc(s, &s->dinvt, (size_t)s->nblocks * 65536) && nd_alloc(s, &s->inv16, (size_t)s->nblocks * 4096) &&
              nd_alloc(s, &s->V, (size_t)t.vec_doubles) && nd_alloc(s, &s->Y, (size_t)t.vec_doubles) &&
              nd_alloc(s, &s->part, (size_t)(s->part_cap = t.vec_doubles / 4 + 256LL * (long long)t.fr.size() + 4096));
    if (ok) {
        const size_t later = sizeof(double) * ((size_t)p->g.ncol * (size_t)(p->g.hstencil + 8)) + sizeof(int) * 8 * (size_t)t.vec_doubles;
        ok = nd_make_schedule(s, -1, later);
        if (!ok) {
            char buf[320];
            snprintf(buf, sizeof buf, "nested dissection: the Schur arena of %.1f GB (packed lower triangles, schedule cut %d) does not fit beside %.1f GB of factor panels",
                     8e-9 * (double)s->sc.arena, s->sc.cut, 8e-9 * (double)s->factor_doubles);
            set_error(buf);
        }
    }
    if (ok && s->mdist) {
        for (int i = 0; i < 3 && ok; ++i) ok = nd_alloc(s, &s->pbuf[i], (size_t)pt.max_panel + 64);
        s->stagev_doubles = max_fp + 64;
        ok = ok && nd_alloc(s, &s->stagev, (size_t)s->stagev_doubles);
    }
    if (!ok) return SPLPAK_E_NOMEM;
    // tables
    std::vector<int> rowsrc((size_t)t.vec_doubles, -1);
    std::vector<FrontDev> fdev;
    for (size_t id = 0; id < t.fr.size(); ++id) {
        const NdFront &f = t.fr[id];
        for (int r = 0; r < f.w; ++r) rowsrc[(size_t)(f.vofs + r)] = t.ownvar[(size_t)(f.rofs + r)];
        if (s->poff[id] >= 0)
            for (int r = f.w; r < f.wp; ++r) padwhere.push_back(s->poff[id] + r + (long long)r * f.ld);
        const int ti = s->mdist ? pt.top_index[id] : -1;
        fdev.push_back(FrontDev{s->poff[id], f.ld, f.bofs, f.own0, f.w, f.wp, f.h, ti >= 0 ? (int)s->tbase[(size_t)ti] : -1, 0});
    }
    s->npad = (int)padwhere.size();
    {
        int maxpos = -1;
        for (int v : t.pos) maxpos = std::max(maxpos, v);
        std::vector<int> ipos((size_t)(maxpos + 1), -1);
        for (size_t i = 0; i < t.pos.size(); ++i)
            if (t.pos[i] >= 0) ipos[(size_t)t.pos[i]] = (int)i;
        if (!nd_upload(s, &s->ipos, ipos)) return SPLPAK_E_NOMEM;
    }
    s->staged_init = !s->mdist && !(splpak::opt_get("SPLPAK_ND_STAGED_INIT") && atoi(splpak::opt_get("SPLPAK_ND_STAGED_INIT")) == 0);
    ok = nd_upload(s, &s->pos, t.pos) && nd_upload(s, &s->front_of, t.front_of) && nd_upload(s, &s->bpos, t.bpos) &&
         nd_upload(s, &s->pmap, t.pmap) && nd_upload(s, &s->rowsrc, rowsrc) && nd_upload(s, &s->padwhere, padwhere) &&
         nd_upload(s, &s->fdev, fdev) && nd_upload(s, &s->topcol_dev, s->topcol);
    if (!ok) return SPLPAK_E_NOMEM;
    s->rowsrc_host.swap(rowsrc);
    s->full_diag = splpak::opt_get("SPLPAK_ND_FULL_DIAG") != nullptr ? 1 : 0;      // (before the job tables: it enters their flop counts)
    // XCD-aware item map of the Schur passes: on (round 4) -- half the fabric traffic per launch for the same factor bits at
    // +0.2 .. 0.4 % time (SPLPAK_ND_XCD=0: the plain map)
    s->xmode = splpak::opt_get("SPLPAK_ND_XCD") ? atoi(splpak::opt_get("SPLPAK_ND_XCD")) : 1;
    if (!nd_build_jobs(s)) { if (true) set_error("nested dissection: job tables"); return SPLPAK_E_UNSUPPORTED; }
    ok = nd_upload_jobs(s);
    if (!ok) return SPLPAK_E_NOMEM;
    if (s->mdist) {             // what this rank reports into the solution: its subtrees' variables and the top fronts it ends the backward sweep of
        std::vector<int> out(s->rowsrc_host);
        for (size_t id = 0; id < t.fr.size(); ++id) {
            const NdFront &f = t.fr[id];
            const bool report = pt.owner[id] >= 0 ? pt.owner[id] == rank : top_owner(pt, 0) == rank;
            if (!report)
                for (int r = 0; r < f.fp; ++r) out[(size_t)(f.vofs + r)] = -1;
        }
        if (!nd_upload(s, &s->rowsrc_out, out)) return SPLPAK_E_NOMEM;
    }
    // the host copies of the big index arrays are no longer needed
    std::vector<int>().swap(t.ownvar);
    std::vector<int>().swap(t.bvar);
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (const char *e = splpak::opt_get("SPLPAK_ND_DUMMY_STREAMS"))        // (experiment: how the streams fall onto the hardware queues)
        for (int i = 0; i < atoi(e); ++i) { hipStream_t q; (void)hipStreamCreateWithFlags(&q, hipStreamNonBlocking); }
    (void)hipStreamCreateWithPriority(&s->sP, hipStreamNonBlocking, hi);
    (void)hipStreamCreateWithFlags(&s->sU, hipStreamNonBlocking);
    for (hipEvent_t *e : {&s->ev0, &s->evJ, &s->evU, &s->evZlast, &s->evDone, &s->evPre, &s->evTail, &s->evR0}) (void)hipEventCreateWithFlags(e, hipEventDisableTiming);
    // item queues of the update launches (two per step at most)
    s->nqueues = 8 * t.nblocks + 64;
    if (const char *e = splpak::opt_get("SPLPAK_ND_SMALL_GRID")) s->small_grid = atoi(e);
    if (const char *e = splpak::opt_get("SPLPAK_ND_WG4")) s->wg4 = atoi(e);
    if (const char *e = splpak::opt_get("SPLPAK_ND_PINNED_SPLIT")) s->pinned_split = atoi(e);
    s->small_queue = splpak::opt_get("SPLPAK_ND_SMALL_QUEUE") != nullptr;
    if (const char *e = splpak::opt_get("SPLPAK_ND_POTRF_WAVES")) s->potrf_waves = atoi(e);
    if (!nd_alloc(s, &s->queues, (size_t)ND_QSTRIDE * s->nqueues) || !nd_alloc(s, &s->resmap, (size_t)128)) return SPLPAK_E_NOMEM;
    (void)hipMemset(s->resmap, 0, 128 * sizeof(unsigned));
    // A few CUs are left to the diagonal-block factorisations of the upper tree levels: v_mfma_f64 runs on the same
    // pipes as f64 VALU code, and the latency-bound potrf workgroups ran 8x slower (1.26 ms instead of 0.16) beside
    // the update waves (rocprofv3, 64^3).  potrf is pinned to those CUs through a CU-masked stream; the update
    // waves are not masked, they step aside when they find themselves there (nd_syrk_kernel).  Only trees whose
    // upper levels are worth it (>= 8 block steps in the root) pay for the extra stream.
    const int want_res = splpak::opt_get("SPLPAK_ND_RES_CUS") ? atoi(splpak::opt_get("SPLPAK_ND_RES_CUS")) : 8;
    if (want_res > 0 && t.fr[(size_t)t.root].nsteps >= 8 && !splpak::opt_get("SPLPAK_NO_PANEL_CU")) {
        hipDeviceProp_t prop;
        (void)hipGetDeviceProperties(&prop, s->device);
        const int ncu = prop.multiProcessorCount;
        std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
        for (int i = 0; i < want_res && i < ncu; ++i) mask[(size_t)i / 32] |= 1u << (i % 32);
        if (hipExtStreamCreateWithCUMask(&s->sR, (uint32_t)mask.size(), mask.data()) == hipSuccess) {
            hipLaunchKernelGGL(nd_whoami_kernel, dim3(64 * (unsigned)want_res), dim3(64), 0, s->sR, s->resmap);
            unsigned hm[128];
            if (hipStreamSynchronize(s->sR) == hipSuccess && hipMemcpy(hm, s->resmap, sizeof hm, hipMemcpyDeviceToHost) == hipSuccess)
                for (unsigned wv : hm) s->nres += __builtin_popcount(wv);
            if (s->nres == 0 || s->nres > 2 * want_res) {      // the mask did not take: no pinning
                (void)hipStreamDestroy(s->sR);
                s->sR = nullptr;
                s->nres = 0;
                (void)hipMemset(s->resmap, 0, 128 * sizeof(unsigned));
            }
        } else {
            (void)hipGetLastError();
            s->sR = nullptr;
        }
    }
    if (s->mdist) {
        (void)hipStreamCreateWithPriority(&s->sCopy, hipStreamNonBlocking, hi);
        for (auto *v : {&s->evReady, &s->evArr, &s->evCol, &s->evBulk, &s->evSF, &s->evSB}) {
            v->assign((size_t)s->pt.nseq + 1, nullptr);
            for (hipEvent_t &e : *v) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
        }
        s->evAdd.assign(s->pt.top.size() + 1, nullptr);
        for (hipEvent_t &e : s->evAdd) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
        (void)hipEventCreateWithFlags(&s->evSub, hipEventDisableTiming);
        (void)hipEventCreateWithFlags(&s->evTop, hipEventDisableTiming);
        if (!s->sCopy || !s->evSub || !s->evTop) { set_error("nested dissection: stream creation failed"); (void)hipGetLastError(); return SPLPAK_E_NODEVICE; }
    }
    if (!s->sP || !s->sU) { set_error("nested dissection: stream creation failed"); (void)hipGetLastError(); return SPLPAK_E_NODEVICE; }
    p->expand_fn = nd_assemble;
    p->prefit_fn = nd_prefit;
    p->factor_fn = nd_factor;
    p->solve_fn = nd_solve;
    {
        char buf[256];
        snprintf(buf, sizeof buf, "; %zu fronts in %zu stages (schedule cut %d), %.1f GB of factor panels, %.1f GB Schur arena (%s)", t.fr.size(), s->sc.st.size(),
                 s->sc.cut, 8e-9 * (double)s->factor_doubles, 8e-9 * (double)s->sarena_doubles, s->sc.packed ? "packed lower triangles" : "square buffers");
        s->desc = s->mdist ? "nested-dissection multifrontal Cholesky distributed over several GPUs: subtrees per GPU, top fronts by block columns (csrc/ndchol.hip, csrc/ndtop.inc)"
                           : "nested-dissection multifrontal Cholesky (csrc/ndtree.hip, csrc/ndchol.hip)";
        s->desc += buf;
    }
    p->fn_name = s->desc.c_str();
    p->fn_code = s->mdist ? 5 : 4;
    p->factor_flop = t.flop;                 // (what the iteration in front of this factorisation may spend is weighed against it, plan.hip)
    if (factor_arena) *factor_arena = s->factor;
    if (factor_doubles) *factor_doubles = s->factor_doubles;
    if (splpak::opt_get("SPLPAK_DEBUG"))
        fprintf(stderr, "[splpak] nested dissection%s: %zu fronts, depth %d, factor %.2f GB, Schur arena %.2f GB (%s, %zu stages, cut %d), %.3e flop, %d reserved CUs\n",
                s->mdist ? " (one rank of a multi-GPU fit)" : "", t.fr.size(), t.maxdepth, 8e-9 * (double)s->factor_doubles, 8e-9 * (double)s->sarena_doubles,
                s->sc.packed ? "packed" : "square", s->sc.st.size(), s->sc.cut, t.flop, s->nres);
    return 0;
}

// ---- batched Cholesky of independent dense 256 x 256 blocks (the block-Jacobi component of the iterative solve, pcg.hip) ----------
// The diagonal-block kernels of the fronts on blocks that belong to no front: `blocks` holds nb column-major 256 x 256 matrices (lower
// triangle read, L written in place), inv16 nb x 16 leaf inverses of 16 x 16, dinv / dinvt the inverse of L row-major and its
// transpose.  The job tables live in device memory the caller provides (block_chol_job_bytes) and are written once per plan.
size_t block_chol_job_bytes(int nb) { return (size_t)nb * (sizeof(PotrfJob) + sizeof(TrinvJob)) + 256; }

hipError_t block_chol_prepare(void *jobs_dev, int nb, double *blocks, double *inv16, double *dinv, double *dinvt, const int *ncols_host)
{
    std::vector<PotrfJob> pj((size_t)nb);
    std::vector<TrinvJob> tj((size_t)nb);
    for (int b = 0; b < nb; ++b) {
        double *A = blocks + (size_t)b * NBLK * NBLK, *iv = inv16 + (size_t)b * 16 * 256;
        pj[(size_t)b] = PotrfJob{A, iv, NBLK, b * NBLK, ncols_host ? ncols_host[b] : NBLK};
        tj[(size_t)b] = TrinvJob{A, iv, dinv + (size_t)b * NBLK * NBLK, dinvt + (size_t)b * NBLK * NBLK, NBLK};
    }
    char *base = static_cast<char *>(jobs_dev);
    hipError_t e = hipMemcpy(base, pj.data(), sizeof(PotrfJob) * (size_t)nb, hipMemcpyHostToDevice);
    if (e != hipSuccess) return e;
    const size_t off = ((sizeof(PotrfJob) * (size_t)nb + 255) / 256) * 256;
    return hipMemcpy(base + off, tj.data(), sizeof(TrinvJob) * (size_t)nb, hipMemcpyHostToDevice);
}

hipError_t block_chol_run(const void *jobs_dev, int nb, int *info_dev, double *minpiv_dev, hipStream_t st)
{
    const char *base = static_cast<const char *>(jobs_dev);
    const size_t off = ((sizeof(PotrfJob) * (size_t)nb + 255) / 256) * 256;
    hipLaunchKernelGGL(nd_potrf_kernel<8>, dim3((unsigned)nb), dim3(512), 0, st, reinterpret_cast<const PotrfJob *>(base), info_dev, minpiv_dev);
    hipLaunchKernelGGL(nd_trinv_kernel, dim3(NBLK / 16, (unsigned)nb), dim3(64), 0, st, reinterpret_cast<const TrinvJob *>(base + off));
    return hipGetLastError();
}

}  // namespace splpak
