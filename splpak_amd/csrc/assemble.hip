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
// No floating-point atomics on this path: every sum has ONE owner and a fixed order, so the normal
// equations, the sparse-area histogram and every refinement residual are bitwise reproducible from
// run to run, and the assembly runs at streaming rates instead of the rate of
// scattered 8-byte atomics (round 1: 3.8 GB of them at 64^3, 19.8 ms).
//
//   keys_kernel            window key per point + per-cell counts (integer atomics only)
//   scan_partials/segments exclusive scan of the counts (two launches over 256 segments)
//   scatter_kernel         counting-sort scatter into cell-ordered SoA copies (+ original index)
//   cell_order_kernel      orders the points INSIDE every cell by original index: the scatter's
//                          cursor order is not reproducible, the per-cell sums below must be
//   gram_block_kernel      per cell: point tables staged in LDS, register-tiled rank-1 updates;
//                          the 4^d x 4^d Gram block (packed lower triangle), B^T W^2 y and the
//                          cell's share of the nearest-node histogram (:886-907) are written with
//                          plain stores into a scratch image (the not-yet-used band storage)
//   stencil_gather_kernel  one wave per node: sums the <= 4^d blocks that contain the node, in a
//                          fixed order, into its stencil row, right-hand side and histogram entry
//   constraint_rows_kernel derivative-constraint rows of data-sparse nodes (:921-1046), gathered
//                          per stencil row from the <= 3^d sparse neighbours
//   residual_block_kernel / constraint_dots_kernel / rho_gather_kernel
//                          rho = A^T W (W y - W A x) - C^T C x for iterative refinement, same
//                          owner-gathers structure
//   expand_kernel          half stencil -> band storage of the Cholesky factorisation
// HBM roofline: 8*(ndim+1+[weighted]) algorithmic bytes per point and streaming pass.
#include "basis.hpp"
#include "kernels.hpp"
#include <cstdlib>

namespace splpak {

namespace {

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));

__device__ inline double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// sum of v[t], v[t + nt], v[t + 2 nt], .. below n, eight loads in flight (eight partial sums, combined in a fixed order):
// the single-workgroup reductions below were bound by one dependent load + add per element (round 3)
__device__ inline double strided_sum8(const double *__restrict__ v, long long n, int t, int nt)
{
    double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    long long i = t;
    for (; i + 7LL * nt < n; i += 8LL * nt) {
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] += v[i + (long long)u * nt];
    }
    for (int u = 0; i < n; i += nt, ++u) a[u] += v[i];
    return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}

// ---------------------------------------------------------------------------
template <int D>
__global__ void __launch_bounds__(256)
keys_kernel(Grid g, long long m, const double *__restrict__ x, int ldx,
            const double *__restrict__ w, int *__restrict__ key, int *__restrict__ count,
            double *__restrict__ scal)
{
    double lrows = 0.0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
        const double wv = w ? w[i] : 1.0;
        int k = g.ncell;                       // zero weight: ignored (:799, :891)
        if (wv != 0.0) {
            k = 0;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                int lo, hi;
                k += window_start(g, d, x[i * ldx + g.perm[d]], lo, hi) * g.cellstride[d];
            }
            lrows += 1.0;
        }
        key[i] = k;
        atomicAdd(&count[k], 1);
    }
    lrows = wave_sum(lrows);
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, wv_id = threadIdx.x >> 6;
    if (lane == 0) red[wv_id] = lrows;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double b = red[0] + red[1] + red[2] + red[3];
        if (b != 0.0) atomicAdd(&scal[SC_NROWS_DATA], b);      // integer-valued: exact in any order
    }
}

// exclusive scan of count[0..n) into offset[0..n] in two launches over SCAN_SEG segments (round 3: one workgroup walking
// 227 000 cells took 0.35 ms per fit at 64^3): segment sums, then every workgroup scans its own segment on top of the
// sum of the segments before it
constexpr int SCAN_SEG = 256;
__global__ void __launch_bounds__(256)
scan_partials_kernel(const int *__restrict__ count, int *__restrict__ part, int n)
{
    __shared__ int red[4];
    const int per = (n + SCAN_SEG - 1) / SCAN_SEG;
    const int lo = blockIdx.x * per, hi = (lo + per < n) ? lo + per : n;
    int s = 0;
    for (int i = lo + (int)threadIdx.x; i < hi; i += 256) s += count[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void __launch_bounds__(256)
scan_segments_kernel(const int *__restrict__ count, const int *__restrict__ part, int *__restrict__ offset, int n)
{
    __shared__ int sc[256];
    __shared__ int sbase;
    const int t = threadIdx.x;
    {   // sum of the segments before this one
        int v = (t < (int)blockIdx.x) ? part[t] : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((t & 63) == 0) sc[t >> 6] = v;
        __syncthreads();
        if (t == 0) sbase = sc[0] + sc[1] + sc[2] + sc[3];
        __syncthreads();
    }
    const int per = (n + SCAN_SEG - 1) / SCAN_SEG;
    const int lo = blockIdx.x * per, hi = (lo + per < n) ? lo + per : n;
    const int chunk = (per + 255) / 256;                  // consecutive entries per thread
    const int a = lo + t * chunk, b = (a + chunk < hi) ? a + chunk : hi;
    int s = 0;
    for (int i = a; i < b; ++i) s += count[i];
    __syncthreads();
    sc[t] = s;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {         // Hillis-Steele inclusive scan
        const int v = (t >= o) ? sc[t - o] : 0;
        __syncthreads();
        sc[t] += v;
        __syncthreads();
    }
    int run = sbase + sc[t] - s;
    for (int i = a; i < b; ++i) { offset[i] = run; run += count[i]; }
    if (blockIdx.x == SCAN_SEG - 1 && t == 255) offset[n] = sbase + sc[255];
}

template <int D>
__global__ void __launch_bounds__(256)
scatter_kernel(Grid g, long long m, const double *__restrict__ x, int ldx,
               const double *__restrict__ y, const double *__restrict__ w,
               const int *__restrict__ key, const int *__restrict__ offset,
               int *__restrict__ cursor, double *__restrict__ xs, double *__restrict__ ys,
               double *__restrict__ ws, int *__restrict__ idx, long long cap)
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
        idx[pos] = (int)i;
    }
}

// Orders the points inside every cell by original index.  The scatter places them in the order in
// which its atomic cursor was served, which differs from run to run; the per-cell sums of the Gram
// and residual kernels are taken in storage order, so this is what makes the whole fit reproducible
// bit for bit.  One workgroup per cell.  Cells of up to ORDER_CAP points are permuted through an image
// in LDS.  Larger cells (round 4; up to ORDER_BIG_CAP points): the rank of every point is counted against
// the cell's original indices streamed through LDS in chunks of 1 024, the ordered indices are parked in
// `ordtmp` (the key array, dead once the scatter has run) and the cell's image is rebuilt from the caller's
// arrays.  Cells beyond ORDER_BIG_CAP (65 536 points in ONE window of the grid) keep the scatter's order:
// their sums are correct, only not reproducible from run to run.
constexpr int ORDER_CAP = 1024;
constexpr int ORDER_BIG_CAP = 1 << 16;
template <int D>
__global__ void __launch_bounds__(256)
cell_order_kernel(Grid g, const int *__restrict__ offset, double *__restrict__ xs, double *__restrict__ ys,
                  double *__restrict__ ws, int *__restrict__ idx, long long cap, const double *__restrict__ x, int ldx,
                  const double *__restrict__ y, const double *__restrict__ w, int *__restrict__ ordtmp)
{
    __shared__ int sidx[ORDER_CAP];
    __shared__ double sv[ORDER_CAP * (D + 2)];
    const int cell = blockIdx.x;
    const long long beg = offset[cell];
    const int n = (int)(offset[cell + 1] - beg);
    if (n < 2 || n > ORDER_BIG_CAP) return;
    const int tid = threadIdx.x;
    if (n > ORDER_CAP) {
        for (int t0 = 0; t0 < n; t0 += ORDER_CAP) {           // a tile of 1 024 points: four per thread
            int me[4], rank[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = t0 + tid + 256 * u;
                me[u] = p < n ? idx[beg + p] : 0x7fffffff;
                rank[u] = 0;
            }
            for (int c0 = 0; c0 < n; c0 += ORDER_CAP) {
                __syncthreads();
                for (int p = tid; p < ORDER_CAP; p += 256) sidx[p] = c0 + p < n ? idx[beg + c0 + p] : 0x7fffffff;
                __syncthreads();
                const int cn = n - c0 < ORDER_CAP ? n - c0 : ORDER_CAP;
                for (int q = 0; q < cn; ++q) {
                    const int v = sidx[q];
#pragma unroll
                    for (int u = 0; u < 4; ++u) rank[u] += v < me[u] ? 1 : 0;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (t0 + tid + 256 * u < n) ordtmp[beg + rank[u]] = me[u];
        }
        __syncthreads();                                      // (the workgroup's own global writes are visible to it)
        for (int p = tid; p < n; p += 256) {
            const int i = ordtmp[beg + p];
            idx[beg + p] = i;
#pragma unroll
            for (int d = 0; d < D; ++d) xs[(long long)d * cap + beg + p] = x[(long long)i * ldx + g.perm[d]];
            ys[beg + p] = y[i];
            ws[beg + p] = w ? w[i] : 1.0;
        }
        return;
    }
    for (int p = tid; p < n; p += 256) {
        sidx[p] = idx[beg + p];
#pragma unroll
        for (int d = 0; d < D; ++d) sv[d * ORDER_CAP + p] = xs[(long long)d * cap + beg + p];
        sv[D * ORDER_CAP + p] = ys[beg + p];
        sv[(D + 1) * ORDER_CAP + p] = ws[beg + p];
    }
    __syncthreads();
    for (int p = tid; p < n; p += 256) {
        const int me = sidx[p];
        int rank = 0;
        for (int q = 0; q < n; ++q) rank += sidx[q] < me;
        const long long pos = beg + rank;
#pragma unroll
        for (int d = 0; d < D; ++d) xs[(long long)d * cap + pos] = sv[d * ORDER_CAP + p];
        ys[pos] = sv[D * ORDER_CAP + p];
        ws[pos] = sv[(D + 1) * ORDER_CAP + p];
        idx[pos] = me;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Stable partition of the points by window (round 5).  keys_kernel / scatter_kernel above bin with one global
// integer atomic per point and pass -- on the cell's counter, then on its cursor -- and scatter (ndim + 3) separate 8-byte
// words per point; the cursor's order is not reproducible, so cell_order_kernel re-sorts every cell by original index:
// 3.6 ms for the 1e7 points of BASELINE config 3 (0.04 of the HBM roofline), 0.42 of config 2's 2.0 ms (269 points on each of
// only 3 721 counters).  Here a counting sort WITHOUT global atomics that is stable by construction -- the points of a cell
// end up in ascending original index, which is exactly the order cell_order_kernel produced: the same bits downstream.
//
//   A block = SP_Q consecutive points, its 16 waves own 512 consecutive points each.
//   sp_count:    window key of every point (kept in `key`), bin = key / cpt (one level: cpt = 1, bin = cell; the last bin holds
//                the zero-weight points, :799); per-wave bin counts in LDS (16-bit halves of words, LDS atomics -- counting is
//                order-free), block totals -> row [block][bin] of the count matrix.
//   sp_colsum / sp_binscan / sp_blockbase: bin bases (exclusive scan of the bin totals) and, in place of every count, where that
//                block's points of that bin start: bins in order, blocks in order inside a bin.
//   sp_scatter:  the per-wave counts again, turned into exclusive prefixes over the waves; then every wave walks its 8 passes
//                of 64 points in order: a point's rank among the EARLIER lanes with its bin comes from the ballots of the
//                bin's bits (12 ballots: the lanes whose bin matches, below the own one), its wave's running count of the bin
//                from LDS -- the lowest lane of every group adds the group's size to it.  One level: the point goes straight
//                to its sorted place (SoA planes).  Two levels (more cells than bins: 64^3 has 226 981): a RECORD (coordinates,
//                y, w, index, cell) goes to the tile-sorted intermediate image, and
//   sp_bin2:     one workgroup per tile of cpt cells sorts the tile's records (contiguous: ~2 800 at config 3) by cell with the
//                same machinery -- per-wave counts, prefixes over waves and sub-blocks, ballot ranks -- writes their sorted
//                places and the offsets of its cells.
// Every count has one writer or is a sum; no order depends on timing: bitwise reproducible without a second sort.
constexpr int SP_NT = 1024, SP_NW = SP_NT / 64, SP_BITS = 12;
// doubles of a record: ndim coordinates, y, w, (index, cell) -- rounded up to an even count: records move as 16-byte words (a
// scattered store costs this chip ~7 cycles per lane and instruction whatever its width)
__host__ __device__ constexpr int sp_rec(int d) { return (d + 3 + 1) & ~1; }
static_assert(SP_NB == 1 << SP_BITS && SP_Q == SP_NW * 512, "bins / block shape");

template <int D>
__device__ inline int sp_cell_key(const Grid &g, const double *__restrict__ x, int ldx, const double *__restrict__ w, long long i)
{
    const double wv = w ? w[i] : 1.0;
    if (wv == 0.0) return g.ncell;             // zero weight: ignored (:799, :891)
    int k = 0;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        int lo, hi;
        k += window_start(g, d, x[i * ldx + g.perm[d]], lo, hi) * g.cellstride[d];
    }
    return k;
}

// lanes of the wave whose `bin` equals this lane's (all lanes take part; inactive ones pass bin = -1 and are matched with nobody real)
__device__ inline unsigned long long sp_match(int bin, int bits)
{
    unsigned long long m = ~0ull;
    for (int b = 0; b < bits; ++b) {
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(((bin >> b) & 1) != 0);
        m &= ((bin >> b) & 1) ? bal : ~bal;
    }
    return m;
}

// per-wave bin counts of a block of up to SP_Q items: whist[wave][bin / 2], two 16-bit counts per word; nb bins in use.
// keyv[j]: the bin of the wave's item j * 64 + lane (-1: none)
template <int NBW>
__device__ inline void sp_wave_counts(unsigned (*whist)[NBW], int nb, const int (&binv)[8])
{
    const int tid = threadIdx.x, wave = tid >> 6;
    const int nw = (nb + 1) >> 1;
    for (int v = 0; v < SP_NW; ++v)
        for (int e = tid; e < nw; e += SP_NT) whist[v][e] = 0u;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (binv[j] >= 0) atomicAdd(&whist[wave][binv[j] >> 1], 1u << (16 * (binv[j] & 1)));
    __syncthreads();
}

// first level, counts: a block only needs its TOTALS per bin -- one histogram for all waves (16 KB of LDS: several workgroups
// per CU; with per-wave counts, 128 KB, the kernel ran one workgroup per CU at 0.38 ms for the 1e7 points of config 3)
template <int D>
__global__ void __launch_bounds__(SP_NT)
sp_count_kernel(Grid g, long long m, const double *__restrict__ x, int ldx, const double *__restrict__ w, int cpt,
                int *__restrict__ key, int *__restrict__ cntm)
{
    __shared__ int hist[SP_NB];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long long base = (long long)blockIdx.x * SP_Q;
    const int n = (int)(m - base < SP_Q ? m - base : SP_Q);
    for (int b = tid; b < SP_NB; b += SP_NT) hist[b] = 0;
    __syncthreads();
    // (all loads of a thread's 8 points in flight together: one dependent round trip per point made this pass latency bound)
    double xv[8][D], wv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int p = wave * 512 + j * 64 + lane;
        const long long i = base + (p < n ? p : 0);
        wv[j] = p < n ? (w ? w[i] : 1.0) : 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) xv[j][d] = x[i * ldx + g.perm[d]];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int p = wave * 512 + j * 64 + lane;
        if (p < n) {
            int k = g.ncell;                       // zero weight: ignored (:799, :891)
            if (wv[j] != 0.0) {
                k = 0;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    int lo, hi;
                    k += window_start(g, d, xv[j][d], lo, hi) * g.cellstride[d];
                }
            }
            key[base + p] = k;
            atomicAdd(&hist[k < g.ncell ? k / cpt : SP_NB - 1], 1);
        }
    }
    __syncthreads();
    int *__restrict__ row = cntm + (long long)blockIdx.x * SP_NB;
    for (int b = tid; b < SP_NB; b += SP_NT) row[b] = hist[b];
    // (the number of data rows -- the points of non-zero weight -- is the base of the last bin: sp_binscan_kernel adds it to the
    //  fit's scalars, no f64 atomic per block.  What this kernel's 0.25-0.3 ms per 1e7 points are, measured by taking its parts
    //  out: 0.23 ms are the loads of the coordinates and weights themselves -- 1.4 TB/s, although all 32 of a thread are in
    //  flight together --, the key stores, LDS atomics and the row together 0.04)
}

// Count matrix [block][bin] -> in place: where the block's points of the bin start (bins in order, blocks in order inside a
// bin), and binbase = exclusive scan of the bin totals.  Four small launches: sums over chunks of SP_ROWS blocks, per bin the
// exclusive prefix over the chunks + its total, the scan of the totals, and the walk down every chunk.
constexpr int SP_ROWS = 16;        // blocks per chunk of the column sums
__global__ void __launch_bounds__(256)
sp_colsum_kernel(int nblk, const int *__restrict__ cntm, int *__restrict__ part)
{
    const int bin = blockIdx.x * 256 + threadIdx.x, ch = blockIdx.y;
    const int b0 = ch * SP_ROWS, b1 = b0 + SP_ROWS < nblk ? b0 + SP_ROWS : nblk;
    int v[SP_ROWS];
#pragma unroll
    for (int r = 0; r < SP_ROWS; ++r) v[r] = b0 + r < b1 ? cntm[(long long)(b0 + r) * SP_NB + bin] : 0;
    int s = 0;
#pragma unroll
    for (int r = 0; r < SP_ROWS; ++r) s += v[r];
    part[(long long)ch * SP_NB + bin] = s;
}
__global__ void __launch_bounds__(256)
sp_chunkscan_kernel(int nchunk, int *__restrict__ part, int *__restrict__ tot)
{
    const int bin = blockIdx.x * 256 + threadIdx.x;
    int run = 0, c = 0;
    for (; c + 8 <= nchunk; c += 8) {             // eight loads in flight (a dependent load per chunk: 1 us each)
        int v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(long long)(c + u) * SP_NB + bin];
#pragma unroll
        for (int u = 0; u < 8; ++u) { part[(long long)(c + u) * SP_NB + bin] = run; run += v[u]; }
    }
    for (; c < nchunk; ++c) {
        const int v = part[(long long)c * SP_NB + bin];
        part[(long long)c * SP_NB + bin] = run;
        run += v;
    }
    tot[bin] = run;
}
__global__ void __launch_bounds__(1024)
sp_binscan_kernel(const int *__restrict__ tot, int *__restrict__ binbase, double *__restrict__ scal)
{
    __shared__ int sc[1024];
    const int t = threadIdx.x;
    int v[4], sum = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) { v[u] = tot[4 * t + u]; sum += v[u]; }
    sc[t] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int a = t >= o ? sc[t - o] : 0;
        __syncthreads();
        sc[t] += a;
        __syncthreads();
    }
    int run = sc[t] - sum;
#pragma unroll
    for (int u = 0; u < 4; ++u) { binbase[4 * t + u] = run; run += v[u]; }
    if (t == 1023) {
        binbase[SP_NB] = run;
        const int nvalid = run - v[3];              // everything below the last bin (which holds the zero-weight points)
        if (nvalid) atomicAdd(&scal[SC_NROWS_DATA], (double)nvalid);      // integer-valued: exact in any order
    }
}
__global__ void __launch_bounds__(256)
sp_blockbase_kernel(int nblk, int *__restrict__ cntm, const int *__restrict__ part, const int *__restrict__ binbase)
{
    const int bin = blockIdx.x * 256 + threadIdx.x, ch = blockIdx.y;
    const int b0 = ch * SP_ROWS, b1 = b0 + SP_ROWS < nblk ? b0 + SP_ROWS : nblk;
    int v[SP_ROWS];
#pragma unroll
    for (int r = 0; r < SP_ROWS; ++r) v[r] = b0 + r < b1 ? cntm[(long long)(b0 + r) * SP_NB + bin] : 0;
    int run = binbase[bin] + part[(long long)ch * SP_NB + bin];
#pragma unroll
    for (int r = 0; r < SP_ROWS; ++r) {
        if (b0 + r < b1) cntm[(long long)(b0 + r) * SP_NB + bin] = run;
        run += v[r];
    }
}

// the stable places of a block's items: place(j, pos) is called once per item j * 64 + lane of the wave with binv[j] >= 0, pos =
// bbase[bin] + items of the bin before it in the block.  whist: the per-wave counts (sp_wave_counts), turned into running
// exclusive prefixes here.
template <int NBW>
__device__ inline void sp_wave_prefix(unsigned (*whist)[NBW], int nb)
{
    for (int b2 = threadIdx.x; b2 < ((nb + 1) >> 1); b2 += SP_NT) {         // exclusive prefixes over the waves, both halves of a word at once
        unsigned run = 0;
#pragma unroll
        for (int v = 0; v < SP_NW; ++v) {
            const unsigned c = whist[v][b2];
            whist[v][b2] = run;
            run += c;
        }
    }
    __syncthreads();
}
// (passes J0 .. J1 - 1 of the wave: a caller that has to LOAD what it places splits the eight passes in two, so that the loads of
//  four passes are in flight together without their registers exceeding the budget)
template <int J0, int J1, int NBW, typename Place>
__device__ inline void sp_places(unsigned (*whist)[NBW], const int *bbase, int bits, const int (&binv)[8], Place &&place)
{
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int j = J0; j < J1; ++j) {
        const int bin = binv[j];
        const unsigned long long mm = sp_match(bin, bits + 1);          // (+ 1: bit `bits` tells -1 from every real bin)
        if (bin >= 0) {
            const unsigned wv = whist[wave][bin >> 1];
            const int before = (int)((wv >> (16 * (bin & 1))) & 0xffffu) + __builtin_popcountll(mm & lt);
            place(j, bbase[bin] + before);
        }
        __builtin_amdgcn_wave_barrier();                               // (every lane has read its running count)
        if (bin >= 0 && (mm & lt) == 0) atomicAdd(&whist[wave][bin >> 1], (unsigned)__builtin_popcountll(mm) << (16 * (bin & 1)));
        __builtin_amdgcn_wave_barrier();
    }
}
template <int NBW, typename Place>
__device__ inline void sp_stable_places(unsigned (*whist)[NBW], const int *bbase, int nb, int bits, const int (&binv)[8], Place &&place)
{
    sp_wave_prefix(whist, nb);
    sp_places<0, 8>(whist, bbase, bits, binv, place);
}

// RECORD of a point on its way through the tile-sorted image: ndim coordinates (internal order), y, w, (index, cell)
template <int D>
__global__ void __launch_bounds__(SP_NT)
sp_scatter_kernel(Grid g, long long m, const double *__restrict__ x, int ldx, const double *__restrict__ y, const double *__restrict__ w,
                  int cpt, int nb, const int *__restrict__ key, const int *__restrict__ cntm, double *__restrict__ rec,
                  double *__restrict__ xs, double *__restrict__ ys, double *__restrict__ ws, int *__restrict__ idx, long long cap)
{
    __shared__ unsigned whist[SP_NW][SP_NB / 2];
    __shared__ int bbase[SP_NB];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long long base = (long long)blockIdx.x * SP_Q;
    const int n = (int)(m - base < SP_Q ? m - base : SP_Q);
    for (int b = tid; b < nb; b += SP_NT) bbase[b] = cntm[(long long)blockIdx.x * SP_NB + b];
    // the wave's 8 x 64 points, all loads of a thread in flight together (the walk below is LDS and stores only)
    int kk[8], binv[8];
    double xv[8][D], yv[8], wv[8];
    // (UNCONDITIONAL loads, the index clamped for the tail of the last block: loads under `valid ? load : 0` wait for the key and
    //  then for each other -- eight serialised round trips per wave, 0.82 of this kernel's 1.05 ms per 1e7 points, found by taking
    //  the rest of the kernel out)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int p = wave * 512 + j * 64 + lane;
        const long long i = base + (p < n ? p : 0);
        kk[j] = key[i];
#pragma unroll
        for (int d = 0; d < D; ++d) xv[j][d] = x[i * ldx + g.perm[d]];
        yv[j] = y[i];
        wv[j] = w ? w[i] : 1.0;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int p = wave * 512 + j * 64 + lane;
        if (p >= n) kk[j] = -1;
        const bool ok = kk[j] >= 0 && kk[j] < g.ncell;
        binv[j] = ok ? kk[j] / cpt : -1;                                // zero weight: neither counted nor placed here
    }
    sp_wave_counts(whist, nb, binv);
    sp_stable_places(whist, bbase, nb, SP_BITS, binv, [&](int j, int pos) {
        const int i = (int)(base + wave * 512 + j * 64 + lane);
        if (rec) {
            constexpr int R = sp_rec(D);
            double v[R];
#pragma unroll
            for (int d = 0; d < D; ++d) v[d] = xv[j][d];
            v[D] = yv[j];
            v[D + 1] = wv[j];
            v[D + 2] = __hiloint2double(kk[j], i);                       // (low word: index, high word: cell)
#pragma unroll
            for (int e = D + 3; e < R; ++e) v[e] = 0.0;
            d2_t *__restrict__ r = reinterpret_cast<d2_t *>(rec + (long long)pos * R);
#pragma unroll
            for (int e = 0; e < R / 2; ++e) r[e] = (d2_t){v[2 * e], v[2 * e + 1]};
        } else {
#pragma unroll
            for (int d = 0; d < D; ++d) xs[(long long)d * cap + pos] = xv[j][d];
            ys[pos] = yv[j];
            ws[pos] = wv[j];
            idx[pos] = i;
        }
    });
}

// second level: workgroup = tile of cpt cells; its records (binbase[tile] .. binbase[tile + 1]) -> sorted places by cell.
// NB2: capacity in cells of a tile (256; SP_NB for grids beyond ~1e6 cells).  A tile of at most SP_STAGE records (config 3: ~2 500)
// is assembled in LDS -- planes of coordinates, y, w, index in sorted order -- and leaves with consecutive stores; larger tiles
// (clustered data) are walked in sub-blocks of SP_Q records and scattered straight into place.
constexpr int SP_STAGE = 2816;      // (x 52 bytes in 4-D + the counters: 157 KB of the 160 KB LDS)
template <int D, int NB2>
__global__ void __launch_bounds__(SP_NT)
sp_bin2_kernel(Grid g, int cpt, int ntile, const int *__restrict__ binbase, const double *__restrict__ rec, int *__restrict__ offset,
               double *__restrict__ xs, double *__restrict__ ys, double *__restrict__ ws, int *__restrict__ idx, long long cap)
{
    constexpr int R = sp_rec(D);
    __shared__ unsigned whist[SP_NW][NB2 / 2];
    __shared__ int bbase[NB2], btot[NB2];                             // [local cell]: where its next sub-block's records start / totals
    __shared__ double stage[NB2 <= 256 ? SP_STAGE * (D + 2) : 1];
    __shared__ int stidx[NB2 <= 256 ? SP_STAGE : 1];
    const int tile = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tb = binbase[tile], te = binbase[tile + 1];
    const int c0 = tile * cpt, nc = min(cpt, g.ncell - c0);
    int bits = 1;
    while ((1 << bits) < nc) ++bits;
    auto cell_at = [&](int q) { return __double2hiint(rec[(long long)q * R + D + 2]) - c0; };       // local cell of record q
    // cell totals of the whole tile, then their exclusive prefix = the cells' offsets
    for (int b = tid; b < nc; b += SP_NT) btot[b] = 0;
    __syncthreads();
    for (int q = tb + tid; q < te; q += SP_NT) atomicAdd(&btot[cell_at(q)], 1);
    __syncthreads();
    if (tid < 64) {                                                    // exclusive scan over the nc cells by one wave
        int carry = tb;
        for (int b0 = 0; b0 < nc; b0 += 64) {
            const int v = b0 + lane < nc ? btot[b0 + lane] : 0;
            int inc = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(inc, o, 64);
                if (lane >= o) inc += t;
            }
            if (b0 + lane < nc) bbase[b0 + lane] = carry + inc - v;
            carry += __shfl(inc, 63, 64);
        }
    }
    __syncthreads();
    for (int b = tid; b < nc; b += SP_NT) offset[c0 + b] = bbase[b];
    if (tile == ntile - 1 && tid == 0) offset[g.ncell] = te;
    const bool staged = NB2 <= 256 && te - tb <= SP_STAGE;
    for (int sb0 = 0; sb0 < te - tb; sb0 += SP_Q) {
        const int n = min(SP_Q, te - tb - sb0);
        int binv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = wave * 512 + j * 64 + lane;
            const int c = cell_at(tb + sb0 + (p < n ? p : 0));          // (unconditional load, index clamped)
            binv[j] = p < n ? c : -1;
        }
        sp_wave_counts(whist, nc, binv);
        // (the sub-block's totals, for the next sub-block's bases -- taken before the counts become prefixes)
        for (int b = tid; b < nc; b += SP_NT) {
            unsigned t = 0;
            for (int v = 0; v < SP_NW; ++v) t += (whist[v][b >> 1] >> (16 * (b & 1))) & 0xffffu;
            btot[b] = (int)t;
        }
        __syncthreads();
        sp_wave_prefix(whist, nc);
        // (the records of four passes are loaded together, UNCONDITIONALLY -- index clamped --: a load per pass inside the walk was
        //  a dependent memory round trip per pass and wave)
        d2_t rv[4][R / 2];
        auto preload = [&](int j0) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int p = wave * 512 + (j0 + jj) * 64 + lane;
                const d2_t *__restrict__ r = reinterpret_cast<const d2_t *>(rec + (long long)(tb + sb0 + (p < n ? p : 0)) * R);
#pragma unroll
                for (int e = 0; e < R / 2; ++e) rv[jj][e] = r[e];
            }
        };
        auto place = [&](int j, int pos) {
            double v[R];
#pragma unroll
            for (int e = 0; e < R / 2; ++e) { v[2 * e] = rv[j & 3][e][0]; v[2 * e + 1] = rv[j & 3][e][1]; }
            if (staged) {
                const int l = pos - tb;
#pragma unroll
                for (int d = 0; d < D + 2; ++d) stage[d * SP_STAGE + l] = v[d];
                stidx[l] = __double2loint(v[D + 2]);
            } else {
#pragma unroll
                for (int d = 0; d < D; ++d) xs[(long long)d * cap + pos] = v[d];
                ys[pos] = v[D];
                ws[pos] = v[D + 1];
                idx[pos] = __double2loint(v[D + 2]);
            }
        };
        preload(0);
        sp_places<0, 4>(whist, bbase, bits, binv, place);
        preload(4);
        sp_places<4, 8>(whist, bbase, bits, binv, place);
        __syncthreads();
        for (int b = tid; b < nc; b += SP_NT) bbase[b] += btot[b];
        __syncthreads();
    }
    if (staged) {
        const int n = te - tb;
        for (int e = tid; e < n; e += SP_NT) {
#pragma unroll
            for (int d = 0; d < D; ++d) xs[(long long)d * cap + tb + e] = stage[d * SP_STAGE + e];
            ys[tb + e] = stage[D * SP_STAGE + e];
            ws[tb + e] = stage[(D + 1) * SP_STAGE + e];
            idx[tb + e] = stidx[e];
        }
    }
}

// nearest-node histogram slot of a point INSIDE its cell's window (local index, dim 0 fastest), or -1
// when the reference's address (:894-902) is not a node of the window: a coordinate so far outside
// the grid that its dimension is skipped in the Horner address (the :899 quirk).  x is in the plan's
// internal dimension order; per dimension the arithmetic is the reference's.
template <int D>
__device__ inline int nearest_slot(const Grid &g, const double *x)
{
#pragma clang fp contract(off)
    int slot = 0;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const double t = g.dxin[d] * (x[d] - g.xmin[d]) + 0.5;
        const int inidim = (t >= 2.0e9) ? 2000000000 : (t <= -2.0e9 ? -2000000000 : (int)t);
        if (inidim < 0 || inidim > g.nodes[d] - 1) return -1;
        int lo, hi;
        const int l = inidim - window_start(g, d, x[d], lo, hi);
        if (l < 0 || l > 3) return -1;
        slot += l << (2 * d);
    }
    return slot;
}

// ---------------------------------------------------------------------------
// Per-cell staging shared by the Gram and the residual kernels: for `np` points
// starting at sorted position `p0`, fill bw[p*LDB + c] = w * ((b0*b1)*b2...) --
// the row of the weighted least-squares matrix restricted to the cell's window
// (:833-837) -- and wy[p] = w*y (:806).  hslot != NULL: also the point's histogram slot
// (nearest_slot; points whose address lies outside the window are added to `hist` directly).
template <int D, int NB, int LDB, int NT>
__device__ inline void stage_points(const Grid &g, const double *__restrict__ xs,
                                    const double *__restrict__ ys,
                                    const double *__restrict__ ws, long long cap, long long p0,
                                    int np, double *tab /*[PCH][D][4]*/, double *bw, double *wy,
                                    double *wt, int *hslot, double *__restrict__ hist)
{
    const int tid = threadIdx.x;
    for (int p = tid; p < np; p += NT) {
        const double wv = ws[p0 + p];
        double xv[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            double b[4];
            xv[d] = xs[(long long)d * cap + p0 + p];
            window_table(g, d, xv[d], 0, b);
#pragma unroll
            for (int k = 0; k < 4; ++k) tab[(p * D + d) * 4 + k] = b[k];
        }
        wy[p] = ys ? wv * ys[p0 + p] : 0.0;      // (ys == NULL: the rows applied to a vector, residual pass as operator -- pcg.hip)
        wt[p] = wv;
        if (hslot) {
            const int sl = nearest_slot<D>(g, xv);
            hslot[p] = sl;
            if (sl < 0) {                       // rare: far outside the grid (:899); the only atomic left
                double xr[MAXD] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int d = 0; d < D; ++d) xr[g.perm[d]] = xv[d];
                atomicAdd(&hist[nearest_node_address(g, xr)], wv);       // :905
            }
        }
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
template <> struct GramCfg<1> { static constexpr int NB = 4,   TR = 1, TC = 1, NTY = 4,  NTX = 4,  NT = 64,   PCH = 64,  WPE = 1; };
template <> struct GramCfg<2> { static constexpr int NB = 16,  TR = 1, TC = 1, NTY = 16, NTX = 16, NT = 256,  PCH = 128, WPE = 1; };
template <> struct GramCfg<3> { static constexpr int NB = 64,  TR = 4, TC = 4, NTY = 16, NTX = 16, NT = 256,  PCH = 64,  WPE = 4; };
template <> struct GramCfg<4> { static constexpr int NB = 256, TR = 4, TC = 4, NTY = 64, NTX = 16, NT = 1024, PCH = 16,  WPE = 1; };

// scratch image of the per-cell blocks: [ncell][TRI] packed lower triangles (row-major: entry (r,c),
// c <= r, at r(r+1)/2 + c), then [ncell][NB] right-hand sides, then [ncell][NB] histogram shares
__host__ __device__ inline long long gram_tri(int nb) { return (long long)nb * (nb + 1) / 2; }

template <int D>
__global__ void __launch_bounds__(GramCfg<D>::NT, GramCfg<D>::WPE)
gram_block_kernel(Grid g, const int *__restrict__ offset, const double *__restrict__ xs,
                  const double *__restrict__ ys, const double *__restrict__ ws, long long cap,
                  double *__restrict__ blk, double *__restrict__ rblk, double *__restrict__ hblk,
                  double *__restrict__ hist, int cell0)
{
    using C = GramCfg<D>;
    constexpr int NB = C::NB, TR = C::TR, TC = C::TC, NT = C::NT, PCH = C::PCH;
    constexpr int CW = C::NTX * TC;            // columns handled by this workgroup
    constexpr long long TRI = (long long)NB * (NB + 1) / 2;
    const int cell = cell0 + blockIdx.x;       // the scratch image holds the cells of one slab, from cell0 on
    const int cpass = blockIdx.y;
    const long long beg = offset[cell], end = offset[cell + 1];
    if (beg == end) return;                    // the gather skips empty cells

    // (the point image doubles as the staging area of the packed triangle on the way out: at least TRI entries when
    // one column pass covers the block)
    constexpr int BWN = (C::NTX * TC >= NB && TRI > (long long)PCH * NB) ? (int)TRI : PCH * NB;
    __shared__ double tab[PCH * D * 4];
    __shared__ double bw[BWN];
    __shared__ double wy[PCH];
    __shared__ double wt[PCH];
    __shared__ int hslot[PCH];

    const int tid = threadIdx.x;
    const int ty = tid / C::NTX, tx = tid % C::NTX;
    const int r0 = ty * TR, c0 = cpass * CW + tx * TC;
    const bool tile_on = (tid < C::NTY * C::NTX) && (r0 + TR - 1 >= c0);
    const bool vec_on = (cpass == 0) && (tid < NB);
    const bool hist_on = hblk != nullptr && cpass == 0;

    double acc[TR][TC];
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
        for (int j = 0; j < TC; ++j) acc[i][j] = 0.0;
    double racc = 0.0, hacc = 0.0;

    for (long long p0 = beg; p0 < end; p0 += PCH) {
        const int np = (int)((end - p0 < PCH) ? (end - p0) : PCH);
        stage_points<D, NB, NB, NT>(g, xs, ys, ws, cap, p0, np, tab, bw, wy, wt, hist_on ? hslot : nullptr, hist);
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
        if (vec_on) {
            for (int p = 0; p < np; ++p) racc += bw[p * NB + tid] * wy[p];
            if (hist_on)
                for (int p = 0; p < np; ++p) hacc += (hslot[p] == tid) ? wt[p] : 0.0;       // :905, in storage order
        }
        __syncthreads();
    }

    double *__restrict__ out = blk + (long long)(cell - cell0) * TRI;
    if constexpr (TRI <= (long long)BWN) {
        // the packed triangle is assembled in LDS (the point image is no longer needed) and leaves with
        // consecutive lanes on consecutive addresses (the 4x4 register tiles written directly put 8 bytes
        // every 32; same kernel time at 64^3 -- the kernel is bound by its per-cell phases, not by the
        // stores -- but a quarter of the write requests)
        if (tile_on) {
#pragma unroll
            for (int i = 0; i < TR; ++i) {
                const int r = r0 + i;
#pragma unroll
                for (int j = 0; j < TC; ++j) {
                    const int c = c0 + j;
                    if (c <= r) bw[r * (r + 1) / 2 + c] = acc[i][j];
                }
            }
        }
        __syncthreads();
        for (int e = tid; e < (int)TRI; e += NT) out[e] = bw[e];
    } else if (tile_on) {
#pragma unroll
        for (int i = 0; i < TR; ++i) {
            const int r = r0 + i;
#pragma unroll
            for (int j = 0; j < TC; ++j) {
                const int c = c0 + j;
                if (c <= r) out[(long long)r * (r + 1) / 2 + c] = acc[i][j];
            }
        }
    }
    if (vec_on) {
        rblk[(long long)(cell - cell0) * NB + tid] = racc;
        if (hist_on) hblk[(long long)(cell - cell0) * NB + tid] = hacc;
    }
}

// The same per-cell blocks for 2-D and 3-D grids (NB = 16 / 64) on the f64 matrix cores, ONE WAVE per cell, no workgroup
// barriers (round 3).  The block is B^T B with B = the cell's weighted rows (points x NB), i.e. a product
// with K = points: per 4 points one v_mfma_f64_16x16x4_f64 per 16 x 16 tile of the lower triangle (10 tiles at NB = 64).
// The operands are never staged: lane (l15, q) of k-step s needs
// B[p = 4 s + q][16 m + l15] = (w_p b0[l15 & 3] b1[l15 >> 2]) * b2[m] -- three table reads from the wave's LDS slice and
// one multiplication per tile block.  The result tiles have consecutive lanes on consecutive entries of a packed row
// (first MFMA operand = the row block), so they leave with plain coalesced stores.  The workgroup-per-cell form above
// read 8 LDS values per 16 FMAs, used 136 of its 256 threads in the triangle and passed five barriers per cell: 3.56 ms
// at C3 for 0.65 ms of matrix-pipe time.
// Round 5 (in-kernel clock stamps per wave, C3: 5.1 us loads + tables, 4.1 us products, 5.1 us right-hand side, 1.5 us stores):
//  - the right-hand side B^T (w^2 y) is summed from the matrix operands the lanes hold anyway (one multiply-add per row
//    block and step, the four lanes of an entry added at the end); as a loop over the points with lane = window function it
//    read five LDS words per point behind the LDS latency and cost as much as the blocks;
//  - the histogram share of a point is one LDS atomic of lane = point onto its nearest node's word of the wave's LDS row (the
//    lanes of one instruction that meet on a word are served one after the other by the LDS unit, the same way every run);
//  - a wave owns a RUN of up to GW_RUN consecutive cells (fewer on a small grid: launch_gram) and loads the points of the next cell before the products of the current
//    one, so that the two dependent round trips (offsets, then points) are paid once per run, not once per cell.
constexpr int GW_RUN = 8;
template <int D>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
gram_wave_kernel(Grid g, const int *__restrict__ offset, const double *__restrict__ xs,
                 const double *__restrict__ ys, const double *__restrict__ ws, long long cap,
                 double *__restrict__ blk, double *__restrict__ rblk, double *__restrict__ hblk,
                 double *__restrict__ hist, int cell0, int ncells, int run)
{
    static_assert(D == 2 || D == 3, "16 or 64 window functions");
    static_assert(GW_RUN + 1 <= 64, "the offsets of a run sit in one register of the wave");
    constexpr int NB = 1 << (2 * D), MT = NB / 16, NTILE = MT * (MT + 1) / 2, PCH = 64, LDT = 4 * D + 1;
    constexpr long long TRI = (long long)NB * (NB + 1) / 2;
    __shared__ double s_tab[4][PCH * LDT];
    __shared__ double s_w[4][PCH];
    __shared__ double s_wy[4][PCH];
    __shared__ double s_hist[4][NB];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, q = lane >> 4;
    const int first = (blockIdx.x * 4 + wave) * run;
    if (first >= ncells) return;
    const int ncl = ncells - first < run ? ncells - first : run;
    double *tab = s_tab[wave], *sw = s_w[wave], *swy = s_wy[wave], *sh = s_hist[wave];
    const bool hist_on = hblk != nullptr;
    const int offv = offset[cell0 + first + (lane <= ncl ? lane : ncl)];
    // the points of a chunk, lane = point (zeros beyond np: zero rows pad the last k-step)
    struct Pts { double x[D], y, w; };
    auto fetch = [&](long long p0, int np) {
        Pts t;
#pragma unroll
        for (int d = 0; d < D; ++d) t.x[d] = 0.0;
        t.y = 0.0;
        t.w = 0.0;
        if (lane < np) {
            t.w = ws[p0 + lane];
#pragma unroll
            for (int d = 0; d < D; ++d) t.x[d] = xs[(long long)d * cap + p0 + lane];
            t.y = ys[p0 + lane];
        }
        return t;
    };
    long long beg = __builtin_amdgcn_readlane(offv, 0), end = __builtin_amdgcn_readlane(offv, 1);
    Pts pre = fetch(beg, (int)(end - beg < PCH ? end - beg : PCH));
    for (int ci = 0; ci < ncl; ++ci) {
        const int rel = first + ci;
        long long nbeg = end, nend = end;
        if (ci + 1 < ncl) nend = __builtin_amdgcn_readlane(offv, ci + 2);
        if (beg == end) {                          // an empty cell leaves a zero block: the gather reads every block unconditionally
            double *__restrict__ z = blk + (long long)rel * TRI;
            for (int e = lane; e < (int)TRI; e += 64) z[e] = 0.0;
            if (lane < NB) {
                rblk[(long long)rel * NB + lane] = 0.0;
                if (hblk) hblk[(long long)rel * NB + lane] = 0.0;
            }
            if (ci + 1 < ncl) pre = fetch(nbeg, (int)(nend - nbeg < PCH ? nend - nbeg : PCH));
            beg = nbeg;
            end = nend;
            continue;
        }
        if (hist_on && lane < NB) sh[lane] = 0.0;
        d4_t acc[NTILE];
        double racc[MT];
#pragma unroll
        for (int t = 0; t < NTILE; ++t) acc[t] = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int m = 0; m < MT; ++m) racc[m] = 0.0;
        for (long long p0 = beg; p0 < end; p0 += PCH) {
            const int np = (int)((end - p0 < PCH) ? (end - p0) : PCH);
            const Pts cur = (p0 == beg) ? pre : fetch(p0, np);
            // ---- lane = point: window tables, weight, w^2 y, histogram share
            {
                double b[D][4];
#pragma unroll
                for (int d = 0; d < D; ++d)
#pragma unroll
                    for (int k = 0; k < 4; ++k) b[d][k] = 0.0;
                if (lane < np) {
#pragma unroll
                    for (int d = 0; d < D; ++d) window_table_value(g, d, cur.x[d], b[d]);   // (the bits of window_table(.., 0, ..), see basis.hpp)
                    if (hist_on) {
                        const int sl = nearest_slot<D>(g, cur.x);
                        if (sl >= 0)
                            __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double *)(sh + sl), cur.w);      // :905
                        else {                              // rare: far outside the grid (:899); the only global atomic left
                            double xr[MAXD] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                            for (int d = 0; d < D; ++d) xr[g.perm[d]] = cur.x[d];
                            atomicAdd(&hist[nearest_node_address(g, xr)], cur.w);       // :905
                        }
                    }
                }
#pragma unroll
                for (int d = 0; d < D; ++d)
#pragma unroll
                    for (int k = 0; k < 4; ++k) tab[lane * LDT + 4 * d + k] = b[d][k];
                sw[lane] = cur.w;
                swy[lane] = cur.w * cur.y;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // the next cell's first chunk is on its way while this one is multiplied
            if (p0 + PCH >= end && ci + 1 < ncl) pre = fetch(nbeg, (int)(nend - nbeg < PCH ? nend - nbeg : PCH));
            // ---- matrix cores: K = the chunk's points, four per step
            const int nsteps = (np + 3) >> 2;
            // (the LDS words of step s + 1 are asked for before the products of step s are issued)
            struct Ops { double w, t0, t1, t2[MT], wy; };
            auto lds_ops = [&](int s4) {
                const int p = 4 * s4 + q;
                Ops o;
                o.w = sw[p];
                o.t0 = tab[p * LDT + (l15 & 3)];
                o.t1 = tab[p * LDT + 4 + (l15 >> 2)];
                if constexpr (D == 3) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) o.t2[m] = tab[p * LDT + 8 + m];
                }
                o.wy = swy[p];
                return o;
            };
            Ops nx = lds_ops(0);
            for (int s4 = 0; s4 < nsteps; ++s4) {
                const Ops o = nx;
                nx = lds_ops(s4 + 1 < nsteps ? s4 + 1 : s4);
                const double u = (o.w * o.t0) * o.t1;
                double op[MT];
                if constexpr (D == 2) op[0] = u;
                else {
#pragma unroll
                    for (int m = 0; m < MT; ++m) op[m] = u * o.t2[m];
                }
                int t = 0;
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = m; n < MT; ++n, ++t)
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[n], op[m], acc[t], 0, 0, 0);
                // right-hand side: this lane's entries B[4 s + q][16 m + l15] times w^2 y of its point, summed over the steps;
                // the four q of an entry meet after the last chunk
#pragma unroll
                for (int m = 0; m < MT; ++m) racc[m] += op[m] * o.wy;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // packed lower triangle: tile (m, n), register v of lane (l15, q) = entry (r, c) = (16 n + q + 4 v, 16 m + l15)
        double *__restrict__ out = blk + (long long)rel * TRI;
        {
            int t = 0;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = m; n < MT; ++n, ++t)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int r = 16 * n + q + 4 * v, c = 16 * m + l15;
                        if (c <= r) out[r * (r + 1) / 2 + c] = acc[t][v];
                    }
        }
        // right-hand side: entry 16 m + l15 = the sum over q of the lanes (l15, q) -> through LDS to lane = entry
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            double r = racc[m];
            r += __shfl_xor(r, 16, 64);
            r += __shfl_xor(r, 32, 64);
            if (q == 0) sw[16 * m + l15] = r;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < NB) {
            rblk[(long long)rel * NB + lane] = sw[lane];
            if (hist_on) hblk[(long long)rel * NB + lane] = sh[lane];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        beg = nbeg;
        end = nend;
    }
}

// The 4-D blocks (256 window functions: 136 tiles of 16 x 16 in the lower triangle) on the matrix cores: one workgroup of TEN
// waves per cell, a wave per 64 x 64 super-block (I, J), J <= I, of the 256 x 256 block -- 16 accumulator tiles with static
// indices, as in the trailing-update kernels (the four diagonal super-blocks compute 6 tiles they do not store: 160 MFMAs per
// 4 points for 136 useful).  Operands as in gram_wave_kernel, with the fourth dimension's factor on top:
// B[p][c] = (((w b0[c0]) b1[c1]) b2[c2]) b3[c3], c = c0 + 4 c1 + 16 c2 + 64 c3: tile block m = (c2, c3) = (m & 3, m >> 2), so a
// super-block (I, J) needs t[j] = u b2[j] once and t[j] b3[I], t[j] b3[J] per step.  Round 3: the workgroup-per-cell VALU form
// took 73.8 ms at 16^4 / 10^7 points (a third of the fit) for ~11 ms of matrix-pipe time.
template <int D>
__global__ void __launch_bounds__(640)
gram_mfma4_kernel(Grid g, const int *__restrict__ offset, const double *__restrict__ xs,
                  const double *__restrict__ ys, const double *__restrict__ ws, long long cap,
                  double *__restrict__ blk, double *__restrict__ rblk, double *__restrict__ hblk,
                  double *__restrict__ hist, int cell0)
{
    static_assert(D == 4, "256 window functions");
    constexpr int NB = 256, PCH = 64, LDT = 4 * D + 1;
    constexpr long long TRI = (long long)NB * (NB + 1) / 2;
    __shared__ double tab[PCH * LDT];
    __shared__ double sw[PCH], swy[PCH];
    __shared__ int sslot[PCH];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int cell = cell0 + blockIdx.x;
    const long long beg = offset[cell], end = offset[cell + 1];
    if (beg == end) return;                    // the gather skips empty cells
    const bool hist_on = hblk != nullptr;
    // super-block of this wave: (0,0) (1,0) (1,1) (2,0) (2,1) (2,2) (3,0) (3,1) (3,2) (3,3)
    int sbI = 0, sbJ = wave;
    while (sbJ > sbI) { sbJ -= sbI + 1; ++sbI; }

    d4_t acc[4][4];                            // [column tile mi of J][row tile ni of I]
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = d4_t{0.0, 0.0, 0.0, 0.0};
    double racc = 0.0, hacc = 0.0;
    for (long long p0 = beg; p0 < end; p0 += PCH) {
        const int np = (int)((end - p0 < PCH) ? (end - p0) : PCH);
        if (tid < PCH) {                       // lane = point: window tables, weight, w^2 y, histogram slot (zero rows beyond np)
            double b[D][4], wv = 0.0, wyv = 0.0;
            int sl = -2;
#pragma unroll
            for (int d = 0; d < D; ++d)
#pragma unroll
                for (int k = 0; k < 4; ++k) b[d][k] = 0.0;
            if (tid < np) {
                double xv[D];
                wv = ws[p0 + tid];
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    xv[d] = xs[(long long)d * cap + p0 + tid];
                    window_table(g, d, xv[d], 0, b[d]);         // (the value form costs this kernel registers it does not have: +4.7 ms at 16^4)
                }
                wyv = wv * ys[p0 + tid];
                if (hist_on) {
                    sl = nearest_slot<D>(g, xv);
                    if (sl < 0) {                       // rare: far outside the grid (:899); the only atomic left
                        double xr[MAXD] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int d = 0; d < D; ++d) xr[g.perm[d]] = xv[d];
                        atomicAdd(&hist[nearest_node_address(g, xr)], wv);       // :905
                    }
                }
            }
#pragma unroll
            for (int d = 0; d < D; ++d)
#pragma unroll
                for (int k = 0; k < 4; ++k) tab[tid * LDT + 4 * d + k] = b[d][k];
            sw[tid] = wv;
            swy[tid] = wyv;
            sslot[tid] = sl;
        }
        __syncthreads();
        const int nsteps = (np + 3) >> 2;
        for (int s4 = 0; s4 < nsteps; ++s4) {
            const int p = 4 * s4 + q;
            const double u = (sw[p] * tab[p * LDT + (l15 & 3)]) * tab[p * LDT + 4 + (l15 >> 2)];
            const double bI = tab[p * LDT + 12 + sbI], bJ = tab[p * LDT + 12 + sbJ];
            double orow[4], ocol[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double t = u * tab[p * LDT + 8 + j];
                orow[j] = t * bI;
                ocol[j] = t * bJ;
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(orow[ni], ocol[mi], acc[mi][ni], 0, 0, 0);
        }
        // right-hand side and histogram shares: thread = window function, points in storage order
        if (tid < NB) {
            for (int p = 0; p < np; ++p) {
                double prod = tab[p * LDT + (tid & 3)];
#pragma unroll
                for (int d = 1; d < D; ++d) prod *= tab[p * LDT + 4 * d + ((tid >> (2 * d)) & 3)];
                racc += (sw[p] * prod) * swy[p];
                if (hist_on) hacc += (sslot[p] == tid) ? sw[p] : 0.0;       // :905
            }
        }
        __syncthreads();
    }
    // packed lower triangle: tile (m, n) = (4 J + mi, 4 I + ni), register v of lane (l15, q) = entry (r, c) = (16 n + q + 4 v, 16 m + l15)
    double *__restrict__ out = blk + (long long)blockIdx.x * TRI;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int m = 4 * sbJ + mi, n = 4 * sbI + ni;
            if (m > n) continue;               // (above the diagonal of a diagonal super-block)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int r = 16 * n + q + 4 * v, c = 16 * m + l15;
                if (c <= r) out[(long long)r * (r + 1) / 2 + c] = acc[mi][ni][v];
            }
        }
    if (tid < NB) {
        rblk[(long long)blockIdx.x * NB + tid] = racc;
        if (hist_on) hblk[(long long)blockIdx.x * NB + tid] = hacc;
    }
}

// Cells whose window contains node `in`: window starts ws_d in [max(in_d - 3, 0), min(in_d, cells_d - 1)],
// enumerated with dimension 0 fastest -- THE fixed summation order of every gather below.
template <int D>
struct CellRange {
    int lo[D], cnt[D], total;
    __device__ CellRange(const Grid &g, const int *in)
    {
        total = 1;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            lo[d] = in[d] - 3 > 0 ? in[d] - 3 : 0;
            const int hi = in[d] < g.cells[d] - 1 ? in[d] : g.cells[d] - 1;
            cnt[d] = hi - lo[d] + 1;
            total *= cnt[d];
        }
    }
    // e-th cell: its linear index, and the node's local basis index r inside that cell's window
    __device__ void get(const Grid &g, const int *in, int e, int &cell, int &r) const
    {
        cell = 0;
        r = 0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int wsd = lo[d] + e % cnt[d];
            e /= cnt[d];
            cell += wsd * g.cellstride[d];
            r += (in[d] - wsd) << (2 * d);
        }
    }
};

// One wave per node i: stencil row nst[i][*], rhs[i] and the node's histogram entry as the sums of
// the per-cell blocks, cell after cell in CellRange order.  Entry (r, c) of a block, c <= r, belongs
// to column offset c - r (digit-wise), i.e. stencil code sum_d (c_d - r_d + 3) 7^d; for one cell the
// lanes c = 0..r hit distinct codes, so the wave accumulates in an LDS row without conflicts.
// ZEROED: the blocks of empty cells were written as zeros (gram_wave_kernel), so no cell's point count is looked up and the
// rows of a batch do not wait for a first round of loads (round 3: two dependent round trips per batch of 8 cells -> one).
template <int D, bool ZEROED = false>
__global__ void __launch_bounds__(256)
stencil_gather_kernel(Grid g, const int *__restrict__ offset, const double *__restrict__ blk,
                      const double *__restrict__ rblk, const double *__restrict__ hblk,
                      double *__restrict__ nst, double *__restrict__ rhs, double *__restrict__ hist,
                      int cell0, int cell1, int node0, int node1)
{
    constexpr int NB = 1 << (2 * D);
    constexpr long long TRI = (long long)NB * (NB + 1) / 2;
    constexpr int HS = (D == 1) ? 4 : (D == 2) ? 25 : (D == 3) ? 172 : 1201;
    __shared__ double sacc[4][HS];
    __shared__ double stmp[4][128];
    // (the wave index as a SCALAR: node, cell walk, row and block addresses below are then scalar arithmetic -- as vector
    //  code with a division per dimension and cell, and seven integer operations per lane for the stencil code, the kernel ran
    //  at the same 10.7 ns per node whether its blocks came from HBM or from the Infinity Cache: issue-bound, round 5)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // the cells [cell0, cell1) of the current slab (whole hyper-rows of the slowest dimension) touch the
    // nodes [node0, node1); with one slab that is everything.  Sums of successive slabs add up in nst
    // (zero on entry), in slab order: still one owner and a fixed order per entry.
    const int node = node0 + blockIdx.x * 4 + wave;
    if (node >= node1) return;                 // whole waves leave; no workgroup barrier below
    double *acc = sacc[wave];
    for (int e = lane; e < HS; e += 64) acc[e] = 0.0;
    int in[D];
#pragma unroll
    for (int d = 0; d < D; ++d) in[d] = (node / g.colstride[d]) % g.nodes[d];
    const CellRange<D> cr(g, in);
    double racc = 0.0, hacc = 0.0;
    // Cells in batches of GB, three phases per batch so that no load waits for a decision that needs an
    // earlier load of the same batch: (1) the cells' point counts, (2) the block rows of the non-empty ones
    // (predicated, never a branch), (3) the LDS accumulation in cell order.  A cell at a time made every
    // wave wait out two dependent memory round trips per cell (3.4 ms at 64^3).
    constexpr int GB = (D <= 3) ? 8 : 4;
    constexpr int NCH = (NB + 63) / 64;              // 64-column chunks of a block row
    // stencil code of entry (r, c) = K(c) - K(r) + centre with K(v) = sum_d v_d 7^d over the 2-bit digits of v: K(c) once per
    // lane, K(r) with the row
    int kc[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int c = lane + 64 * ch;
        int k = 0, m7 = 1;
#pragma unroll
        for (int d = 0; d < D; ++d) { k += ((c >> (2 * d)) & 3) * m7; m7 *= 7; }
        kc[ch] = k + (HS - 1);
    }
    if constexpr (D == 3 && ZEROED) {
        // 3-D: the walk as three nested loops with the row address carried along a row of cells (next cell: + TRI, one row up
        // in the triangle: - r) -- a dozen scalar instructions per cell.  The generic walk below spends ~140 (64-bit products
        // for every address, the counters' carries, the slab tests): the kernel took 1.8 ms of its 2.9 ms at 64^3 with every
        // load, LDS access and store removed (round 5).  Same cells in the same order.
        // Right-hand side and histogram shares: lane = cell, two gathers for the node instead of two one-lane loads per cell
        {
            double rv = 0.0, hv = 0.0;
            if (lane < cr.total) {
                int cell, r;
                cr.get(g, in, lane, cell, r);
                if (cell >= cell0 && cell < cell1) {
                    const long long cb = (long long)(cell - cell0);
                    rv = rblk[cb * NB + r];
                    if (hblk) hv = hblk[cb * NB + r];
                }
            }
            // summed cell after cell, as the generic walk does (the bits of rounds 1-4: a butterfly over the lanes moved the
            // second refinement correction of config 3 from 2.3e-8 to 3.9e-8 and the stopping rule's estimate across its 1e-11)
            double *tmp = stmp[wave];
            tmp[lane] = rv;
            tmp[64 + lane] = hv;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int e = 0; e < cr.total; ++e) {
                racc += tmp[e];
                hacc += tmp[64 + e];
            }
        }
        const int nx = cr.cnt[0], ny = cr.cnt[1], nz = cr.cnt[2];
        const int rx0 = in[0] - cr.lo[0];
        for (int iz = 0; iz < nz; ++iz) {
            const int cz = cr.lo[2] + iz;
            const int cellz = cz * g.cellstride[2];
            const bool zok = cellz >= cell0 && cellz < cell1;       // (a slab = whole hyper-rows of the slowest dimension)
            if (!zok) continue;
            for (int iy = 0; iy < ny; iy += 2) {
                double v[2][4];
                int rs[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const bool yok = iy + h < ny;
                    const int cy = cr.lo[1] + (yok ? iy + h : iy);
                    const int cell = cellz + cy * g.cellstride[1] + cr.lo[0];
                    int r = ((in[2] - cz) << 4) + ((in[1] - cy) << 2) + rx0;
                    rs[h] = yok ? r : -1;
                    long long off = (long long)(cell - cell0) * TRI + (long long)r * (r + 1) / 2;
#pragma unroll
                    for (int xi = 0; xi < 4; ++xi) {
                        const bool ok = yok && xi < nx;
                        v[h][xi] = (ok && lane <= r) ? blk[off + lane] : 0.0;
                        off += TRI - r;
                        r -= 1;
                    }
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    int r = rs[h];
                    if (r < 0) continue;
                    int k = (in[2] - cz) * 49 + ((r >> 2) & 3) * 7 + rx0;
#pragma unroll
                    for (int xi = 0; xi < 4; ++xi) {
                        if (xi < nx && lane <= r) acc[kc[0] - k] += v[h][xi];
                        r -= 1;
                        k -= 1;
                    }
                }
            }
        }
    } else {
    // the cells in CellRange order (dimension 0 fastest) by counters, not by divisions
    int ix[D];
#pragma unroll
    for (int d = 0; d < D; ++d) ix[d] = 0;
    for (int e0 = 0; e0 < cr.total; e0 += GB) {
        int cl[GB], rr[GB], kr[GB], o0[GB], o1[GB];
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            int cell = 0, r = 0, k = 0;
            const bool have = e0 + u < cr.total;
            if (have) {
                int m7 = 1;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const int wsd = cr.lo[d] + ix[d];
                    cell += wsd * g.cellstride[d];
                    r += (in[d] - wsd) << (2 * d);
                    k += (in[d] - wsd) * m7;
                    m7 *= 7;
                }
                bool carry = true;
#pragma unroll
                for (int d = 0; d < D; ++d)
                    if (carry) {
                        if (++ix[d] < cr.cnt[d]) carry = false;
                        else ix[d] = 0;
                    }
            }
            kr[u] = k;
            const bool ok = have && cell >= cell0 && cell < cell1;
            cl[u] = ok ? cell : cell0;               // a harmless cell for the lookups below
            rr[u] = ok ? r : -1;
            if (!ZEROED) {
                o0[u] = offset[cl[u]];
                o1[u] = offset[cl[u] + 1];
            } else
                o0[u] = 0, o1[u] = 1;
        }
        double v[GB][NCH], rv[GB], hv[GB];
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            if (o0[u] == o1[u]) rr[u] = -1;          // empty cell: its block was never written
            const int r = rr[u];
            const long long cb = (long long)(cl[u] - cell0);
            const double *__restrict__ row = blk + cb * TRI + (long long)(r < 0 ? 0 : r) * ((r < 0 ? 0 : r) + 1) / 2;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int c = lane + 64 * ch;
                v[u][ch] = (c <= r) ? row[c] : 0.0;
            }
            rv[u] = (lane == 0 && r >= 0) ? rblk[cb * NB + r] : 0.0;
            hv[u] = (lane == 0 && r >= 0 && hblk) ? hblk[cb * NB + r] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            const int r = rr[u];
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int c = lane + 64 * ch;
                if (c <= r) acc[kc[ch] - kr[u]] += v[u][ch];
            }
            racc += rv[u];
            hacc += hv[u];
        }
    }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double *__restrict__ out = nst + (long long)node * g.hstencil;
    for (int e = lane; e < HS; e += 64) out[e] += acc[e];
    if (lane == 0) {
        rhs[node] += racc;
        if (hblk) {
            int refnode = 0;                   // the histogram is kept in the caller's dimension order
#pragma unroll
            for (int d = 0; d < D; ++d) refnode += in[d] * g.refstride[d];
            hist[refnode] += hacc;             // on top of the out-of-window points added by gram_block_kernel
        }
    }
}

// totlwt (:906) = the sum of the histogram (every counted point is in it), in a fixed order
__global__ void __launch_bounds__(1024)
hist_total_kernel(const double *__restrict__ hist, int n, double *__restrict__ scal)
{
    __shared__ double part[1024];
    const int t = threadIdx.x;
    const double s = strided_sum8(hist, n, t, 1024);
    part[t] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (t < o) part[t] += part[t + o];
        __syncthreads();
    }
    if (t == 0) scal[SC_TOTLWT] = part[0];
}

template <int D>
struct ResCfg;
template <> struct ResCfg<4> { static constexpr int NB = 256, NT = 256, PCH = 16; };

// per-cell share of rho = A^T W (W y - W A x): rcell[cell][c] (plain stores; empty cells are skipped by the gather).
// Workgroup per cell: the 4-D form (256 window functions); 1-D .. 3-D grids use residual_wave_kernel below.
template <int D>
__global__ void __launch_bounds__(ResCfg<D>::NT)
residual_block_kernel(Grid g, const int *__restrict__ offset, const double *__restrict__ xs,
                      const double *__restrict__ ys, const double *__restrict__ ws, long long cap,
                      const double *__restrict__ xvec, double *__restrict__ rcell, double *__restrict__ ssq)
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
    if (tid < NB) xloc[tid] = xvec[local_col<D>(g, colbase, tid)];
    double racc = 0.0, e2 = 0.0;
    for (long long p0 = beg; p0 < end; p0 += PCH) {
        const int np = (int)((end - p0 < PCH) ? (end - p0) : PCH);
        stage_points<D, NB, LDB, NT>(g, xs, ys, ws, cap, p0, np, tab, bw, wy, wt, nullptr, nullptr);
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
    if (tid < NB) rcell[(long long)cell * NB + tid] = racc;
    if (ssq) {                                   // sum of squared row residuals (the reference's errsum; a diagnostic):
        e2 = wave_sum(e2);                       // the cell's share, its waves added in a fixed order (no atomics: reproducible)
        if ((tid & 63) == 0) wt[tid >> 6] = e2;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int wv = 0; wv < NT / 64; ++wv) t += wt[wv];
            ssq[cell] = t;
        }
    }
}

// The same per-cell share for 1-D .. 3-D grids (NB = 4^D <= 64), ONE WAVE per cell, four cells per workgroup, no
// workgroup barriers (round 3: the workgroup-per-cell form above spent 1.67 ms per pass at C3 -- 227 000 workgroups of 256
// threads for 44 points each, staged through five __syncthreads -- for 0.4 GB of points; four passes per fit).
//   phase 1  lane = point:  the D window tables (parked in the wave's LDS slice), t = (w b) . x against the cell's 4^D
//            coefficients (LDS broadcast reads), e = w y - t
//   phase 2  lane = (window function c, point group):  racc_c += (w b)_c * e  over the points
// Sums in a fixed order: reproducible bits.
template <int D>
__global__ void __launch_bounds__(256)
residual_wave_kernel(Grid g, const int *__restrict__ offset, const double *__restrict__ xs,
                     const double *__restrict__ ys, const double *__restrict__ ws, long long cap,
                     const double *__restrict__ xvec, double *__restrict__ rcell, double *__restrict__ ssq)
{
    static_assert(D >= 1 && D <= 3, "one lane per window function");
    constexpr int NB = 1 << (2 * D), G = 64 / NB, PCH = 64, LDT = 4 * D + 1;
    __shared__ double s_tab[4][PCH * LDT];
    __shared__ double s_we[4][PCH];
    __shared__ double s_wt[4][PCH];
    __shared__ double s_x[4][NB];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int cell = blockIdx.x * 4 + wave;
    if (cell >= g.ncell) return;
    const long long beg = offset[cell], end = offset[cell + 1];
    if (beg == end) return;
    double *tab = s_tab[wave], *we = s_we[wave], *sw = s_wt[wave], *xl = s_x[wave];
    int colbase = 0;
#pragma unroll
    for (int d = 0; d < D; ++d) colbase += ((cell / g.cellstride[d]) % g.cells[d]) * g.colstride[d];
    if (lane < NB) xl[lane] = xvec[local_col<D>(g, colbase, lane)];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int c = lane % NB, grp = lane / NB;
    double racc = 0.0, e2 = 0.0;
    for (long long p0 = beg; p0 < end; p0 += PCH) {
        const int np = (int)((end - p0 < PCH) ? (end - p0) : PCH);
        if (lane < np) {
            double b[D][4];
#pragma unroll
            for (int d = 0; d < D; ++d) {
                window_table_value(g, d, xs[(long long)d * cap + p0 + lane], b[d]);
#pragma unroll
                for (int k = 0; k < 4; ++k) tab[lane * LDT + 4 * d + k] = b[d][k];
            }
            // t = (w b) . x with the row entries rounded exactly as the Gram kernel rounds them, ((w b0) b1) b2: the
            // refinement iterates with the operator whose Gram matrix was factored.  (Measured at C3: the contraction
            // factor is the same 2.2e-4 with the factorised window sum -- it is set by the Gram sums and the
            // factorisation, not by the rounding of the row entries; the consistent form costs nothing measurable.)
            const double wv = ws[p0 + lane];
            double t = 0.0;
            if constexpr (D == 1) {
#pragma unroll
                for (int k0 = 0; k0 < 4; ++k0) t = fma(wv * b[0][k0], xl[k0], t);
            } else {
                double u[4][4];
#pragma unroll
                for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
                    for (int k0 = 0; k0 < 4; ++k0) u[k1][k0] = (wv * b[0][k0]) * b[1][k1];
                if constexpr (D == 2) {
#pragma unroll
                    for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
                        for (int k0 = 0; k0 < 4; ++k0) t = fma(u[k1][k0], xl[k0 + 4 * k1], t);
                } else {
#pragma unroll
                    for (int k2 = 0; k2 < 4; ++k2)
#pragma unroll
                        for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
                            for (int k0 = 0; k0 < 4; ++k0) t = fma(u[k1][k0] * b[2][k2], xl[k0 + 4 * k1 + 16 * k2], t);
                }
            }
            const double e = (ys ? wv * ys[p0 + lane] : 0.0) - t; // row residual  w y - (w b) . x  (ys == NULL: -(w b) . x, the rows as an operator)
            we[lane] = e;
            sw[lane] = wv;
            e2 = fma(e, e, e2);
        } else {                                                  // zero rows pad the last group of the loop below
#pragma unroll
            for (int k = 0; k < 4 * D; ++k) tab[lane * LDT + k] = 0.0;
            we[lane] = 0.0;
            sw[lane] = 0.0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // four points per trip, their LDS words asked for together (one point per trip waited out the LDS latency for each:
        // 5 us of a wave's time per cell at C3, as in the Gram kernel's right-hand side before round 5); the padding rows
        // add +0.0, the order of the sum is the points' order as before
        const int ntrip = (np + 4 * G - 1) / (4 * G);
        for (int it = 0; it < ntrip; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int p = grp + G * (4 * it + k);
                double prod = sw[p] * tab[p * LDT + (c & 3)];
#pragma unroll
                for (int d = 1; d < D; ++d) prod *= tab[p * LDT + 4 * d + ((c >> (2 * d)) & 3)];
                racc = fma(prod, we[p], racc);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int o = NB; o < 64; o <<= 1) racc += __shfl_xor(racc, o, 64);
    if (lane < NB) rcell[(long long)cell * NB + lane] = racc;
    if (ssq) {
        e2 = wave_sum(e2);
        if (lane == 0) ssq[cell] = e2;
    }
}

// The 4-D form of the same share (256 window functions), one workgroup of four waves per cell (round 3: the staged form above,
// which builds the full 16 x 256 row image of a chunk in LDS, took 7.5 ms per pass at 16^4 / 10^7 points):
//   phase 1  wave k3, lane = point: b3[k3] * (factorised window sum of the slab k3 of the cell's coefficients) -> 4 partials
//   phase 2  thread = window function: racc_c += (w b)_c * e over the points
template <int D>
__global__ void __launch_bounds__(256)
residual_cell4_kernel(Grid g, const int *__restrict__ offset, const double *__restrict__ xs,
                      const double *__restrict__ ys, const double *__restrict__ ws, long long cap,
                      const double *__restrict__ xvec, double *__restrict__ rcell, double *__restrict__ ssq)
{
    static_assert(D == 4, "256 window functions");
    constexpr int NB = 256, PCH = 64, LDT = 4 * D + 1;
    __shared__ double tab[PCH * LDT];
    __shared__ double sw[PCH], sy[PCH], se[PCH];
    __shared__ double part[4][PCH];
    __shared__ double xl[NB];
    const int cell = blockIdx.x;
    const long long beg = offset[cell], end = offset[cell + 1];
    if (beg == end) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int colbase = 0;
#pragma unroll
    for (int d = 0; d < D; ++d) colbase += ((cell / g.cellstride[d]) % g.cells[d]) * g.colstride[d];
    xl[tid] = xvec[local_col<D>(g, colbase, tid)];
    double racc = 0.0, e2 = 0.0;
    for (long long p0 = beg; p0 < end; p0 += PCH) {
        const int np = (int)((end - p0 < PCH) ? (end - p0) : PCH);
        if (tid < np) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                double b[4];
                window_table_value(g, d, xs[(long long)d * cap + p0 + tid], b);
#pragma unroll
                for (int k = 0; k < 4; ++k) tab[tid * LDT + 4 * d + k] = b[k];
            }
            sw[tid] = ws[p0 + tid];
            sy[tid] = ys ? ys[p0 + tid] : 0.0;      // (ys == NULL: the rows as an operator, rho = -N x: pcg.hip)
        }
        __syncthreads();
        if (lane < np) {                        // slab k3 = wave of the window sum of point `lane`
            const double *__restrict__ tb = tab + lane * LDT;
            double r3 = 0.0;
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
                double r2 = 0.0;
#pragma unroll
                for (int k1 = 0; k1 < 4; ++k1) {
                    double r = 0.0;
#pragma unroll
                    for (int k0 = 0; k0 < 4; ++k0) r = fma(tb[k0], xl[k0 + 4 * k1 + 16 * k2 + 64 * wave], r);
                    r2 = fma(tb[4 + k1], r, r2);
                }
                r3 = fma(tb[8 + k2], r2, r3);
            }
            part[wave][lane] = tb[12 + wave] * r3;
        }
        __syncthreads();
        if (tid < np) {
            const double t = ((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid];
            const double e = sw[tid] * sy[tid] - sw[tid] * t;      // row residual  w y - (w b) . x
            se[tid] = e;
            e2 = fma(e, e, e2);
        }
        __syncthreads();
        for (int p = 0; p < np; ++p) {
            const double *__restrict__ tb = tab + p * LDT;
            const double prod = (((sw[p] * tb[tid & 3]) * tb[4 + ((tid >> 2) & 3)]) * tb[8 + ((tid >> 4) & 3)]) * tb[12 + (tid >> 6)];
            racc = fma(prod, se[p], racc);
        }
        __syncthreads();
    }
    rcell[(long long)cell * NB + tid] = racc;
    if (ssq && wave == 0) {
        e2 = wave_sum(e2);
        if (lane == 0) ssq[cell] = e2;
    }
}

// ---------------------------------------------------------------------------
// Derivative-constraint rows (:921-1046).  A data-sparse node n (histogram below spcrit = 0.75 of the
// expected weight, :936) emits D(D+1)/2 rows, one per pair idm <= jdm, whose entries sit on the 3^D
// nodes around n:  row(n, pair)[j] = rowwt * prod_d bas1(nderiv_d; x_n; node j).
struct SparseNode {
    bool sparse;
    double dcwght;
};

template <int D>
__device__ inline SparseNode sparse_node(const Grid &g, const int *in, const double *__restrict__ hist,
                                         double totlwt, double xtrap)
{
#pragma clang fp contract(off)
    long long nrect = 1;
#pragma unroll
    for (int d = 0; d < D; ++d) nrect *= (g.nodes[d] - 1);
    const double wtprrc = totlwt / (double)nrect;                 // :910
    double expect = wtprrc;
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (in[d] == 0 || in[d] == g.nodes[d] - 1) expect = 0.5 * expect;     // :928
    int refnode = 0;                                              // the histogram is in the caller's order
#pragma unroll
    for (int d = 0; d < D; ++d) refnode += in[d] * g.refstride[d];
    const double have = hist[refnode];
    SparseNode s;
    s.sparse = have < 0.75 * expect;                              // spcrit, :696, :936
    s.dcwght = xtrap * (expect - have);                           // :938, :960
    return s;
}

// derivative orders and weight of constraint row `pair` (idm <= jdm enumerated row by row) of node `in`
template <int D>
__device__ inline double constraint_pattern(const Grid &g, const int *in, int idm, int jdm, double dcwght, int *nder)
{
#pragma clang fp contract(off)
#pragma unroll
    for (int d = 0; d < D; ++d) nder[d] = 0;
    bool boundary = true;
    double rowwt = 2.0 * dcwght;                                  // :983
    if (jdm == idm) {
        rowwt = dcwght;
        nder[jdm] = 2;
        if (in[idm] != 0 && in[idm] != g.nodes[idm] - 1) boundary = false;
    }
    if (boundary) { nder[idm] = 1; nder[jdm] = 1; }                // :998-999
    return rowwt;
}

// one dimension's factor of a constraint-row entry: derivative `nder` of the basis function of node n + o at node n
__device__ inline double constraint_factor(const Grid &g, int d, int n, int o, int nder)
{
#pragma clang fp contract(off)
    const int ib = n + o;
    if (ib < 0 || ib > g.nodes[d] - 1) return 0.0;
    const double xnode = g.xmin[d] + (double)n * g.dx[d];         // :943
    const double xb = g.xmin[d] + (double)ib * g.dx[d];
    return basis_1d(basis_kind(ib, g.nodes[d]), nder, xnode, xb, g.dxin[d]);
}

// The factors depend on the grid alone: constraint_table_kernel tabulates them once per plan,
//   ctab[9 * (nodes_0 + .. + nodes_{d-1}) + (n * 3 + o + 1) * 3 + nder],
// and the row kernels multiply table entries instead of evaluating 3 piecewise cubics per entry (round 3: the constraint
// rows took 2.7 ms per fit at 64^3).  Same values, same order of the product: identical bits.  ctab == NULL: evaluate.
__global__ void __launch_bounds__(256)
constraint_table_kernel(Grid g, double *__restrict__ ctab)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x, base = 0;
    for (int d = 0; d < g.ndim; ++d) {
        const int cnt = 9 * g.nodes[d];
        if (t < cnt) {
            const int nder = t % 3, o = (t / 3) % 3 - 1, n = t / 9;
            ctab[base + t] = constraint_factor(g, d, n, o, nder);
            return;
        }
        t -= cnt;
        base += cnt;
    }
}

// entry of the constraint row of node n (coordinates nn) at node j = nn + off (off_d in [-1,1]); 0 outside the grid
template <int D>
__device__ inline double constraint_entry(const Grid &g, const double *__restrict__ ctab, const int *nn, const int *off,
                                          const int *nder, double rowwt)
{
#pragma clang fp contract(off)
    double basm = 1.0;
    int base = 0;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const int ib = nn[d] + off[d];
        if (ib < 0 || ib > g.nodes[d] - 1) return 0.0;
        basm *= ctab ? ctab[base + (nn[d] * 3 + off[d] + 1) * 3 + nder[d]] : constraint_factor(g, d, nn[d], off[d], nder[d]);
        base += 9 * g.nodes[d];
    }
    return rowwt * basm;                                          // :1011
}

// Pre-pass: which nodes are data sparse, and their constraint weight -- once per fit (the histogram is
// final), so that the row gathers below look a neighbour up with one byte instead of re-deriving it
template <int D>
__global__ void __launch_bounds__(256)
sparse_mark_kernel(Grid g, const double *__restrict__ hist, const double *__restrict__ scal, double xtrap,
                   double *__restrict__ dcw, unsigned char *__restrict__ spf)
{
    const int node = blockIdx.x * blockDim.x + threadIdx.x;
    if (node >= g.ncol) return;
    int in[D];
#pragma unroll
    for (int d = 0; d < D; ++d) in[d] = (node / g.colstride[d]) % g.nodes[d];
    const SparseNode sn = sparse_node<D>(g, in, hist, scal[SC_TOTLWT], xtrap);
    dcw[node] = sn.dcwght;
    spf[node] = sn.sparse ? 1 : 0;
}

// rows of the constraint system: ndim (ndim + 1) / 2 per data-sparse node (:974-1000) -> scal_out[SC_NROWS_CONS]
__global__ void __launch_bounds__(1024)
count_sparse_kernel(const unsigned char *__restrict__ spf, int ncol, int rows_per_node, double *__restrict__ scal_out)
{
    __shared__ int red[16];
    int c = 0;
    for (int i = threadIdx.x; i < ncol; i += 1024) c += spf[i] != 0 ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int v = 0; v < 16; ++v) t += red[v];
        scal_out[SC_NROWS_CONS] += (double)t * (double)rows_per_node;
    }
}

// One wave per stencil row i: nst[i][code(j - i)] += sum over the sparse nodes n within one node of
// both i and j, and over n's rows, of row[i] * row[j] -- neighbours and rows in a fixed order, the
// row's owner adds with plain read-modify-writes.  Also counts the rows (scal_out[SC_NROWS_CONS]).
template <int D>
__global__ void __launch_bounds__(256)
constraint_rows_kernel(Grid g, const double *__restrict__ dcw, const unsigned char *__restrict__ spf, const double *__restrict__ ctab,
                       double *__restrict__ nst, double *__restrict__ scal_out)
{
    constexpr int NE = (D == 1) ? 3 : (D == 2) ? 9 : (D == 3) ? 27 : 81;
    constexpr int HS = (D == 1) ? 4 : (D == 2) ? 25 : (D == 3) ? 172 : 1201;
    __shared__ double sacc[4][HS];
    __shared__ double sct[4][D * 27], sdn[4][NE];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int node = blockIdx.x * 4 + wave;
    if (node >= g.ncol) return;
    double *acc = sacc[wave];
    int in[D];
#pragma unroll
    for (int d = 0; d < D; ++d) in[d] = (node / g.colstride[d]) % g.nodes[d];
    // lanes look the 3^D neighbours up in parallel: sparse[r] = which of the neighbours 64 r .. 64 r + 63 are data sparse.  Most
    // rows of a well-covered grid leave here; the others walk the SET bits only (the walk used to re-read the flag of every
    // neighbour, one dependent load after the other: 27 of them for 1.6 sparse neighbours at C3)
    unsigned long long sparse[(NE + 63) / 64];
#pragma unroll
    for (int r = 0; r < (NE + 63) / 64; ++r) {
        const int ne = lane + 64 * r;
        bool mine = false;
        if (ne < NE) {
            int t = ne, col = 0;
            bool ok = true;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int nd_ = in[d] + t % 3 - 1;
                t /= 3;
                ok = ok && nd_ >= 0 && nd_ <= g.nodes[d] - 1;
                col += nd_ * g.colstride[d];
            }
            mine = ok && spf[col] != 0;
        }
        sparse[r] = __builtin_amdgcn_ballot_w64(mine);
    }
    {
        unsigned long long anyb = 0;
#pragma unroll
        for (int r = 0; r < (NE + 63) / 64; ++r) anyb |= sparse[r];
        if (anyb == 0) return;
    }
    for (int e = lane; e < HS; e += 64) acc[e] = 0.0;
    // The factors this row can meet -- nodes in[d] - 1 .. in[d] + 1 of every dimension, 27 D of them -- and the weights of the 3^D
    // neighbours go to LDS first: the walk below then multiplies LDS words instead of chasing three dependent table loads per
    // entry through L2 (round 5: 1.84 ms of a 9.7 ms assembly at 64^3 for 17 000 data-sparse nodes).  Same values, same order.
    double *ct = sct[wave], *dn = sdn[wave];
    {
        for (int e = lane; e < D * 27; e += 64) {
            const int d = e / 27, r = e % 27, n = in[d] + r / 9 - 1;
            int base = 0;
            for (int q = 0; q < d; ++q) base += 9 * g.nodes[q];
            double v = 0.0;
            if (n >= 0 && n <= g.nodes[d] - 1) v = ctab ? ctab[base + n * 9 + r % 9] : constraint_factor(g, d, n, (r % 9) / 3 - 1, r % 3);
            ct[e] = v;
        }
        for (int ne = lane; ne < NE; ne += 64) {
            int t = ne, col = 0;
#pragma unroll
            for (int d = 0; d < D; ++d) { col += (in[d] + t % 3 - 1) * g.colstride[d]; t /= 3; }
            dn[ne] = ((sparse[ne >> 6] >> (ne & 63)) & 1ull) ? dcw[col] : 0.0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // entry of the constraint row of node nn at nn + off, from the LDS copy (constraint_entry's product, factor by factor)
    auto entry = [&](const int *nn, const int *off, const int *nder, double rowwt) -> double {
#pragma clang fp contract(off)
        double basm = 1.0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int ib = nn[d] + off[d];
            if (ib < 0 || ib > g.nodes[d] - 1) return 0.0;
            basm *= ct[d * 27 + (nn[d] - in[d] + 1) * 9 + (off[d] + 1) * 3 + nder[d]];
        }
        return rowwt * basm;
    };
    bool any = false;
    for (int ne = 0; ne < NE; ++ne) {            // neighbour n = i + offn, offn_d in [-1,1], dim 0 fastest
        if (!((sparse[ne >> 6] >> (ne & 63)) & 1ull)) continue;        // (in the grid and data sparse)
        int nn[D], offi[D], t = ne;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int o = t % 3 - 1;
            t /= 3;
            nn[d] = in[d] + o;
            offi[d] = -o;                        // i = n + offi
        }
        SparseNode sn;
        sn.sparse = true;
        sn.dcwght = dn[ne];
        any = true;
        // (the rows are counted by count_sparse_kernel: one f64 atomicAdd per data-sparse node on ONE word -- a compare-and-swap
        //  loop on this build -- serialised 17 000 of them at config 3 and 177 000 at 4-D 28^4)
        for (int idm = 0; idm < D; ++idm)
            for (int jdm = idm; jdm < D; ++jdm) {
                int nder[D];
                const double rowwt = constraint_pattern<D>(g, nn, idm, jdm, sn.dcwght, nder);
                const double ci = entry(nn, offi, nder, rowwt);
                if (ci == 0.0) continue;         // wave-uniform
                for (int je = lane; je < NE; je += 64) {
                    int offj[D], tt = je, code = 0, m7 = 1;
                    bool lower = true, decided = false;
#pragma unroll
                    for (int d = 0; d < D; ++d) { offj[d] = tt % 3 - 1; tt /= 3; }
                    // column j = n + offj must not exceed row i = n + offi in the linear order
                    // (highest dimension most significant)
#pragma unroll
                    for (int d = D - 1; d >= 0; --d) {
                        if (!decided && offj[d] != offi[d]) { lower = offj[d] < offi[d]; decided = true; }
                    }
#pragma unroll
                    for (int d = 0; d < D; ++d) { code += (offj[d] - offi[d] + 3) * m7; m7 *= 7; }
                    if (!lower) continue;
                    const double cj = entry(nn, offj, nder, rowwt);
                    if (cj != 0.0) acc[code] += ci * cj;
                }
            }
    }
    if (!any) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double *__restrict__ out = nst + (long long)node * g.hstencil;
    for (int e = lane; e < HS; e += 64)
        if (acc[e] != 0.0) out[e] += acc[e];
}

// residual mode, step 1: t[n][pair] = row(n, pair) . x for every sparse node (0 otherwise); one wave per node
template <int D>
__global__ void __launch_bounds__(256)
constraint_dots_kernel(Grid g, const double *__restrict__ dcw, const unsigned char *__restrict__ spf, const double *__restrict__ ctab,
                       const double *__restrict__ xvec, double *__restrict__ tbuf, double *__restrict__ ssq)
{
    constexpr int NE = (D == 1) ? 3 : (D == 2) ? 9 : (D == 3) ? 27 : 81;
    constexpr int NP = D * (D + 1) / 2;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int node = blockIdx.x * 4 + wave;
    if (node >= g.ncol) return;
    int nn[D];
#pragma unroll
    for (int d = 0; d < D; ++d) nn[d] = (node / g.colstride[d]) % g.nodes[d];
    SparseNode sn;
    sn.sparse = spf[node] != 0;
    sn.dcwght = dcw[node];
    double *__restrict__ out = tbuf + (long long)node * NP;
    if (!sn.sparse) {
        if (lane < NP) out[lane] = 0.0;
        return;
    }
    int pair = 0;
    double e2 = 0.0;
    for (int idm = 0; idm < D; ++idm)
        for (int jdm = idm; jdm < D; ++jdm, ++pair) {
            int nder[D];
            const double rowwt = constraint_pattern<D>(g, nn, idm, jdm, sn.dcwght, nder);
            double t = 0.0;
            for (int je = lane; je < NE; je += 64) {
                int offj[D], tt = je, col = 0;
                bool ok = true;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    offj[d] = tt % 3 - 1;
                    tt /= 3;
                    const int ib = nn[d] + offj[d];
                    ok = ok && ib >= 0 && ib <= g.nodes[d] - 1;
                    col += ib * g.colstride[d];
                }
                if (ok) t += constraint_entry<D>(g, ctab, nn, offj, nder, rowwt) * xvec[col];
            }
            t = wave_sum(t);
            if (lane == 0) out[pair] = t;
            e2 += t * t;                         // constraint rows have rhs 0
        }
    if (ssq && lane == 0) ssq[node] = e2;      // the node's share of the squared constraint residuals (summed in a fixed order later)
}

// rho[i] = sum over the cells that contain node i of their share (CellRange order)
//          - sum over sparse neighbours n and their rows of row(n,pair)[i] * t[n][pair]   (tbuf != NULL)
template <int D>
__global__ void __launch_bounds__(256)
rho_gather_kernel(Grid g, const int *__restrict__ offset, const double *__restrict__ rcell,
                  const double *__restrict__ dcw, const unsigned char *__restrict__ spf, const double *__restrict__ ctab,
                  const double *__restrict__ tbuf, double *__restrict__ rho)
{
    constexpr int NB = 1 << (2 * D);
    constexpr int NE = (D == 1) ? 3 : (D == 2) ? 9 : (D == 3) ? 27 : 81;
    constexpr int NP = D * (D + 1) / 2;
    const int node = blockIdx.x * blockDim.x + threadIdx.x;
    if (node >= g.ncol) return;
    int in[D];
#pragma unroll
    for (int d = 0; d < D; ++d) in[d] = (node / g.colstride[d]) % g.nodes[d];
    const CellRange<D> cr(g, in);
    double acc = 0.0;
    for (int e = 0; e < cr.total; ++e) {
        int cell, r;
        cr.get(g, in, e, cell, r);
        if (offset[cell] != offset[cell + 1]) acc += rcell[(long long)cell * NB + r];
    }
    if (tbuf) {
        for (int ne = 0; ne < NE; ++ne) {
            int nn[D], offi[D], t = ne, ncol_n = 0;
            bool ok = true;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int o = t % 3 - 1;
                t /= 3;
                nn[d] = in[d] + o;
                offi[d] = -o;
                ok = ok && nn[d] >= 0 && nn[d] <= g.nodes[d] - 1;
                ncol_n += nn[d] * g.colstride[d];
            }
            if (!ok) continue;
            if (spf[ncol_n] == 0) continue;
            SparseNode sn;
            sn.sparse = true;
            sn.dcwght = dcw[ncol_n];
            int pair = 0;
            for (int idm = 0; idm < D; ++idm)
                for (int jdm = idm; jdm < D; ++jdm, ++pair) {
                    int nder[D];
                    const double rowwt = constraint_pattern<D>(g, nn, idm, jdm, sn.dcwght, nder);
                    const double ci = constraint_entry<D>(g, ctab, nn, offi, nder, rowwt);
                    if (ci != 0.0) acc -= ci * tbuf[(long long)ncol_n * NP + pair];
                }
        }
    }
    rho[node] = acc;
}

// Componentwise backward error of the returned coefficients with respect to the rows:
//   omega = max_i |rho_i| / ((|N| |x|)_i + |r_i|),   rho = A^T W (W y - W A x) - C^T C x  (from the rows),
// N = the assembled normal equations (half stencil, both triangles visited), r = A^T W^2 y.  The
// denominator is the size of the terms whose sum rho_i is (the basis functions are non-negative, so
// |A|^T |A| = N on the data rows): omega is at rounding level exactly when x minimises the
// least-squares functional to working precision, whatever the grading of the constraint weights.
// (Thread per node on purpose: neighbouring threads walk neighbouring rows code by code, so every 128-byte line of nst that
// is fetched serves 16 iterations of the same wave from the L1.  A wave per node with the lanes over the codes -- coalesced
// for the node's own row -- reads every line of the transposed part for ONE entry: 3.0 instead of 1.3 ms at 64^3, round 3.)
// den[i] = (|N| |x|)_i + |rhs_i| from the half stencil: sum over code < centre of |N(i, jl) x_jl| (jl = i + off(code), the entry
// of row i) and |N(ju, i) x_ju| (ju = i - off(code), the entry of row ju at the same code), code ascending.
// A thread per row that walked its 171 codes touched a new 64-byte sector of another row at every step (N(ju, i) of
// consecutive codes lie in consecutive ROWS): 45 M sector fetches, 1.2 ms at 64^3 -- and beside another kernel it starved
// that one.  Here a workgroup owns 256 consecutive rows and stages, for the seven codes of one (o_1, .., o_{D-1}) at a time
// (they differ in the offset along dimension 0 only), the 7-double pieces of its own rows and of the 256 + 6 rows
// i0 - base - 3 .. that hold the transposed entries, and the two windows of x: every fetched sector is used whole.  Same terms
// in the same order per row.
template <int D>
__global__ void __launch_bounds__(256)
backward_denominators_kernel(Grid g, const double *__restrict__ nst, const double *__restrict__ xvec,
                             const double *__restrict__ rhs, double *__restrict__ den)
{
    constexpr int TR = 256, HALO = TR + 6;
    __shared__ double HA[TR * 7], HB[HALO * 7], xa[HALO], xb[HALO];
    const int tid = threadIdx.x;
    const long long i0 = (long long)blockIdx.x * TR;
    const long long i = i0 + tid;
    const bool live = i < g.ncol;
    const int centre = g.hstencil - 1;
    int in[D];
#pragma unroll
    for (int d = 0; d < D; ++d) in[d] = live ? (int)((i / g.colstride[d]) % g.nodes[d]) : 0;
    double s = live ? fabs(nst[i * g.hstencil + centre] * xvec[i]) + fabs(rhs[i]) : 0.0;
    const int ngroups = centre / 7 + 1;
    for (int gi = 0; gi < ngroups; ++gi) {
        const int c0 = 7 * gi, nk = gi + 1 < ngroups ? 7 : centre - c0;      // (the last group: the codes below the centre)
        int oh[D], t = gi;
        long long base = 0;
        oh[0] = 0;
#pragma unroll
        for (int d = 1; d < D; ++d) {
            oh[d] = t % 7 - 3;
            t /= 7;
            base += (long long)oh[d] * g.colstride[d];
        }
        __syncthreads();
        for (int e = tid; e < TR * 7; e += 256) {
            const long long r = i0 + e / 7;
            const int k = e % 7;
            HA[e] = (r < g.ncol && k < nk) ? nst[r * g.hstencil + c0 + k] : 0.0;
        }
        for (int e = tid; e < HALO * 7; e += 256) {
            const long long r = i0 - base - 3 + e / 7;
            const int k = e % 7;
            HB[e] = (r >= 0 && r < g.ncol && k < nk) ? nst[r * g.hstencil + c0 + k] : 0.0;
        }
        for (int e = tid; e < HALO; e += 256) {
            const long long ra = i0 + base - 3 + e, rb = i0 - base - 3 + e;
            xa[e] = (ra >= 0 && ra < g.ncol) ? xvec[ra] : 0.0;
            xb[e] = (rb >= 0 && rb < g.ncol) ? xvec[rb] : 0.0;
        }
        __syncthreads();
        if (!live) continue;
        bool okh_l = true, okh_u = true;                // the higher dimensions' share of the two in-grid tests
#pragma unroll
        for (int d = 1; d < D; ++d) {
            okh_l = okh_l && in[d] + oh[d] >= 0 && in[d] + oh[d] <= g.nodes[d] - 1;
            okh_u = okh_u && in[d] - oh[d] >= 0 && in[d] - oh[d] <= g.nodes[d] - 1;
        }
        for (int k = 0; k < nk; ++k) {
            const int ox = k - 3;
            const bool okl = okh_l && in[0] + ox >= 0 && in[0] + ox <= g.nodes[0] - 1;
            const bool oku = okh_u && in[0] - ox >= 0 && in[0] - ox <= g.nodes[0] - 1;
            if (okl) s += fabs(HA[tid * 7 + k] * xa[tid + ox + 3]);                     // N(i, jl), jl = i + base + ox < i
            if (oku) s += fabs(HB[(tid - ox + 3) * 7 + k] * xb[tid - ox + 3]);          // N(ju, i), ju = i - base - ox > i
        }
    }
    if (live) den[i] = s;
}

__global__ void __launch_bounds__(256)
backward_error_kernel(int ncol, const double *__restrict__ den, const double *__restrict__ rho, unsigned long long *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double om = 0.0;
    if (i < ncol) {
        const double s = den[i];
        om = s > 0.0 ? fabs(rho[i]) / s : fabs(rho[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) om = fmax(om, __shfl_xor(om, o, 64));
    if ((threadIdx.x & 63) == 0 && om > 0.0) atomicMax(out, (unsigned long long)__double_as_longlong(om));
}

// ---------------------------------------------------------------------------
template <int D>
__global__ void __launch_bounds__(256)
expand_kernel(Grid g, const double *__restrict__ nst, double *__restrict__ ab, long long lda, DistMap dm)
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
        const int J = j / NBLK;                 // block column of the entry: stored here only if this rank owns it
        if (!dm_owned(dm, J)) continue;
        ab[dm_shift(dm, J) + (long long)i + (long long)j * lda] = nst[t];
    }
}

__global__ void pad_diag_kernel(double *ab, long long lda, int n, int npad, DistMap dm)
{
    const int i = n + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npad && dm_owned(dm, i / NBLK)) ab[dm_shift(dm, i / NBLK) + (long long)i + (long long)i * lda] = 1.0;
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

long long gram_scratch_doubles(const Grid &g)
{
    return (long long)g.ncell * (gram_tri(g.nb) + 2LL * g.nb);
}

// cells per bin of the stable partition's first level (1: the bins ARE the cells); 0: the grid has too many cells for it
static int sp_cells_per_bin(const Grid &g)
{
    const long long cpt = ((long long)g.ncell + SP_NB - 2) / (SP_NB - 1);
    return cpt <= SP_NB - 1 ? (int)(cpt < 1 ? 1 : cpt) : 0;
}

long long bin_record_doubles(const Grid &g, long long max_ndata)
{
    const int cpt = sp_cells_per_bin(g);
    return cpt > 1 ? max_ndata * (long long)sp_rec(g.ndim) : 0;
}

hipError_t launch_bin_points(const Grid &g, long long m, const double *x, int ldx, const double *y,
                             const double *w, const SortScratch &s, double *scal, hipStream_t st)
{
    const bool old_form = splpak::opt_get("SPLPAK_BIN_ATOMIC") != nullptr;      // A/B switch: rounds 1-4 (global atomics + in-cell re-sort)
    const int cpt = sp_cells_per_bin(g);
    if (!old_form && cpt > 0 && s.cntm && s.binbase && s.sppart && (cpt == 1 || s.rec)) {
        if (m <= 0) return hipMemsetAsync(s.offset, 0, sizeof(int) * (size_t)(g.ncell + 2), st);
        const int nblk = (int)((m + SP_Q - 1) / SP_Q), nchunk = (nblk + SP_ROWS - 1) / SP_ROWS;
        double *rec = cpt > 1 ? s.rec : nullptr;
        const int ntile = (g.ncell + cpt - 1) / cpt;                   // bins in use (one level: the cells)
        int *tot = s.binbase + SP_NB + 8;
        DISPATCH_D(g.ndim, hipLaunchKernelGGL(sp_count_kernel<D>, dim3((unsigned)nblk), dim3(SP_NT), 0, st, g, m, x, ldx, w, cpt, s.key, s.cntm));
        hipLaunchKernelGGL(sp_colsum_kernel, dim3(SP_NB / 256, (unsigned)nchunk), dim3(256), 0, st, nblk, (const int *)s.cntm, s.sppart);
        hipLaunchKernelGGL(sp_chunkscan_kernel, dim3(SP_NB / 256), dim3(256), 0, st, nchunk, s.sppart, tot);
        hipLaunchKernelGGL(sp_binscan_kernel, dim3(1), dim3(1024), 0, st, (const int *)tot, s.binbase, scal);
        hipLaunchKernelGGL(sp_blockbase_kernel, dim3(SP_NB / 256, (unsigned)nchunk), dim3(256), 0, st, nblk, s.cntm, (const int *)s.sppart, (const int *)s.binbase);
        DISPATCH_D(g.ndim, hipLaunchKernelGGL(sp_scatter_kernel<D>, dim3((unsigned)nblk), dim3(SP_NT), 0, st, g, m, x, ldx, y, w, cpt, ntile, (const int *)s.key,
                                              (const int *)s.cntm, rec, s.xs, s.ys, s.ws, s.idx, s.cap));
        if (cpt == 1)      // the bins are the cells: their bases are the offsets (bins ncell .. SP_NB - 2 are empty)
            return hipMemcpyAsync(s.offset, s.binbase, sizeof(int) * (size_t)(g.ncell + 1), hipMemcpyDeviceToDevice, st);
        if (cpt <= 256) {
            DISPATCH_D(g.ndim, hipLaunchKernelGGL((sp_bin2_kernel<D, 256>), dim3((unsigned)ntile), dim3(SP_NT), 0, st, g, cpt, ntile, (const int *)s.binbase,
                                                  (const double *)rec, s.offset, s.xs, s.ys, s.ws, s.idx, s.cap));
        } else {
            DISPATCH_D(g.ndim, hipLaunchKernelGGL((sp_bin2_kernel<D, SP_NB>), dim3((unsigned)ntile), dim3(SP_NT), 0, st, g, cpt, ntile, (const int *)s.binbase,
                                                  (const double *)rec, s.offset, s.xs, s.ys, s.ws, s.idx, s.cap));
        }
        return hipGetLastError();
    }
    hipError_t e = hipMemsetAsync(s.count, 0, sizeof(int) * (size_t)(g.ncell + 2), st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(s.cursor, 0, sizeof(int) * (size_t)(g.ncell + 1), st);
    if (e != hipSuccess) return e;
    if (m > 0) {
        dim3 gr(grid_for(m, 256)), bl(256);
        DISPATCH_D(g.ndim, hipLaunchKernelGGL(keys_kernel<D>, gr, bl, 0, st, g, m, x, ldx, w, s.key, s.count, scal));
    }
    hipLaunchKernelGGL(scan_partials_kernel, dim3(SCAN_SEG), dim3(256), 0, st, (const int *)s.count, s.scanpart, g.ncell + 1);
    hipLaunchKernelGGL(scan_segments_kernel, dim3(SCAN_SEG), dim3(256), 0, st, (const int *)s.count, (const int *)s.scanpart, s.offset, g.ncell + 1);
    if (m > 0) {
        dim3 gr(grid_for(m, 256)), bl(256);
        DISPATCH_D(g.ndim, hipLaunchKernelGGL(scatter_kernel<D>, gr, bl, 0, st, g, m, x, ldx, y, w,
                                              s.key, s.offset, s.cursor, s.xs, s.ys, s.ws, s.idx, s.cap));
        DISPATCH_D(g.ndim, hipLaunchKernelGGL(cell_order_kernel<D>, dim3((unsigned)g.ncell), dim3(256), 0, st, g,
                                              (const int *)s.offset, s.xs, s.ys, s.ws, s.idx, s.cap, x, ldx, y, w, s.key));
    }
    return hipGetLastError();
}

long long gram_scratch_min_doubles(const Grid &g)
{   // one hyper-row of cells along the slowest dimension
    return (long long)g.cellstride[g.ndim - 1] * (gram_tri(g.nb) + 2LL * g.nb);
}

// -> true when the blocks of empty cells were written as zeros (the gather then reads every block unconditionally)
template <int D>
static bool gram_cells(const Grid &g, const SortScratch &s, double *blk, double *rblk, double *hblk, double *hist, int cell0,
                       int ncells, hipStream_t st)
{
    const bool old_form = splpak::opt_get("SPLPAK_GRAM_VALU") != nullptr;       // A/B switch: the workgroup-per-cell form
    if constexpr (D == 2 || D == 3) {
        if (!old_form) {
            // cells per wave: GW_RUN where that still leaves four rounds of waves for the chip (two per SIMD), fewer on a small grid
            int run = (int)(ncells / (4LL * 2048));
            run = run < 1 ? 1 : run > GW_RUN ? GW_RUN : run;
            hipLaunchKernelGGL(gram_wave_kernel<D>, dim3((unsigned)((ncells + 4 * run - 1) / (4 * run))), dim3(256), 0, st, g, (const int *)s.offset,
                               (const double *)s.xs, (const double *)s.ys, (const double *)s.ws, s.cap, blk, rblk, hblk, hist,
                               cell0, ncells, run);
            return true;
        }
    }
    if constexpr (D == 4) {
        if (!old_form) {
            hipLaunchKernelGGL(gram_mfma4_kernel<D>, dim3((unsigned)ncells), dim3(640), 0, st, g, (const int *)s.offset,
                               (const double *)s.xs, (const double *)s.ys, (const double *)s.ws, s.cap, blk, rblk, hblk, hist, cell0);
            return false;                      // (empty cells write nothing: a 263 KB zero block each would be too much)
        }
    }
    using C = GramCfg<D>;
    dim3 gr((unsigned)ncells, (unsigned)(C::NB / (C::NTX * C::TC)));
    hipLaunchKernelGGL(gram_block_kernel<D>, gr, dim3(C::NT), 0, st, g, (const int *)s.offset, (const double *)s.xs,
                       (const double *)s.ys, (const double *)s.ws, s.cap, blk, rblk, hblk, hist, cell0);
    return false;
}

hipError_t launch_gram(const Grid &g, const SortScratch &s, double *scratch, long long scratch_doubles, bool smooth,
                       double *nst, double *rhs, double *hist, double *scalH, hipStream_t st)
{
    // The per-cell blocks are produced and gathered slab by slab: as many whole hyper-rows of cells (along
    // the slowest dimension) as the scratch holds -- all of them at 64^3 (3.9 GB), two or three slabs of
    // the 189 GB that the 29^4 cells of the 4-D 32^4 grid would need at once.
    const long long per_cell = gram_tri(g.nb) + 2LL * g.nb;
    const int hrow = g.cellstride[g.ndim - 1];              // cells per hyper-row
    const int nhrow = g.cells[g.ndim - 1];
    long long fit_rows = scratch_doubles / (per_cell * hrow);
    if (fit_rows < 1) return hipErrorInvalidValue;
    if (fit_rows > nhrow) fit_rows = nhrow;
    const int nstride = g.colstride[g.ndim - 1];            // nodes per hyper-row of nodes
    for (int h0 = 0; h0 < nhrow; h0 += (int)fit_rows) {
        const int h1 = (h0 + fit_rows < nhrow) ? h0 + (int)fit_rows : nhrow;
        const int cell0 = h0 * hrow, cell1 = h1 * hrow, ncells = cell1 - cell0;
        const int node0 = h0 * nstride, node1 = (h1 + 3) * nstride;     // window starts h0..h1-1 touch nodes h0..h1+2
        double *blk = scratch;
        double *rblk = blk + (long long)ncells * gram_tri(g.nb);
        double *hblk = smooth ? rblk + (long long)ncells * g.nb : nullptr;
        DISPATCH_D(g.ndim, {
            const bool zeroed = gram_cells<D>(g, s, blk, rblk, hblk, hist, cell0, ncells, st) && !splpak::opt_get("SPLPAK_GATHER_LOOKUP");
            const dim3 gg((unsigned)((node1 - node0 + 3) / 4));
            if (zeroed)
                hipLaunchKernelGGL((stencil_gather_kernel<D, true>), gg, dim3(256), 0, st, g,
                                   (const int *)s.offset, (const double *)blk, (const double *)rblk, (const double *)hblk,
                                   nst, rhs, hist, cell0, cell1, node0, node1);
            else
                hipLaunchKernelGGL((stencil_gather_kernel<D, false>), gg, dim3(256), 0, st, g,
                                   (const int *)s.offset, (const double *)blk, (const double *)rblk, (const double *)hblk,
                                   nst, rhs, hist, cell0, cell1, node0, node1);
        });
    }
    if (smooth) hipLaunchKernelGGL(hist_total_kernel, dim3(1), dim3(1024), 0, st, (const double *)hist, g.ncol, scalH);
    return hipGetLastError();
}

hipError_t launch_sparse_mark(const Grid &g, const double *hist, const double *scal, double xtrap, double *dcw,
                              unsigned char *spf, hipStream_t st)
{
    DISPATCH_D(g.ndim, hipLaunchKernelGGL(sparse_mark_kernel<D>, dim3((unsigned)((g.ncol + 255) / 256)), dim3(256), 0, st,
                                          g, hist, scal, xtrap, dcw, spf));
    return hipGetLastError();
}

hipError_t launch_constraint_rows(const Grid &g, const double *dcw, const unsigned char *spf, const double *ctab, double *nst,
                                  double *scal_out, hipStream_t st)
{
    dim3 gr((unsigned)((g.ncol + 3) / 4)), bl(256);
    DISPATCH_D(g.ndim, hipLaunchKernelGGL(constraint_rows_kernel<D>, gr, bl, 0, st, g, dcw, spf, ctab, nst, scal_out));
    hipLaunchKernelGGL(count_sparse_kernel, dim3(1), dim3(1024), 0, st, spf, g.ncol, g.ndim * (g.ndim + 1) / 2, scal_out);
    return hipGetLastError();
}

hipError_t launch_hist_total(const Grid &g, const double *hist, double *scal, hipStream_t st)
{
    hipLaunchKernelGGL(hist_total_kernel, dim3(1), dim3(1024), 0, st, hist, g.ncol, scal);
    return hipGetLastError();
}

hipError_t launch_count_sparse(const Grid &g, const unsigned char *spf, double *scal_out, hipStream_t st)
{
    hipLaunchKernelGGL(count_sparse_kernel, dim3(1), dim3(1024), 0, st, spf, g.ncol, g.ndim * (g.ndim + 1) / 2, scal_out);
    return hipGetLastError();
}

long long constraint_table_doubles(const Grid &g)
{
    long long n = 0;
    for (int d = 0; d < g.ndim; ++d) n += 9LL * g.nodes[d];
    return n;
}

hipError_t launch_constraint_table(const Grid &g, double *ctab, hipStream_t st)
{
    const long long n = constraint_table_doubles(g);
    hipLaunchKernelGGL(constraint_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g, ctab);
    return hipGetLastError();
}

// out[0] = sum of v[0 .. n) in a fixed order: thread t sums the entries t, t + 1024, ..; then a tree over the threads
__global__ void __launch_bounds__(1024)
sum_fixed_kernel(const double *__restrict__ v, long long n, double *__restrict__ out)
{
    __shared__ double red[1024];
    red[threadIdx.x] = strided_sum8(v, n, (int)threadIdx.x, 1024);
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

template <int D>
static void residual_cells(const Grid &g, const SortScratch &s, const double *xvec, double *rcell, double *e2c, hipStream_t st)
{
    if constexpr (D <= 3) {
        hipLaunchKernelGGL(residual_wave_kernel<D>, dim3((unsigned)((g.ncell + 3) / 4)), dim3(256), 0, st, g,
                           (const int *)s.offset, (const double *)s.xs, (const double *)s.ys, (const double *)s.ws,
                           s.cap, xvec, rcell, e2c);
    } else {
        const bool old_form = splpak::opt_get("SPLPAK_RESIDUAL_STAGED") != nullptr;      // A/B switch
        if (old_form)
            hipLaunchKernelGGL(residual_block_kernel<D>, dim3((unsigned)g.ncell), dim3(ResCfg<D>::NT), 0, st, g,
                               (const int *)s.offset, (const double *)s.xs, (const double *)s.ys, (const double *)s.ws,
                               s.cap, xvec, rcell, e2c);
        else
            hipLaunchKernelGGL(residual_cell4_kernel<D>, dim3((unsigned)g.ncell), dim3(256), 0, st, g,
                               (const int *)s.offset, (const double *)s.xs, (const double *)s.ys, (const double *)s.ws,
                               s.cap, xvec, rcell, e2c);
    }
}

hipError_t launch_residual(const Grid &g, const SortScratch &s, const double *xvec, double *rcell,
                           const double *dcw, const unsigned char *spf, const double *ctab, bool constraints,
                           double *tbuf, double *rho, double *ssq, double *e2buf, hipStream_t st)
{
    dim3 gn((unsigned)((g.ncol + 3) / 4)), bl(256);
    // sum of squared row residuals (ssq != NULL): every cell and every data-sparse node leaves its share in e2buf
    // ([ncell] + [ncol]), one workgroup adds them in a fixed order -- no floating-point atomics, reproducible bits
    double *e2c = (ssq && e2buf) ? e2buf : nullptr, *e2n = e2c ? e2buf + g.ncell : nullptr;
    if (e2c) {
        hipError_t e = hipMemsetAsync(e2buf, 0, sizeof(double) * ((size_t)g.ncell + (size_t)g.ncol), st);
        if (e != hipSuccess) return e;
    }
    DISPATCH_D(g.ndim, {
        residual_cells<D>(g, s, xvec, rcell, e2c, st);
        if (constraints)
            hipLaunchKernelGGL(constraint_dots_kernel<D>, gn, bl, 0, st, g, dcw, spf, ctab, xvec, tbuf, e2n);
        hipLaunchKernelGGL(rho_gather_kernel<D>, dim3((unsigned)((g.ncol + 255) / 256)), bl, 0, st, g,
                           (const int *)s.offset, (const double *)rcell, dcw, spf, ctab,
                           constraints ? (const double *)tbuf : (const double *)nullptr, rho);
    });
    if (e2c)
        hipLaunchKernelGGL(sum_fixed_kernel, dim3(1), dim3(1024), 0, st, (const double *)e2buf, (long long)g.ncell + g.ncol, ssq);
    return hipGetLastError();
}

hipError_t launch_sum_fixed(const double *v, long long n, double *out, hipStream_t st)
{
    hipLaunchKernelGGL(sum_fixed_kernel, dim3(1), dim3(1024), 0, st, v, n, out);
    return hipGetLastError();
}

hipError_t launch_backward_denominators(const Grid &g, const double *nst, const double *xvec, const double *rhs, double *den, hipStream_t st)
{
    dim3 gr((unsigned)((g.ncol + 255) / 256)), bl(256);           // (256 rows per workgroup: the kernel's tile)
    DISPATCH_D(g.ndim, hipLaunchKernelGGL(backward_denominators_kernel<D>, gr, bl, 0, st, g, nst, xvec, rhs, den));
    return hipGetLastError();
}

hipError_t launch_backward_error(const Grid &g, const double *den, const double *rho, double *out, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(out, 0, sizeof(double), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(backward_error_kernel, dim3((unsigned)((g.ncol + 255) / 256)), dim3(256), 0, st, g.ncol, den, rho,
                       reinterpret_cast<unsigned long long *>(out));
    return hipGetLastError();
}

hipError_t launch_expand(const Grid &g, const double *nst, const Band &b, const DistMap &dm, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(b.ab, 0, b.bytes, st);
    if (e != hipSuccess) return e;
    const long long total = (long long)g.ncol * g.hstencil;
    dim3 gr(grid_for(total, 256, 256LL * 64)), bl(256);
    DISPATCH_D(g.ndim, hipLaunchKernelGGL(expand_kernel<D>, gr, bl, 0, st, g, nst, b.ab, b.lda, dm));
    if (b.npad > b.n)
        hipLaunchKernelGGL(pad_diag_kernel, dim3((b.npad - b.n + 255) / 256), dim3(256), 0, st, b.ab,
                           b.lda, b.n, b.npad, dm);
    return hipGetLastError();
}

}  // namespace splpak
