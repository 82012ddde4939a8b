// Batched spline / partial-derivative evaluation: one thread per query.
//
// Replaces a loop of scalar splde/splfe calls (src/splpak.F90:1089-1240,
// :1258-1275).  Per query the reference calls bascmp 4^ndim times and recomputes
// every 1-D factor 4^(ndim-1) times; here each thread builds the separable
// 4 x ndim table once (basis.hpp) and walks the 4^ndim window with the same
// summation order as the reference's odometer (dimension 1 fastest, :1228-1232),
// so results differ from the reference only by FMA contraction.
//
// HBM roofline: 8*(ndim+1) algorithmic bytes per query (ndim coordinates in, one
// value out).  Two paths with identical arithmetic: the direct kernel gathers the
// coefficients from global memory (L2-bound for scattered queries), the binned
// path sorts large batches by grid region and gathers from LDS (see below).
#include "basis.hpp"
#include "kernels.hpp"
#include <type_traits>
#include <cstdlib>

namespace splpak {

struct NDeriv { int v[MAXD]; };

// Sum of the 4^D window products, factorised: the innermost dimension is contracted with its four
// factors first, then the partial sums with the factors of the next dimension, and so on -- 64 + 16 + 4
// fused multiply-adds in 3-D instead of the 64 * 3 multiplications of the plain triple product (the
// reference forms every product basm = prod_d bas1_d and adds coef*basm, :1215-1236; the two orders
// differ by rounding only).  load4(k1, k2, k3, c) delivers the 4 coefficients of the window row
// (k0 = 0..3).  Shared by the direct and the binned kernels so that both produce bit-identical values.
template <int D, typename L4>
__device__ inline double window_sum(const double (&b)[D][4], L4 &&load4)
{
    auto row = [&](int k1, int k2, int k3) {
        double c[4];
        load4(k1, k2, k3, c);
        double t = c[0] * b[0][0];
        t = fma(c[1], b[0][1], t);
        t = fma(c[2], b[0][2], t);
        t = fma(c[3], b[0][3], t);
        return t;
    };
    if constexpr (D == 1) {
        return row(0, 0, 0);
    } else if constexpr (D == 2) {
        double sum = 0.0;
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) sum = fma(row(k1, 0, 0), b[1][k1], sum);
        return sum;
    } else if constexpr (D == 3) {
        double sum = 0.0;
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) {
            double r = 0.0;
#pragma unroll
            for (int k1 = 0; k1 < 4; ++k1) r = fma(row(k1, k2, 0), b[1][k1], r);
            sum = fma(r, b[2][k2], sum);
        }
        return sum;
    } else {
        double sum = 0.0;
        for (int k3 = 0; k3 < 4; ++k3) {
            double q = 0.0;
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
                double r = 0.0;
#pragma unroll
                for (int k1 = 0; k1 < 4; ++k1) r = fma(row(k1, k2, k3), b[1][k1], r);
                q = fma(r, b[2][k2], q);
            }
            sum = fma(q, b[3][k3], sum);
        }
        return sum;
    }
}

// the 4-entry factor table of dimension d: the branch-free value form when no derivative is asked for
template <bool VAL>
__device__ inline int eval_table(const Grid &g, int d, double x, int nder, double (&b)[4])
{
    if constexpr (VAL) {
        int lo, hi, it;
        bool interior;
        double u, t;
        const int ws = window_start_frac(g, d, x, lo, hi, interior, u, t, it);
        // every lane computes the closed form of an interior window (16 operations); a wave that holds queries whose window
        // in this dimension is NOT interior also computes, for those lanes, the form of a window next to an end of the grid
        // (the first / last three cells: end functions put into the closed form, window_values_near) and, if it holds
        // queries OUTSIDE the grid (or the grid has fewer than 8 nodes), the general form for these.  Which form a query
        // gets depends on the query alone.
        window_values_interior(u, b);
        if (__builtin_amdgcn_ballot_w64(!interior) != 0) {
            const int nod = g.nodes[d];
            const bool near = !interior && nod >= 8 && t >= 0.0 && it <= nod - 2;
            double bn[4];
            window_values_near(t, it, nod, b, bn);
            if (__builtin_amdgcn_ballot_w64(!interior && !near) != 0) {
                double bg[4];
                window_values<false>(g, d, x, ws, lo, hi, bg);
#pragma unroll
                for (int k = 0; k < 4; ++k) bn[k] = near ? bn[k] : bg[k];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) b[k] = interior ? b[k] : bn[k];
        }
        return ws;
    } else {
        return window_table(g, d, x, nder, b);
    }
}

// ---- direct path: one thread per query, coefficient gathers from global memory (L2) -------------
template <int D, typename T, bool VAL>
__global__ void __launch_bounds__(256)
eval_kernel(Grid g, long long nq, const T *__restrict__ xq, int ldxq, NDeriv nd,
            const T *__restrict__ coef, T *__restrict__ out)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += stride) {
        double b[D][4];
        int base = 0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const double x = (double)xq[i * ldxq + d];
            const int ws = eval_table<VAL>(g, d, x, nd.v[d], b[d]);
            base += ws * g.colstride[d];
        }
        // The 4 coefficients of a window row are contiguous: two 16-byte loads instead of four
        // 8-byte ones (the kernel is bound by gather instructions / L2 lines, not bytes).
        const int s1 = D > 1 ? g.colstride[1] : 0, s2 = D > 2 ? g.colstride[2] : 0, s3 = D > 3 ? g.colstride[3] : 0;
        const double sum = window_sum<D>(b, [&](int k1, int k2, int k3, double (&c)[4]) {
            const long long idx = base + k1 * s1 + k2 * s2 + k3 * s3;
            if constexpr (sizeof(T) == 8) {
                typedef double d2v __attribute__((ext_vector_type(2), aligned(8)));
                const d2v lo = *reinterpret_cast<const d2v *>(coef + idx);
                const d2v hi = *reinterpret_cast<const d2v *>(coef + idx + 2);
                c[0] = lo[0]; c[1] = lo[1]; c[2] = hi[0]; c[3] = hi[1];
            } else {
                typedef float f4v __attribute__((ext_vector_type(4), aligned(4)));
                const f4v v = *reinterpret_cast<const f4v *>(coef + idx);
                c[0] = v[0]; c[1] = v[1]; c[2] = v[2]; c[3] = v[3];
            }
        });
        out[i] = (T)sum;
    }
}

// ---- binned path ---------------------------------------------------------------------------------
// Random queries make every window row a separate 128-byte L2 line (~19 lines = 2.4 KB of L2
// traffic per 3-D query, for 512 useful bytes): the direct kernel sits at the L2 gather ceiling.
// The binned path sorts a chunk of queries by REGION -- a box of window starts whose
// coefficients (box + 3 nodes per dimension, 4096 doubles = 32 KB) fit in LDS -- and evaluates
// every region's queries from an LDS copy of its coefficients: the gathers become LDS reads, the
// global traffic per query is its coordinates, a permutation index and the result.
//   pass A  bin_count_kernel    region histogram of the chunk + per-workgroup region counts
//           bin_scan_kernel     offsets and workgroups per region
//           bin_wgbase_kernel   where every pass-B workgroup's runs start (prefix over workgroups)
//   pass B  bin_scatter_kernel  coordinates + original index copied into region order
//   pass C  eval_binned_kernel  one workgroup per (region, 2048 queries)
// The arithmetic per query is window_table + window_sum exactly as in the direct kernel, so both
// paths return identical bits; only the order in which queries are processed differs.
template <int D> struct TileShape;
template <> struct TileShape<2> { static constexpr int T[4] = {64, 64, 1, 1}; };
template <> struct TileShape<3> { static constexpr int T[4] = {16, 16, 16, 1}; };
// 4-D: 8 x 8 x 8 x 16 coefficients (64 KB) serve 5 x 5 x 5 x 13 window starts -- 648 regions at 32^4 instead of the 1 296 of an
// 8^4 tile (round 3): half the bins in the sort passes, 2.6 x the window starts per tile fill
template <> struct TileShape<4> { static constexpr int T[4] = {8, 8, 8, 16}; };
// LDS strides of the tile dimensions.  4-D: padded (8 -> 67 -> 539 instead of 64 -> 512) so that the tile offset of a window
// start, taken mod 32 doubles = its LDS bank class for ds_read_b64, is uniform over the 5 x 5 x 5 x 13 starts of a region (50-52
// per class; the dense strides give 20 classes, five of them double: SQ_LDS_BANK_CONFLICT was 80 % of SQ_LDS_IDX_ACTIVE and
// the LDS pipe 90 % of the evaluation pass, round 3).  The evaluation pass then deals its queries to the lanes BY CLASS
// (eval_binned_kernel), which makes every window read conflict free.
template <int D> struct TileStride { static constexpr int S[4] = {1, TileShape<D>::T[0], TileShape<D>::T[0] * TileShape<D>::T[1],
                                                                  TileShape<D>::T[0] * TileShape<D>::T[1] * TileShape<D>::T[2]}; };
template <> struct TileStride<4> { static constexpr int S[4] = {1, 8, 67, 539}; };
template <> struct TileStride<3> { static constexpr int S[4] = {1, 17, 274, 274 * 16}; };     // 13^3 starts: 67-70 per class (dense: 26 classes)
template <int D> constexpr int tile_elems() { return TileStride<D>::S[D - 1] * TileShape<D>::T[D - 1]; }
template <int D> constexpr int tile_cells() { return TileShape<D>::T[0] * TileShape<D>::T[1] * TileShape<D>::T[2] * TileShape<D>::T[3]; }
constexpr int BIN_MAX = 2048;          // regions per grid handled by the LDS histograms
constexpr int EVAL_QPW = 2048;         // queries per workgroup in pass C

struct Regions { int nreg[MAXD]; int nbins; };

// A sorted query is ONE record of D + 1 doubles: its coordinates and, in the low half of the last word, its position in the
// caller's batch (round 3: coordinate planes + a separate permutation made pass B issue D + 1 scattered 8-byte stores per
// query -- 99 B of HBM writes for the 36-byte payload of a 4-D query, runs of 1.6 queries per workgroup and region; a record
// is one 32- / 40-byte store and one load in pass C).
typedef double rec2u_t __attribute__((ext_vector_type(2), aligned(8)));     // 40-byte records: 8-byte aligned pieces
typedef double rec2a_t __attribute__((ext_vector_type(2), aligned(16)));    // 32-byte records: aligned 16-byte pieces
template <int D>
__device__ inline void store_record(double *__restrict__ dst, const double (&x)[D], int idx)
{
    using rec2_t = typename std::conditional<(D + 1) % 2 == 0, rec2a_t, rec2u_t>::type;
    double v[D + 1];
#pragma unroll
    for (int d = 0; d < D; ++d) v[d] = x[d];
    v[D] = __longlong_as_double((long long)idx);
    constexpr int N = D + 1;
#pragma unroll
    for (int k = 0; k + 1 < N; k += 2) {
        rec2_t t;
        t[0] = v[k]; t[1] = v[k + 1];
        *reinterpret_cast<rec2_t *>(dst + k) = t;
    }
    if constexpr (N % 2 == 1) dst[N - 1] = v[N - 1];
}
template <int D>
__device__ inline int load_record(const double *__restrict__ src, double (&x)[D])
{
    constexpr int N = D + 1;
    using rec2_t = typename std::conditional<(D + 1) % 2 == 0, rec2a_t, rec2u_t>::type;
    double v[N];
#pragma unroll
    for (int k = 0; k + 1 < N; k += 2) {
        const rec2_t t = *reinterpret_cast<const rec2_t *>(src + k);
        v[k] = t[0]; v[k + 1] = t[1];
    }
    if constexpr (N % 2 == 1) v[N - 1] = src[N - 1];
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = v[d];
    return (int)__double_as_longlong(v[D]);
}

template <int D>
__device__ inline int region_of(const Grid &g, const Regions &rg, const double *__restrict__ x)
{
    int r = 0, m = 1;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        int lo, hi;
        const int ws = window_start(g, d, x[d], lo, hi);
        r += (ws / (TileShape<D>::T[d] - 3)) * m;
        m *= rg.nreg[d];
    }
    return r;
}

template <int D> struct ScatterShape { static constexpr int QPT = D == 4 ? 4 : 8; };   // queries per thread in passes A and B

// Pass A.  Workgroup w counts the SAME 256*QPT queries that workgroup w of pass B will place, and
// leaves its per-region counts in row w of `cnt`; the column-wise prefix of that matrix
// (bin_wgbase_kernel) then tells every pass-B workgroup where each of its runs starts.  No workgroup
// ever waits on a global atomic (round 2: 2 050 workgroups taking turns on 125 cursor words cost 42 of
// the 80 us of pass B), and the sorted order is a function of the input alone.
template <int D, typename T>
__global__ void __launch_bounds__(256)
bin_count_kernel(Grid g, Regions rg, int n, const T *__restrict__ xq, int ldxq, int *__restrict__ cnt, int ldw)
{
    constexpr int QPT = ScatterShape<D>::QPT;
    __shared__ int lh[BIN_MAX];
    for (int b = threadIdx.x; b < rg.nbins; b += 256) lh[b] = 0;
    __syncthreads();
    const int base = blockIdx.x * (256 * QPT);
#pragma unroll
    for (int j = 0; j < QPT; ++j) {
        const int i = base + j * 256 + threadIdx.x;
        if (i < n) {
            double x[D];
#pragma unroll
            for (int d = 0; d < D; ++d) x[d] = (double)xq[(long long)i * ldxq + d];
            atomicAdd(&lh[region_of<D>(g, rg, x)], 1);
        }
    }
    __syncthreads();
    (void)ldw;
    for (int b = threadIdx.x; b < rg.nbins; b += 256) cnt[(long long)blockIdx.x * rg.nbins + b] = lh[b];   // row = this workgroup, consecutive regions: coalesced
}

// The count matrix is cnt[workgroup][region] (round 3: the transposed layout made every workgroup of pass A write, and of
// pass B read, one 4-byte word per 32-byte sector -- 680 MB of traffic each for the 85 MB matrix of a 4-D batch: 1 296
// regions x 16 384 workgroups; profiles/r03_eval_pmc.json).  Column sums and prefixes over a row-major matrix: a thread
// owns a region (consecutive threads = consecutive regions = coalesced rows), workgroups own chunks of BIN_ROWS rows.
constexpr int BIN_ROWS = 128;
// part[c][b] = sum of cnt[w][b] over the rows w of chunk c; grid (ceil(nbins / 256), nchunk)
__global__ void __launch_bounds__(256)
bin_colsum_kernel(int nwg, int nbins, const int *__restrict__ cnt, int *__restrict__ part)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nbins) return;
    const int w0 = blockIdx.y * BIN_ROWS, w1 = (w0 + BIN_ROWS < nwg) ? w0 + BIN_ROWS : nwg;
    int sum = 0;
    for (int w = w0; w < w1; ++w) sum += cnt[(long long)w * nbins + b];
    part[(long long)blockIdx.y * nbins + b] = sum;
}
// hist[b] = sum_c part[c][b];  part[c][b] <- sum_{c' < c} part[c'][b]   (thread = region)
__global__ void __launch_bounds__(256)
bin_total_kernel(int nchunk, int nbins, int *__restrict__ part, int *__restrict__ hist)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nbins) return;
    int run = 0;
    for (int c = 0; c < nchunk; ++c) {
        const int v = part[(long long)c * nbins + b];
        part[(long long)c * nbins + b] = run;
        run += v;
    }
    hist[b] = run;
}
// cnt[w][b] <- off[b] + (queries of region b in the workgroups before w): first sorted position of workgroup w's run
__global__ void __launch_bounds__(256)
bin_wgbase_kernel(int nwg, int nbins, const int *__restrict__ off, const int *__restrict__ part, int *__restrict__ cnt)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nbins) return;
    const int w0 = blockIdx.y * BIN_ROWS, w1 = (w0 + BIN_ROWS < nwg) ? w0 + BIN_ROWS : nwg;
    int run = off[b] + part[(long long)blockIdx.y * nbins + b];
    for (int w = w0; w < w1; ++w) {
        const int c = cnt[(long long)w * nbins + b];
        cnt[(long long)w * nbins + b] = run;
        run += c;
    }
}

// ints: hist[nbins] | off[nbins+1] | cursor[nbins] | wgoff[nbins+1]
__global__ void __launch_bounds__(256)
bin_scan_kernel(int nbins, const int *__restrict__ hist, int *__restrict__ off, int *__restrict__ cursor,
                int *__restrict__ wgoff)
{
    __shared__ int sq[256], sw[256];
    const int per = (nbins + 255) / 256;
    const int b0 = threadIdx.x * per;
    int q = 0, w = 0;
    for (int b = b0; b < b0 + per && b < nbins; ++b) {
        q += hist[b];
        w += (hist[b] + EVAL_QPW - 1) / EVAL_QPW;
    }
    sq[threadIdx.x] = q;
    sw[threadIdx.x] = w;
    __syncthreads();
    for (int s = 1; s < 256; s <<= 1) {
        const int aq = threadIdx.x >= s ? sq[threadIdx.x - s] : 0;
        const int aw = threadIdx.x >= s ? sw[threadIdx.x - s] : 0;
        __syncthreads();
        sq[threadIdx.x] += aq;
        sw[threadIdx.x] += aw;
        __syncthreads();
    }
    q = sq[threadIdx.x] - q;          // exclusive
    w = sw[threadIdx.x] - w;
    for (int b = b0; b < b0 + per && b < nbins; ++b) {
        off[b] = q;
        cursor[b] = q;
        wgoff[b] = w;
        q += hist[b];
        w += (hist[b] + EVAL_QPW - 1) / EVAL_QPW;
    }
    if (threadIdx.x == 255) {
        off[nbins] = sq[255];
        wgoff[nbins] = sw[255];
    }
}

// Pass B.  The workgroup sorts its queries by region in LDS first, so that the copy to global
// memory walks every region's run with consecutive lanes on consecutive addresses (the
// straightforward per-query scatter issued one 8-byte store request per coordinate and was
// bound by the request rate, not by bytes).
template <int D, typename T>
__global__ void __launch_bounds__(256)
bin_scatter_kernel(Grid g, Regions rg, int n, const T *__restrict__ xq, int ldxq,
                   const int *__restrict__ wgbase, int ldw, double *__restrict__ xs)
{
    constexpr int QPT = ScatterShape<D>::QPT, QPW = 256 * QPT;
    __shared__ double sx[QPW * D];
    __shared__ int sidx[QPW];
    __shared__ unsigned short srid[QPW];
    extern __shared__ int lds_bins[];          // lh[nbins] | lbase[nbins]: sized by the launch, not by BIN_MAX
    int *lh = lds_bins, *lbase = lds_bins + rg.nbins;
    __shared__ int sscan[256];
    for (int b = threadIdx.x; b < rg.nbins; b += 256) lh[b] = 0;
    __syncthreads();
    const int base = blockIdx.x * QPW;
    int rid[QPT], rank[QPT];
    double xr[QPT][D];
#pragma unroll
    for (int j = 0; j < QPT; ++j) {
        const int i = base + j * 256 + threadIdx.x;
        rid[j] = -1;
        rank[j] = 0;
        if (i < n) {
#pragma unroll
            for (int d = 0; d < D; ++d) xr[j][d] = (double)xq[(long long)i * ldxq + d];
            rid[j] = region_of<D>(g, rg, xr[j]);
            rank[j] = atomicAdd(&lh[rid[j]], 1);
        }
    }
    __syncthreads();
    // exclusive scan of the local counts; lh[b] <- local start, lbase[b] <- global start - local start
    const int per = (rg.nbins + 255) / 256;
    const int b0 = threadIdx.x * per;
    int q = 0;
    for (int b = b0; b < b0 + per && b < rg.nbins; ++b) q += lh[b];
    sscan[threadIdx.x] = q;
    __syncthreads();
    for (int s = 1; s < 256; s <<= 1) {
        const int aq = threadIdx.x >= s ? sscan[threadIdx.x - s] : 0;
        __syncthreads();
        sscan[threadIdx.x] += aq;
        __syncthreads();
    }
    q = sscan[threadIdx.x] - q;
    for (int b = b0; b < b0 + per && b < rg.nbins; ++b) {
        const int c = lh[b];
        lh[b] = q;
        lbase[b] = wgbase[(long long)blockIdx.x * rg.nbins + b] - q;
        q += c;
    }
    const int total = sscan[255];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < QPT; ++j) {
        if (rid[j] < 0) continue;
        const int i = base + j * 256 + threadIdx.x;
        const int lp = lh[rid[j]] + rank[j];
#pragma unroll
        for (int d = 0; d < D; ++d) sx[d * QPW + lp] = xr[j][d];
        sidx[lp] = i;
        srid[lp] = (unsigned short)rid[j];
    }
    __syncthreads();
    // copy-out: consecutive lanes walk a region's run, one record each (consecutive 32- / 40-byte pieces)
    for (int lp = threadIdx.x; lp < total; lp += 256) {
        const long long gpos = lp + lbase[srid[lp]];
        double x[D];
#pragma unroll
        for (int d = 0; d < D; ++d) x[d] = sx[d * QPW + lp];
        store_record<D>(xs + gpos * (D + 1), x, sidx[lp]);
    }
}

constexpr int EVAL_WG = 1024;          // threads per workgroup in pass C (value path): 16 waves share one 32 KB tile (A/B: 256 -> 512 threads +3 %, 1024 +5 %)
// T = storage type of the coefficients and the results (double, or float for the REAL32 entry points: widened when the
// tile is filled / narrowed when a result is stored; the sorted coordinates are always double, the arithmetic too)
// (8 waves per SIMD: two of these 16-wave workgroups per CU need <= 64 registers -- the 4-D instantiation came out at 65 and
// ran ONE workgroup per CU until round 3)
template <int D, bool VAL, typename T>
__global__ void __launch_bounds__(EVAL_WG, 8)
eval_binned_kernel(Grid g, Regions rg, NDeriv nd, const T *__restrict__ coef,
                   const double *__restrict__ xs,
                   const int *__restrict__ off, const int *__restrict__ wgoff, T *__restrict__ out)
{
    constexpr int TILE_ELEMS = tile_elems<D>();
    __shared__ double tile[TILE_ELEMS];
    using TS = TileShape<D>;
    const int wg = blockIdx.x;
    if (wg >= wgoff[rg.nbins]) return;
    int lo = 0, hi = rg.nbins;               // wgoff[lo] <= wg < wgoff[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (wgoff[mid] <= wg) lo = mid; else hi = mid;
    }
    const int r = lo;                         // wgoff[r] <= wg < wgoff[r+1]: a non-empty region
    const int part = wg - wgoff[r];
    int a[D];                                  // first node of the region's tile
    {
        int rr = r;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            a[d] = (rr % rg.nreg[d]) * (TS::T[d] - 3);
            rr /= rg.nreg[d];
        }
    }
    using TT = TileStride<D>;
    __shared__ int s_cnt[32], s_off[32], s_fre[33];
    constexpr bool DEAL = D == 4;             // queries dealt to the lanes by LDS bank class (below; 3-D: 403 -> 476 us, the class sort costs more than the conflicts)
    __shared__ unsigned short s_list[DEAL ? EVAL_QPW : 1], s_ovf[DEAL ? EVAL_QPW : 1];
    if (DEAL && threadIdx.x < 32) s_cnt[threadIdx.x] = 0;
    for (int e = threadIdx.x; e < tile_cells<D>(); e += EVAL_WG) {
        int rem = e, idx = 0, te = 0;
        bool ok = true;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int l = rem % TS::T[d];
            rem /= TS::T[d];
            const int node = a[d] + l;
            ok = ok && node < g.nodes[d];
            idx += node * g.colstride[d];
            te += l * TT::S[d];
        }
        tile[te] = ok ? (double)coef[idx] : 0.0;
    }
    __syncthreads();
    const int qb = off[r] + part * EVAL_QPW;
    const int qe = min(off[r + 1], qb + EVAL_QPW);
    constexpr int t1 = TT::S[1], t2 = TT::S[2], t3 = TT::S[3];
    auto evaluate = [&](const double (&x)[D], int p) {
        double b[D][4];
        int base = 0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int ws = eval_table<VAL>(g, d, x[d], nd.v[d], b[d]);
            base += (ws - a[d]) * TT::S[d];
        }
        const double sum = window_sum<D>(b, [&](int k1, int k2, int k3, double (&c)[4]) {
            // four ds_read_b64 (2 LDS cycles each, 64 banks) instead of the two ds_read2_b64 the
            // compiler would merge them into (8 cycles each, 32 banks): volatile keeps them apart
            typedef const volatile __attribute__((address_space(3))) double *lds_cvd;
            lds_cvd q = (lds_cvd)tile + (base + k1 * t1 + k2 * t2 + k3 * t3);
            c[0] = q[0]; c[1] = q[1]; c[2] = q[2]; c[3] = q[3];
        });
        out[p] = (T)sum;
    };
    if constexpr (DEAL) {
        // Queries dealt to the lanes by bank class: lane h of every 32-lane half takes the queries whose tile offset is
        // h mod 32 (counting sort of the workgroup's <= 2 048 queries by that class in LDS).  The 64 window rows of a query are
        // read at the same constant offsets from its base by every lane, so lanes with distinct base classes never meet on a
        // bank: 2 LDS cycles per read instead of the ~10 of random windows.  Same arithmetic per query: identical bits.
        // A class holds 64 +- 8 of the 2 048 queries; every lane has exactly two rounds (64 slots per class = one per
        // half-wave and round), so what a class holds beyond 64 goes to the free slots of the short classes -- those few
        // lanes meet the lane of their own class on a bank (one extra LDS cycle), nobody idles.
        static_assert(EVAL_QPW == 2 * EVAL_WG && EVAL_WG == 32 * 32, "two rounds of 32 half-waves x 32 classes");
        const int nq = qe - qb;
        int key[2] = {-1, -1}, rk[2] = {0, 0};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int jj = (int)threadIdx.x + u * EVAL_WG;
            if (jj < nq) {
                double x[D];
                (void)load_record<D>(xs + (long long)(qb + jj) * (D + 1), x);
                int base = 0;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    int lo, hi;
                    base += (window_start(g, d, x[d], lo, hi) - a[d]) * TT::S[d];
                }
                key[u] = base & 31;
                rk[u] = atomicAdd(&s_cnt[key[u]], 1);
            }
        }
        __syncthreads();
        if (threadIdx.x < 32) {             // exclusive scans over the 32 classes: surplus (beyond 64) and free slots
            const int n = s_cnt[threadIdx.x];
            const int sur = n > 64 ? n - 64 : 0, fre = n < 64 ? 64 - n : 0;
            int is = sur, ifr = fre;
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) {
                const int ts = __shfl_up(is, o, 32), tf = __shfl_up(ifr, o, 32);
                if ((int)threadIdx.x >= o) { is += ts; ifr += tf; }
            }
            s_off[threadIdx.x] = is - sur;          // first surplus position of the class
            s_fre[threadIdx.x] = ifr - fre;         // first surplus entry its free slots take
            if (threadIdx.x == 31) s_fre[32] = is;  // surplus entries in all
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (key[u] >= 0) {
                const unsigned short id = (unsigned short)((int)threadIdx.x + u * EVAL_WG);
                if (rk[u] < 64) s_list[key[u] * 64 + rk[u]] = id;
                else s_ovf[s_off[key[u]] + rk[u] - 64] = id;
            }
        __syncthreads();
        const int h = threadIdx.x & 31, w = threadIdx.x >> 5;
        const int n_h = s_cnt[h], nsur = s_fre[32];
        int jq[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int slot = w + 32 * u;
            int id = -1;
            if (slot < n_h) id = s_list[h * 64 + slot];           // (slot < 64 always)
            else {
                const int e = s_fre[h] + (slot - n_h);
                if (e < nsur) id = s_ovf[e];
            }
            jq[u] = id;
        }
        double x0[D], x1[D];
        int p0 = 0, p1 = 0;
        if (jq[0] >= 0) p0 = load_record<D>(xs + (long long)(qb + jq[0]) * (D + 1), x0);
        if (jq[1] >= 0) p1 = load_record<D>(xs + (long long)(qb + jq[1]) * (D + 1), x1);
        if (jq[0] >= 0) evaluate(x0, p0);
        if (jq[1] >= 0) evaluate(x1, p1);
        return;
    }
    // the coordinates (and the destination) of the NEXT round are in flight while the current one is
    // evaluated: a round's global loads would otherwise be exposed once per round
    int j = qb + threadIdx.x;
    double xn[D];
    int pn = 0;
    if (j < qe) pn = load_record<D>(xs + (long long)j * (D + 1), xn);
    while (j < qe) {
        double x[D];
#pragma unroll
        for (int d = 0; d < D; ++d) x[d] = xn[d];
        const int p = pn;
        const int jn = j + EVAL_WG;
        if (jn < qe) pn = load_record<D>(xs + (long long)jn * (D + 1), xn);
        evaluate(x, p);
        j = jn;
    }
}

// ---- run path (3-D / 4-D value and single-pattern evaluation; round 3) ----------------------------------------
// The region sort above moves every query three times (count, place, evaluate) and spends a tenth of its time on the
// prefixes in between.  Here the place pass stops at what it has in LDS anyway: every workgroup writes ITS 2 048 queries,
// sorted by region, as one contiguous image (64 KB of records, fully coalesced) and leaves the starts of its runs in a row
// of `starts`; the evaluation workgroup (region r, group k) walks the runs (w, r) of the ~nbins workgroups of its
// group.  No count pass, no prefix kernels, no global order: 24 + 32 bytes per query in the place pass, 32 + 8 in the
// evaluation pass.  Same arithmetic per query as everywhere else: identical bits.
constexpr int RUN_QPW = 2048;          // queries per place-pass workgroup

// RUN_NT threads per workgroup: 62 KB of LDS allow two workgroups per CU, i.e. 16 waves with 512 threads each (8 with 256:
// too few for a pass that waits on memory)
constexpr int RUN_NT = 512;
template <int D, typename T>
__global__ void __launch_bounds__(RUN_NT)
run_place_kernel(Grid g, Regions rg, int n, const T *__restrict__ xq, int ldxq, double *__restrict__ img,
                 int *__restrict__ starts)
{
    constexpr int NT = RUN_NT, NW = NT / 64, QPT = RUN_QPW / NT, QPW = RUN_QPW;
    __shared__ double sx[QPW * D];
    __shared__ int sidx[QPW];
    extern __shared__ int lds_bins[];          // lstart[nbins + 1] | lcount[nbins]
    int *lst = lds_bins, *lcn = lds_bins + rg.nbins + 1;
    __shared__ int sscan[NW];
    for (int b = threadIdx.x; b < rg.nbins; b += NT) lcn[b] = 0;
    __syncthreads();
    const int base = blockIdx.x * QPW;
    int rid[QPT], rank[QPT];
    double xr[QPT][D];
#pragma unroll
    for (int j = 0; j < QPT; ++j) {
        const int i = base + j * NT + threadIdx.x;
        rid[j] = -1;
        rank[j] = 0;
        if (i < n) {
#pragma unroll
            for (int d = 0; d < D; ++d) xr[j][d] = (double)xq[(long long)i * ldxq + d];
            rid[j] = region_of<D>(g, rg, xr[j]);
            rank[j] = atomicAdd(&lcn[rid[j]], 1);
        }
    }
    __syncthreads();
    const int per = (rg.nbins + NT - 1) / NT;
    const int b0 = threadIdx.x * per;
    int q = 0;
    for (int b = b0; b < b0 + per && b < rg.nbins; ++b) q += lcn[b];
    // exclusive scan over the threads: within the waves by shuffles, then the wave totals (two barriers instead of the sixteen
    // of a Hillis-Steele scan in LDS: a third of this workgroup's time)
    int incl = q;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if ((int)(threadIdx.x & 63) >= o) incl += t;
    }
    if ((threadIdx.x & 63) == 63) sscan[threadIdx.x >> 6] = incl;
    __syncthreads();
    int woff = 0;
    for (int v = 0; v < (int)(threadIdx.x >> 6); ++v) woff += sscan[v];
    int total = 0;
#pragma unroll
    for (int v = 0; v < NW; ++v) total += sscan[v];
    q = woff + incl - q;
    for (int b = b0; b < b0 + per && b < rg.nbins; ++b) {
        lst[b] = q;
        q += lcn[b];
    }
    if (threadIdx.x == 0) lst[rg.nbins] = total;
    __syncthreads();
    for (int b = threadIdx.x; b <= rg.nbins; b += NT) starts[(long long)blockIdx.x * (rg.nbins + 1) + b] = lst[b];
#pragma unroll
    for (int j = 0; j < QPT; ++j) {
        if (rid[j] < 0) continue;
        const int lp = lst[rid[j]] + rank[j];
#pragma unroll
        for (int d = 0; d < D; ++d) sx[d * QPW + lp] = xr[j][d];
        sidx[lp] = base + j * NT + threadIdx.x;
    }
    __syncthreads();
    for (int lp = threadIdx.x; lp < total; lp += NT) {         // the sorted image: consecutive lanes, consecutive records
        double x[D];
#pragma unroll
        for (int d = 0; d < D; ++d) x[d] = sx[d * QPW + lp];
        store_record<D>(img + ((long long)blockIdx.x * QPW + lp) * (D + 1), x, sidx[lp]);
    }
}

// grp = place-pass workgroups per evaluation workgroup (<= RUN_GROUP_MAX): chosen by the host so that a region's queries in a
// group are ~2 000 (3-D 64^3: 128 x 16.4; 4-D 32^4: 615 x 3.2).
constexpr int RUN_GROUP_MAX = 1024;
template <int D, bool VAL, typename T>
__global__ void __launch_bounds__(EVAL_WG, 8)
eval_runs_kernel(Grid g, Regions rg, NDeriv nd, const T *__restrict__ coef, const double *__restrict__ img,
                 const int *__restrict__ starts, int nwg, int grp, T *__restrict__ out)
{
    constexpr bool DEAL = D == 4;             // queries dealt to the lanes by LDS bank class (see eval_binned_kernel)
    static_assert(EVAL_WG == 1024 && RUN_GROUP_MAX <= EVAL_WG, "one place-pass workgroup per thread in the prefix");
    __shared__ double tile[tile_elems<D>()];
    __shared__ int pre[RUN_GROUP_MAX + 1];
    __shared__ unsigned short rst[RUN_GROUP_MAX];
    __shared__ int wsum[16];
    __shared__ int s_cnt[32], s_sur[33], s_fre[33];
    __shared__ unsigned short s_list[DEAL ? EVAL_QPW : 1];      // [class][64 slots] (two of these workgroups share a CU's LDS: 79 KB each)
    using TS = TileShape<D>;
    using TT = TileStride<D>;
    const int r = blockIdx.x % rg.nbins, k = blockIdx.x / rg.nbins;
    const int w0 = k * grp;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // this region's runs in the group's workgroups: start inside the workgroup's image, inclusive prefix of the lengths
    {
        int c = 0, st = 0;
        if (tid < grp && w0 + tid < nwg) {
            const int *__restrict__ row = starts + (long long)(w0 + tid) * (rg.nbins + 1) + r;
            st = row[0];
            c = row[1] - st;
        }
        if (tid < grp) rst[tid] = (unsigned short)st;
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int woff = 0;
        for (int v = 0; v < wave; ++v) woff += wsum[v];
        if (tid < grp) pre[tid + 1] = woff + incl;
        if (tid == 0) pre[0] = 0;
    }
    __syncthreads();
    const int total = pre[grp];
    if (total == 0) return;
    int a[D];                                  // first node of the region's tile
    {
        int rr = r;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            a[d] = (rr % rg.nreg[d]) * (TS::T[d] - 3);
            rr /= rg.nreg[d];
        }
    }
    for (int e = tid; e < tile_cells<D>(); e += EVAL_WG) {
        int rem = e, idx = 0, te = 0;
        bool ok = true;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int l = rem % TS::T[d];
            rem /= TS::T[d];
            const int node = a[d] + l;
            ok = ok && node < g.nodes[d];
            idx += node * g.colstride[d];
            te += l * TT::S[d];
        }
        tile[te] = ok ? (double)coef[idx] : 0.0;
    }
    __syncthreads();
    constexpr int t1 = TT::S[1], t2 = TT::S[2], t3 = TT::S[3];
    auto locate = [&](int qi) -> const double * {      // record qi of the group's queries of this region
        int lo = 0, hi = grp;                          // pre[lo] <= qi < pre[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pre[mid] <= qi) lo = mid; else hi = mid;
        }
        return img + ((long long)(w0 + lo) * RUN_QPW + rst[lo] + (qi - pre[lo])) * (D + 1);
    };
    auto evaluate = [&](const double (&x)[D], int p) {
        double b[D][4];
        int base = 0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int ws = eval_table<VAL>(g, d, x[d], nd.v[d], b[d]);
            base += (ws - a[d]) * TT::S[d];
        }
        const double sum = window_sum<D>(b, [&](int k1, int k2, int k3, double (&c)[4]) {
            typedef const volatile __attribute__((address_space(3))) double *lds_cvd;
            lds_cvd qq = (lds_cvd)tile + (base + k1 * t1 + k2 * t2 + k3 * t3);
            c[0] = qq[0]; c[1] = qq[1]; c[2] = qq[2]; c[3] = qq[3];
        });
        out[p] = (T)sum;
    };
    if constexpr (DEAL) {
        // batches of <= 2 048 queries, dealt to the lanes by the bank class of their tile offset (eval_binned_kernel has the
        // reasoning): 64 slots per class = two rounds per lane; what a class holds beyond 64 fills the free slots of the
        // short classes
        for (int q0 = 0; q0 < total; q0 += EVAL_QPW) {
            const int nqb = total - q0 < EVAL_QPW ? total - q0 : EVAL_QPW;
            if (tid < 32) s_cnt[tid] = 0;
            __syncthreads();
            int key[2] = {-1, -1}, rk[2] = {0, 0};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int jj = tid + u * EVAL_WG;
                if (jj < nqb) {
                    double x[D];
                    (void)load_record<D>(locate(q0 + jj), x);
                    int base = 0;
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        int lo, hi;
                        base += (window_start(g, d, x[d], lo, hi) - a[d]) * TT::S[d];
                    }
                    key[u] = base & 31;
                    rk[u] = atomicAdd(&s_cnt[key[u]], 1);
                }
            }
            __syncthreads();
            if (tid < 32) {             // exclusive scans over the 32 classes: surplus (beyond 64) and free slots
                const int n = s_cnt[tid];
                const int sur = n > 64 ? n - 64 : 0, fre = n < 64 ? 64 - n : 0;
                int is = sur, ifr = fre;
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) {
                    const int ts = __shfl_up(is, o, 32), tf = __shfl_up(ifr, o, 32);
                    if (tid >= o) { is += ts; ifr += tf; }
                }
                s_sur[tid] = is - sur;
                s_fre[tid] = ifr - fre;
                if (tid == 31) { s_sur[32] = is; s_fre[32] = ifr; }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (key[u] >= 0) {
                    const unsigned short id = (unsigned short)(tid + u * EVAL_WG);
                    if (rk[u] < 64) s_list[key[u] * 64 + rk[u]] = id;
                    else {                               // surplus entry e takes the e-th free slot (classes in order)
                        const int e = s_sur[key[u]] + rk[u] - 64;
                        int lo = 0, hi = 32;             // s_fre[lo] <= e < s_fre[hi]
                        while (hi - lo > 1) {
                            const int mid = (lo + hi) >> 1;
                            if (s_fre[mid] <= e) lo = mid; else hi = mid;
                        }
                        s_list[lo * 64 + s_cnt[lo] + (e - s_fre[lo])] = id;
                    }
                }
            __syncthreads();
            const int h = tid & 31, w = tid >> 5;
            const int n_h = s_cnt[h], nsur = s_sur[32];
            int filled = n_h < 64 ? n_h : 64;           // own entries + the surplus entries that took this class's free slots
            if (n_h < 64) {
                int ex = nsur - s_fre[h];
                ex = ex < 0 ? 0 : (ex > 64 - n_h ? 64 - n_h : ex);
                filled += ex;
            }
            int jq[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int slot = w + 32 * u;
                jq[u] = slot < filled ? (int)s_list[h * 64 + slot] : -1;
            }
            double x0[D], x1[D];
            int p0 = 0, p1 = 0;
            if (jq[0] >= 0) p0 = load_record<D>(locate(q0 + jq[0]), x0);
            if (jq[1] >= 0) p1 = load_record<D>(locate(q0 + jq[1]), x1);
            if (jq[0] >= 0) evaluate(x0, p0);
            if (jq[1] >= 0) evaluate(x1, p1);
            __syncthreads();
        }
    } else {
        int qi = tid;
        double xn[D];
        int pn = 0;
        if (qi < total) pn = load_record<D>(locate(qi), xn);
        while (qi < total) {
            double x[D];
#pragma unroll
            for (int d = 0; d < D; ++d) x[d] = xn[d];
            const int p = pn;
            const int qn = qi + EVAL_WG;
            if (qn < total) pn = load_record<D>(locate(qn), xn);
            evaluate(x, p);
            qi = qn;
        }
    }
}

// ---- persistent region path (round 4) ---------------------------------------------------------------------------------
// The run path above moves every query twice through HBM as a 32-byte record and writes its 8-byte result into a sector that
// other regions' workgroups fill at other times: 149 B per query by the counters for 32 algorithmic ones.  Here:
//
//   pr_place_kernel   a workgroup reads 8 192 queries, bins them by (region, interior or not) in LDS and writes their
//                     coordinates as D planes in that order (staged through LDS: consecutive stores), their LOCAL indices
//                     (16 bit) and the starts of the runs.  24 B read, 26 B written per 3-D query.
//   pr_eval_kernel    PERSISTENT: one workgroup of 16 waves per CU.  A workgroup holds the coefficient tile of ONE region in
//                     LDS; its waves, each an independent worker, take chunks of that region's runs (two levels: the
//                     workgroup takes superchunks from the region's counter in global memory, its waves take chunks from
//                     a counter in LDS) and walk them as one stream (four runs in flight, so the lanes the tail of one run
//                     leaves idle start the next ones: no barrier, no prefix, no search), the boundary runs first, then
//                     the interior ones, whose lanes all take the 16-operation closed form of the basis table.  When a
//                     region is used up the workgroup moves to the one with the most work left per workgroup there.
//                     The coordinates of element i + 1 are requested before the window of element i is read.  Results go
//                     to the SAME sorted places: consecutive lanes, consecutive words.  24 B read, 8 B written.
//   pr_unsort_kernel  per place-pass workgroup: sorted results + local indices -> the caller's order, through LDS.  10 B read,
//                     8 B written.
//
// By construction 100 B of HBM traffic per 3-D query, every byte of it in consecutive runs of at least 200 B.  (The first
// form of this path sorted only the 16-bit indices and let the evaluation pass gather the coordinates and scatter the results
// through the L2 of "its" XCD: 60 B by construction, but the windows of 8 192 workers do not stay in a 4 MB L2 -- 108 B
// fetched and 124 B written per query by the counters, every 8-byte result a read-for-ownership and an eviction of a line.)
// Arithmetic per query: eval_table + window_sum as everywhere else -- identical bits.  Used for 3-D grids of at most 64 regions
// of 16 (or 8) window starts per dimension: 64^3 nodes give 4 x 4 x 4 regions with tiles of 19^3 coefficients (55 KB).
struct PRegions { int nreg[MAXD]; int sper[MAXD]; int text[MAXD]; int tstr[MAXD]; int nbins; int telems; int tcells; int deal; };

constexpr int PR_Q = 8192;         // queries per place-pass workgroup (16-bit local indices; half a coordinate plane of them = 32 KB of LDS)
constexpr int PR_NT = 1024;        // threads of a place-pass workgroup
constexpr int PR_EW = 1024;        // threads of an evaluation workgroup: 16 waves = 4 per SIMD, ONE workgroup per CU (two of 640 threads
                                   // never shared a CU: 10 waves go to the SIMDs as 3,3,2,2 and two such sets can need 6 x 88 registers
                                   // on one SIMD -- the second workgroup of every CU only started when the first had ended)
constexpr int PR_WPE = 4;          // waves per SIMD the evaluation kernel is compiled for (128 registers; it takes 88)
// place-pass shape by dimension count.  MAXBINS: most regions of a grid; NCLS: LDS slot classes the queries of a bin are dealt
// by (3-D: the 16 sixteen-byte slots of ds_read_b128; 4-D: 32 eight-byte slots, ds_read_b64); JMAX: rounds of the deal.
// 2 MAXBINS JMAX = 2 PR_NT: every thread of the place pass prefixes two (bin, round) counters.
template <int D> struct PRCfg { static constexpr int MAXBINS = 64, NCLS = 16, JMAX = 16; };
template <> struct PRCfg<4> { static constexpr int MAXBINS = 256, NCLS = 32, JMAX = 4; };
constexpr int PR_MAXBINS = 256;    // (scratch sizes: the largest of them)

template <int D, typename T>
__global__ void __launch_bounds__(PR_NT)
pr_place_kernel(Grid g, PRegions rg, long long nq, const T *__restrict__ xq, int ldxq, unsigned short *__restrict__ sidx,
                int *__restrict__ starts, T *__restrict__ xs, unsigned char *__restrict__ scls)
{
    // bins: 2 per region -- [2 r] the queries whose windows are interior ones in every dimension (closed-form basis table,
    // eval_table), [2 r + 1] the others (end functions / clipped windows in some dimension): the evaluation pass walks the
    // interior runs first and the others afterwards, so that its waves are homogeneous (a wave with one non-interior lane
    // pays the general table for that dimension)
    // Inside a bin the queries are DEALT by the LDS slot class of their window (the 16-byte slot, mod 16, of the address the
    // evaluation pass reads the window from: ds_read_b128 serves 16 lanes per cycle from 16 slots): first the first query of
    // every class, then the second of every class, ... -- neighbouring lanes of the evaluation pass then read from
    // different slots where a random order has ~3 of 16 lanes on the busiest one (SQ_LDS_BANK_CONFLICT was 70 % of that
    // pass's LDS cycles, and the LDS its bottleneck).  A query's rank within its (bin, class) is its ROUND; it marks its
    // class in the round's mask, the rounds' populations (popcounts) are prefixed over bins x rounds, and its place is the
    // start of its round + the number of classes below its own in that round (class order inside a round: 16 consecutive
    // queries then straddle two rounds with fewer repeats than in the order the counters would give, 8.4 against 9.8 LDS
    // cycles per read in a simulation of 64^3).  Ranks beyond JMAX - 1 (clustered queries) share the last round, in counter order.
    constexpr int QPT = PR_Q / PR_NT, NB2 = 2 * PRCfg<D>::MAXBINS, HALF = PR_Q / 2, NCLS = PRCfg<D>::NCLS, JMAX = PRCfg<D>::JMAX, NKEY = NB2 * JMAX;
    static_assert(NKEY == 2 * PR_NT, "the prefix below gives every thread two (bin, round) counters");
    constexpr int TAB_BYTES = NB2 * NCLS * 4 + 2 * NKEY * 4;
    constexpr int SB_BYTES = HALF * (int)sizeof(T) > TAB_BYTES ? HALF * (int)sizeof(T) : TAB_BYTES;
    __shared__ unsigned short ssort[PR_Q];
    // 4-D: the LDS slot class of every query's window (tile offset mod 32) travels with the image as a plane of bytes -- the
    // evaluation pass deals its chunks to the lanes by it (pr_eval4_kernel); runs of ~16 queries are too short to be dealt here
    __shared__ __attribute__((aligned(16))) unsigned char scl[D == 4 ? PR_Q : 4];
    __shared__ __attribute__((aligned(16))) unsigned char sbuf[SB_BYTES];      // the counters, then the staging half plane
    T *splane = reinterpret_cast<T *>(sbuf);
    int *ccnt = reinterpret_cast<int *>(sbuf);                                   // [bin][class]: queries so far
    unsigned *dmask = reinterpret_cast<unsigned *>(sbuf + NB2 * NCLS * 4);       // [bin][round]: the classes present (last round: a counter)
    int *rstart = reinterpret_cast<int *>(sbuf + NB2 * NCLS * 4 + NKEY * 4);     // [bin][round]: where the round starts in the image
    __shared__ int lst[NB2 + 1];
    __shared__ int wsum[PR_NT / 64];
    __shared__ unsigned char rtab[D][256];         // region index of a window start, per dimension (window starts < 256: host check)
    const int tid = threadIdx.x;
    const int nb2 = 2 * rg.nbins;
    for (int e = tid; e < D * 256; e += PR_NT) {
        const int d = e >> 8, ws = e & 255;
        const int rd = ws / rg.sper[d];
        rtab[d][ws] = (unsigned char)(rd < rg.nreg[d] - 1 ? rd : rg.nreg[d] - 1);
    }
    for (int e = tid; e < TAB_BYTES / 4; e += PR_NT) reinterpret_cast<int *>(sbuf)[e] = 0;
    __syncthreads();
    const long long base = (long long)blockIdx.x * PR_Q;
    int rid[QPT], rank[QPT];                       // rid: bin * JMAX + round; rank: within the last round, or -1 - class
    unsigned long long clsw = 0;                   // (4-D) the classes of this thread's queries, 8 bits each
    T xr[QPT][D];
#pragma unroll
    for (int j = 0; j < QPT; ++j) {
        const long long i = base + j * PR_NT + tid;
        rid[j] = -1;
        rank[j] = 0;
        if (i < nq) {
            int r = 0, m = 1, tb = 0;
            bool inter = true;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                int lo, hi;
                bool in_d;
                xr[j][d] = __builtin_nontemporal_load(xq + i * ldxq + d);
                const int ws = window_start_value(g, d, (double)xr[j][d], lo, hi, in_d);
                inter = inter && in_d;
                const int rd = (int)rtab[d][ws];
                r += rd * m;
                m *= rg.nreg[d];
                tb += (ws - rd * rg.sper[d]) * rg.tstr[d];          // the window's first entry in the region's tile
            }
            // (3-D tiles: an odd start reads the second copy of the tile, rg.telems entries further and one entry down)
            const int ta = (rg.telems > 0 && (tb & 1)) ? rg.telems + tb - 1 : tb;
            const int cls = rg.telems > 0 ? (ta >> 1) & (NCLS - 1) : tb & (NCLS - 1);
            if constexpr (D == 4) clsw |= (unsigned long long)cls << (8 * j);
            const int bin = 2 * r + (inter ? 0 : 1);
            int round = atomicAdd(&ccnt[bin * NCLS + cls], 1);
            if (!rg.deal) round = JMAX - 1;
            if (round < JMAX - 1) {
                atomicOr(&dmask[bin * JMAX + round], 1u << cls);
                rid[j] = bin * JMAX + round;
                rank[j] = -1 - cls;                                  // (place: by the mask)
            } else {
                rid[j] = bin * JMAX + JMAX - 1;
                rank[j] = (int)atomicAdd(&dmask[rid[j]], 1u);        // (place: by this counter)
            }
        }
    }
    __syncthreads();
    {
        // exclusive prefix over the NKEY = 2 PR_NT (bin, round) counters, in place
        // (two neighbouring rounds of one bin: the odd one may be the bin's last round, which holds a count, not a mask)
        const unsigned m0 = dmask[2 * tid], m1 = dmask[2 * tid + 1];
        const int c0 = __builtin_popcount(m0), c1 = ((2 * tid + 1) % JMAX == JMAX - 1) ? (int)m1 : __builtin_popcount(m1);
        int incl = c0 + c1;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if ((tid & 63) >= o) incl += t;
        }
        if ((tid & 63) == 63) wsum[tid >> 6] = incl;
        __syncthreads();
        int before = 0;
        for (int w = 0; w < (tid >> 6); ++w) before += wsum[w];
        const int excl = before + incl - (c0 + c1);
        rstart[2 * tid] = excl;
        rstart[2 * tid + 1] = excl + c0;
        if (tid == PR_NT - 1) lst[NB2] = excl + c0 + c1;
    }
    __syncthreads();
    for (int e = tid; e < NB2; e += PR_NT) lst[e] = rstart[e * JMAX];
    int lp[QPT];
#pragma unroll
    for (int j = 0; j < QPT; ++j) {
        lp[j] = -1;
        if (rid[j] >= 0) {
            lp[j] = rstart[rid[j]] + (rank[j] >= 0 ? rank[j] : __builtin_popcount(dmask[rid[j]] & ((1u << (-1 - rank[j])) - 1u)));
            ssort[lp[j]] = (unsigned short)(j * PR_NT + tid);
            if constexpr (D == 4) scl[lp[j]] = (unsigned char)(clsw >> (8 * j));
        }
    }
    __syncthreads();                               // (the counters give way to the staging buffer; lst is complete)
    const int total = lst[NB2];
    // the coordinates go to their sorted places in the workgroup's image (D planes of PR_Q entries: the evaluation pass reads
    // them with consecutive lanes on consecutive entries), half a plane at a time through LDS so that the stores are
    // consecutive (straight scattered 8-byte stores into the image: 1.03 ms per 5e7 queries instead of 0.25 without them)
    if constexpr (D == 4) {
        // 4-D: the image holds RECORDS of the four coordinates (32 bytes) instead of planes -- the evaluation pass takes its
        // elements in an order dealt by LDS bank class, i.e. scattered over a chunk, and a scattered element then costs one
        // sector instead of four lines (measured with planes: 2.8 of the pass's 5 ms per 1e8 queries went into these loads).
        // Staged through LDS a quarter of the image at a time.
        constexpr int PIECE = PR_Q / 4;
        static_assert(PIECE * 4 * (int)sizeof(T) <= SB_BYTES, "a quarter of the records fits the staging buffer");
        T *__restrict__ dstr = xs + (long long)blockIdx.x * PR_Q * D;
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {
            if (pc > 0) __syncthreads();
#pragma unroll
            for (int j = 0; j < QPT; ++j) {
                const int l = lp[j] - pc * PIECE;
                if (lp[j] >= 0 && l >= 0 && l < PIECE) {
#pragma unroll
                    for (int d = 0; d < D; ++d) splane[l * D + d] = xr[j][d];
                }
            }
            __syncthreads();
            const int nrec = total - pc * PIECE < PIECE ? total - pc * PIECE : PIECE;
            for (int e = tid; e < nrec * D; e += PR_NT) __builtin_nontemporal_store(splane[e], dstr + (long long)pc * PIECE * D + e);
        }
    } else {
#pragma unroll
    for (int d = 0; d < D; ++d) {
        T *__restrict__ dstp = xs + ((long long)blockIdx.x * D + d) * PR_Q;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            if (d + hf > 0) __syncthreads();
#pragma unroll
            for (int j = 0; j < QPT; ++j) {
                const int l = lp[j] - hf * HALF;
                if (lp[j] >= 0 && l >= 0 && l < HALF) splane[l] = xr[j][d];
            }
            __syncthreads();
            for (int e = tid; e < HALF && hf * HALF + e < total; e += PR_NT) __builtin_nontemporal_store(splane[e], dstp + hf * HALF + e);
        }
    }
    }
    for (int e = tid; e <= nb2; e += PR_NT) starts[(long long)blockIdx.x * (nb2 + 1) + e] = lst[e];
    unsigned *__restrict__ dst = reinterpret_cast<unsigned *>(sidx + base);
    for (int e = tid; 2 * e < total; e += PR_NT) {
        const unsigned lo = ssort[2 * e], hi = 2 * e + 1 < total ? ssort[2 * e + 1] : 0u;
        dst[e] = lo | (hi << 16);
    }
    if constexpr (D == 4) {
        static_assert(QPT <= 8, "eight class bytes per thread");
        unsigned *__restrict__ dc = reinterpret_cast<unsigned *>(scls + base);
        const unsigned *sc4 = reinterpret_cast<const unsigned *>(scl);
        for (int e = tid; 4 * e < total; e += PR_NT) dc[e] = sc4[e];          // (the bytes beyond `total` in the last word are never read)
    }
}

template <int D, int SPER> struct PTile {            // tile of a region: SPER window starts + 3 per dimension, compile-time LDS strides
    static constexpr int TE = SPER + 3;
    // 3-D: the rows of a window are read as two 16-byte halves (ds_read_b128: 16 lanes per LDS cycle, 16-byte slots -- a
    // random set of 16 slots out of 16 collides less than 32 out of 32, and half as many instructions), which must be
    // 16-byte aligned: even strides, and a SECOND copy of the tile one entry further for the windows with an odd start
    static constexpr bool W128 = D == 3;
    static constexpr int S1 = W128 ? ((TE + 1) & ~1) : (TE | 1);      // (else) odd row stride: the window rows of a lane spread over the LDS banks
    static constexpr int S2 = W128 ? S1 * TE : S1 * TE + 1;
    static constexpr int S3 = S2 * TE + 1;
    static constexpr int stride(int d) { return d == 0 ? 1 : (d == 1 ? S1 : (d == 2 ? S2 : S3)); }
    static constexpr int COPY = ((D == 1 ? TE : (D == 2 ? S1 * TE : (D == 3 ? S2 * TE : S3 * TE))) + 8 + 1) & ~1;
    static constexpr int ELEMS = W128 ? 2 * COPY : COPY;
};

template <int D, int SPER, bool VAL, typename T>
__global__ void __launch_bounds__(PR_EW, PR_WPE)
pr_eval_kernel(Grid g, PRegions rg, NDeriv nd, const T *__restrict__ coef, const T *__restrict__ xs,
               const int *__restrict__ starts, int nwg_all, int c0, int *__restrict__ queue, T *__restrict__ outs)
{
    using PT = PTile<D, SPER>;
    __shared__ __attribute__((aligned(16))) double pr_tile[PT::ELEMS];
    __shared__ int s_region, s_next, s_done, s_base[8], s_ready[8];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int t1 = D > 1 ? PT::S1 : 0, t2 = D > 2 ? PT::S2 : 0, t3 = D > 3 ? PT::S3 : 0;
    // Work distribution.  A workgroup holds the tile of ONE region; its waves, each on its own, take chunks of that region's
    // runs (c0 place-pass workgroups of interior runs, 4 c0 of the others; take_chunk below) until the region is used up, then
    // the workgroup moves to the region with the most work left per workgroup already there.
    // Why per wave: the SIMD issues from its oldest wave first, so the waves of a workgroup given equal shares finish one
    // after the other -- with a workgroup-wide barrier per work item the first wave waited 28-42 % of its life at barriers
    // (in-kernel clocks) while the last ones ran alone on their SIMDs with nothing to hide their latencies behind.
    //   queue[r]          SUPERCHUNKS (SC chunks) of region r taken (may overshoot by one per workgroup)
    //   queue[nbins + r]  workgroups at region r
    const int nbins = rg.nbins;
    const int c1 = 4 * c0;
    const int nch0 = (nwg_all + c0 - 1) / c0, nch1 = (nwg_all + c1 - 1) / c1;
    // lane L of a wave: cost of region L (window starts it covers, those with a boundary window in some dimension 2.5-fold)
    // and whether it has boundary windows at all (regions without them have no chunks of the second kind)
    double cost_l = 0.0;
    bool edge_l = false;
    if (lane < nbins) {
        double vol = 1.0, inter = 1.0;
        int rr = lane;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int rd = rr % rg.nreg[d], S = g.nodes[d] - 3;
            rr /= rg.nreg[d];
            const int lo = rd * rg.sper[d], hi = min(S, lo + rg.sper[d]);
            const int ilo = max(lo, 2), ihi = min(hi, g.nodes[d] - 5);
            vol *= hi - lo;
            inter *= max(0, ihi - ilo);
        }
        cost_l = vol + 1.5 * (vol - inter);
        edge_l = inter < vol;
    }
    const unsigned long long edge_mask = __builtin_amdgcn_ballot_w64(edge_l);
    if (wave == 0) {
        // the first region of this workgroup: workgroups are dealt to the regions in proportion to the costs
        double cum = cost_l;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            const double up = __shfl_up(cum, sft);
            if (lane >= sft) cum += up;
        }
        const double total = __shfl(cum, 63);
        const double target = ((double)blockIdx.x + 0.5) * total / (double)gridDim.x;
        int r0 = __builtin_popcountll(__builtin_amdgcn_ballot_w64(cum <= target));
        if (r0 > nbins - 1) r0 = nbins - 1;
        if (lane == 0) {
            atomicAdd(queue + nbins + r0, 1);
            s_region = r0;
        }
    }
    __syncthreads();
    int a[D];                                        // first node of the current region's tile
    for (;;) {
    const int r = __builtin_amdgcn_readfirstlane(s_region);
    if (r < 0) break;
    {
        int rr = r;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            a[d] = (rr % rg.nreg[d]) * rg.sper[d];
            rr /= rg.nreg[d];
        }
        for (int e = tid; e < rg.tcells; e += PR_EW) {
            int rem = e, idx = 0, te = 0;
            bool ok = true;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int l = rem % rg.text[d];
                rem /= rg.text[d];
                const int node = a[d] + l;
                ok = ok && node < g.nodes[d];
                idx += node * g.colstride[d];
                te += l * PT::stride(d);
            }
            const double cv = ok ? (double)coef[idx] : 0.0;
            pr_tile[te] = cv;
            if (PT::W128 && te > 0) pr_tile[PT::COPY + te - 1] = cv;      // the copy for odd window starts: entry i holds tile entry i + 1
        }
    }
    constexpr int SC = 16;                           // chunks of a superchunk
    const int nedge = ((edge_mask >> r) & 1ull) ? nch1 : 0;      // chunks of boundary runs come first
    const int ntot = nedge + nch0, nsc = (ntot + SC - 1) / SC;
    if (tid == 0) {
        const int b0 = atomicAdd(queue + r, 1);
        s_next = 0;
        s_done = b0 >= nsc ? 1 : 0;
        s_base[0] = b0 < nsc ? b0 : -1;
        s_ready[0] = 1;
        for (int k = 1; k < 8; ++k) s_ready[k] = 0;
    }
    __syncthreads();
    constexpr int NR = 4;
    const int nb1 = 2 * nbins + 1;
    // A wave's stream of runs goes on across its chunks (the first form drained the lanes and paid two memory round trips
    // at the start of every chunk: a sixth of a wave's life by the in-kernel clocks).
    // Chunks are handed out in two levels: the region's counter in global memory counts SUPERCHUNKS of SC chunks (an atomic
    // add on one address from all XCDs took ~0.4 us of that address's time: with one per chunk and wave the 64 counters were
    // the bottleneck at 2 place-pass workgroups per chunk and cost 10 us per chunk at 8); inside the workgroup the waves
    // take chunk numbers from a counter in LDS.  Superchunk k of the workgroup is published (s_base, s_ready = k + 1) by the
    // wave that took its chunk SC/2 of superchunk k - 1: half a superchunk ahead of its first use.
    auto take_chunk = [&]() -> int {
        int v = 0;
        if (lane == 0) v = atomicAdd(&s_next, 1);
        v = __builtin_amdgcn_readfirstlane(v);
        const int k = v / SC, o = v % SC;
        for (;;) {
            if (__hip_atomic_load(&s_ready[k & 7], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == k + 1) break;
            if (__hip_atomic_load(&s_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return -1;       // (set after every valid superchunk was published)
            __builtin_amdgcn_s_sleep(4);
        }
        const int b = __hip_atomic_load(&s_base[k & 7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (o == SC / 2) {
            // this wave publishes superchunk k + 1, whatever happens to its own chunk: the waves that hold numbers of it wait for that
            int nb = -1;
            if (b >= 0) {
                int t = 0;
                if (lane == 0) t = atomicAdd(queue + r, 1);
                t = __builtin_amdgcn_readfirstlane(t);
                if (t < nsc) nb = t;
            }
            if (lane == 0) {
                __hip_atomic_store(&s_base[(k + 1) & 7], nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (nb < 0) __hip_atomic_store(&s_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(&s_ready[(k + 1) & 7], k + 2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (b < 0) return -1;
        const int c = b * SC + o;
        return c < ntot ? c : -1;
    };
    {
        bool more = true;
        int pw = 0, pend = 0, pcol = 0;              // the run whose descriptor is asked for next, the end of its chunk, its column
        int rst[NR], rlen[NR], rwa[NR];              // runs in flight: first element, length, place-pass workgroup (< 0: none)
        int qv = 0, qw = -1;                         // the run after them, as loaded (lanes 0 and 1; made uniform when it moves up)
        // (a VECTOR load, lanes 0 and 1 fetching the two words: as a scalar load -- the address is uniform -- it counts on
        //  lgkmcnt, and the lgkmcnt(0) waits of the window reads then wait for IT: a trip to L2 in front of every window.
        //  The loaded register is only looked at when the run moves up, several runs later: reading it into scalars at once
        //  was a full memory round trip at every run boundary)
        auto fetch_run = [&](int &v, int &w) {
            if (pw == pend && more) {
                const int c = take_chunk();
                if (c < 0)
                    more = false;
                else {
                    const int ph = c < nedge ? 1 : 0;
                    pw = ph ? c * c1 : (c - nedge) * c0;
                    pend = pw + (ph ? c1 : c0);
                    pend = pend < nwg_all ? pend : nwg_all;
                    pcol = 2 * r + ph;
                }
            }
            v = 0;
            w = -1;
            if (pw < pend) {
                v = starts[(long long)pw * nb1 + pcol + (lane & 1)];
                w = pw;
                ++pw;
            }
        };
        {
            int fv[NR];
#pragma unroll
            for (int j = 0; j < NR; ++j) fetch_run(fv[j], rwa[j]);
            fetch_run(qv, qw);
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                rst[j] = __builtin_amdgcn_readlane(fv[j], 0);
                rlen[j] = __builtin_amdgcn_readlane(fv[j], 1) - rst[j];
            }
        }
        // this lane's next element of the stream, relative to the start of run 0.  Not simply the lane number: ds_read_b128
        // serves the lanes in four fixed sets of 16 per half wave ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}); each set takes 16
        // CONSECUTIVE elements of the stream, which the place pass dealt so that neighbours read different LDS slots
        int pos;
        {
            const int m = lane & 31;
            pos = (lane & 32) + (m < 4 ? m : (m < 12 ? m + 12 : (m < 16 ? m - 8 : (m < 20 ? m + 8 : (m < 28 ? m - 12 : m)))));
        }
        // locates the lane's next element (drops the runs every lane has passed) and requests its coordinates
        auto next_element = [&](bool &act, long long &oq, T (&xn)[D]) {
            while (rwa[0] >= 0 && __builtin_amdgcn_ballot_w64(pos < rlen[0]) == 0) {
                pos -= rlen[0];
#pragma unroll
                for (int j = 0; j + 1 < NR; ++j) { rst[j] = rst[j + 1]; rlen[j] = rlen[j + 1]; rwa[j] = rwa[j + 1]; }
                rst[NR - 1] = __builtin_amdgcn_readlane(qv, 0);
                rlen[NR - 1] = __builtin_amdgcn_readlane(qv, 1) - rst[NR - 1];
                rwa[NR - 1] = qw;
                fetch_run(qv, qw);
            }
            act = false;
            int off = 0, wa = 0, rel = pos;
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const bool here = !act && rel < rlen[j];
                if (here) { off = rst[j] + rel; wa = rwa[j]; }
                act = act || here;
                rel -= rlen[j];
            }
            if (act) {
                oq = (long long)wa * PR_Q + off;
#pragma unroll
                for (int d = 0; d < D; ++d) xn[d] = __builtin_nontemporal_load(xs + ((long long)wa * D + d) * PR_Q + off);
                pos += 64;
            }
        };
        // the coordinates of element i + 1 are requested between the basis tables and the window sum of element i: they
        // arrive while the window is read from LDS (without this the two memory round trips and the LDS phase of a wave simply
        // added up: 0.5 + 0.5 ms of a 1.2 ms pass per 5e7 queries, measured by leaving either out)
        bool act = false;
        long long oq = 0;
        T xc[D];
#pragma unroll
        for (int d = 0; d < D; ++d) xc[d] = (T)0;
        next_element(act, oq, xc);
        while (__builtin_amdgcn_ballot_w64(act) != 0) {
            double b[D][4];
            int base = 0;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int ws = eval_table<VAL>(g, d, (double)xc[d], nd.v[d], b[d]);
                base += (ws - a[d]) * PT::stride(d);
            }
            if (!act) base = 0;
            const bool act_c = act;
            const long long oq_c = oq;
            next_element(act, oq, xc);
            double sum;
            if constexpr (D == 3) {
                // window_sum<3> with the LDS reads written as inline assembly, one k2 plane (16 reads, 32 registers) at a time:
                // left to the compiler, all 64 reads of a window are issued up front (128 registers: 2 waves per SIMD, or
                // spills), and behind a function call the coordinates requested above would be waited for at the call (the
                // compiler drains every counter there).  Same operations in the same order as window_sum<3>: identical bits.
                // an odd window start reads the second copy of the tile, one entry down: every address is 16-byte aligned
                const unsigned la = (unsigned)(size_t)(const __attribute__((address_space(3))) double *)pr_tile +
                                    (unsigned)((base & 1) ? PT::COPY + base - 1 : base) * 8u;
                sum = 0.0;
                // (the reads need the window starts only and would move above the basis tables, which then spill: the first
                //  read names the tables as operands it does not use)
                asm volatile("; tables ready %0 %1 %2 %3 %4 %5" :: "v"(b[0][0]), "v"(b[0][3]), "v"(b[1][0]), "v"(b[1][3]), "v"(b[2][0]), "v"(b[2][3]));
                __builtin_amdgcn_sched_barrier(0);
                typedef double d2v __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) {
                    d2v c[4][2];
                    // (the first read of a plane names the running sum as an operand it does not use: the multiply-adds of the
                    //  plane before stay in front of it)
                    asm volatile("ds_read_b128 %0, %1 offset:%2 ; after %3" : "=v"(c[0][0]) : "v"(la), "n"((k2 * t2) * 8), "v"(sum));
#pragma unroll
                    for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
                        for (int h = (k1 == 0 ? 1 : 0); h < 2; ++h)
                            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c[k1][h]) : "v"(la), "n"((k1 * t1 + k2 * t2 + 2 * h) * 8));
                    // (the values pass through the wait as in/out operands: what uses them stays behind it)
                    asm volatile("s_waitcnt lgkmcnt(0)"
                                 : "+v"(c[0][0]), "+v"(c[0][1]), "+v"(c[1][0]), "+v"(c[1][1]), "+v"(c[2][0]), "+v"(c[2][1]), "+v"(c[3][0]), "+v"(c[3][1])
                                 :: "memory");
                    double rr = 0.0;
#pragma unroll
                    for (int k1 = 0; k1 < 4; ++k1) {
                        double t = c[k1][0].x * b[0][0];
                        t = fma(c[k1][0].y, b[0][1], t);
                        t = fma(c[k1][1].x, b[0][2], t);
                        t = fma(c[k1][1].y, b[0][3], t);
                        rr = fma(t, b[1][k1], rr);
                    }
                    sum = fma(rr, b[2][k2], sum);
                    __builtin_amdgcn_sched_barrier(0);          // (the multiply-adds of a plane stay in front of the next plane's reads)
                }
            } else {
                sum = window_sum<D>(b, [&](int k1, int k2, int k3, double (&c)[4]) {
                    typedef const volatile __attribute__((address_space(3))) double *lds_cvd;
                    lds_cvd qq = (lds_cvd)pr_tile + (base + k1 * t1 + k2 * t2 + k3 * t3);
                    c[0] = qq[0]; c[1] = qq[1]; c[2] = qq[2]; c[3] = qq[3];
                });
            }
            if (act_c) __builtin_nontemporal_store((T)sum, outs + oq_c);
        }
    }      // chunks
    __syncthreads();                                 // every wave is done with the tile
    if (wave == 0) {
        // the next region: the one with the most chunks left per workgroup that would then be there
        int left = 0, there = 0;
        if (lane < nbins) {
            const int taken = __hip_atomic_load(queue + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            there = __hip_atomic_load(queue + nbins + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            left = (nch0 + (((edge_mask >> lane) & 1ull) ? nch1 : 0) + SC - 1) / SC - taken;
        }
        const float score = left > 0 ? (float)left / (float)(there + 1) : 0.0f;
        unsigned long long key = ((unsigned long long)__float_as_uint(score) << 32) | (unsigned)lane;
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) {
            const unsigned long long o = __shfl_xor(key, sft);
            key = o > key ? o : key;
        }
        if (lane == 0) {
            const int rn = (key >> 32) != 0 ? (int)(key & 63u) : -1;
            atomicSub(queue + nbins + r, 1);
            if (rn >= 0) atomicAdd(queue + nbins + rn, 1);
            s_region = rn;
        }
    }
    __syncthreads();
    }      // regions
}

// ---- persistent region path, 4-D (round 5) --------------------------------------------------------------------------------
// BASELINE config 5's evaluation half (4-D 32^4, 1e8 queries) ran the three-pass global region sort of round 3: seven launches
// per 2^24 queries, 287 B of fabric traffic per query for 40 algorithmic ones, the tile of a region (64 KB) loaded once per
// 2 048 queries: 8.07 ms per 1e8 queries.  Here the 3-D scheme above: pr_place_kernel<4> (regions of 8 window starts per
// dimension: 32^4 nodes give 4^4 = 256 regions, two bins each; the image holds 32-byte RECORDS of the coordinates and a byte
// per query with the LDS bank class of its window), this kernel, pr_unsort_kernel -- 32 + 35, 35 + 8 and 10 + 8 bytes per query:
// 6.2 ms per 1e8 queries (place 1.45, this kernel 4.4, unsort 0.33).
//   One persistent workgroup of 12 waves per CU (3 per SIMD: 168 registers, no spills; 16 waves left 128 and spilled in the round
//   loop) holds the tile of ONE region in LDS: (8 + 3)^4 coefficients with odd strides = 118 KB.  Its waves, each on its own,
//   take CHUNKS of the region's runs -- c0 consecutive place-pass workgroups of one bin -- from the region's counter (the number
//   of the next chunk is requested before the current one is worked on).  A chunk's run descriptors sit one per lane; a prefix
//   over the lanes makes the chunk ONE stream of elements, element e of which is found by a six-step search over the lanes'
//   prefixes -- runs of a 4-D place-pass workgroup are ~16 queries long (8 192 queries over 512 bins), too short for the 3-D
//   kernel's four-runs-in-flight walk.
//   What the counters and A/B builds showed, step by step (tools/eval4_bench.py, per 1e8 queries):
//   * elements in stream order: 5.3 ms in this kernel, the LDS pipe busy for 4.7 of them, 69 % of its cycles bank conflicts
//     (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE): the 256 window reads of a query (ds_read_b64: two groups of 32 lanes, 64 banks
//     of 4 bytes) all have the bank class of its tile offset mod 32, and 32 random classes per group collide 2-4 ways.
//   * so a wave DEALS every 256 elements of its chunk to its lanes by class, as round 3's region kernel dealt a workgroup's
//     2 048 queries: a counting sort in the wave's own LDS gives lanes h and h + 32 the elements of class h, two per round; what
//     a class holds beyond its 2 R slots goes to the free slots of short classes (conflicts: 69 % -> 43 % of the LDS cycles).
//   * dealt elements are scattered over the chunk: with coordinate PLANES every element cost four lines (2.8 ms); records: one
//     sector.  The 64 results of a round go to ~60 different lines: stored from the round they cost 1.2-1.4 ms (plain or
//     non-temporal stores alike); they are parked in the wave's LDS by element and leave in stream order, consecutive lanes on
//     consecutive slots.
//   * with the window reads taken out the kernel takes 2.3 ms (914 vector instructions per query: 340 multiply-adds, the
//     basis tables -- 45 % of the queries of a 32^4 grid have a window next to an end in some dimension --, two element
//     searches), with them 4.4: the reads do not overlap the arithmetic.  Software-pipelining them in units of 8 with
//     s_waitcnt lgkmcnt(8) broke: the compiler reloads fields of the by-value Grid argument with s_load inside the loop, which
//     count on lgkmcnt and return out of order (wrong values on a 20^4 grid), and the registers of a second buffer spilled.
//   * interleaving TWO elements per lane (one element's next 8 reads in flight while the other's are multiplied, waits of
//     lgkmcnt(0) only): 6.26 against 6.21 ms -- the reads are not waited for, the LDS pipe itself is the limit (its 2 cycles per
//     read become ~3.5 with the conflicts left, plus the deal's own traffic).  "Pure rounds + a mixed round for the surplus"
//     would cut the conflicts but adds a quarter more rounds of arithmetic at deals of 256: not built.
//   A workgroup whose region is used up moves to the region with the most chunks left.  No spinning anywhere: counters,
//   barriers, and loops every wave leaves when its region has no chunk left.
// Arithmetic per query: eval_table + window_sum<4> as everywhere else -- identical bits.
constexpr int PR4_EW = 768;        // threads of a 4-D evaluation workgroup: 12 waves = 3 per SIMD (168 registers each), one workgroup per CU
template <int SPER, bool VAL, typename T>
__global__ void __launch_bounds__(PR4_EW, 3)
pr_eval4_kernel(Grid g, PRegions rg, NDeriv nd, const T *__restrict__ coef, const T *__restrict__ xs, const unsigned char *__restrict__ scls,
                const int *__restrict__ starts, int nwg_all, int c0, int *__restrict__ queue, T *__restrict__ outs)
{
    constexpr int D = 4, PR_EW = PR4_EW, NWAVE = PR_EW / 64, SUB = 256, RMAX = SUB / 64, CAPMAX = 2 * RMAX;
    using PT = PTile<D, SPER>;
    __shared__ __attribute__((aligned(16))) double pr_tile[PT::ELEMS];
    __shared__ double w_res[NWAVE][SUB];             // results of a deal, by element: they leave in stream order (consecutive stores)
    __shared__ unsigned short w_list[NWAVE][32][CAPMAX], w_ovf[NWAVE][SUB];
    __shared__ int w_cnt[NWAVE][32];
    __shared__ int s_region;
    const int tid = threadIdx.x, lane = tid & 63, hcl = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int t1 = PT::S1, t2 = PT::S2, t3 = PT::S3;
    const int nbins = rg.nbins, nb1 = 2 * nbins + 1;
    const int nch = (nwg_all + c0 - 1) / c0;         // chunks per bin
    unsigned short (*list)[CAPMAX] = w_list[wave];
    unsigned short *ovf = w_ovf[wave];
    double *res = w_res[wave];
    int *cnt = w_cnt[wave];
    if (lane < 32) cnt[lane] = 0;
    // does region r hold windows that are not interior ones (its second bin is non-empty)?
    auto has_edge = [&](int r) {
        bool edge = false;
        int rr = r;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int rd = rr % rg.nreg[d], S = g.nodes[d] - 3;
            rr /= rg.nreg[d];
            const int lo = rd * rg.sper[d], hi = min(S, lo + rg.sper[d]);
            edge = edge || lo < 2 || hi > g.nodes[d] - 5;
        }
        return edge;
    };
    if (tid == 0) {
        // first region: workgroup b starts at region b mod nbins (more workgroups than regions: they share from the start)
        const int r0 = (int)(blockIdx.x % (unsigned)nbins);
        atomicAdd(queue + nbins + r0, 1);
        s_region = r0;
    }
    __syncthreads();
    int a[D];
    for (;;) {
        const int r = __builtin_amdgcn_readfirstlane(s_region);
        if (r < 0) break;
        {
            int rr = r;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                a[d] = (rr % rg.nreg[d]) * rg.sper[d];
                rr /= rg.nreg[d];
            }
            for (int e = tid; e < rg.tcells; e += PR_EW) {
                int rem = e, idx = 0, te = 0;
                bool ok = true;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const int l = rem % rg.text[d];
                    rem /= rg.text[d];
                    const int node = a[d] + l;
                    ok = ok && node < g.nodes[d];
                    idx += node * g.colstride[d];
                    te += l * PT::stride(d);
                }
                pr_tile[te] = ok ? (double)coef[idx] : 0.0;
            }
        }
        __syncthreads();
        const int nedge = has_edge(r) ? nch : 0;     // chunks of boundary runs come first (the costlier ones: no long tail)
        const int ntot = nedge + nch;
        auto take = [&]() {
            int v = 0;
            if (lane == 0) v = atomicAdd(queue + r, 1);
            return v;                                // (lane 0's register; made uniform when it is looked at)
        };
        int cnext = take();
        for (;;) {
            const int c = __builtin_amdgcn_readfirstlane(cnext);
            if (c >= ntot) break;
            cnext = take();                          // the next chunk's number travels while this chunk is worked on
            const int ph = c < nedge ? 1 : 0, cc = ph ? c : c - nedge;
            const int w0 = cc * c0, nw = min(c0, nwg_all - w0), col = 2 * r + ph;
            // one run per lane (c0 <= 64): start in its workgroup's image, inclusive prefix of the lengths
            int st_l = 0, len_l = 0;
            if (lane < nw) {
                const int *sp = starts + (long long)(w0 + lane) * nb1 + col;
                st_l = sp[0];
                len_l = sp[1] - st_l;
            }
            int pre = len_l;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(pre, o, 64);
                if (lane >= o) pre += t;
            }
            const int total = __builtin_amdgcn_readlane(pre, 63);
            const int excl = pre - len_l;
            // element e of the chunk -> its slot in the images (every lane takes part: the prefixes are read across the lanes)
            auto locate = [&](int e) -> unsigned {
                int lo = 0, hi = 63;
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int mid = (lo + hi) >> 1;
                    const int v = __shfl(pre, mid, 64);
                    if (v <= e) lo = mid + 1; else hi = mid;
                }
                const int off = __shfl(st_l, lo, 64) + (e - __shfl(excl, lo, 64));
                return (unsigned)(w0 + lo) * (unsigned)PR_Q + (unsigned)off;
            };
            auto request = [&](unsigned oq, bool act, T (&x)[D]) {
                // (records of the four coordinates: one or two 16-byte loads)
                typedef T rec4 __attribute__((ext_vector_type(4)));
                rec4 v = {(T)0, (T)0, (T)0, (T)0};
                if (act) v = __builtin_nontemporal_load(reinterpret_cast<const rec4 *>(xs) + oq);
                x[0] = v[0]; x[1] = v[1]; x[2] = v[2]; x[3] = v[3];
            };
            for (int E0 = 0; E0 < total; E0 += SUB) {
                const int n = min(SUB, total - E0), R = (n + 63) >> 6, cap = 2 * R;
                // ---- deal the n elements to the lanes by class
                int ck[RMAX], rk[RMAX];
                unsigned oqk[RMAX];                  // (kept: the results are stored to these slots at the end of the deal)
#pragma unroll
                for (int k = 0; k < RMAX; ++k) {
                    const int i = k * 64 + lane;
                    const bool valid = k < R && i < n;
                    oqk[k] = locate(valid ? E0 + i : 0);
                    ck[k] = valid ? (int)scls[oqk[k]] : -1;
                }
#pragma unroll
                for (int k = 0; k < RMAX; ++k) rk[k] = ck[k] >= 0 ? atomicAdd(&cnt[ck[k]], 1) : 0;
                __builtin_amdgcn_wave_barrier();
                const int n_h = __hip_atomic_load(&cnt[hcl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                const int sur = n_h > cap ? n_h - cap : 0, fre = n_h < cap ? cap - n_h : 0;
                int is = sur, ifr = fre;
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) {
                    const int ts = __shfl_up(is, o, 32), tf = __shfl_up(ifr, o, 32);
                    if (hcl >= o) { is += ts; ifr += tf; }
                }
                const int sur0 = is - sur, fre0 = ifr - fre, nsur = __shfl(is, 31, 32);
#pragma unroll
                for (int k = 0; k < RMAX; ++k) {
                    const int so = __shfl(sur0, ck[k] >= 0 ? ck[k] : 0, 32);
                    if (ck[k] >= 0) {
                        const unsigned short i = (unsigned short)(k * 64 + lane);
                        if (rk[k] < cap) list[ck[k]][rk[k]] = i;
                        else ovf[so + rk[k] - cap] = i;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if (lane < 32) cnt[lane] = 0;         // (for the next deal; every lane has read its class's count)
                // lanes h and h + 32 walk the slots of class h (even / odd ones); a slot beyond the class's own elements takes
                // one of the surplus elements of the long classes, if any is left
                auto slot_element = [&](int t) -> int {
                    const int sl = 2 * t + half;
                    if (t >= R) return -1;
                    if (sl < n_h) return (int)list[hcl][sl];
                    const int f = fre0 + (sl - n_h);
                    return f < nsur ? (int)ovf[f] : -1;
                };
                int inext = slot_element(0);
                unsigned oq = locate(E0 + (inext >= 0 ? inext : 0));
                bool act = inext >= 0;
                T xc[D];
                request(oq, act, xc);
                for (int t = 0; t < R; ++t) {
                    double b[D][4];
                    int base = 0;
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        const int ws = eval_table<VAL>(g, d, (double)xc[d], nd.v[d], b[d]);
                        base += (ws - a[d]) * PT::stride(d);
                    }
                    if (!act) base = 0;
                    const bool act_c = act;
                    const int i_c = inext;
                    inext = slot_element(t + 1);
                    oq = locate(E0 + (inext >= 0 ? inext : 0));
                    act = inext >= 0;
                    request(oq, act, xc);
                    // window_sum<4> with the LDS reads as inline assembly, one (k2, k3) plane -- 16 ds_read_b64, 32 registers --
                    // at a time: left to the compiler the 256 reads of a window are hoisted and spill (450 registers to scratch
                    // in the first build of this kernel).  Same operations in the same order as window_sum<4>: identical bits.
                    // The k3 loop stays rolled (the plane's offsets are immediates on top of a base that advances by t3).
                    unsigned la = (unsigned)(size_t)(const __attribute__((address_space(3))) double *)pr_tile + (unsigned)base * 8u;
                    double sum = 0.0;
                    asm volatile("; tables ready %0 %1 %2 %3" :: "v"(b[0][0]), "v"(b[1][0]), "v"(b[2][0]), "v"(b[3][0]));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
                    for (int k3 = 0; k3 < 4; ++k3) {
                        const double b3 = k3 == 0 ? b[3][0] : (k3 == 1 ? b[3][1] : (k3 == 2 ? b[3][2] : b[3][3]));
                        double q = 0.0;
#pragma unroll
                        for (int k2 = 0; k2 < 4; ++k2) {
                            double c[4][4];
                            asm volatile("ds_read_b64 %0, %1 offset:%2 ; after %3" : "=v"(c[0][0]) : "v"(la), "n"((k2 * t2) * 8), "v"(q));
#pragma unroll
                            for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
                                for (int k0 = (k1 == 0 ? 1 : 0); k0 < 4; ++k0)
                                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(c[k1][k0]) : "v"(la), "n"((k0 + k1 * t1 + k2 * t2) * 8));
                            asm volatile("s_waitcnt lgkmcnt(0)"
                                         : "+v"(c[0][0]), "+v"(c[0][1]), "+v"(c[0][2]), "+v"(c[0][3]), "+v"(c[1][0]), "+v"(c[1][1]), "+v"(c[1][2]), "+v"(c[1][3]),
                                           "+v"(c[2][0]), "+v"(c[2][1]), "+v"(c[2][2]), "+v"(c[2][3]), "+v"(c[3][0]), "+v"(c[3][1]), "+v"(c[3][2]), "+v"(c[3][3])
                                         :: "memory");
                            double rr = 0.0;
#pragma unroll
                            for (int k1 = 0; k1 < 4; ++k1) {
                                double tt = c[k1][0] * b[0][0];
                                tt = fma(c[k1][1], b[0][1], tt);
                                tt = fma(c[k1][2], b[0][2], tt);
                                tt = fma(c[k1][3], b[0][3], tt);
                                rr = fma(tt, b[1][k1], rr);
                            }
                            q = fma(rr, b[2][k2], q);
                            __builtin_amdgcn_sched_barrier(0);      // (the multiply-adds of a plane stay in front of the next plane's reads)
                        }
                        sum = fma(q, b3, sum);
                        la += (unsigned)(t3 * 8);
                    }
                    // (the 64 results of a round belong to ~60 different lines of the images: stored from here they cost 1.2-1.4
                    //  of the pass's 4.5 ms per 1e8 queries, plain or non-temporal; they go through the wave's LDS instead)
                    if (act_c) res[i_c] = sum;
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < RMAX; ++k)
                    if (ck[k] >= 0) __builtin_nontemporal_store((T)res[k * 64 + lane], outs + oqk[k]);
                __builtin_amdgcn_wave_barrier();     // (the lists are rewritten by the next deal)
            }
        }
        __syncthreads();                             // every wave is done with the tile
        if (tid < 64) {
            // the next region: the one with the most chunks left per workgroup that would then be there
            unsigned long long key = 0;
            for (int q0 = 0; q0 < nbins; q0 += 64) {
                const int q = q0 + lane;
                if (q < nbins) {
                    const int taken = __hip_atomic_load(queue + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int there = __hip_atomic_load(queue + nbins + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int left = (has_edge(q) ? 2 * nch : nch) - taken;
                    const float score = left > 0 ? (float)left / (float)(there + 1) : 0.0f;
                    const unsigned long long k = ((unsigned long long)__float_as_uint(score) << 32) | (unsigned)q;
                    key = k > key ? k : key;
                }
            }
#pragma unroll
            for (int sft = 32; sft > 0; sft >>= 1) {
                const unsigned long long o = __shfl_xor(key, sft);
                key = o > key ? o : key;
            }
            if (lane == 0) {
                const int rn = (key >> 32) != 0 ? (int)(key & 0xffffffffu) : -1;
                atomicSub(queue + nbins + r, 1);
                if (rn >= 0) atomicAdd(queue + nbins + rn, 1);
                s_region = rn;
            }
        }
        __syncthreads();
    }
}

// results of a place-pass workgroup's queries, from the sorted order of its image back to the caller's order: staged through
// LDS, so that both the read and the write are consecutive
template <typename T>
__global__ void __launch_bounds__(PR_NT)
pr_unsort_kernel(long long nq, const unsigned short *__restrict__ sidx, const int *__restrict__ starts, int nb1, const T *__restrict__ outs,
                 T *__restrict__ out)
{
    __shared__ T lv[PR_Q];
    const long long base = (long long)blockIdx.x * PR_Q;
    const int total = starts[(long long)blockIdx.x * nb1 + nb1 - 1];
    for (int p = threadIdx.x; p < total; p += PR_NT) lv[sidx[base + p]] = __builtin_nontemporal_load(outs + base + p);
    __syncthreads();
    const long long left = nq - base;
    const int n = left < PR_Q ? (int)left : PR_Q;
    // (every query of the workgroup has a region: total == n)
    for (int j = threadIdx.x; j < n; j += PR_NT) __builtin_nontemporal_store(lv[j], out + base + j);
}

// pass C of the fused value / gradient / Hessian evaluation (defined with eval_derivs_kernel below)
template <int D, int ORDER>
__global__ void eval_derivs_binned_kernel(Grid g, Regions rg, const double *__restrict__ coef, const double *__restrict__ xs,
                                          const int *__restrict__ off,
                                          const int *__restrict__ wgoff, double *__restrict__ out, int ldout);

// scratch of the binned path: per thread, grown on demand, released by splpak_shutdown
namespace {
struct EvalScratch {
    double *xs = nullptr;         // sorted records of a chunk: (capd + 1) doubles each
    int *ints = nullptr;          // hist | off | cursor | wgoff
    int *cnt = nullptr;           // per-workgroup region counts / run bases
    long long cap = 0;            // queries per chunk the buffers hold
    long long cnt_ints = 0;       // ints allocated for cnt (count matrix: workgroups of a chunk x regions of the grid)
    int capd = 0;
    hipEvent_t last = nullptr;    // end of the previous use (another stream must wait for it)
    int dev = -1;
};
struct RunScratch {               // run path: per-workgroup sorted images of a chunk, starts of their runs
    double *img = nullptr;
    int *starts = nullptr;
    long long img_doubles = 0, start_ints = 0;
    hipEvent_t last = nullptr;
    int dev = -1;
};
thread_local EvalScratch g_scratch;
thread_local RunScratch g_runs;
thread_local int g_eval_mode = 0;             // 0 auto, 1 direct, 2 binned
thread_local long long g_eval_chunk = 0;      // queries per chunk, 0 = default
}  // namespace

static void run_scratch_shutdown()
{
    RunScratch &s = g_runs;
    if (s.img) (void)hipFree(s.img);
    if (s.starts) (void)hipFree(s.starts);
    if (s.last) (void)hipEventDestroy(s.last);
    s = RunScratch();
}

static void pscratch_shutdown();

void eval_scratch_shutdown()
{
    run_scratch_shutdown();
    pscratch_shutdown();
    EvalScratch &s = g_scratch;
    if (s.xs) (void)hipFree(s.xs);
    if (s.ints) (void)hipFree(s.ints);
    if (s.cnt) (void)hipFree(s.cnt);
    if (s.last) (void)hipEventDestroy(s.last);
    s = EvalScratch();
}

void set_eval_mode(int mode, long long chunk)
{
    g_eval_mode = mode;
    g_eval_chunk = chunk;
}

// persistent region path: regions of the grid, or false when it does not apply (more than 64 regions / tiles beyond 64 KB)
template <int D>
static bool make_pregions(const Grid &g, PRegions &rg, int sper)
{
    rg.nbins = 1;
    rg.tcells = 1;
    for (int d = 0; d < MAXD; ++d) { rg.nreg[d] = 1; rg.sper[d] = 1; rg.text[d] = 1; rg.tstr[d] = 0; }
    for (int d = 0; d < D; ++d) {
        const int S = g.nodes[d] - 3;                // window starts 0 .. nodes - 4
        if (g.nodes[d] < 8 || S > 256) return false;
        rg.sper[d] = sper;
        rg.nreg[d] = (S + sper - 1) / sper;
        rg.text[d] = sper + 3;
        rg.nbins *= rg.nreg[d];
        rg.tcells *= rg.text[d];
    }
    if (rg.nbins < 8 || rg.nbins > PRCfg<D>::MAXBINS) return false;
    rg.deal = splpak::opt_get("SPLPAK_PR_NODEAL") ? 0 : 1;
    // (4-D: the runs of a place-pass workgroup hold ~16 queries for 32 slot classes -- nothing to deal; SPLPAK_PR_DEAL4=1 deals anyway)
    if (D == 4 && !splpak::opt_get("SPLPAK_PR_DEAL4")) rg.deal = 0;
    // what the place pass needs to know of the evaluation pass's LDS tile: its strides and the distance of its second copy
    if constexpr (D == 4) {
        if (sper != 8) return false;                 // (a tile of 19^4 coefficients is 1 MB)
        using PT = PTile<D, 8>;
        for (int d = 0; d < D; ++d) rg.tstr[d] = PT::stride(d);
        rg.telems = 0;
    } else if (sper == 16) { using PT = PTile<D, 16>; for (int d = 0; d < D; ++d) rg.tstr[d] = PT::stride(d); rg.telems = PT::W128 ? PT::COPY : 0; }
    else { using PT = PTile<D, 8>; for (int d = 0; d < D; ++d) rg.tstr[d] = PT::stride(d); rg.telems = PT::W128 ? PT::COPY : 0; }
    return true;
}

namespace {
struct PScratch {
    unsigned short *sidx = nullptr;
    unsigned char *scls = nullptr;                // (4-D) LDS slot class of every sorted query
    int *starts = nullptr, *claim = nullptr;
    void *xs = nullptr, *outs = nullptr;          // sorted coordinate planes [workgroup][D][PR_Q], sorted results
    long long cap_q = 0, cap_st = 0, cap_xs = 0;
    hipEvent_t last = nullptr;
    int dev = -1, ncu = 0;
};
thread_local PScratch g_pscratch;
}  // namespace

static void pscratch_shutdown()
{
    PScratch &s = g_pscratch;
    if (s.sidx) (void)hipFree(s.sidx);
    if (s.scls) (void)hipFree(s.scls);
    if (s.starts) (void)hipFree(s.starts);
    if (s.claim) (void)hipFree(s.claim);
    if (s.xs) (void)hipFree(s.xs);
    if (s.outs) (void)hipFree(s.outs);
    if (s.last) (void)hipEventDestroy(s.last);
    s = PScratch();
}

// persistent region path (see pr_place_kernel); hipErrorNotSupported = not for this grid: take the run path
template <int D, typename T>
static hipError_t eval_persistent(const Grid &g, long long nq, const T *xq, int ldxq, const NDeriv &nd, const T *coef, T *out, hipStream_t st)
{
    if (splpak::opt_get("SPLPAK_EVAL_NO_PERSISTENT")) return hipErrorNotSupported;
    // regions of 16 window starts per dimension (tiles of 19^3 = 55 KB: 64^3 nodes give 4 x 4 x 4 regions), of 8 for smaller grids
    PRegions rg;
    int sper = D == 4 ? 8 : 16;
    if (!make_pregions<D>(g, rg, sper)) {
        sper = 8;
        if (D == 4 || !make_pregions<D>(g, rg, sper)) return hipErrorNotSupported;
    }
    const long long nwg_ll = (nq + PR_Q - 1) / PR_Q;
    if (nwg_ll > 0x3fffffffLL / (2 * rg.nbins + 1)) return hipErrorNotSupported;
    const int nwg = (int)nwg_ll;
    if (nwg_ll * PR_Q > 0xffffffffLL) return hipErrorNotSupported;          // (32-bit query indices in the evaluation kernel)
    const long long need_q = nwg_ll * PR_Q, need_st = nwg_ll * (2 * rg.nbins + 1);
    PScratch &s = g_pscratch;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const long long need_xs = need_q * (D + 1) * (long long)sizeof(T);
    if (s.dev != dev || s.cap_q < need_q || s.cap_st < need_st || s.cap_xs < need_xs) {
        pscratch_shutdown();
        hipError_t e = hipMalloc(&s.sidx, sizeof(unsigned short) * (size_t)need_q);
        if (e == hipSuccess) e = hipMalloc(&s.scls, (size_t)need_q + 16);
        if (e == hipSuccess) e = hipMalloc(&s.starts, sizeof(int) * (size_t)need_st);
        if (e == hipSuccess) e = hipMalloc(&s.xs, sizeof(T) * (size_t)need_q * D);
        if (e == hipSuccess) e = hipMalloc(&s.outs, sizeof(T) * (size_t)need_q);
        if (e == hipSuccess) e = hipMalloc(&s.claim, sizeof(int) * 2 * PR_MAXBINS);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.last, hipEventDisableTiming);
        if (e != hipSuccess) { pscratch_shutdown(); (void)hipGetLastError(); return hipErrorNotSupported; }      // (no room: the other paths need less)
        s.cap_q = need_q;
        s.cap_st = need_st;
        s.cap_xs = need_xs;
        s.dev = dev;
    } else
        (void)hipStreamWaitEvent(st, s.last, 0);
    bool value_only = true;
    for (int d = 0; d < D; ++d) value_only = value_only && nd.v[d] == 0;
    hipError_t e = hipMemsetAsync(s.claim, 0, sizeof(int) * 2 * PR_MAXBINS, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((pr_place_kernel<D, T>), dim3((unsigned)nwg), dim3(PR_NT), 0, st, g, rg, nq, xq, ldxq, s.sidx, s.starts, (T *)s.xs, s.scls);
    // persistent workers: one workgroup per CU
    if (s.ncu <= 0) {
        int v = 0;
        s.ncu = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    const int ncu = s.ncu;
    const unsigned nworkers = (unsigned)ncu;
    // (chunks of 4 place-pass workgroups of interior runs, 16 of boundary runs: ~400 queries; the waves of a workgroup take
    //  them from a counter in LDS, so small chunks cost nothing and keep the tails short)
    static const int c0_env = splpak::opt_get("SPLPAK_PR_C0") ? atoi(splpak::opt_get("SPLPAK_PR_C0")) : 0;
    const dim3 grid(nworkers);
    if constexpr (D == 4) {
        // chunks of 32 place-pass workgroups of one bin: ~500 queries (a chunk's runs sit one per lane: at most 64)
        const int c0 = c0_env > 0 ? (c0_env < 64 ? c0_env : 64) : 32;
        if (value_only)
            hipLaunchKernelGGL((pr_eval4_kernel<8, true, T>), grid, dim3(PR4_EW), 0, st, g, rg, nd, coef, (const T *)s.xs, (const unsigned char *)s.scls,
                               (const int *)s.starts, nwg, c0, s.claim, (T *)s.outs);
        else
            hipLaunchKernelGGL((pr_eval4_kernel<8, false, T>), grid, dim3(PR4_EW), 0, st, g, rg, nd, coef, (const T *)s.xs, (const unsigned char *)s.scls,
                               (const int *)s.starts, nwg, c0, s.claim, (T *)s.outs);
    } else {
        const int c0 = c0_env > 0 ? c0_env : 4;
#define PR_GO(SP, VL)                                                                                                                        \
    hipLaunchKernelGGL((pr_eval_kernel<D, SP, VL, T>), grid, dim3(PR_EW), 0, st, g, rg, nd, coef, (const T *)s.xs, (const int *)s.starts, nwg, \
                       c0, s.claim, (T *)s.outs)
        if (sper == 16) { if (value_only) PR_GO(16, true); else PR_GO(16, false); }
        else { if (value_only) PR_GO(8, true); else PR_GO(8, false); }
#undef PR_GO
    }
    hipLaunchKernelGGL((pr_unsort_kernel<T>), dim3((unsigned)nwg), dim3(PR_NT), 0, st, nq, (const unsigned short *)s.sidx, (const int *)s.starts,
                       2 * rg.nbins + 1, (const T *)s.outs, out);
    (void)hipEventRecord(s.last, st);
    return hipGetLastError();
}

// run path (see run_place_kernel); hipErrorNotSupported = not for this grid / batch, take the region sort
template <int D, typename T>
static hipError_t eval_runs(const Grid &g, const Regions &rg, long long nq, const T *xq, int ldxq, const NDeriv &nd,
                            const T *coef, T *out, hipStream_t st)
{
    // (runs of RUN_QPW / nbins records: 16 at 64^3, 3 at 4-D 32^4; with more regions than that the evaluation pass would
    // gather single records and its groups outgrow the prefix)
    // Measured at 4-D 32^4 (648 regions, runs of 3 records = one 128-byte line): place 0.30 ms instead of count + prefixes +
    // place 0.70, but the evaluation pass 1.30 instead of 0.82 ms (fragments, a 10-step search per record, twice with the class
    // dealing) -- 1.09 against 1.18e10 evals/s: the sort stays for grids of more than 256 regions.
    static const int max_bins = splpak::opt_get("SPLPAK_EVAL_RUNS_MAXBINS") ? atoi(splpak::opt_get("SPLPAK_EVAL_RUNS_MAXBINS")) : 256;
    if (rg.nbins > RUN_GROUP_MAX || rg.nbins > max_bins || splpak::opt_get("SPLPAK_EVAL_SORT")) return hipErrorNotSupported;
    // place-pass workgroups per evaluation workgroup: ~1 950 queries of a region (two rounds of 1 024 threads; the 4-D
    // class dealing works in batches of 2 048)
    int grp = (int)(0.95 * rg.nbins + 0.5);
    if (D == 3 && grp < 128) grp = 128;
    if (grp < 32) grp = 32;
    if (grp > RUN_GROUP_MAX) grp = RUN_GROUP_MAX;
    long long chunk = g_eval_chunk > 0 ? g_eval_chunk : (1LL << 24);
    if (chunk > (1LL << 26)) chunk = 1LL << 26;
    if (chunk > nq) chunk = nq;
    const long long nwg_max = (chunk + RUN_QPW - 1) / RUN_QPW;
    const long long need_img = nwg_max * RUN_QPW * (D + 1), need_st = nwg_max * (rg.nbins + 1);
    RunScratch &s = g_runs;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (s.dev != dev || s.img_doubles < need_img || s.start_ints < need_st) {
        run_scratch_shutdown();
        hipError_t e = hipMalloc(&s.img, sizeof(double) * (size_t)need_img);
        if (e != hipSuccess && release_cached_plan_for_memory()) {
            (void)hipGetLastError();
            e = hipMalloc(&s.img, sizeof(double) * (size_t)need_img);
        }
        if (e == hipSuccess) e = hipMalloc(&s.starts, sizeof(int) * (size_t)need_st);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.last, hipEventDisableTiming);
        if (e != hipSuccess) { run_scratch_shutdown(); (void)hipGetLastError(); return hipErrorOutOfMemory; }
        s.img_doubles = need_img;
        s.start_ints = need_st;
        s.dev = dev;
    } else {
        (void)hipStreamWaitEvent(st, s.last, 0);
    }
    bool value_only = true;
    for (int d = 0; d < D; ++d) value_only = value_only && nd.v[d] == 0;
    for (long long c0 = 0; c0 < nq; c0 += chunk) {
        const int n = (int)(nq - c0 < chunk ? nq - c0 : chunk);
        const T *xc = xq + c0 * ldxq;
        const unsigned nwg = (unsigned)((n + RUN_QPW - 1) / RUN_QPW);
        hipLaunchKernelGGL((run_place_kernel<D, T>), dim3(nwg), dim3(RUN_NT), sizeof(int) * (2 * rg.nbins + 1), st, g, rg, n, xc, ldxq,
                           s.img, s.starts);
        const unsigned ngroups = (nwg + (unsigned)grp - 1) / (unsigned)grp;
        if (value_only)
            hipLaunchKernelGGL((eval_runs_kernel<D, true, T>), dim3(ngroups * (unsigned)rg.nbins), dim3(EVAL_WG), 0, st, g, rg, nd, coef,
                               (const double *)s.img, (const int *)s.starts, (int)nwg, grp, out + c0);
        else
            hipLaunchKernelGGL((eval_runs_kernel<D, false, T>), dim3(ngroups * (unsigned)rg.nbins), dim3(EVAL_WG), 0, st, g, rg, nd, coef,
                               (const double *)s.img, (const int *)s.starts, (int)nwg, grp, out + c0);
    }
    (void)hipEventRecord(s.last, st);
    return hipGetLastError();
}

// order == 0: one nderiv pattern (nd) -> out[nq]; order 1 / 2: value + gradient (+ Hessian) -> out[nq][ldout]
template <int D, typename T = double>
static hipError_t eval_binned(const Grid &g, const Regions &rg, long long nq, const T *xq, int ldxq,
                              const NDeriv &nd, const T *coef, T *out, hipStream_t st,
                              int order = 0, int ldout = 1)
{
    if constexpr (D == 3 || D == 4) {
        if (order == 0) {
            const hipError_t e = eval_persistent<D, T>(g, nq, xq, ldxq, nd, coef, out, st);
            if (e != hipErrorNotSupported) return e;
        }
    }
    if constexpr (D >= 3) {
        if (order == 0) {
            const hipError_t e = eval_runs<D, T>(g, rg, nq, xq, ldxq, nd, coef, out, st);
            if (e != hipErrorNotSupported) return e;
        }
    }
    // default chunk: 2^24 queries (measured best at 64^3: large enough that the ~8 000 evaluation
    // workgroups of a chunk keep every CU full to the end; chunks small enough to stay in the Infinity
    // Cache were not faster -- the passes are bound by instructions, not by HBM)
    long long chunk = g_eval_chunk > 0 ? g_eval_chunk : (1LL << 24);
    bool value_only = true;
    for (int d = 0; d < D; ++d) value_only = value_only && nd.v[d] == 0;
    if (chunk > (1LL << 28)) chunk = 1LL << 28;
    if (chunk > nq) chunk = nq;
    EvalScratch &s = g_scratch;
    int dev = 0;
    (void)hipGetDevice(&dev);
    // row length of the count matrix for THIS chunk and dimension count, and what it needs for THIS grid's regions:
    // the scratch is regrown when a later grid has more regions than the one it was sized for (round-2 advice:
    // a 40^3 spline followed by a 64^3 one wrote past the allocation)
    const int ldw = (int)(chunk / (256 * ScatterShape<D>::QPT) + 2);
    const long long cnt_need = (long long)ldw * rg.nbins + (long long)(ldw / BIN_ROWS + 2) * rg.nbins;     // count matrix + chunk sums
    if (s.dev != dev || s.cap < chunk || s.capd < D || s.cnt_ints < cnt_need) {
        eval_scratch_shutdown();
        hipError_t e = hipMalloc(&s.xs, sizeof(double) * (size_t)chunk * (D + 1));
        if (e != hipSuccess && release_cached_plan_for_memory()) {      // the one-shot fit's cached plan is in the way
            (void)hipGetLastError();
            e = hipMalloc(&s.xs, sizeof(double) * (size_t)chunk * (D + 1));
        }
        if (e == hipSuccess) e = hipMalloc(&s.ints, sizeof(int) * (4 * BIN_MAX + 8));
        // per-workgroup region counts of pass A -> run bases of pass B: [workgroups of a chunk][regions]
        if (e == hipSuccess) e = hipMalloc(&s.cnt, sizeof(int) * (size_t)cnt_need);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.last, hipEventDisableTiming);
        if (e != hipSuccess) { eval_scratch_shutdown(); return e; }
        s.cap = chunk;
        s.cnt_ints = cnt_need;
        s.capd = D;
        s.dev = dev;
    } else {
        (void)hipStreamWaitEvent(st, s.last, 0);
    }
    int *hist = s.ints, *off = hist + BIN_MAX, *cursor = off + BIN_MAX + 1, *wgoff = cursor + BIN_MAX;
    for (long long c0 = 0; c0 < nq; c0 += chunk) {
        const int n = (int)(nq - c0 < chunk ? nq - c0 : chunk);
        const T *xc = xq + c0 * ldxq;
        const unsigned nbs = (unsigned)((n + 256 * ScatterShape<D>::QPT - 1) / (256 * ScatterShape<D>::QPT));
        hipLaunchKernelGGL((bin_count_kernel<D, T>), dim3(nbs), dim3(256), 0, st, g, rg, n, xc, ldxq, s.cnt, ldw);
        int *part = s.cnt + (long long)ldw * rg.nbins;
        const unsigned nchunk = (nbs + BIN_ROWS - 1) / BIN_ROWS, nbg = (unsigned)((rg.nbins + 255) / 256);
        hipLaunchKernelGGL(bin_colsum_kernel, dim3(nbg, nchunk), dim3(256), 0, st, (int)nbs, rg.nbins, (const int *)s.cnt, part);
        hipLaunchKernelGGL(bin_total_kernel, dim3(nbg), dim3(256), 0, st, (int)nchunk, rg.nbins, part, hist);
        hipLaunchKernelGGL(bin_scan_kernel, dim3(1), dim3(256), 0, st, rg.nbins, (const int *)hist, off, cursor, wgoff);
        hipLaunchKernelGGL(bin_wgbase_kernel, dim3(nbg, nchunk), dim3(256), 0, st, (int)nbs, rg.nbins, (const int *)off, (const int *)part, s.cnt);
        hipLaunchKernelGGL((bin_scatter_kernel<D, T>), dim3(nbs), dim3(256), 2 * sizeof(int) * rg.nbins, st, g, rg, n, xc, ldxq,
                           (const int *)s.cnt, ldw, s.xs);
        const unsigned nw = (unsigned)(n / EVAL_QPW + rg.nbins + 1);
        if constexpr (sizeof(T) == 8) {
            if (order == 1)
                hipLaunchKernelGGL((eval_derivs_binned_kernel<D, 1>), dim3(nw), dim3(256), 0, st, g, rg, coef, (const double *)s.xs,
                                   (const int *)off, (const int *)wgoff, out + c0 * ldout, ldout);
            else if (order == 2)
                hipLaunchKernelGGL((eval_derivs_binned_kernel<D, 2>), dim3(nw), dim3(256), 0, st, g, rg, coef, (const double *)s.xs,
                                   (const int *)off, (const int *)wgoff, out + c0 * ldout, ldout);
        }
        if (order == 0 && value_only)
            hipLaunchKernelGGL((eval_binned_kernel<D, true, T>), dim3(nw), dim3(EVAL_WG), 0, st, g, rg, nd, coef,
                               (const double *)s.xs, (const int *)off, (const int *)wgoff, out + c0);
        else if (order == 0)
            hipLaunchKernelGGL((eval_binned_kernel<D, false, T>), dim3(nw), dim3(EVAL_WG), 0, st, g, rg, nd, coef,
                               (const double *)s.xs, (const int *)off, (const int *)wgoff, out + c0);
    }
    (void)hipEventRecord(s.last, st);
    return hipGetLastError();
}

// regions of the grid for dimension count D; false when the binned path does not apply
template <int D>
static bool make_regions(const Grid &g, Regions &rg)
{
    long long nb = 1;
    for (int d = 0; d < MAXD; ++d) rg.nreg[d] = 1;
    for (int d = 0; d < D; ++d) {
        const int R = TileShape<D>::T[d] - 3;
        rg.nreg[d] = (g.nodes[d] - 3 + R - 1) / R;
        nb *= rg.nreg[d];
    }
    rg.nbins = (int)nb;
    return nb >= 1 && nb <= BIN_MAX;
}

// ---- fused value + gradient (+ Hessian) ------------------------------------------------------------
// all derivative patterns of total order <= ORDER from ONE pass over the window, instead
// of one splde call (:1089-1240) per pattern.  Output per query, ldout apart:
//   [ f, df/dx_1 .. df/dx_D, (ORDER 2:) d2f/dx_1dx_1, d2f/dx_1dx_2, .., d2f/dx_1dx_D, d2f/dx_2dx_2, .. ]
// Each entry is the reference's sum  sum_window coef * prod_d bas1(nderiv_d; x_d)  for its nderiv
// pattern; the 1-D factors come from the same window_table as everywhere else.
// acc[*] for one query from its factor tables b[a][d][k] (a = derivative order) and a loader of window
// rows: load4(k, c) delivers the 4 coefficients (k_0 = 0..3) of the row with window indices k[1..D-1].
// Shared by the direct and the binned kernel: identical bits.
template <int D, int ORDER, typename L4>
__device__ inline void derivs_accumulate(const double (&b)[ORDER + 1][D][4], L4 &&load4,
                                         double (&acc)[1 + D + (ORDER == 2 ? D * (D + 1) / 2 : 0)])
{
    constexpr int NOUT = 1 + D + (ORDER == 2 ? D * (D + 1) / 2 : 0);
#pragma unroll
    for (int j = 0; j < NOUT; ++j) acc[j] = 0.0;
    // window rows (k_0 = 0..3 contiguous): contract dimension 1 with its value / first / second
    // derivative factors first, then combine with the factors of the other dimensions
    constexpr int NROW = D == 1 ? 1 : (D == 2 ? 4 : (D == 3 ? 16 : 64));
    for (int e = 0; e < NROW; ++e) {
        int k[D];
        k[0] = 0;
#pragma unroll
        for (int d = 1; d < D; ++d) k[d] = (e >> (2 * (d - 1))) & 3;
        double c[4];
        load4(k, c);
        double r[ORDER + 1];                  // r[a] = sum_k0 c[k0] * (a-th derivative factor of dim 1)
#pragma unroll
        for (int a = 0; a <= ORDER; ++a) {
            double t = 0.0;
#pragma unroll
            for (int k0 = 0; k0 < 4; ++k0) t = fma(c[k0], b[a][0][k0], t);
            r[a] = t;
        }
        double v0[D], v1[D], pex[D];          // dims >= 1: pex[d] = prod_{f >= 1, f != d} v0[f]
        double full = 1.0;                    // prod_{f >= 1} v0[f]
        v0[0] = v1[0] = pex[0] = 1.0;
#pragma unroll
        for (int d = 1; d < D; ++d) {
            v0[d] = b[0][d][k[d]];
            v1[d] = b[1][d][k[d]];
            full *= v0[d];
        }
#pragma unroll
        for (int d = 1; d < D; ++d) {
            double pd = 1.0;
#pragma unroll
            for (int f = 1; f < D; ++f)
                if (f != d) pd *= v0[f];
            pex[d] = pd;
        }
        acc[0] = fma(r[0], full, acc[0]);
        acc[1] = fma(r[1], full, acc[1]);
#pragma unroll
        for (int d = 1; d < D; ++d) acc[1 + d] = fma(r[0], v1[d] * pex[d], acc[1 + d]);
        if constexpr (ORDER == 2) {
            int j = 1 + D;
#pragma unroll
            for (int d = 0; d < D; ++d)
#pragma unroll
                for (int f = d; f < D; ++f) {
                    double term;
                    if (d == 0 && f == 0) {
                        term = r[2] * full;
                    } else if (d == 0) {
                        term = r[1] * (v1[f] * pex[f]);
                    } else if (f == d) {
                        term = r[0] * (b[2][d][k[d]] * pex[d]);
                    } else {
                        double pdf = 1.0;
#pragma unroll
                        for (int h = 1; h < D; ++h)
                            if (h != d && h != f) pdf *= v0[h];
                        term = r[0] * (v1[d] * v1[f] * pdf);
                    }
                    acc[j] += term;
                    ++j;
                }
        }
    }
}

template <int D, int ORDER, typename T>
__global__ void __launch_bounds__(256)
eval_derivs_kernel(Grid g, long long nq, const T *__restrict__ xq, int ldxq, const T *__restrict__ coef,
                   T *__restrict__ out, int ldout)
{
    constexpr int NOUT = 1 + D + (ORDER == 2 ? D * (D + 1) / 2 : 0);
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += stride) {
        double b[ORDER + 1][D][4];
        int base = 0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const double x = (double)xq[i * ldxq + d];
            int ws = 0;
#pragma unroll
            for (int a = 0; a <= ORDER; ++a) ws = window_table(g, d, x, a, b[a][d]);
            base += ws * g.colstride[d];
        }
        double acc[NOUT];
        derivs_accumulate<D, ORDER>(b, [&](const int (&k)[D], double (&c)[4]) {
            int off = 0;
#pragma unroll
            for (int d = 1; d < D; ++d) off += k[d] * g.colstride[d];
            if constexpr (sizeof(T) == 8) {
                typedef double d2v __attribute__((ext_vector_type(2), aligned(8)));
                const d2v lo2 = *reinterpret_cast<const d2v *>(coef + base + off);
                const d2v hi2 = *reinterpret_cast<const d2v *>(coef + base + off + 2);
                c[0] = lo2[0]; c[1] = lo2[1]; c[2] = hi2[0]; c[3] = hi2[1];
            } else {
#pragma unroll
                for (int k0 = 0; k0 < 4; ++k0) c[k0] = (double)coef[base + off + k0];
            }
        }, acc);
#pragma unroll
        for (int j = 0; j < NOUT; ++j) out[i * ldout + j] = (T)acc[j];
    }
}

// binned form (pass C of the region sort, see eval_binned_kernel): the window rows come from the LDS tile
template <int D, int ORDER>
__global__ void __launch_bounds__(256)
eval_derivs_binned_kernel(Grid g, Regions rg, const double *__restrict__ coef, const double *__restrict__ xs,
                          const int *__restrict__ off,
                          const int *__restrict__ wgoff, double *__restrict__ out, int ldout)
{
    constexpr int NOUT = 1 + D + (ORDER == 2 ? D * (D + 1) / 2 : 0);
    constexpr int TILE_ELEMS = tile_cells<D>();          // (dense strides here: the padded ones belong to eval_binned_kernel)
    __shared__ double tile[TILE_ELEMS];
    using TS = TileShape<D>;
    const int wg = blockIdx.x;
    if (wg >= wgoff[rg.nbins]) return;
    int lo = 0, hi = rg.nbins;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (wgoff[mid] <= wg) lo = mid; else hi = mid;
    }
    const int r = lo, part = wg - wgoff[r];
    int a[D];
    {
        int rr = r;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            a[d] = (rr % rg.nreg[d]) * (TS::T[d] - 3);
            rr /= rg.nreg[d];
        }
    }
    for (int e = threadIdx.x; e < TILE_ELEMS; e += 256) {
        int rem = e, idx = 0;
        bool ok = true;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int l = rem % TS::T[d];
            rem /= TS::T[d];
            const int node = a[d] + l;
            ok = ok && node < g.nodes[d];
            idx += node * g.colstride[d];
        }
        tile[e] = ok ? coef[idx] : 0.0;
    }
    __syncthreads();
    const int qb = off[r] + part * EVAL_QPW;
    const int qe = min(off[r + 1], qb + EVAL_QPW);
    int tstr[D];
    {
        int m = 1;
#pragma unroll
        for (int d = 0; d < D; ++d) { tstr[d] = m; m *= TS::T[d]; }
    }
    for (int j = qb + threadIdx.x; j < qe; j += 256) {
        double b[ORDER + 1][D][4];
        int base = 0;
        double xr[D];
        const long long p = load_record<D>(xs + (long long)j * (D + 1), xr);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const double x = xr[d];
            int ws = 0;
#pragma unroll
            for (int aa = 0; aa <= ORDER; ++aa) ws = window_table(g, d, x, aa, b[aa][d]);
            base += (ws - a[d]) * tstr[d];
        }
        double acc[NOUT];
        derivs_accumulate<D, ORDER>(b, [&](const int (&k)[D], double (&c)[4]) {
            int o = base;
#pragma unroll
            for (int d = 1; d < D; ++d) o += k[d] * tstr[d];
            typedef const volatile __attribute__((address_space(3))) double *lds_cvd;
            lds_cvd q = (lds_cvd)tile + o;
            c[0] = q[0]; c[1] = q[1]; c[2] = q[2]; c[3] = q[3];
        }, acc);
#pragma unroll
        for (int jj = 0; jj < NOUT; ++jj) out[p * ldout + jj] = acc[jj];
    }
}

template <typename T>
static hipError_t launch_eval_derivs_t(const Grid &g, long long nq, const T *xq, int ldxq, int order,
                                       const T *coef, T *out, int ldout, hipStream_t st)
{
    if (nq <= 0) return hipSuccess;
    if constexpr (sizeof(T) == 8) {
        // same rule as the single-pattern evaluation: large batches on 3-D / 4-D grids go through the region sort
        Regions rg;
        bool can = false;
        if (g.ndim == 2) can = make_regions<2>(g, rg);
        if (g.ndim == 3) can = make_regions<3>(g, rg);
        if (g.ndim == 4) can = make_regions<4>(g, rg);
        const bool want = g_eval_mode == 2 || (g_eval_mode == 0 && g.ndim >= 3 && nq >= (1LL << 20) && g.ncol > 32768);
        if (can && want) {
            NDeriv nd0{};
            hipError_t e = g.ndim == 2   ? eval_binned<2>(g, rg, nq, xq, ldxq, nd0, coef, out, st, order, ldout)
                           : g.ndim == 3 ? eval_binned<3>(g, rg, nq, xq, ldxq, nd0, coef, out, st, order, ldout)
                                         : eval_binned<4>(g, rg, nq, xq, ldxq, nd0, coef, out, st, order, ldout);
            if (e != hipErrorOutOfMemory) return e;
            (void)hipGetLastError();
        }
    }
    long long blocks = (nq + 255) / 256;
    if (blocks > 256LL * 32) blocks = 256LL * 32;
    dim3 gr((unsigned)blocks), bl(256);
#define SPLPAK_DERIVS(DD)                                                                                         \
    if (order == 1) hipLaunchKernelGGL((eval_derivs_kernel<DD, 1, T>), gr, bl, 0, st, g, nq, xq, ldxq, coef, out, ldout); \
    else hipLaunchKernelGGL((eval_derivs_kernel<DD, 2, T>), gr, bl, 0, st, g, nq, xq, ldxq, coef, out, ldout);
    switch (g.ndim) {
    case 1: SPLPAK_DERIVS(1) break;
    case 2: SPLPAK_DERIVS(2) break;
    case 3: SPLPAK_DERIVS(3) break;
    default: SPLPAK_DERIVS(4) break;
    }
#undef SPLPAK_DERIVS
    return hipGetLastError();
}

hipError_t launch_eval_derivs(const Grid &g, long long nq, const double *xq, int ldxq, int order,
                              const double *coef, double *out, int ldout, hipStream_t st)
{
    return launch_eval_derivs_t<double>(g, nq, xq, ldxq, order, coef, out, ldout, st);
}

hipError_t launch_eval_derivs_f32(const Grid &g, long long nq, const float *xq, int ldxq, int order,
                                  const float *coef, float *out, int ldout, hipStream_t st)
{
    return launch_eval_derivs_t<float>(g, nq, xq, ldxq, order, coef, out, ldout, st);
}

template <typename T>
static hipError_t launch_eval_t(const Grid &g, long long nq, const T *xq, int ldxq,
                                const int *nderiv, const T *coef, T *out, hipStream_t st)
{
    if (nq <= 0) return hipSuccess;
    NDeriv nd;
    for (int d = 0; d < MAXD; ++d) {
        int v = (nderiv && d < g.ndim) ? nderiv[d] : 0;
        nd.v[d] = v < 0 ? 0 : (v > 2 ? 2 : v);
    }
    {
        // binned path: large batches on grids whose coefficients are far beyond the L1 (auto), or forced (both storage kinds:
        // the REAL32 entry points widen their inputs in the sort passes and the tile fill, same arithmetic as their direct kernel)
        Regions rg;
        bool can = false;
        if (g.ndim == 2) can = make_regions<2>(g, rg);
        if (g.ndim == 3) can = make_regions<3>(g, rg);
        if (g.ndim == 4) can = make_regions<4>(g, rg);
        // auto: 2-D windows are 4 rows (4-5 L2 lines) and the direct kernel wins; from 3-D on (16+ rows)
        // the sort pays for itself once the batch is large and the coefficients are far beyond L1
        const bool want = g_eval_mode == 2 || (g_eval_mode == 0 && g.ndim >= 3 && nq >= (1LL << 20) && g.ncol > 32768);
        if (can && want) {
            hipError_t e = g.ndim == 2   ? eval_binned<2, T>(g, rg, nq, xq, ldxq, nd, coef, out, st)
                           : g.ndim == 3 ? eval_binned<3, T>(g, rg, nq, xq, ldxq, nd, coef, out, st)
                                         : eval_binned<4, T>(g, rg, nq, xq, ldxq, nd, coef, out, st);
            // no room for the sort scratch: the binned path is an optimisation, fall through to the direct one
            if (e != hipErrorOutOfMemory) return e;
            (void)hipGetLastError();
        }
    }
    const int threads = 256;
    long long blocks = (nq + threads - 1) / threads;
    if (blocks > 256LL * 32) blocks = 256LL * 32;   // grid-stride the rest
    dim3 gr((unsigned)blocks), bl(threads);
    bool value_only = true;
    for (int d = 0; d < g.ndim; ++d) value_only = value_only && nd.v[d] == 0;
#define SPLPAK_EVAL(DD)                                                                                              \
    if (value_only) hipLaunchKernelGGL((eval_kernel<DD, T, true>), gr, bl, 0, st, g, nq, xq, ldxq, nd, coef, out);   \
    else hipLaunchKernelGGL((eval_kernel<DD, T, false>), gr, bl, 0, st, g, nq, xq, ldxq, nd, coef, out);
    switch (g.ndim) {
    case 1: SPLPAK_EVAL(1) break;
    case 2: SPLPAK_EVAL(2) break;
    case 3: SPLPAK_EVAL(3) break;
    default: SPLPAK_EVAL(4) break;
    }
#undef SPLPAK_EVAL
    return hipGetLastError();
}

hipError_t launch_eval(const Grid &g, long long nq, const double *xq, int ldxq, const int *nderiv,
                       const double *coef, double *out, hipStream_t st)
{
    return launch_eval_t<double>(g, nq, xq, ldxq, nderiv, coef, out, st);
}

hipError_t launch_eval_f32(const Grid &g, long long nq, const float *xq, int ldxq, const int *nderiv,
                           const float *coef, float *out, hipStream_t st)
{
    return launch_eval_t<float>(g, nq, xq, ldxq, nderiv, coef, out, st);
}

}  // namespace splpak
