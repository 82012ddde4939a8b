// Host-side launchers of the HIP kernels (implemented in the .hip files).
#pragma once
#include <cstdlib>
#include "common.hpp"
#include <vector>

namespace splpak {

// after a failed allocation: release the plan / staging buffers the one-shot fit entry caches (plan.hip); true = retry
bool release_cached_plan_for_memory();

// ---- eval.hip
hipError_t launch_eval(const Grid &g, long long nq, const double *xq, int ldxq, const int *nderiv,
                       const double *coef, double *out, hipStream_t st);
// value + gradient (order 1) (+ Hessian upper triangle, order 2) per query, ldout apart
hipError_t launch_eval_derivs(const Grid &g, long long nq, const double *xq, int ldxq, int order,
                              const double *coef, double *out, int ldout, hipStream_t st);
hipError_t launch_eval_derivs_f32(const Grid &g, long long nq, const float *xq, int ldxq, int order,
                                  const float *coef, float *out, int ldout, hipStream_t st);
// evaluation path of the calling thread: 0 auto, 1 direct (global gathers), 2 binned (LDS tiles);
// chunk = queries sorted per pass of the binned path (0 = default)
void set_eval_mode(int mode, long long chunk);
void eval_scratch_shutdown();
hipError_t launch_eval_f32(const Grid &g, long long nq, const float *xq, int ldxq, const int *nderiv,
                           const float *coef, float *out, hipStream_t st);

// ---- synth.hip
hipError_t launch_synth_points(int ndim, long long first, long long n, double *x, double *y,
                               double *w, hipStream_t st);
hipError_t launch_synth_queries(int ndim, long long skip_draws, long long nq, double *xq,
                                hipStream_t st);

// ---- assemble.hip
// scalars (device doubles) written by the assembly kernels
enum { SC_TOTLWT = 0, SC_NROWS_DATA = 1, SC_NROWS_CONS = 2, SC_ERRFLAG = 3, SC_SUMW2 = 4 /* pcg.hip: sum of w^2 */, SC_COUNT = 8 };

struct SortScratch {
    int *key;        // [max_ndata] cell key per point (ncell = zero-weight sentinel)
    int *count;      // [ncell + 2]
    int *offset;     // [ncell + 2] exclusive scan of count
    int *cursor;     // [ncell + 1]
    int *scanpart;   // [256] segment sums of the scan
    double *xs;      // [ndim][cap] sorted coordinates, SoA (internal dimension order)
    double *ys;      // [cap]
    double *ws;      // [cap]
    int *idx;        // [cap] original index of the sorted points (orders the points inside a cell)
    long long cap;   // max_ndata
    // stable partition (round 5; sp_* kernels of assemble.hip): per-block bin counts / bases, bin bases, intermediate records
    int *cntm;       // [ceil(cap / SP_Q)][SP_NB]
    int *binbase;    // [SP_NB + 1]
    int *sppart;     // [chunks of 16 blocks][SP_NB]
    double *rec;     // [cap][ndim + 3] records sorted by tile (grids of more cells than bins; else NULL)
};
constexpr int SP_Q = 8192;         // points per block of the stable partition
constexpr int SP_NB = 4096;        // bins (the last one holds the zero-weight points)
// doubles of SortScratch::rec a grid needs (0: one level suffices)
long long bin_record_doubles(const Grid &g, long long max_ndata);

// binning: window keys + per-cell counts + scalars -> scan -> counting-sort scatter -> points of every
// cell ordered by original index
hipError_t launch_bin_points(const Grid &g, long long m, const double *x, int ldx, const double *y,
                             const double *w, const SortScratch &s, double *scal, hipStream_t st);
// doubles of scratch launch_gram uses when it can hold every cell's blocks at once (per-cell Gram blocks,
// right-hand sides, histogram shares), and the least it can work with (one hyper-row of cells)
long long gram_scratch_doubles(const Grid &g);
long long gram_scratch_min_doubles(const Grid &g);
// per-cell Gram blocks -> (owner gathers) half-stencil normal equations nst[ncol][hstencil], rhs[ncol] and,
// when smooth, the nearest-node histogram hist[ncol] (caller's order; must be zero on entry) + its total
hipError_t launch_gram(const Grid &g, const SortScratch &s, double *scratch, long long scratch_doubles, bool smooth,
                       double *nst, double *rhs, double *hist, double *scalH, hipStream_t st);
// derivative-constraint rows of the data-sparse nodes (:921-1046): nst += C^T C, rows counted into
// scal_out[SC_NROWS_CONS]
// dcw[node] / spf[node]: constraint weight xtrap (expect - have) and "data sparse" flag of every node (:923-960)
hipError_t launch_sparse_mark(const Grid &g, const double *hist, const double *scal, double xtrap, double *dcw,
                              unsigned char *spf, hipStream_t st);
// ctab: the per-dimension factors of the constraint-row entries, tabulated once per plan (launch_constraint_table;
// constraint_table_doubles entries); NULL = evaluate them in place
hipError_t launch_constraint_rows(const Grid &g, const double *dcw, const unsigned char *spf, const double *ctab, double *nst,
                                  double *scal_out, hipStream_t st);
// the two single-workgroup reductions of the assembly on their own (the rows-only assembly of pcg-only plans, rowsop.hip):
// scal[SC_TOTLWT] = sum of the histogram; scal_out[SC_NROWS_CONS] += rows of the data-sparse nodes
hipError_t launch_hist_total(const Grid &g, const double *hist, double *scal, hipStream_t st);
hipError_t launch_count_sparse(const Grid &g, const unsigned char *spf, double *scal_out, hipStream_t st);
long long constraint_table_doubles(const Grid &g);
hipError_t launch_constraint_table(const Grid &g, double *ctab, hipStream_t st);
// refinement residual rho = A^T W (W y - W A x) [- C^T C x when `constraints`]; rcell: [ncell][nb] scratch,
// tbuf: [ncol][ndim(ndim+1)/2] scratch; ssq != NULL: also the sum of squared row residuals, from the per-cell / per-node
// shares in e2buf ([ncell + ncol] scratch) added in a fixed order
hipError_t launch_sum_fixed(const double *v, long long n, double *out, hipStream_t st);      // out[0] = sum of v[0 .. n) in a fixed order
hipError_t launch_residual(const Grid &g, const SortScratch &s, const double *xvec, double *rcell,
                           const double *dcw, const unsigned char *spf, const double *ctab, bool constraints,
                           double *tbuf, double *rho, double *ssq, double *e2buf, hipStream_t st);
// out[0] = max_i |rho_i| / ((|N||x|)_i + |rhs_i|): componentwise backward error with respect to the rows -- the denominators
// den[i] first (they need the coefficients only: beside the residual pass, on another stream), then the maximum of the ratios
hipError_t launch_backward_denominators(const Grid &g, const double *nst, const double *xvec, const double *rhs, double *den, hipStream_t st);
hipError_t launch_backward_error(const Grid &g, const double *den, const double *rho, double *out, hipStream_t st);

// coef[reference column] = xvec[internal column] (a plain copy when the plan did not reorder the dimensions)
hipError_t launch_to_reference_order(const Grid &g, const double *xvec, double *coef, hipStream_t st);

// ---- bandchol.hip
constexpr int NBLK = 256;     // block size of the band factorisation

// Distribution of the block columns of the band over R ranks (GPUs): chunks of c consecutive block
// columns are dealt round-robin; a rank stores its own block columns packed.  R = 1: everything local.
struct DistMap {
    int R, r, c;
    long long ld;             // column stride of the band storage (lda + 1)
};
__host__ __device__ inline int dm_owner(const DistMap &m, int J) { return (J / m.c) % m.R; }
__host__ __device__ inline bool dm_owned(const DistMap &m, int J) { return dm_owner(m, J) == m.r; }
__host__ __device__ inline int dm_slot(const DistMap &m, int J) { return ((J / m.c) / m.R) * m.c + J % m.c; }
// A(i,j) of block column J = (ab + dm_shift(J))[i + j lda]
__host__ __device__ inline long long dm_shift(const DistMap &m, int J) { return 256LL * m.ld * (dm_slot(m, J) - J); }
// number of block columns a rank stores when nblk are dealt
inline int dm_local_blocks(const DistMap &m, int nblk)
{
    int n = 0;
    for (int J = 0; J < nblk; ++J) n += dm_owned(m, J) ? 1 : 0;
    return n;
}
struct Band {
    double *ab;       // dense-view base: A(i,j) = ab[i + j*lda], j <= i <= j + halfbw
    double *dinv;     // [nblk][NBLK*NBLK] inverses of the diagonal blocks of L (row-major)
    double *dinvt;    // [nblk][NBLK*NBLK] transposes of dinv (backward sweep)
    double *mfwd;     // [nblk-1][NBLK*NBLK] Linv_{k+1} L_{k+1,k} (row-major): forward-sweep coupling blocks; NULL = two-kernel sweeps
    double *mbwd;     // [nblk-1][NBLK*NBLK] Linv_k^T L_{k+1,k}^T: backward-sweep coupling blocks
    double *inv64;    // [nblk][16][16*16] inverses of the 16x16 diagonal leaves of L (column-major), 16 KB stride per block
    long long lda;    // column stride of the dense view (ld - 1)
    int n;            // logical order
    int npad;         // padded to a multiple of NBLK (identity on the padding)
    int nblk;         // npad / NBLK
    int bw;           // block half-bandwidth: ceil(halfbw / NBLK)
    size_t bytes;     // allocation size of ab
    mutable void *pipe = nullptr;   // look-ahead pipeline of band_cholesky (streams, events, queues): created on first
                                    // use, released by band_pipeline_destroy -- owned by whoever owns the Band
};
size_t band_bytes(int n, int halfbw, Band *desc);
struct DistMap;
// zero the band, put 1 on the padded diagonal, scatter the half-stencil into it (the block columns
// DistMap gives to this rank)
hipError_t launch_expand(const Grid &g, const double *nst, const Band &b, const DistMap &dm, hipStream_t st);

struct CholStats {            // optional per-kernel accounting (HIP events)
    bool enabled = false;
    double syrk_launches = 0, syrk_ms = 0, syrk_flop = 0, factor_ms = 0;   // timed bulk trailing-update launches
    double bulk_launches = 0, bulk_flop = 0;                               // all bulk launches
    double total_flop = 0;                                                 // all trailing-update launches
};
// in-place L L^T; *info_dev (device int) is set to 1 + column of the first
// non-positive pivot (0 = success); min pivot is tracked in minpiv_dev
// Only block columns [kbeg, kend) are eliminated (kend < 0: to the end), leaving the rest updated but unfactored;
// inverses / sweep coupling blocks are produced for the first nfinish block columns (< 0: all).
hipError_t band_cholesky(const Band &b, int *info_dev, double *minpiv_dev, hipStream_t st,
                         CholStats *stats, int kbeg = 0, int kend = -1, int nfinish = -1);
// Block half-bandwidth below which a factorisation is chain-bound and takes the narrow form (and, with enough
// block columns, both ends at once: twoend.hip).  Measured crossover on MI355X: 44^3 nodes (24 blocks) 112 ms
// two-ended against 131 ms, 48^3 (28 blocks) equal, 56^3 (38 blocks) 393 against 370 ms.  SPLPAK_PIN_BW (the
// threshold for pinning potrf in the four-stream pipeline) lowers it too, so that SPLPAK_PIN_BW=1 still forces
// that pipeline onto small grids.
inline int narrow_band_limit()
{
    if (const char *e = splpak::opt_get("SPLPAK_NARROW_BW")) return atoi(e);
    if (const char *e = splpak::opt_get("SPLPAK_PIN_BW")) return atoi(e);
    return 28;
}
// the same for narrow (chain-bound) bands: one extra stream, one update launch per step (see bandchol.hip)
hipError_t band_cholesky_narrow(const Band &b, int *info_dev, double *minpiv_dev, hipStream_t st, int kbeg = 0,
                                int kend = -1, int nfinish = -1);
// release a Band's pipeline (streams / events / queues); NULL is fine
void band_pipeline_destroy(void *pipe);
// x <- (L L^T)^{-1} x; x and tmp of length npad (padding entries of x must be 0)
hipError_t band_solve(const Band &b, double *x, double *tmp, hipStream_t st);
// The two sweeps of band_solve, by block ranges (twoend.hip).  Forward over blocks [kb, ke): y -> tmp, every row
// below a solved block (also those from ke on) is updated in x.  Backward: blocks >= kgiven already hold their
// solution in x (kgiven >= nblk: none); the blocks [kstop, kgiven) are solved from tmp into x.  resume = the
// blocks >= kgiven were solved by an earlier call that stopped at kstop = kgiven (their contributions to the
// right-hand sides are in place, except what the step that solves block kgiven - 1 applies itself).
hipError_t band_forward(const Band &b, double *x, double *tmp, int kb, int ke, hipStream_t st);
hipError_t band_backward(const Band &b, double *x, double *tmp, int kgiven, hipStream_t st, int kstop = 0,
                         bool resume = false);

// pieces of the distributed band factorisation / sweeps (driven by dist.hip)
hipError_t launch_potrf_block(double *abJ, long long lda, int k0, int *info, double *minpiv, double *inv16, hipStream_t st);
hipError_t launch_trsm_panel(const double *Lkk, double *X, long long lda, const double *inv16, int nrows, hipStream_t st);
hipError_t launch_pack_panel(const double *src, long long lda, double *dst, int nrows, hipStream_t st);
long long syrk64d_items(const DistMap &dm, int row0, int jb, int je, int re);
hipError_t launch_syrk64d(double *abl, long long lda, const DistMap &dm, const double *P, long long ldp, int row0,
                          int jb, int je, int re, hipStream_t st);
hipError_t launch_trtri_owned(const double *abl, long long lda, const DistMap &dm, const int *blocks_dev, int nown,
                              const double *inv16, double *dinv, double *dinvt, hipStream_t st);
hipError_t launch_blockmv(const double *M, const double *v, double *out, hipStream_t st);
hipError_t launch_fwd_update(const double *Lpanel, long long lda, const double *yk, double *vbelow, int nrows, hipStream_t st);
hipError_t launch_bwd_column(const double *Lpanel, long long lda, const double *xbelow, int nrows, const double *dinvt_k,
                             const double *yk, double *part, double *xk, hipStream_t st);
hipError_t launch_vec_add(long long n, double *dst, const double *src, hipStream_t st);
hipError_t launch_mask_owned(int n, const DistMap &dm, double *x, hipStream_t st);

// small vector helpers (vecops in bandchol.hip)
hipError_t launch_axpy_absmax(int n, double *x, const double *dx, double *absmax2, hipStream_t st);
// absmax2[0] = max |a|, absmax2[1] = max |b| (device doubles, non-negative => ordered like their bit patterns)
hipError_t launch_absmax2(int n, const double *a, const double *b, double *absmax2, hipStream_t st);

}  // namespace splpak
