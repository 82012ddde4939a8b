// The fit plan (internal): shared by plan.hip (single-GPU driver, C ABI) and dist.hip (several GPUs).
#pragma once
#include "kernels.hpp"
#include "../../include/splpak_hip.h"
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <vector>

namespace splpak {
// Barrier of the host threads of a one-process multi-GPU fit (one thread per GPU: dist.hip, ndchol.hip).
struct HostBarrier {
    std::mutex mu;
    std::condition_variable cv;
    int n = 1, count = 0, phase = 0;
    // false: another rank has failed and will never arrive (abort flag) -- the caller gives up too
    bool wait(const std::atomic<int> &abort)
    {
        std::unique_lock<std::mutex> lk(mu);
        const int ph = phase;
        if (++count == n) {
            count = 0;
            ++phase;
            cv.notify_all();
            return true;
        }
        while (phase == ph) {
            cv.wait_for(lk, std::chrono::milliseconds(20));
            if (phase == ph && abort.load()) return false;
        }
        return true;
    }
    void reset()
    {
        std::lock_guard<std::mutex> lk(mu);
        count = 0;
        ++phase;
    }
};
struct RowsOp;         // the rows of a 4-D fit applied to a vector, tile by tile (rowsop.hip)
struct PcgState;       // iterative solve of the least-squares problem (pcg.hip)
struct NdGroup;        // the ranks of a one-process multi-GPU nested-dissection factorisation (ndchol.hip)
}  // namespace splpak

struct splpak_plan {
    splpak::Grid g{};
    double xtrap = 0;
    long long max_ndata = 0;
    splpak::SortScratch s{};
    splpak::Band band{};
    double *comm = nullptr;
    bool own_comm = false;
    long long comm_len = 0;
    // views into comm
    double *nst = nullptr, *rhs = nullptr, *scalG = nullptr, *hist = nullptr, *scalH = nullptr,
           *rho = nullptr;
    long long lenG = 0, lenH = 0, lenR = 0;
    double *xvec = nullptr, *tmp = nullptr, *small = nullptr;   // small: [absmax(2) | minpiv(1) | backward error(1) | pad]
    double *gscratch = nullptr;   // per-cell Gram blocks: the band storage itself when it is large enough (it is only
                                  // filled after the gather), a buffer of its own otherwise
    long long gscratch_doubles = 0;
    double *ctab = nullptr;       // per-dimension factors of the constraint-row entries (grid only: filled once, launch_constraint_table)
    double *dcw = nullptr;        // [ncol] constraint weight of every node, spf: [ncol] "data sparse" flags (:923-960)
    unsigned char *spf = nullptr;
    double *rcell = nullptr;      // [ncell][nb] per-cell shares of the refinement residual
    double *e2buf = nullptr;      // [ncell + ncol] per-cell / per-node shares of the sum of squared row residuals (reserr)
    double *tbuf = nullptr;       // [ncol][ndim(ndim+1)/2] constraint-row dot products of the refinement residual
    int *info = nullptr;
    splpak_allreduce_fn ar = nullptr;
    void *ar_user = nullptr;
    int rank = 0, world = 1;
    void *ar_owned = nullptr;     // hook state the plan owns (splpak_plan_set_rccl): free()d with the plan
    int ar_flags = 0;             // SPLPAK_AR_*: what the hook accepts (splpak_plan_set_allreduce_ex)
    int setup_rc = 0;             // a failure while the ranks were set up (nd_set_ranks): returned by the next fit, collectively
    bool comm_failed = false;     // the hook reported a failure during the current fit (SPLPAK_E_COMM, not a device fault)
    int max_refine = 4;           // nominal number of refinement steps; a solve that is still contracting goes on (max_refine_hard)
    int max_refine_hard = 80;     // (30 until round 5: a 1-D grid of 2 048 nodes contracting by 0.63 per step needed ~50; fuzz seed 506 trial 58)
    double tol = 1e-11;           // on the ESTIMATED remaining error; the parity bar is 1e-10
    splpak::CholStats stats;
    // stage timing of the assembly and of one residual pass (HIP events on the fit's stream, kernel timing only)
    hipEvent_t evStage[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // [8], [9]: the diagnostic residual pass
    double stage_ms[6] = {0, 0, 0, 0, 0, 0};    // bin, gram blocks + gather, constraint rows, expand (+ band memset), residual pass, solve
    std::vector<void *> owned;
    size_t owned_bytes = 0;       // device bytes of `owned` (+ what the factorisation hooks report: fn_bytes)
    size_t (*fn_bytes)(void *user) = nullptr;
    // distributed band (dist.hip): this plan holds the block columns DistMap deals to rank dm.r
    splpak::DistMap dm{1, 0, 1, 0};
    int nown = 0;                 // block columns stored here
    int *own_blocks = nullptr;    // [nown] their global indices, ascending (device)
    std::vector<int> own_blocks_host;
    int device = 0;
    // factorisation / solve of the normal equations: NULL = the single-GPU band Cholesky of bandchol.hip;
    // dist.hip installs the distributed versions (same contract as band_cholesky / band_solve)
    hipError_t (*factor_fn)(splpak_plan *p, int *info_dev, double *minpiv_dev, hipStream_t st, void *user) = nullptr;
    hipError_t (*solve_fn)(splpak_plan *p, double *x, double *tmp, hipStream_t st, void *user) = nullptr;
    // half stencil -> band storage, when the factorisation keeps its own (twoend.hip); NULL = launch_expand
    hipError_t (*expand_fn)(splpak_plan *p, hipStream_t st, void *user) = nullptr;
    // called first thing in a fit (work that depends on nothing of it: clearing the factor storage beside the assembly); optional
    hipError_t (*prefit_fn)(splpak_plan *p, hipStream_t st, void *user) = nullptr;
    void *fn_user = nullptr;
    void (*fn_destroy)(void *user) = nullptr;      // releases fn_user with the plan (NULL: not the plan's to release)
    // iterative solve (pcg.hip): NULL = none.  solver_mode: 0 a factorisation only, 2 the iteration only (no factor storage: grids
    // no factorisation fits, or by request), 3 the iteration first and the factorisation when it stagnates
    splpak::Options opt;                      // the switches as they were when the plan was created (+ splpak_plan_set_option)
    splpak::RowsOp *rowsop = nullptr;         // 4-D grids: the tiled residual pass (NULL: the cell-by-cell passes of assemble.hip)
    splpak::PcgState *pcg = nullptr;
    int solver_mode = 0;
    double factor_flop = 0.0;     // flop of the plan's factorisation where known (nested dissection), else 0
    bool rows_only = false;       // iteration-only 4-D plans: the normal equations are never assembled (right-hand side, histogram and
                                  // backward-error denominators come from the rows: rowsop.hip); no half stencil, no Gram scratch
    const char *fn_name = nullptr;                 // what the hooks are (splpak_plan_factorisation); fn_code: 2 two-ended band, 4 nested dissection, 3 distributed band
    int fn_code = 0;
};


namespace splpak {
// reference-order validation shared by fit and evaluation (:716-750, :1166-1210): 0 or 101/102/103
int build_grid(int ndim, const int *nodes, const double *xmin, const double *xmax, Grid &g, long long *ncol_out,
               bool reorder = false);
int device_ready();
// rccl.hip: RCCL for the one-process multi-GPU plan (SPLPAK_MPLAN_RCCL=1)
int rccl_comms_for_devices(int n, const int *devices, void **comms);      // ncclCommInitAll; 0 or an SPLPAK_E_* code
int rccl_allreduce_sum(void *comm, double *buf, long long count, hipStream_t st);
void rccl_comm_free(void *comm);
// plan for rank r of R (chunks of c block columns); R = 1 is the ordinary single-GPU plan
// ndgrp != NULL (R > 1): the plan is rank r of a one-process multi-GPU fit whose grid takes the nested-dissection
// factorisation -- distributed over the group's ranks by subtrees and, above them, by block columns (ndchol.hip)
int plan_create_dist(int ndim, const int *nodes, const double *xmin, const double *xmax, double xtrap,
                     long long max_ndata, void *comm_buf_dev, long long comm_len, int R, int r, int c,
                     splpak_plan **plan, bool allow_nd = false, NdGroup *ndgrp = nullptr);
// large 3-D / 4-D grids on one GPU: nested-dissection multifrontal factorisation (ndchol.hip) instead of the band
bool nd_wanted(const Grid &g, const Band &band);
int nd_attach(splpak_plan *p, double **factor_arena, long long *factor_doubles, NdGroup *grp = nullptr, int rank = 0);
// One-process multi-GPU fit: the group its ranks' plans share (abort: the fit's abort flag, set when a rank fails).
// nd_group_finalize: after every rank's plan has been created -- the tables that hold the peers' addresses.
NdGroup *nd_group_create(int R, int chunk, std::atomic<int> *abort);
int nd_group_finalize(NdGroup *g);
void nd_group_destroy(NdGroup *g);
// before a fit: a fit that was abandoned may have left arrivals counted in the group's barrier (all rank threads are joined)
void nd_group_reset(NdGroup *g);
// the whole-grid test of nd_wanted for a grid given by its nodes (no plan yet)
bool nd_wanted_for(int ndim, const int *nodes, const double *xmin, const double *xmax);
// the sharded fit's ranks are known: distribute the nested-dissection factorisation by subtrees (SPLPAK_ND_DIST=1; ndchol.hip)
int nd_set_ranks(splpak_plan *p, int rank, int world);
// sum over the ranks of the sharded fit through the caller's hook (plan.hip); 0 = fine (also with one rank)
int plan_allreduce(splpak_plan *p, double *buf, long long count, hipStream_t st);
// narrow bands on one GPU: install the two-ended factorisation (twoend.hip) when it shortens the chain
void twoend_attach(splpak_plan *p);
void twoend_detach(splpak_plan *p);
// rowsop.hip
int rowsop_create(const Grid &g, bool side_stream, RowsOp **out);
void rowsop_destroy(RowsOp *r);
size_t rowsop_bytes(const RowsOp *r);
hipError_t rowsop_apply(const Grid &g, RowsOp *r, const SortScratch &rows, const double *xvec, const double *dcw, const unsigned char *spf,
                        const double *ctab, bool constraints, double *rho, hipStream_t st);
hipError_t rowsop_histogram(const Grid &g, RowsOp *r, const SortScratch &rows, double *hist, hipStream_t st);
hipError_t rowsop_data_diagonal(const Grid &g, RowsOp *r, const SortScratch &rows, const double *xvec, double *diag, hipStream_t st);
hipError_t rowsop_backward_denominators(const Grid &g, RowsOp *r, const SortScratch &rows, const double *xvec, const double *rhs, const double *dcw,
                                        const unsigned char *spf, const double *ctab, bool constraints, double *absx, double *tmp, double *den,
                                        hipStream_t st);
// rho = A^T W (W y - W A x) [- C^T C x] of the plan's binned points (rows.ys == NULL: y = 0), by whichever pass the plan has
hipError_t rowsop_residual(const Grid &g, RowsOp *r, const SortScratch &rows, const double *xvec, const double *dcw, const unsigned char *spf,
                           const double *ctab, bool constraints, double *rho, double *ssq, double *e2buf, hipStream_t st);
inline hipError_t plan_rows_residual(splpak_plan *p, const SortScratch &rows, const double *xvec, bool constraints, double *rho, hipStream_t st)
{
    if (p->rowsop && p->ctab) return rowsop_apply(p->g, p->rowsop, rows, xvec, p->dcw, p->spf, p->ctab, constraints, rho, st);
    return launch_residual(p->g, rows, xvec, p->rcell, p->dcw, p->spf, p->ctab, constraints, p->tbuf, rho, nullptr, nullptr, st);
}
// ndchol.hip: batched Cholesky + triangular inverses of independent dense 256 x 256 blocks
size_t block_chol_job_bytes(int nb);
hipError_t block_chol_prepare(void *jobs_dev, int nb, double *blocks, double *inv16, double *dinv, double *dinvt, const int *ncols_host);
hipError_t block_chol_run(const void *jobs_dev, int nb, int *info_dev, double *minpiv_dev, hipStream_t st);
// pcg.hip
int pcg_attach(splpak_plan *p, PcgState **out);
void pcg_destroy(PcgState *s);
size_t pcg_bytes(const PcgState *s);
void pcg_stats(const PcgState *s, double *out6);
double *pcg_scratch(PcgState *s, int which);      // two vectors of ncol doubles, free between solves
bool pcg_singular(const PcgState *s);             // after pcg_prepare: a box of the ASSEMBLED normal equations is not positive definite
hipError_t pcg_sum_w2(splpak_plan *p, hipStream_t st);
hipError_t pcg_prepare(splpak_plan *p, PcgState *s, double sumw2, bool smooth, bool from_rows, hipStream_t st);
bool pcg_boxes_from_rows(const PcgState *s);      // the state can build its boxes from the rows (4-D: the fit may leave the normal equations unassembled)
int pcg_solve(splpak_plan *p, PcgState *s, double *v, double tol, bool smooth, hipStream_t st);
int twoend_debug_solve(int n, int halfbw, const double *a_lower, const double *bvec, double *x_out, int *hinfo_out);
}  // namespace splpak
