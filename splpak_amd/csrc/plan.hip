// Fit driver and the C ABI (include/splpak_hip.h).
//
// The fit that the reference performs as "one dense row at a time through a dense
// Householder solver" (splcw :512-1060 -> suprls :1375-1695) is done here as
//   1. bin the points by 4-wide node window (counting sort),       assemble.hip
//   2. per-window Gram blocks -> banded normal equations N, r,      assemble.hip
//   3. derivative-constraint rows of data-sparse nodes -> N,        assemble.hip
//   4. blocked band Cholesky on the f64 matrix cores,               bandchol.hip
//   5. solve + iterative refinement with the residual recomputed FROM THE ROWS,
//      rho = A^T W (W y - W A x) - C^T C x, which brings the normal-equation
//      solution back to the accuracy of an orthogonal factorisation
//      (1e-13..2e-12 max-norm vs the reference).
// Multi-GPU: every rank runs 1-2 on its shard of the points; the
// histogram, then (N, r), then each refinement residual are sum-all-reduced
// through the caller's hook (RCCL via torch.distributed); 3 is applied by rank 0
// before the reduction so all ranks hold bit-identical normal equations; 4-5 are
// replicated.
#include "plan.hpp"
#include "basis.hpp"

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <vector>

namespace splpak {

static thread_local std::string g_err;

void set_error(const std::string &msg) { g_err = msg; }

bool hip_ok(hipError_t e, const char *what)
{
    if (e == hipSuccess) return true;
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    (void)hipGetLastError();
    return false;
}

// reference-order validation shared by fit and evaluation (:716-750, :1166-1210).
// returns 0 or 101/102/103
int build_grid(int ndim, const int *nodes, const double *xmin, const double *xmax, Grid &g,
               long long *ncol_out, bool reorder)
{
    std::memset(&g, 0, sizeof(g));
    if (ndim < 1) return 101;
    if (ndim > MAXD) return SPLPAK_E_UNSUPPORTED;
    g.ndim = ndim;
    // checks in the reference's order (first failing dimension wins), reference-order copies
    long long rs = 1;
    int refstride_of[MAXD] = {0, 0, 0, 0};
    for (int d = 0; d < ndim; ++d) {
        if (nodes[d] < 4) return 102;
        const double xrng = xmax[d] - xmin[d];
        if (xrng == 0.0) return 103;
        g.ref_nodes[d] = nodes[d];
        g.ref_xmin[d] = xmin[d];
        g.ref_dxin[d] = 1.0 / (xrng / (double)(nodes[d] - 1));
        refstride_of[d] = (int)rs;
        rs *= nodes[d];
        if (rs > (1LL << 30)) { if (ncol_out) *ncol_out = rs; return SPLPAK_E_UNSUPPORTED; }
    }
    for (int d = 0; d < MAXD; ++d) g.perm[d] = d;
    if (reorder)        // ascending node counts, stable: identity for isotropic grids
        for (int i = 1; i < ndim; ++i)
            for (int j = i; j > 0 && nodes[g.perm[j]] < nodes[g.perm[j - 1]]; --j) std::swap(g.perm[j], g.perm[j - 1]);
    long long ncol = 1, ncell = 1, cs = 1, ls = 1;
    int nb = 1, h = 1, hb = 0;
    for (int d = 0; d < ndim; ++d) {
        const int r = g.perm[d];
        const int nod = nodes[r];
        const double xrng = xmax[r] - xmin[r];
        g.nodes[d] = nod;
        g.xmin[d] = xmin[r];
        g.dx[d] = xrng / (double)(nod - 1);        // :747
        g.dxin[d] = 1.0 / g.dx[d];                 // :748
        g.colstride[d] = (int)cs;
        g.refstride[d] = refstride_of[r];
        g.cells[d] = nod - 3;
        g.cellstride[d] = (int)ls;
        hb += 3 * (int)cs;
        cs *= nod;
        ls *= (nod - 3);
        ncol *= nod;
        ncell *= (nod - 3);
        nb *= 4;
        h *= 7;
    }
    for (int d = ndim; d < MAXD; ++d) {
        g.nodes[d] = 4; g.dx[d] = g.dxin[d] = 1.0; g.cells[d] = 1;
        g.ref_nodes[d] = 4; g.ref_dxin[d] = 1.0;
    }
    g.ncol = (int)ncol;
    g.ncell = (int)ncell;
    g.nb = nb;
    g.hstencil = (h + 1) / 2;
    g.halfbw = hb;
    if (ncol_out) *ncol_out = ncol;
    return 0;
}

}  // namespace splpak

using namespace splpak;

static long long comm_len_of(const Grid &g)
{
    const long long npad = ((g.ncol + NBLK - 1) / NBLK) * (long long)NBLK;
    return (long long)g.ncol * g.hstencil + g.ncol + SC_COUNT   // G: nst, rhs, scalG
           + (long long)g.ncol + SC_COUNT                        // H: hist, scalH
           + npad + SC_COUNT;                                    // R: rho, scalR (residual sum of squares)
}

template <typename T>
static bool dev_alloc(splpak_plan *p, T **ptr, size_t count)
{
    void *q = nullptr;
    if (count == 0) count = 1;
    hipError_t e = hipMalloc(&q, count * sizeof(T));
    if (e != hipSuccess && release_cached_plan_for_memory()) {      // the one-shot entry's cached plan (35 GB at 64^3) is in the way
        (void)hipGetLastError();
        e = hipMalloc(&q, count * sizeof(T));
    }
    if (e != hipSuccess) {
        char buf[160];
        snprintf(buf, sizeof buf, "hipMalloc of %.3f GB failed: %s", (double)(count * sizeof(T)) / 1e9,
                 hipGetErrorString(e));
        set_error(buf);
        (void)hipGetLastError();
        return false;
    }
    p->owned.push_back(q);
    p->owned_bytes += count * sizeof(T);
    *ptr = static_cast<T *>(q);
    return true;
}

namespace splpak {
int device_ready()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no usable HIP device: the splpak HIP path has no CPU fallback");
        (void)hipGetLastError();
        return SPLPAK_E_NODEVICE;
    }
    return 0;
}
}  // namespace splpak

extern "C" {

int64_t splpak_plan_comm_len(int32_t ndim, const int32_t *nodes)
{
    double xmin[MAXD] = {0, 0, 0, 0}, xmax[MAXD] = {1, 1, 1, 1};
    Grid g;
    if (!nodes || build_grid(ndim, nodes, xmin, xmax, g, nullptr) != 0) return -1;
    return comm_len_of(g);
}

int32_t splpak_plan_create(int32_t ndim, const int32_t *nodes, const double *xmin,
                           const double *xmax, double xtrap, int64_t max_ndata,
                           void *comm_buf_dev, int64_t comm_len, splpak_plan **plan)
{
    return plan_create_dist(ndim, nodes, xmin, xmax, xtrap, max_ndata, comm_buf_dev, comm_len, 1, 0, 1, plan, true);
}

}  // extern "C"

int splpak::plan_create_dist(int ndim, const int *nodes, const double *xmin, const double *xmax, double xtrap,
                             long long max_ndata, void *comm_buf_dev, long long comm_len, int R, int r, int c,
                             splpak_plan **plan, bool allow_nd, NdGroup *ndgrp)
{
    if (!plan || !nodes || !xmin || !xmax) { set_error("null argument"); return SPLPAK_E_BADARG; }
    *plan = nullptr;
    const Options snap = options_snapshot();        // the switches of this plan: process defaults over the environment, as of now
    OptionsScope opt_scope(&snap);
    Grid g;
    long long ncol = 0;
    const int v = build_grid(ndim, nodes, xmin, xmax, g, &ncol, splpak::opt_get("SPLPAK_NO_REORDER") == nullptr);
    if (v != 0) {
        if (v == SPLPAK_E_UNSUPPORTED) set_error("ndim > 4 or more than 2^30 nodes is not supported");
        return v;
    }
    if (max_ndata < 1) return 105;
    if (max_ndata > (int64_t)std::numeric_limits<int32_t>::max() - 1024) {
        // the binning (keys, per-cell offsets and cursors, the scan) is 32-bit
        set_error("max_ndata per plan (= per GPU) is limited to 2^31 - 1025 points; shard the points over more plans");
        return SPLPAK_E_UNSUPPORTED;
    }
    if (int r = device_ready()) return r;

    splpak_plan *p = new splpak_plan();
    p->opt = snap;
    p->g = g;
    p->xtrap = xtrap;
    p->max_ndata = max_ndata;
    bool ok = true;
    // sort scratch
    p->s.cap = max_ndata;
    ok = ok && dev_alloc(p, &p->s.key, (size_t)max_ndata);
    ok = ok && dev_alloc(p, &p->s.count, (size_t)g.ncell + 2);
    ok = ok && dev_alloc(p, &p->s.scanpart, (size_t)256);
    ok = ok && dev_alloc(p, &p->s.offset, (size_t)g.ncell + 2);
    ok = ok && dev_alloc(p, &p->s.cursor, (size_t)g.ncell + 2);
    ok = ok && dev_alloc(p, &p->s.xs, (size_t)max_ndata * g.ndim);
    ok = ok && dev_alloc(p, &p->s.ys, (size_t)max_ndata);
    ok = ok && dev_alloc(p, &p->s.ws, (size_t)max_ndata);
    ok = ok && dev_alloc(p, &p->s.idx, (size_t)max_ndata);
    {   // stable partition of the points (assemble.hip sp_*): count matrix, bin bases, tile-sorted records for grids of more cells than bins
        const size_t nblk = (size_t)((max_ndata + SP_Q - 1) / SP_Q);
        ok = ok && dev_alloc(p, &p->s.cntm, nblk * SP_NB);
        ok = ok && dev_alloc(p, &p->s.binbase, (size_t)2 * SP_NB + 16);      // bin bases | bin totals
        ok = ok && dev_alloc(p, &p->s.sppart, ((nblk + 15) / 16) * SP_NB);
        const long long rd = bin_record_doubles(g, max_ndata);
        if (rd > 0) ok = ok && dev_alloc(p, &p->s.rec, (size_t)rd);
    }
    {   // per-cell shares of the residual passes of assemble.hip: 1-D .. 3-D grids; a 4-D grid's passes go tile by tile (rowsop.hip)
        // and leave this scratch out (1.45 GB at 32^4) unless the A/B switches ask for the cell-by-cell forms
        const char *rt = splpak::opt_get("SPLPAK_ROWS_TILES");
        const bool tiled = g.ndim == 4 && !splpak::opt_get("SPLPAK_NO_CONSTRAINT_TABLE") && !(rt && atoi(rt) == 0) && !splpak::opt_get("SPLPAK_RESIDUAL_CELLS");
        if (!tiled) ok = ok && dev_alloc(p, &p->rcell, (size_t)g.ncell * g.nb);
    }
    ok = ok && dev_alloc(p, &p->tbuf, (size_t)g.ncol * (g.ndim * (g.ndim + 1) / 2));
    ok = ok && dev_alloc(p, &p->dcw, (size_t)g.ncol);
    ok = ok && dev_alloc(p, &p->ctab, (size_t)constraint_table_doubles(g));
    ok = ok && dev_alloc(p, &p->spf, (size_t)g.ncol);
    ok = ok && dev_alloc(p, &p->e2buf, (size_t)g.ncell + (size_t)g.ncol);
    // band: all of it (R = 1) or the block columns dealt to rank r of R
    band_bytes(g.ncol, g.halfbw, &p->band);
    (void)hipGetDevice(&p->device);
    p->dm = DistMap{R < 1 ? 1 : R, r, c < 1 ? 1 : c, p->band.lda + 1};
    for (int J = 0; J < p->band.nblk; ++J)
        if (dm_owned(p->dm, J)) p->own_blocks_host.push_back(J);
    p->nown = (int)p->own_blocks_host.size();
    if (p->dm.R > 1) p->band.bytes = (size_t)(p->nown > 0 ? p->nown : 1) * NBLK * (size_t)p->dm.ld * sizeof(double) + 4096;
    // Large 3-D / 4-D grids on one GPU: the nested-dissection multifrontal factorisation (ndchol.hip) instead of the
    // band.  It owns its storage; the factor arena stands in for the band as the home of the Gram scratch.
    // Several GPUs driven by one process (ndgrp): the same factorisation, distributed -- every rank keeps its own subtrees and
    // its block columns of the fronts above them (round 4; ndchol.hip, ndtop.inc).
    // How the least-squares problem is solved (round 6): a factorisation (band / nested dissection) whenever one fits the device;
    // the iteration of pcg.hip for the grids none fits (4-D from about 29^4 on one GPU) or by request -- SPLPAK_SOLVER =
    // direct | pcg | pcg+direct (the iteration first, the factorisation when it stagnates) | auto.
    int mode = 0;
    if (const char *e = splpak::opt_get("SPLPAK_SOLVER")) {
        if (!std::strcmp(e, "direct")) mode = 1;
        else if (!std::strcmp(e, "pcg")) mode = 2;
        else if (!std::strcmp(e, "pcg+direct")) mode = 3;
    }
    if (p->dm.R > 1) mode = 1;                      // (the one-process multi-GPU plans distribute a factorisation)
    // Left to itself a LARGE 4-D grid (from 20^4 columns on: the factorisation takes seconds) tries the iteration first: where the
    // constraint rows are dense (>= 2 per column: config 5's density of points) or absent it answers in a fraction of the
    // factorisation's time (24^4: 0.5 s against 4.6 s, 28^4: ~1.4 s against 18 s); where it stagnates (1.2 .. 1.7 rows per column)
    // the attempt costs 0.6 .. 1.4 s before the factorisation takes over (tools/pcg/density_sweep.py).
    if (mode == 0 && g.ndim == 4 && g.ncol >= 160000) mode = 3;
    const bool auto_mode = mode == 0 || (mode == 3 && !splpak::opt_get("SPLPAK_SOLVER"));
    bool direct = mode != 2;
    const bool use_nd = direct && allow_nd && (p->dm.R == 1 || ndgrp != nullptr) && nd_wanted(g, p->band);
    if (use_nd && ok) {
        double *arena = nullptr;
        long long arena_doubles = 0;
        const int rc = nd_attach(p, &arena, &arena_doubles, p->dm.R > 1 ? ndgrp : nullptr, r);
        if (rc == SPLPAK_E_NOMEM && auto_mode && p->dm.R == 1) {
            // no factorisation of this grid fits the device: the iteration instead
            if (p->fn_destroy) p->fn_destroy(p->fn_user);
            p->fn_user = nullptr; p->fn_destroy = nullptr; p->fn_bytes = nullptr; p->fn_name = nullptr; p->fn_code = 0;
            p->factor_fn = nullptr; p->solve_fn = nullptr; p->expand_fn = nullptr; p->prefit_fn = nullptr;
            (void)hipGetLastError();
            direct = false;
        } else if (rc != 0) {
            splpak_plan_destroy(p);
            return rc;
        } else {
            p->band.ab = arena;
            p->band.bytes = (size_t)arena_doubles * sizeof(double);
        }
    } else if (direct && ok) {
        const bool okb = dev_alloc(p, &p->band.ab, p->band.bytes / sizeof(double));
        if (!okb && auto_mode && p->dm.R == 1) direct = false;
        else ok = ok && okb;
    }
    if (!direct) { p->band.ab = nullptr; p->band.bytes = 0; }
    // Iteration-only plans of 4-D grids never assemble the normal equations (round 6): the iteration applies the ROWS, and the
    // right-hand side, the histogram and the backward-error denominators come from the rows too (rowsop.hip) -- no half stencil
    // (10 GB at 32^4, and its all-reduce in a sharded fit), no Gram scratch (8 GB), no Gram / gather / constraint-row kernels
    // (0.30 of the 0.32 s of config 5's assembly).  pcg_assemble = 1 keeps the assembled form (A/B).
    {
        const char *rt = splpak::opt_get("SPLPAK_ROWS_TILES");
        p->rows_only = !direct && g.ndim == 4 && !splpak::opt_get("SPLPAK_NO_CONSTRAINT_TABLE") && !(rt && atoi(rt) == 0) &&
                       !splpak::opt_get("SPLPAK_PCG_ASSEMBLE");
    }
    {
        // scratch of the per-cell Gram blocks: everything at once if <= 8 GB (or if the band storage, which is
        // idle until the gather is done, holds it); otherwise slabs of what there is (launch_gram)
        const long long full = p->rows_only ? 8 : gram_scratch_doubles(g), least = p->rows_only ? 8 : gram_scratch_min_doubles(g);
        long long want = full;
        const char *cap = splpak::opt_get("SPLPAK_GRAM_SCRATCH_MB");
        const long long budget = cap ? atoll(cap) * (1LL << 17) : (1LL << 30);      // doubles (default 8 GB)
        if (want > budget) want = budget > least ? budget : least;
        const long long band_doubles = (long long)(p->band.bytes / sizeof(double));
        if (band_doubles >= want && !cap) {
            p->gscratch = p->band.ab;
            p->gscratch_doubles = band_doubles < full ? band_doubles : full;
        } else {
            ok = ok && dev_alloc(p, &p->gscratch, (size_t)want);
            p->gscratch_doubles = want;
        }
    }
    const size_t nloc = (use_nd || !direct) ? 0 : (size_t)(p->nown > 0 ? p->nown : 1);      // (the band's block inverses: not with nested dissection)
    ok = ok && dev_alloc(p, &p->band.dinv, nloc * NBLK * NBLK);
    ok = ok && dev_alloc(p, &p->band.dinvt, nloc * NBLK * NBLK);
    ok = ok && dev_alloc(p, &p->band.inv64, nloc * 4 * 64 * 64);
    if (p->dm.R == 1) {
        ok = ok && dev_alloc(p, &p->band.mfwd, nloc * NBLK * NBLK);
        ok = ok && dev_alloc(p, &p->band.mbwd, nloc * NBLK * NBLK);
    } else {
        ok = ok && dev_alloc(p, &p->own_blocks, nloc);
        if (ok && p->nown > 0 && !use_nd)
            ok = hip_ok(hipMemcpy(p->own_blocks, p->own_blocks_host.data(), sizeof(int) * (size_t)p->nown, hipMemcpyHostToDevice),
                        "hipMemcpy of the block list");
    }
    // communication buffer (rows-only plans: no half stencil in it -- right-hand side first)
    p->comm_len = comm_len_of(g) - (p->rows_only ? (long long)g.ncol * g.hstencil : 0);
    if (comm_buf_dev) {
        if (comm_len < p->comm_len) {
            set_error("comm buffer too small");
            splpak_plan_destroy(p);
            return SPLPAK_E_BADARG;
        }
        p->comm = static_cast<double *>(comm_buf_dev);
    } else {
        ok = ok && dev_alloc(p, &p->comm, (size_t)p->comm_len);
        p->own_comm = true;
    }
    ok = ok && dev_alloc(p, &p->xvec, (size_t)p->band.npad);
    ok = ok && dev_alloc(p, &p->tmp, (size_t)p->band.npad);
    ok = ok && dev_alloc(p, &p->small, 8);
    ok = ok && dev_alloc(p, &p->info, 2);
    if (!ok) {
        splpak_plan_destroy(p);
        return SPLPAK_E_NOMEM;
    }
    p->lenG = (p->rows_only ? 0 : (long long)g.ncol * g.hstencil) + g.ncol + SC_COUNT;
    p->lenH = (long long)g.ncol + SC_COUNT;
    p->lenR = p->band.npad + SC_COUNT;
    p->nst = p->rows_only ? nullptr : p->comm;
    p->rhs = p->comm + (p->rows_only ? 0 : (long long)g.ncol * g.hstencil);
    p->scalG = p->rhs + g.ncol;
    p->hist = p->scalG + SC_COUNT;
    p->scalH = p->hist + g.ncol;
    p->rho = p->scalH + SC_COUNT;
    if (splpak::opt_get("SPLPAK_NO_CONSTRAINT_TABLE")) p->ctab = nullptr;      // (A/B switch; the allocation stays with the plan's list)
    else if (!hip_ok(launch_constraint_table(g, p->ctab, nullptr), "constraint table") ||
             !hip_ok(hipStreamSynchronize(nullptr), "constraint table")) {
        splpak_plan_destroy(p);
        return SPLPAK_E_NODEVICE;
    }
    if (p->ctab) {
        const int rc = rowsop_create(g, !direct, &p->rowsop);
        if (rc != 0) {
            splpak_plan_destroy(p);
            return rc;
        }
    }
    if (direct) twoend_attach(p);
    p->solver_mode = !direct ? 2 : (mode == 3 ? 3 : 0);
    if (!direct || mode == 3) {
        const int rc = pcg_attach(p, &p->pcg);
        if (rc != 0) {
            splpak_plan_destroy(p);
            return rc;
        }
    }
    *plan = p;
    return 0;
}

extern "C" {

void splpak_plan_destroy(splpak_plan *p)
{
    if (!p) return;
    if (p->fn_destroy) p->fn_destroy(p->fn_user);
    pcg_destroy(p->pcg);
    rowsop_destroy(p->rowsop);
    band_pipeline_destroy(p->band.pipe);
    for (hipEvent_t e : p->evStage) if (e) (void)hipEventDestroy(e);
    for (void *q : p->owned) (void)hipFree(q);
    std::free(p->ar_owned);
    delete p;
}

int32_t splpak_plan_set_allreduce_ex(splpak_plan *p, splpak_allreduce_fn fn, void *user, int32_t rank,
                                     int32_t world, int32_t flags)
{
    if (!p) { set_error("null plan"); return SPLPAK_E_BADARG; }
    p->ar = fn;
    p->ar_user = user;
    p->rank = rank;
    p->world = world < 1 ? 1 : world;
    p->ar_flags = flags;
    // (a failure here leaves the plan without usable job tables: it is remembered and every rank's next fit returns it
    //  through the first reduction's error flag -- round-3 advice: the status used to be dropped)
    p->setup_rc = nd_set_ranks(p, p->rank, p->world);
    return p->setup_rc;
}

void splpak_plan_set_allreduce(splpak_plan *p, splpak_allreduce_fn fn, void *user, int32_t rank,
                               int32_t world)
{
    // the hook of rounds 1-2: windows of the plan's communication buffer only -> the factorisation stays replicated
    (void)splpak_plan_set_allreduce_ex(p, fn, user, rank, world, 0);
}

void splpak_plan_set_refine(splpak_plan *p, int32_t max_steps, double tol)
{
    if (!p) return;
    p->max_refine = max_steps < 0 ? 0 : max_steps;
    p->max_refine_hard = p->max_refine == 0 ? 0 : (p->max_refine > 80 ? p->max_refine : 80);
    p->tol = tol;
}

void splpak_plan_enable_kernel_timing(splpak_plan *p, int32_t on)
{
    if (p) p->stats.enabled = on != 0;
}

void splpak_plan_kernel_timing(const splpak_plan *p, double *out4)
{
    if (!p || !out4) return;
    out4[5] = p->stats.bulk_launches;
    out4[6] = p->stats.bulk_flop;
    out4[0] = p->stats.syrk_launches;
    out4[1] = p->stats.syrk_ms;
    out4[2] = p->stats.syrk_flop;
    out4[3] = p->stats.factor_ms;
    out4[4] = p->stats.total_flop;
}

void splpak_plan_stage_timing(const splpak_plan *p, double *out6)
{
    if (!p || !out6) return;
    for (int i = 0; i < 6; ++i) out6[i] = p->stage_ms[i];
}

int32_t splpak_set_default_option(const char *name, const char *value)
{
    if (options_set_default(name, value) != 0) { set_error(std::string("unknown option: ") + (name ? name : "(null)")); return SPLPAK_E_BADARG; }
    return 0;
}

int32_t splpak_plan_set_option(splpak_plan *p, const char *name, const char *value)
{
    if (!p) { set_error("null plan"); return SPLPAK_E_BADARG; }
    std::string canon;
    if (!option_canonical(name, canon)) { set_error(std::string("unknown option: ") + (name ? name : "(null)")); return SPLPAK_E_BADARG; }
    // what shapes the plan's storage and job tables was consumed when the plan was created
    static const char *const at_creation[] = {"SPLPAK_SOLVER", "SPLPAK_ND", "SPLPAK_ND_SPLIT", "SPLPAK_ND_CUT", "SPLPAK_ND_HALVES", "SPLPAK_ND_KB", "SPLPAK_ND_RES_CUS",
                                              "SPLPAK_NO_REORDER", "SPLPAK_GRAM_SCRATCH_MB", "SPLPAK_PCG_MAXIT", "SPLPAK_MPLAN_RCCL", "SPLPAK_RCCL_LIB"};
    for (const char *c : at_creation)
        if (canon == c) {
            set_error(canon + " shapes the plan and is read when it is created: set it with splpak_set_default_option (or the environment) before splpak_plan_create");
            return SPLPAK_E_UNSUPPORTED;
        }
    if (value) p->opt.kv[canon] = value; else p->opt.kv.erase(canon);
    return 0;
}

int32_t splpak_plan_get_option(const splpak_plan *p, const char *name, char *buf, int32_t buflen)
{
    if (!p) { set_error("null plan"); return SPLPAK_E_BADARG; }
    std::string canon;
    if (!option_canonical(name, canon)) { set_error(std::string("unknown option: ") + (name ? name : "(null)")); return SPLPAK_E_BADARG; }
    const char *v = p->opt.get(canon.c_str());
    if (buf && buflen > 0) {
        std::strncpy(buf, v ? v : "", (size_t)buflen - 1);
        buf[buflen - 1] = '\0';
    }
    return v ? 1 : 0;
}

void splpak_plan_pcg_stats(const splpak_plan *p, double *out6)
{
    if (!out6) return;
    for (int i = 0; i < 6; ++i) out6[i] = 0.0;
    if (p && p->pcg) pcg_stats(p->pcg, out6);
}

const double *splpak_plan_hist_dev(const splpak_plan *p) { return p ? p->hist : nullptr; }

int64_t splpak_plan_device_bytes(const splpak_plan *p)
{
    if (!p) return 0;
    return (int64_t)(p->owned_bytes + (p->fn_bytes ? p->fn_bytes(p->fn_user) : 0) + pcg_bytes(p->pcg) + rowsop_bytes(p->rowsop));
}

int32_t splpak_plan_factorisation(const splpak_plan *p, char *buf, int32_t buflen)
{
    if (!p) return SPLPAK_E_BADARG;
    int code = 0;
    const char *what = "band Cholesky, four-stream look-ahead pipeline (csrc/bandchol.hip)";
    if (p->solver_mode == 2) { code = 6; what = "preconditioned conjugate gradients on the rows, separable preconditioner (csrc/pcg.hip); no factorisation"; }
    else if (p->fn_name) { code = p->fn_code; what = p->fn_name; }
    else if (p->dm.R > 1) { code = 3; what = "band Cholesky distributed over several GPUs by block columns (csrc/dist.hip)"; }
    else if (p->band.bw < narrow_band_limit()) { code = 1; what = "band Cholesky, narrow (chain-bound) form (csrc/bandchol.hip)"; }
    if (buf && buflen > 0) {
        std::strncpy(buf, what, (size_t)buflen - 1);
        buf[buflen - 1] = '\0';
    }
    return code;
}

// SPLPAK_DEBUG_SUMS: sum and absolute sum of a device buffer, printed with a label (diagnosing the sharded fit)
static void debug_sum(const splpak_plan *p, const char *what, const double *buf, long long count, hipStream_t st)
{
    if (!splpak::opt_get("SPLPAK_DEBUG_SUMS")) return;
    std::vector<double> h((size_t)count);
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h.data(), buf, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost);
    double s = 0.0, a = 0.0;
    for (double v : h) { s += v; a += std::fabs(v); }
    fprintf(stderr, "[splpak rank %d] %s: sum %.17g abs %.17g\n", p->rank, what, s, a);
}

static int do_allreduce(splpak_plan *p, double *buf, long long count, hipStream_t st)
{
    if (!p->ar || (p->world <= 1 && !(p->ar_flags & SPLPAK_AR_ALWAYS))) return 0;
    debug_sum(p, "before all-reduce", buf, count, st);
    // The buffer is complete before the hook sees it and the reduced values are in place before the fit goes on,
    // whatever the hook's own ordering is worth: the rehearsal of `bench.py --gpus 2` on ONE device over gloo summed
    // buffers the fit's kernels were still writing (round 3; a host synchronisation costs ~10 us, a fit issues
    // 3 + refinement steps of these)
    // (a hook that declares SPLPAK_AR_STREAM_ORDERED -- the native RCCL one: ncclAllReduce is enqueued on the stream it is
    //  handed -- is ordered with the fit's kernels by the stream itself: no host synchronisation on either side)
    const bool ordered = (p->ar_flags & SPLPAK_AR_STREAM_ORDERED) != 0;
    if (!ordered) SPLPAK_HIP_TRY(hipStreamSynchronize(st), SPLPAK_E_NODEVICE);
    const int r = p->ar(buf, count, (void *)st, p->ar_user);
    if (r != 0) { set_error("all-reduce callback failed"); p->comm_failed = true; return SPLPAK_E_COMM; }
    if (!ordered) SPLPAK_HIP_TRY(hipStreamSynchronize(st), SPLPAK_E_NODEVICE);
    debug_sum(p, "after  all-reduce", buf, count, st);
    return 0;
}

}  // extern "C" (the next function has C++ linkage: it is called from ndchol.hip)
namespace splpak { int plan_allreduce(splpak_plan *p, double *buf, long long count, hipStream_t st) { return do_allreduce(p, buf, count, st); } }
extern "C" {

int32_t splpak_plan_fit_dev(splpak_plan *p, const double *x, int32_t l1xdat, const double *y,
                            const double *w, int64_t ndata, double *coef_dev, void *stream,
                            double *info)
{
    if (!p || !coef_dev) { set_error("null argument"); return SPLPAK_E_BADARG; }
    OptionsScope opt_scope(&p->opt);                                  // every switch a fit reads comes from the plan's snapshot
    if (ndata < 1 && p->world <= 1) return 105;                       // :759-764
    // A failure of ONE rank's arguments must not leave the others waiting in a collective: with more
    // than one rank it is carried through the first reduction as a flag and every rank returns.
    int lerr = 0;
    if (ndata < 0) ndata = 0;
    if (ndata > 0 && (!x || !y)) { set_error("null data pointer"); lerr = SPLPAK_E_BADARG; }
    else if (ndata > p->max_ndata) { set_error("ndata exceeds the plan's max_ndata"); lerr = SPLPAK_E_BADARG; }
    else if (l1xdat < p->g.ndim) { set_error("l1xdat < ndim"); lerr = SPLPAK_E_BADARG; }
    if (lerr == 0 && p->setup_rc != 0) { set_error("the plan's rank set-up failed (splpak_plan_set_allreduce)"); lerr = p->setup_rc; }
    if (lerr != 0 && p->world <= 1) return lerr;
    if (lerr != 0) ndata = 0;
    p->comm_failed = false;
    // a failure inside the factorisation / solve hooks: a communication failure is not a device fault (round-3 advice)
#define SPLPAK_HOOK_TRY(expr)                                                        \
    do {                                                                             \
        const hipError_t he_ = (expr);                                               \
        if (p->comm_failed) { (void)hipGetLastError(); return SPLPAK_E_COMM; }       \
        if (!::splpak::hip_ok(he_, #expr)) return SPLPAK_E_NODEVICE;                 \
    } while (0)
    hipStream_t st = (hipStream_t)stream;
    const Grid &g = p->g;
    const Band &b = p->band;
    const bool smooth = p->xtrap != 0.0;                              // swght, :769
    using clk = std::chrono::steady_clock;
    auto t0 = clk::now();
    if (info) for (int i = 0; i < 10; ++i) info[i] = 0.0;

    // ---- assembly -------------------------------------------------------
    const bool stamps = p->stats.enabled;
    if (stamps)
        for (hipEvent_t &e : p->evStage)
            if (!e) SPLPAK_HIP_TRY(hipEventCreate(&e), SPLPAK_E_NODEVICE);
    auto stamp = [&](int i) { if (stamps) (void)hipEventRecord(p->evStage[i], st); };
    // A 4-D plan that has the iteration IN FRONT of a factorisation assembles the normal equations only when the factorisation is
    // going to need them (round 6): the iteration applies the rows, its boxes are built from the rows (bj_build_kernel), and whether
    // it is tried at all is known from the histogram -- so the fit starts as an iteration-only plan's does (3 ms) and falls back to
    // the assembly (62 ms at 24^4: 40 % of such a fit) where the iteration is not tried or gives up.  One rank, no reduction hook.
    const bool lazy = !p->rows_only && p->pcg && p->solver_mode == 3 && p->rowsop && p->ctab && smooth && p->world <= 1 && !p->ar &&
                      pcg_boxes_from_rows(p->pcg) && !splpak::opt_get("SPLPAK_PCG_EAGER");
    {   // (lazy: the half stencil is cleared when -- if -- it is assembled)
        const long long skip = lazy ? (long long)(p->rhs - p->comm) : 0;
        SPLPAK_HIP_TRY(hipMemsetAsync(p->comm + skip, 0, sizeof(double) * (size_t)(p->lenG + p->lenH - skip), st), SPLPAK_E_NODEVICE);
    }
    // (after the memset: the early clear of the factor arena that prefit starts on another stream is ordered behind this point of
    //  `st`, and the two used to share the memory system -- 0.07 ms of clearing took 0.7 ms beside it)
    if (p->prefit_fn) SPLPAK_HIP_TRY(p->prefit_fn(p, st, p->fn_user), SPLPAK_E_NODEVICE);
    stamp(0);
    SPLPAK_HIP_TRY(launch_bin_points(g, ndata, x, l1xdat, y, w, p->s, p->scalH, st), SPLPAK_E_NODEVICE);
    if (p->pcg) SPLPAK_HIP_TRY(pcg_sum_w2(p, st), SPLPAK_E_NODEVICE);       // (rides the histogram's all-reduce)
    stamp(1);
    bool rows_fit = p->rows_only || lazy;              // the normal equations are not assembled (yet)
    auto assemble_now = [&]() -> int {                 // (lazy fits: everything the eager order would have written; same kernels, same bits)
        SPLPAK_HIP_TRY(hipMemsetAsync(p->comm, 0, sizeof(double) * (size_t)(p->lenG + p->lenH), st), SPLPAK_E_NODEVICE);
        SPLPAK_HIP_TRY(launch_gram(g, p->s, p->gscratch, p->gscratch_doubles, smooth, p->nst, p->rhs, p->hist, p->scalH, st), SPLPAK_E_NODEVICE);
        // (the weights of the constraint rows again, from THIS histogram: the rows' one differs from it in the last bits)
        SPLPAK_HIP_TRY(launch_sparse_mark(g, p->hist, p->scalH, p->xtrap, p->dcw, p->spf, st), SPLPAK_E_NODEVICE);
        SPLPAK_HIP_TRY(launch_constraint_rows(g, p->dcw, p->spf, p->ctab, p->nst, p->scalG, st), SPLPAK_E_NODEVICE);
        rows_fit = false;
        return 0;
    };
    if (rows_fit) {
        // the histogram from the rows (tile by tile); the right-hand side follows below, when the reduced histogram has gone
        if (smooth) {
            SPLPAK_HIP_TRY(rowsop_histogram(g, p->rowsop, p->s, p->hist, st), SPLPAK_E_NODEVICE);
            SPLPAK_HIP_TRY(launch_hist_total(g, p->hist, p->scalH, st), SPLPAK_E_NODEVICE);
        }
        SPLPAK_HIP_TRY(hipMemsetAsync(p->xvec, 0, sizeof(double) * (size_t)b.npad, st), SPLPAK_E_NODEVICE);
        SPLPAK_HIP_TRY(rowsop_apply(g, p->rowsop, p->s, p->xvec, p->dcw, p->spf, p->ctab, false, p->rhs, st), SPLPAK_E_NODEVICE);   // A^T W^2 y
    } else
        SPLPAK_HIP_TRY(launch_gram(g, p->s, p->gscratch, p->gscratch_doubles, smooth, p->nst, p->rhs, p->hist, p->scalH, st), SPLPAK_E_NODEVICE);
    stamp(2);
    double hs[2 * SC_COUNT];
    if (p->world > 1) {
        const double one = 1.0;
        if (lerr != 0)
            SPLPAK_HIP_TRY(hipMemcpyAsync(p->scalH + SC_ERRFLAG, &one, sizeof(double), hipMemcpyHostToDevice, st), SPLPAK_E_NODEVICE);
        if (int r = do_allreduce(p, p->hist, p->lenH, st)) return r;
        SPLPAK_HIP_TRY(hipMemcpyAsync(hs + SC_COUNT, p->scalH, sizeof(double) * SC_COUNT, hipMemcpyDeviceToHost, st), SPLPAK_E_NODEVICE);
        SPLPAK_HIP_TRY(hipStreamSynchronize(st), SPLPAK_E_NODEVICE);
        if (hs[SC_COUNT + SC_ERRFLAG] != 0.0) {
            if (lerr != 0) return lerr;
            set_error("another rank of the sharded fit rejected its arguments");
            return SPLPAK_E_COMM;
        }
    }
    if (smooth && (p->rank == 0 || p->pcg))      // (every rank of an iterating fit: the preconditioner's second moment)
        SPLPAK_HIP_TRY(launch_sparse_mark(g, p->hist, p->scalH, p->xtrap, p->dcw, p->spf, st), SPLPAK_E_NODEVICE);
    if (smooth && p->rank == 0) {
        if (rows_fit) SPLPAK_HIP_TRY(launch_count_sparse(g, p->spf, p->scalG, st), SPLPAK_E_NODEVICE);      // (the rows are only counted)
        else SPLPAK_HIP_TRY(launch_constraint_rows(g, p->dcw, p->spf, p->ctab, p->nst, p->scalG, st), SPLPAK_E_NODEVICE);
    }
    stamp(3);
    if (int r = do_allreduce(p, p->rows_only ? p->rhs : p->nst, p->lenG, st)) return r;

    SPLPAK_HIP_TRY(hipMemcpyAsync(hs, p->scalG, sizeof(double) * SC_COUNT, hipMemcpyDeviceToHost, st), SPLPAK_E_NODEVICE);
    // scalG and scalH are not adjacent (hist sits between): fetch scalH separately
    SPLPAK_HIP_TRY(hipMemcpyAsync(hs + SC_COUNT, p->scalH, sizeof(double) * SC_COUNT, hipMemcpyDeviceToHost, st), SPLPAK_E_NODEVICE);
    SPLPAK_HIP_TRY(hipStreamSynchronize(st), SPLPAK_E_NODEVICE);
    const double rows_data = hs[SC_COUNT + SC_NROWS_DATA];
    const double rows_cons = hs[SC_NROWS_CONS];
    auto t1 = clk::now();
    if (info) {
        info[0] = rows_data;
        info[1] = rows_cons;
        info[5] = std::chrono::duration<double>(t1 - t0).count();
    }
    // suprls error 33 "array has too few rows" (:1650-1654) -> 107 (:1053-1058)
    if (rows_data + rows_cons < (double)g.ncol) {
        SPLPAK_HIP_TRY(hipMemsetAsync(coef_dev, 0, sizeof(double) * (size_t)g.ncol, st), SPLPAK_E_NODEVICE);
        SPLPAK_HIP_TRY(hipStreamSynchronize(st), SPLPAK_E_NODEVICE);
        set_error("fewer rows than coefficients (suprls 33)");
        return 107;
    }

    // ---- solve + refinement, around any solver of N z = v -----------------
    const double inf = std::numeric_limits<double>::infinity();
    int steps = 0;
    double last_rel = 0.0, ratio = 0.0;
    // converged: the (estimated) remaining error is below tol, or the corrections sit at the rounding
    // floor; diverged: they stopped contracting while still large.  A solve that is still contracting
    // after the nominal number of steps goes on up to max_refine_hard; if even that leaves an estimated
    // error above the parity bar the fit is reported as failed (107) instead of returning coefficients
    // that silently miss it.
    bool converged = false, diverged = false, stagnated = false;
    // solve(v, first): v <- N^-1 v; 0, a status to return (negative, SPLPAK_E_COMM), or 1 = this solver gives up (the iteration)
    auto solve_and_refine = [&](auto &&solve) -> int {
        steps = 0;
        last_rel = 0.0;
        ratio = 0.0;
        double prev_rel = inf;
        converged = p->max_refine == 0;
        diverged = stagnated = false;
        SPLPAK_HIP_TRY(hipMemsetAsync(p->xvec, 0, sizeof(double) * (size_t)b.npad, st), SPLPAK_E_NODEVICE);
        SPLPAK_HIP_TRY(hipMemcpyAsync(p->xvec, p->rhs, sizeof(double) * (size_t)g.ncol, hipMemcpyDeviceToDevice, st), SPLPAK_E_NODEVICE);
        stamp(6);
        if (int r = solve(p->xvec, true)) return r;
        stamp(7);
        for (int it = 0; it < p->max_refine_hard && !converged; ++it) {
            // (the scalars behind rho travel with it through the all-reduce: zeroed too, or every collective doubles them)
            SPLPAK_HIP_TRY(hipMemsetAsync(p->rho, 0, sizeof(double) * (size_t)(b.npad + SC_COUNT), st), SPLPAK_E_NODEVICE);
            SPLPAK_HIP_TRY(plan_rows_residual(p, p->s, p->xvec, smooth && p->rank == 0, p->rho, st), SPLPAK_E_NODEVICE);
            if (int r = do_allreduce(p, p->rho, p->lenR, st)) return r;
            if (int r = solve(p->rho, false)) return r;
            SPLPAK_HIP_TRY(launch_axpy_absmax(g.ncol, p->xvec, p->rho, p->small, st), SPLPAK_E_NODEVICE);
            double am[2];
            SPLPAK_HIP_TRY(hipMemcpyAsync(am, p->small, 2 * sizeof(double), hipMemcpyDeviceToHost, st), SPLPAK_E_NODEVICE);
            SPLPAK_HIP_TRY(hipStreamSynchronize(st), SPLPAK_E_NODEVICE);
            ++steps;
            last_rel = (am[1] > 0.0) ? am[0] / am[1] : 0.0;
            if (splpak::opt_get("SPLPAK_DEBUG"))
                fprintf(stderr, "[splpak] refinement step %d: |dx|/|x| = %.3e\n", steps, last_rel);
            if (!(last_rel == last_rel)) break;                   // NaN
            if (last_rel <= p->tol) { converged = true; break; }
            if (it >= 1) {
                // linear convergence: after this step the error is ~ dx * ratio / (1 - ratio); stop as soon
                // as that estimate is below the tolerance instead of paying for one more solve
                ratio = last_rel / prev_rel;
                if (ratio < 0.9 && last_rel * ratio / (1.0 - ratio) <= p->tol) { converged = true; break; }
                if (ratio >= 0.9) {                               // stagnation: fine at the rounding floor, a failure if the
                    diverged = last_rel > 1e-8;                   // corrections are still large; in between (1e-10 .. 1e-8) the
                    converged = !diverged;                        // MEASURED backward error decides below (round-2 advice: the
                    stagnated = converged && last_rel > 1e-10;    // estimate alone let coefficients that miss the bar through)
                    break;
                }
                // 0.5 .. 0.9: an ill-conditioned grid whose corrections still shrink -- go on (up to max_refine_hard):
                // stopping here left 1-D grids of 2 000-3 000 nodes 1e-7 .. 1e-10 away from the converged solution
                // (randomized sweep, tools/fuzz_parity.py big)
            }
            prev_rel = last_rel;
            if (it + 1 >= p->max_refine && it + 1 < p->max_refine_hard && splpak::opt_get("SPLPAK_DEBUG"))
                fprintf(stderr, "[splpak] still contracting after %d steps: continuing\n", it + 1);
        }
        return 0;
    };

    // ---- the iteration (pcg.hip), where the plan has it --------------------
    bool solved = false;
    auto t2 = t1;
    // Where a factorisation stands behind the iteration, the attempt is skipped in the regime in which it is known to stagnate or
    // crawl (between 0 and ~1.6 constraint rows per column; it works with none and from ~1.7 on)
    bool try_iteration = p->pcg != nullptr;
    if (try_iteration && p->solver_mode == 3 && !splpak::opt_get("SPLPAK_PCG_ALWAYS")) {
        const double rpc = rows_cons / (double)g.ncol;
        // (a factorisation of seconds -- 24^4: 4.5 s, 28^4: 18 s -- is worth a patient attempt where the iteration only crawls: 24^4 at
        //  1.5 / 1.4 / 1.33 rows per column 1.0 / 2.1 / 3.1 s; at 1.27 it gives up after 2.8 s.  45 TFLOP/s: what the factorisation sustains)
        const double fac_s = p->factor_flop / 45.0e12;
        const double lo = fac_s >= 10.0 ? 1.3 : (fac_s >= 1.0 ? 1.35 : 1.6);
        if (rows_cons > 0.0 && rpc < lo) {
            try_iteration = false;
            if (splpak::opt_get("SPLPAK_DEBUG")) fprintf(stderr, "[splpak] %.2f constraint rows per column: the factorisation without an attempt of the iteration\n", rpc);
        }
    }
    if (lazy && !try_iteration)
        if (int r = assemble_now()) return r;
    if (try_iteration) {
        SPLPAK_HIP_TRY(pcg_prepare(p, p->pcg, hs[SC_COUNT + SC_SUMW2], smooth, rows_fit, st), SPLPAK_E_NODEVICE);
        if (pcg_singular(p->pcg)) {
            // A box taken out of the ASSEMBLED normal equations -- a principal submatrix of N -- is not positive definite by the pivot
            // test of the factorisations: neither is N (a column without data and, with xtrap = 0, without a constraint row; the
            // reference's "system is singular", suprls 34 -> 107).  The iteration would still run to a minimiser with arbitrary
            // values on what the rows do not see: the factorisation gets to say 107, or the plan that has none says it here
            // (randomised sweep tools/pcg/fuzz_pcg.py: 1-D, 150 nodes, 361 points, xtrap = 0)
            if (p->solver_mode == 2) {
                SPLPAK_HIP_TRY(hipMemsetAsync(coef_dev, 0, sizeof(double) * (size_t)g.ncol, st), SPLPAK_E_NODEVICE);
                SPLPAK_HIP_TRY(hipStreamSynchronize(st), SPLPAK_E_NODEVICE);
                set_error("normal equations not positive definite (suprls 34): a block of them failed the pivot test");
                return 107;
            }
            try_iteration = false;
        }
    }
    if (try_iteration) {
        const double tol_first = splpak::opt_get("SPLPAK_PCG_TOL1") ? atof(splpak::opt_get("SPLPAK_PCG_TOL1")) : 1e-11;
        const double tol_next = splpak::opt_get("SPLPAK_PCG_TOL2") ? atof(splpak::opt_get("SPLPAK_PCG_TOL2")) : 1e-3;
        const int r = solve_and_refine([&](double *v, bool first) -> int { return pcg_solve(p, p->pcg, v, first ? tol_first : tol_next, smooth, st); });
        if (r != 0 && r != 1) return r;
        double est = last_rel;
        if (steps >= 2 && ratio > 0.0 && ratio < 1.0) est = last_rel * ratio / (1.0 - ratio);
        solved = r == 0 && last_rel == last_rel && !diverged && (converged || est <= 1e-10);
        if (!solved && p->solver_mode == 2) {
            SPLPAK_HIP_TRY(hipMemsetAsync(coef_dev, 0, sizeof(double) * (size_t)g.ncol, st), SPLPAK_E_NODEVICE);
            SPLPAK_HIP_TRY(hipStreamSynchronize(st), SPLPAK_E_NODEVICE);
            double ps[6];
            pcg_stats(p->pcg, ps);
            char buf[320];
            snprintf(buf, sizeof buf, "the iterative solve did not converge (%.0f iterations in %.0f solves, last preconditioned residual %.1e, last correction %.1e) "
                     "and no factorisation of this grid fits the device: data too clustered for the separable preconditioner", ps[0], ps[1], ps[3], last_rel);
            set_error(buf);
            if (info) { info[2] = steps; info[3] = last_rel; }
            return 107;
        }
        if (!solved && splpak::opt_get("SPLPAK_DEBUG")) fprintf(stderr, "[splpak] the iteration gave up: factorisation instead\n");
    }

    // ---- factorisation --------------------------------------------------
    if (!solved && rows_fit && lazy)
        if (int r = assemble_now()) return r;
    if (!solved) {
        SPLPAK_HIP_TRY(hipMemsetAsync(p->info, 0, 2 * sizeof(int), st), SPLPAK_E_NODEVICE);
        SPLPAK_HIP_TRY(hipMemcpyAsync(p->small + 2, &inf, sizeof(double), hipMemcpyHostToDevice, st), SPLPAK_E_NODEVICE);
        stamp(4);
        SPLPAK_HIP_TRY(p->expand_fn ? p->expand_fn(p, st, p->fn_user) : launch_expand(g, p->nst, b, p->dm, st), SPLPAK_E_NODEVICE);
        stamp(5);
        SPLPAK_HOOK_TRY(p->factor_fn ? p->factor_fn(p, p->info, p->small + 2, st, p->fn_user) : band_cholesky(b, p->info, p->small + 2, st, &p->stats));
        int hinfo = 0;
        double minpiv = 0.0;
        SPLPAK_HIP_TRY(hipMemcpyAsync(&hinfo, p->info, sizeof(int), hipMemcpyDeviceToHost, st), SPLPAK_E_NODEVICE);
        SPLPAK_HIP_TRY(hipMemcpyAsync(&minpiv, p->small + 2, sizeof(double), hipMemcpyDeviceToHost, st), SPLPAK_E_NODEVICE);
        SPLPAK_HIP_TRY(hipStreamSynchronize(st), SPLPAK_E_NODEVICE);
        t2 = clk::now();
        if (info) {
            info[4] = minpiv;
            info[6] = std::chrono::duration<double>(t2 - t1).count();
        }
        if (hinfo != 0) {
            // not positive definite: the reference's "system is singular" (suprls 34 -> 107)
            SPLPAK_HIP_TRY(hipMemsetAsync(coef_dev, 0, sizeof(double) * (size_t)g.ncol, st), SPLPAK_E_NODEVICE);
            SPLPAK_HIP_TRY(hipStreamSynchronize(st), SPLPAK_E_NODEVICE);
            set_error("normal equations not positive definite (suprls 34)");
            return 107;
        }
        const int r = solve_and_refine([&](double *v, bool) -> int {
            SPLPAK_HOOK_TRY(p->solve_fn ? p->solve_fn(p, v, p->tmp, st, p->fn_user) : band_solve(b, v, p->tmp, st));
            return 0;
        });
        if (r != 0) return r;
    }
    // estimated error left after the last step (exact 0 when it met the tolerance outright)
    double est_err = last_rel;
    if (steps >= 2 && ratio > 0.0 && ratio < 1.0) est_err = last_rel * ratio / (1.0 - ratio);
    const bool unconverged = !converged && !diverged && last_rel == last_rel && est_err > 1e-10;
    SPLPAK_HIP_TRY(launch_to_reference_order(g, p->xvec, coef_dev, st), SPLPAK_E_NODEVICE);   // internal -> caller's dimension order
    // Diagnostics from one more pass over the rows at the returned coefficients:
    //  * residual norm ||rows * coef - rhs||_2 over data AND constraint rows: what the reference computes
    //    as `reserr` (suprls :1693) and then drops (splcw :690, :1052)
    //  * optimality residual: the gradient rho = A^T W (W y - W A x) - C^T C x of the least-squares functional,
    //    recomputed from the rows, as a componentwise backward error max_i |rho_i| / ((|N||x|)_i + |A^T W^2 y|_i)
    //    -- 0 at the minimiser the reference computes; a MEASURED statement about the returned
    //    coefficients (the refinement's stopping rule is an estimate)
    double ssq = 0.0, omega = 0.0;
    if (info || stagnated) {
        double *scalR = p->rho + b.npad;
        // the backward error's denominators (|N| |x| + |rhs|: a pass over the half stencil) need the coefficients only; into the
        // solves' scratch vector.  (On a stream of their own beside the residual pass they gained nothing -- the two kernels
        // slowed each other down by what the overlap saved -- and one more stream per plan is not free: round 5)
        if (rows_fit) {
            // from the rows: |A|^T W^2 |A| |x| + |C|^T |C| |x| + |rhs| (this rank's points; the residual's all-reduce below does not
            // carry it -- a sharded rows-only fit normalises by its own shard's terms + the constraint rows on rank 0, a lower bound
            // of the sum, i.e. a pessimistic backward error)
            SPLPAK_HIP_TRY(rowsop_backward_denominators(g, p->rowsop, p->s, p->xvec, p->rhs, p->dcw, p->spf, p->ctab, smooth && p->rank == 0,
                                                        pcg_scratch(p->pcg, 0), pcg_scratch(p->pcg, 1), p->tmp, st), SPLPAK_E_NODEVICE);
        } else
            SPLPAK_HIP_TRY(launch_backward_denominators(g, p->nst, p->xvec, p->rhs, p->tmp, st), SPLPAK_E_NODEVICE);
        SPLPAK_HIP_TRY(hipMemsetAsync(p->rho, 0, sizeof(double) * (size_t)(b.npad + SC_COUNT), st), SPLPAK_E_NODEVICE);
        hipEvent_t r0 = stamps ? p->evStage[8] : nullptr, r1 = stamps ? p->evStage[9] : nullptr;   // (created with the other stage events)
        if (r0 && r1) (void)hipEventRecord(r0, st);
        if (p->rowsop && p->ctab && (!p->rcell || !splpak::opt_get("SPLPAK_RESIDUAL_CELLS")))      // (4-D: tile by tile, as the refinement's passes; 8.3 -> 1 ms at 32^4)
            SPLPAK_HIP_TRY(rowsop_residual(g, p->rowsop, p->s, p->xvec, p->dcw, p->spf, p->ctab, smooth && p->rank == 0, p->rho, scalR, p->e2buf, st),
                           SPLPAK_E_NODEVICE);
        else
            SPLPAK_HIP_TRY(launch_residual(g, p->s, p->xvec, p->rcell, p->dcw, p->spf, p->ctab, smooth && p->rank == 0,
                                           p->tbuf, p->rho, scalR, p->e2buf, st), SPLPAK_E_NODEVICE);
        if (r0 && r1) {
            (void)hipEventRecord(r1, st);
            (void)hipEventSynchronize(r1);
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, r0, r1) == hipSuccess) p->stage_ms[4] = ms;
        }
        if (int r = do_allreduce(p, p->rho, p->lenR, st)) return r;
        SPLPAK_HIP_TRY(launch_backward_error(g, p->tmp, p->rho, p->small + 3, st), SPLPAK_E_NODEVICE);
        SPLPAK_HIP_TRY(hipMemcpyAsync(&ssq, scalR, sizeof(double), hipMemcpyDeviceToHost, st), SPLPAK_E_NODEVICE);
        SPLPAK_HIP_TRY(hipMemcpyAsync(&omega, p->small + 3, sizeof(double), hipMemcpyDeviceToHost, st), SPLPAK_E_NODEVICE);
    }
    SPLPAK_HIP_TRY(hipStreamSynchronize(st), SPLPAK_E_NODEVICE);
    auto t3 = clk::now();
    if (stamps) {
        // all stages are complete (the stream was synchronised above); residual pass: timed separately below
        const int pairs[5][2] = {{0, 1}, {1, 2}, {2, 3}, {4, 5}, {6, 7}};
        const int slot[5] = {0, 1, 2, 3, 5};
        for (int i = 0; i < 5; ++i) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p->evStage[pairs[i][0]], p->evStage[pairs[i][1]]) == hipSuccess) p->stage_ms[slot[i]] = ms;
            else (void)hipGetLastError();
        }
    }
    if (info) {
        info[2] = steps;
        info[3] = last_rel;
        info[7] = std::chrono::duration<double>(t3 - t2).count();
        info[8] = std::sqrt(ssq);
        info[9] = omega;
    }
    // a correction that is still large means the factor did not precondition the problem
    // (numerically singular normal equations): the reference's "suprls failure"
    if (!(last_rel == last_rel) || diverged) {
        set_error("iterative refinement diverged: numerically singular normal equations");
        return 107;
    }
    if (stagnated && !(omega <= 1e-10)) {
        char buf[200];
        snprintf(buf, sizeof buf, "iterative refinement stagnated at corrections of %.2e with a backward error of %.1e > 1e-10", last_rel, omega);
        set_error(buf);
        return 107;
    }
    if (unconverged) {
        char buf[200];
        snprintf(buf, sizeof buf, "iterative refinement did not converge in %d steps: last correction %.2e, contraction %.2f, "
                 "estimated error %.1e > 1e-10", steps, last_rel, ratio, est_err);
        set_error(buf);
        return 107;
    }
    return 0;
#undef SPLPAK_HOOK_TRY
}

// ---------------------------------------------------------------------------
// one-shot host entry points
// ---------------------------------------------------------------------------

// The one-shot entry keeps its plan (band factor storage, sort scratch, staging buffers: 28 GB at
// 64^3) between calls: a caller that fits the same grid again -- the reference's usage pattern is one
// `initialize` per data set -- pays the allocation once.  Released by splpak_shutdown(); disabled by
// SPLPAK_NO_PLAN_CACHE.  Calls from several threads are serialised.
namespace {
struct HostFitCache {
    std::mutex mu;
    splpak_plan *plan = nullptr;
    int ndim = 0, nodes[MAXD] = {0, 0, 0, 0}, dev = -1;
    double xmin[MAXD] = {0, 0, 0, 0}, xmax[MAXD] = {0, 0, 0, 0}, xtrap = 0.0;
    double *dx = nullptr, *dy = nullptr, *dw = nullptr, *dc = nullptr;
    long long cap_x = 0, cap_y = 0, cap_w = 0, cap_c = 0;
    void release()
    {
        for (double **q : {&dx, &dy, &dw, &dc}) { if (*q) (void)hipFree(*q); *q = nullptr; }
        cap_x = cap_y = cap_w = cap_c = 0;
        if (plan) splpak_plan_destroy(plan);
        plan = nullptr;
    }
};
HostFitCache g_hostfit;
}  // namespace

}  // extern "C"

// An allocation failed: give back what the one-shot entry keeps between calls (round-2 advice).  Not while a one-shot
// fit is running (it holds the lock and has released its old plan itself).  true = something was released.
static thread_local bool t_in_fit_host = false;      // this thread holds g_hostfit.mu (try_lock on a mutex one owns is undefined)

bool splpak::release_cached_plan_for_memory()
{
    if (t_in_fit_host) return false;
    std::unique_lock<std::mutex> lock(g_hostfit.mu, std::try_to_lock);
    if (!lock.owns_lock() || !g_hostfit.plan) return false;
    g_hostfit.release();
    return true;
}

extern "C" {

static int32_t fit_host(int32_t ndim, const double *xdata, int32_t l1xdat, const double *ydata,
                        const double *wdata, int64_t ndata, const double *xmin, const double *xmax,
                        const int32_t *nodes, double xtrap, double *coef, int64_t ncf, int64_t nwrk,
                        double *hist_out, double *info)
{
    if (!nodes || !xmin || !xmax) { set_error("null argument"); return SPLPAK_E_BADARG; }
    Grid g;
    long long ncol = 0;
    const int v = build_grid(ndim, nodes, xmin, xmax, g, &ncol);       // 101, 102, 103
    if (v != 0) return v;
    if (ncol > ncf) return 104;                                        // :751-756
    if (ndata < 1) return 105;                                         // :759-764
    if (nwrk >= 0) {                                                   // :772-781
        const long long nwrk1 = (xtrap != 0.0) ? ncol + 1 : 1;
        if (nwrk - nwrk1 + 1 < 1) return 106;
    }
    if (!xdata || !ydata || !coef) { set_error("null argument"); return SPLPAK_E_BADARG; }
    if (l1xdat < ndim) { set_error("l1xdat < ndim"); return SPLPAK_E_BADARG; }
    if (wdata && wdata[0] < 0.0) wdata = nullptr;                      // :581-588, :796
    if (int r = device_ready()) return r;

    HostFitCache &hc = g_hostfit;
    std::lock_guard<std::mutex> lock(hc.mu);
    struct InFit { InFit() { t_in_fit_host = true; } ~InFit() { t_in_fit_host = false; } } in_fit;
    int dev = 0;
    (void)hipGetDevice(&dev);
    // the switches are part of the cache key as a whole (a plan keeps the snapshot it was created with)
    const Options cur = options_snapshot();
    bool same = hc.plan && hc.dev == dev && hc.ndim == ndim && hc.xtrap == xtrap && hc.plan->max_ndata >= ndata && hc.plan->opt == cur;
    for (int d = 0; same && d < ndim; ++d)
        same = hc.nodes[d] == nodes[d] && hc.xmin[d] == xmin[d] && hc.xmax[d] == xmax[d];
    if (!same) {
        hc.release();
        int rc = splpak_plan_create(ndim, nodes, xmin, xmax, xtrap, ndata, nullptr, 0, &hc.plan);
        if (rc != 0) { hc.plan = nullptr; return rc; }
        hc.dev = dev;
        hc.ndim = ndim;
        hc.xtrap = xtrap;
        for (int d = 0; d < ndim; ++d) { hc.nodes[d] = nodes[d]; hc.xmin[d] = xmin[d]; hc.xmax[d] = xmax[d]; }
    }
    splpak_plan *p = hc.plan;
    auto grow = [&](double **q, long long &cap, long long need) {
        if (*q && cap >= need) return true;
        if (*q) (void)hipFree(*q);
        *q = nullptr;
        cap = 0;
        if (!hip_ok(hipMalloc((void **)q, sizeof(double) * (size_t)need), "hipMalloc of the staging buffers")) return false;
        cap = need;
        return true;
    };
    const bool ok = grow(&hc.dx, hc.cap_x, (long long)ndata * l1xdat) && grow(&hc.dy, hc.cap_y, ndata) &&
                    (!wdata || grow(&hc.dw, hc.cap_w, ndata)) && grow(&hc.dc, hc.cap_c, ncol);
    if (!ok) { hc.release(); return SPLPAK_E_NOMEM; }
    hipError_t e = hipMemcpy(hc.dx, xdata, sizeof(double) * (size_t)ndata * l1xdat, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(hc.dy, ydata, sizeof(double) * (size_t)ndata, hipMemcpyHostToDevice);
    if (e == hipSuccess && wdata) e = hipMemcpy(hc.dw, wdata, sizeof(double) * (size_t)ndata, hipMemcpyHostToDevice);
    if (!hip_ok(e, "hipMemcpy H2D")) { hc.release(); return SPLPAK_E_NODEVICE; }
    int rc = splpak_plan_fit_dev(p, hc.dx, l1xdat, hc.dy, wdata ? hc.dw : nullptr, ndata, hc.dc, nullptr, info);
    if (rc == 0 || rc == 107) {
        e = hipMemcpy(coef, hc.dc, sizeof(double) * (size_t)ncol, hipMemcpyDeviceToHost);
        if (e == hipSuccess && hist_out && xtrap != 0.0)
            e = hipMemcpy(hist_out, p->hist, sizeof(double) * (size_t)ncol, hipMemcpyDeviceToHost);
        if (!hip_ok(e, "hipMemcpy D2H")) rc = SPLPAK_E_NODEVICE;
    }
    if (rc < 0 || splpak::opt_get("SPLPAK_NO_PLAN_CACHE")) hc.release();
    return rc;
}

int32_t splpak_fit_f64(int32_t ndim, const double *xdata, int32_t l1xdat, const double *ydata,
                       const double *wdata, int64_t ndata, const double *xmin, const double *xmax,
                       const int32_t *nodes, double xtrap, double *coef, int64_t ncf, int64_t nwrk,
                       double *hist_out, double *info)
{
    return fit_host(ndim, xdata, l1xdat, ydata, wdata, ndata, xmin, xmax, nodes, xtrap, coef, ncf,
                    nwrk, hist_out, info);
}

int32_t splpak_fit_f32(int32_t ndim, const float *xdata, int32_t l1xdat, const float *ydata,
                       const float *wdata, int64_t ndata, const float *xmin, const float *xmax,
                       const int32_t *nodes, float xtrap, float *coef, int64_t ncf, int64_t nwrk,
                       float *hist_out, double *info)
{
    // REAL32 storage, f64 arithmetic: widen on the host (the arrays are small
    // next to the factorisation), run the f64 path, narrow the results.
    if (ndim < 1) return 101;
    if (ndim > MAXD) return SPLPAK_E_UNSUPPORTED;
    if (!nodes || !xmin || !xmax) { set_error("null argument"); return SPLPAK_E_BADARG; }
    double xmn[MAXD], xmx[MAXD];
    long long ncol = 1;
    for (int d = 0; d < ndim; ++d) { xmn[d] = xmin[d]; xmx[d] = xmax[d]; ncol *= nodes[d] > 0 ? nodes[d] : 1; }
    const bool have = xdata && ydata && coef && ndata >= 1 && ncol <= ncf;
    std::vector<double> X, Y, W, Cf, H;
    if (have) {
        X.assign(xdata, xdata + (size_t)ndata * l1xdat);
        Y.assign(ydata, ydata + (size_t)ndata);
        if (wdata && wdata[0] >= 0.0f) W.assign(wdata, wdata + (size_t)ndata);
        Cf.resize((size_t)ncol);
        if (hist_out) H.resize((size_t)ncol);
    }
    const int rc = fit_host(ndim, have ? X.data() : nullptr, l1xdat, have ? Y.data() : nullptr,
                            W.empty() ? nullptr : W.data(), ndata, xmn, xmx, nodes, (double)xtrap,
                            have ? Cf.data() : nullptr, ncf, nwrk, H.empty() ? nullptr : H.data(), info);
    if (have && (rc == 0 || rc == 107)) {
        for (long long i = 0; i < ncol; ++i) coef[i] = (float)Cf[(size_t)i];
        if (hist_out && xtrap != 0.0f)
            for (long long i = 0; i < ncol; ++i) hist_out[i] = (float)H[(size_t)i];
    }
    return rc;
}

// returns 0/101/102/103/104 exactly like splde's checks (:1166-1194)
static int eval_validate(int32_t ndim, const int32_t *nderiv, const double *xmin, const double *xmax,
                         const int32_t *nodes, Grid &g)
{
    int v = build_grid(ndim, nodes, xmin, xmax, g, nullptr);
    if (v != 0) return v;
    if (nderiv)
        for (int d = 0; d < ndim; ++d)
            if (nderiv[d] < 0 || nderiv[d] > 2) v = 104;
    return v;
}

// host only: the 4-entry value table of a 1-D grid at n points, as the evaluation kernels select it and in the general form
int32_t splpak_debug_window_values(int32_t nodes, double xmin, double xmax, int64_t n, const double *x, int32_t *ws_out,
                                   double *used4, double *general4, int32_t *form_out)
{
    if (!x || !ws_out || !used4 || !general4 || !form_out || n < 0) { set_error("null argument"); return SPLPAK_E_BADARG; }
    Grid g;
    const int v = build_grid(1, &nodes, &xmin, &xmax, g, nullptr);
    if (v != 0) return v;
    for (int64_t i = 0; i < n; ++i) {
        int form = 0;
        ws_out[i] = window_table_selected(g, 0, x[i], used4 + 4 * i, form);
        form_out[i] = form;
        const int wg = window_table_value(g, 0, x[i], general4 + 4 * i);
        if (wg != ws_out[i]) { set_error("window starts of the two forms differ"); return SPLPAK_E_BADARG; }
    }
    return 0;
}

int32_t splpak_eval_dev_f64(int32_t ndim, int64_t nq, const double *xq_dev, int32_t ldxq,
                            const int32_t *nderiv, const double *coef_dev, const double *xmin,
                            const double *xmax, const int32_t *nodes, double *out_dev, void *stream)
{
    if (!nodes || !xmin || !xmax) { set_error("null argument"); return SPLPAK_E_BADARG; }
    Grid g;
    const int v = eval_validate(ndim, nderiv, xmin, xmax, nodes, g);
    if (v != 0 && v != 104) {
        if (v > 0 && out_dev && nq > 0) (void)hipMemsetAsync(out_dev, 0, sizeof(double) * (size_t)nq, (hipStream_t)stream);
        return v;
    }
    if (nq <= 0) return v;
    if (!xq_dev || !coef_dev || !out_dev) { set_error("null argument"); return SPLPAK_E_BADARG; }
    if (ldxq < ndim) { set_error("ldxq smaller than ndim"); return SPLPAK_E_BADARG; }
    if (int r = device_ready()) return r;
    SPLPAK_HIP_TRY(launch_eval(g, nq, xq_dev, ldxq, nderiv, coef_dev, out_dev, (hipStream_t)stream), SPLPAK_E_NODEVICE);
    return v;
}

int32_t splpak_eval_dev_f32(int32_t ndim, int64_t nq, const float *xq_dev, int32_t ldxq,
                            const int32_t *nderiv, const float *coef_dev, const float *xmin_f,
                            const float *xmax_f, const int32_t *nodes, float *out_dev, void *stream)
{
    if (!nodes || !xmin_f || !xmax_f) { set_error("null argument"); return SPLPAK_E_BADARG; }
    if (ndim < 1) return 101;
    if (ndim > MAXD) return SPLPAK_E_UNSUPPORTED;
    double xmin[MAXD], xmax[MAXD];
    for (int d = 0; d < ndim; ++d) { xmin[d] = (double)xmin_f[d]; xmax[d] = (double)xmax_f[d]; }
    Grid g;
    const int v = eval_validate(ndim, nderiv, xmin, xmax, nodes, g);
    if (v != 0 && v != 104) {
        if (v > 0 && out_dev && nq > 0) (void)hipMemsetAsync(out_dev, 0, sizeof(float) * (size_t)nq, (hipStream_t)stream);
        return v;
    }
    if (nq <= 0) return v;
    if (!xq_dev || !coef_dev || !out_dev) { set_error("null argument"); return SPLPAK_E_BADARG; }
    if (ldxq < ndim) { set_error("ldxq smaller than ndim"); return SPLPAK_E_BADARG; }
    if (int r = device_ready()) return r;
    SPLPAK_HIP_TRY(launch_eval_f32(g, nq, xq_dev, ldxq, nderiv, coef_dev, out_dev, (hipStream_t)stream), SPLPAK_E_NODEVICE);
    return v;
}

static int derivs_nout(int ndim, int order) { return 1 + ndim + (order == 2 ? ndim * (ndim + 1) / 2 : 0); }

int32_t splpak_eval_derivs_dev_f64(int32_t ndim, int64_t nq, const double *xq_dev, int32_t ldxq, int32_t order,
                                   const double *coef_dev, const double *xmin, const double *xmax,
                                   const int32_t *nodes, double *out_dev, int32_t ldout, void *stream)
{
    if (!nodes || !xmin || !xmax) { set_error("null argument"); return SPLPAK_E_BADARG; }
    Grid g;
    const int v = eval_validate(ndim, nullptr, xmin, xmax, nodes, g);
    if (v != 0) {
        if (v > 0 && out_dev && nq > 0 && ldout > 0)
            (void)hipMemsetAsync(out_dev, 0, sizeof(double) * (size_t)nq * (size_t)ldout, (hipStream_t)stream);
        return v;
    }
    if (order < 1 || order > 2) { set_error("order must be 1 (gradient) or 2 (gradient and Hessian)"); return SPLPAK_E_BADARG; }
    if (ldxq < ndim || ldout < derivs_nout(ndim, order)) { set_error("ldxq or ldout too small"); return SPLPAK_E_BADARG; }
    if (nq <= 0) return 0;
    if (!xq_dev || !coef_dev || !out_dev) { set_error("null argument"); return SPLPAK_E_BADARG; }
    if (int r = device_ready()) return r;
    SPLPAK_HIP_TRY(launch_eval_derivs(g, nq, xq_dev, ldxq, order, coef_dev, out_dev, ldout, (hipStream_t)stream), SPLPAK_E_NODEVICE);
    return 0;
}

}  // extern "C"

template <typename T>
static int32_t eval_host(int32_t ndim, int64_t nq, const T *xq, int32_t ldxq, const int32_t *nderiv,
                         const T *coef, const T *xmin_t, const T *xmax_t, const int32_t *nodes, T *out)
{
    if (!nodes || !xmin_t || !xmax_t) { set_error("null argument"); return SPLPAK_E_BADARG; }
    if (ndim < 1) return 101;
    if (ndim > MAXD) return SPLPAK_E_UNSUPPORTED;
    double xmin[MAXD], xmax[MAXD];
    for (int d = 0; d < ndim; ++d) { xmin[d] = (double)xmin_t[d]; xmax[d] = (double)xmax_t[d]; }
    Grid g;
    const int v = eval_validate(ndim, nderiv, xmin, xmax, nodes, g);
    if (v != 0 && v != 104) {
        if (v > 0 && out) for (int64_t i = 0; i < nq; ++i) out[i] = (T)0;
        return v;
    }
    if (nq <= 0) return v;
    if (!xq || !coef || !out) { set_error("null argument"); return SPLPAK_E_BADARG; }
    if (ldxq < ndim) { set_error("ldxq smaller than ndim"); return SPLPAK_E_BADARG; }
    if (int r = device_ready()) return r;
    T *dq = nullptr, *dc = nullptr, *dout = nullptr;
    splpak_plan holder;   // only used as an allocation owner
    bool ok = dev_alloc(&holder, &dq, (size_t)nq * ldxq) && dev_alloc(&holder, &dc, (size_t)g.ncol) &&
              dev_alloc(&holder, &dout, (size_t)nq);
    int rc = v;
    if (!ok) rc = SPLPAK_E_NOMEM;
    if (ok) {
        hipError_t e = hipMemcpy(dq, xq, sizeof(T) * (size_t)nq * ldxq, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(dc, coef, sizeof(T) * (size_t)g.ncol, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            if constexpr (sizeof(T) == 8)
                e = launch_eval(g, nq, (const double *)dq, ldxq, nderiv, (const double *)dc, (double *)dout, nullptr);
            else
                e = launch_eval_f32(g, nq, (const float *)dq, ldxq, nderiv, (const float *)dc, (float *)dout, nullptr);
        }
        if (e == hipSuccess) e = hipMemcpy(out, dout, sizeof(T) * (size_t)nq, hipMemcpyDeviceToHost);
        if (!hip_ok(e, "evaluation")) rc = SPLPAK_E_NODEVICE;
    }
    for (void *q : holder.owned) (void)hipFree(q);
    return rc;
}

template <typename T>
static int32_t eval_derivs_host(int32_t ndim, int64_t nq, const T *xq, int32_t ldxq, int32_t order, const T *coef,
                                const T *xmin_t, const T *xmax_t, const int32_t *nodes, T *out, int32_t ldout)
{
    if (!nodes || !xmin_t || !xmax_t) { set_error("null argument"); return SPLPAK_E_BADARG; }
    if (ndim < 1) return 101;
    if (ndim > MAXD) return SPLPAK_E_UNSUPPORTED;
    double xmin[MAXD], xmax[MAXD];
    for (int d = 0; d < ndim; ++d) { xmin[d] = (double)xmin_t[d]; xmax[d] = (double)xmax_t[d]; }
    Grid g;
    const int v = eval_validate(ndim, nullptr, xmin, xmax, nodes, g);
    if (v != 0) {
        if (v > 0 && out && ldout > 0) for (int64_t i = 0; i < nq * ldout; ++i) out[i] = (T)0;
        return v;
    }
    if (order < 1 || order > 2) { set_error("order must be 1 (gradient) or 2 (gradient and Hessian)"); return SPLPAK_E_BADARG; }
    if (ldxq < ndim || ldout < derivs_nout(ndim, order)) { set_error("ldxq or ldout too small"); return SPLPAK_E_BADARG; }
    if (nq <= 0) return 0;
    if (!xq || !coef || !out) { set_error("null argument"); return SPLPAK_E_BADARG; }
    if (int r = device_ready()) return r;
    T *dq = nullptr, *dc = nullptr, *dout = nullptr;
    splpak_plan holder;   // only used as an allocation owner
    bool ok = dev_alloc(&holder, &dq, (size_t)nq * ldxq) && dev_alloc(&holder, &dc, (size_t)g.ncol) &&
              dev_alloc(&holder, &dout, (size_t)nq * ldout);
    int rc = 0;
    if (!ok) rc = SPLPAK_E_NOMEM;
    if (ok) {
        hipError_t e = hipMemcpy(dq, xq, sizeof(T) * (size_t)nq * ldxq, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(dc, coef, sizeof(T) * (size_t)g.ncol, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemset(dout, 0, sizeof(T) * (size_t)nq * ldout);
        if (e == hipSuccess) {
            if constexpr (sizeof(T) == 8)
                e = launch_eval_derivs(g, nq, (const double *)dq, ldxq, order, (const double *)dc, (double *)dout, ldout, nullptr);
            else
                e = launch_eval_derivs_f32(g, nq, (const float *)dq, ldxq, order, (const float *)dc, (float *)dout, ldout, nullptr);
        }
        if (e == hipSuccess) e = hipMemcpy(out, dout, sizeof(T) * (size_t)nq * ldout, hipMemcpyDeviceToHost);
        if (!hip_ok(e, "evaluation")) rc = SPLPAK_E_NODEVICE;
    }
    for (void *q : holder.owned) (void)hipFree(q);
    return rc;
}

extern "C" {

int32_t splpak_eval_derivs_f64(int32_t ndim, int64_t nq, const double *xq, int32_t ldxq, int32_t order,
                               const double *coef, const double *xmin, const double *xmax,
                               const int32_t *nodes, double *out, int32_t ldout)
{
    return eval_derivs_host<double>(ndim, nq, xq, ldxq, order, coef, xmin, xmax, nodes, out, ldout);
}

int32_t splpak_eval_derivs_f32(int32_t ndim, int64_t nq, const float *xq, int32_t ldxq, int32_t order,
                               const float *coef, const float *xmin, const float *xmax,
                               const int32_t *nodes, float *out, int32_t ldout)
{
    return eval_derivs_host<float>(ndim, nq, xq, ldxq, order, coef, xmin, xmax, nodes, out, ldout);
}

int32_t splpak_eval_f64(int32_t ndim, int64_t nq, const double *xq, int32_t ldxq,
                        const int32_t *nderiv, const double *coef, const double *xmin,
                        const double *xmax, const int32_t *nodes, double *out)
{
    return eval_host<double>(ndim, nq, xq, ldxq, nderiv, coef, xmin, xmax, nodes, out);
}

int32_t splpak_eval_f32(int32_t ndim, int64_t nq, const float *xq, int32_t ldxq,
                        const int32_t *nderiv, const float *coef, const float *xmin,
                        const float *xmax, const int32_t *nodes, float *out)
{
    return eval_host<float>(ndim, nq, xq, ldxq, nderiv, coef, xmin, xmax, nodes, out);
}

int32_t splpak_synth_points_f64(int32_t ndim, int64_t first_point, int64_t ndata, double *xdata_dev,
                                double *ydata_dev, double *wdata_dev, void *stream)
{
    if (ndim < 1 || ndim > MAXD) return SPLPAK_E_UNSUPPORTED;
    if (int r = device_ready()) return r;
    SPLPAK_HIP_TRY(launch_synth_points(ndim, first_point, ndata, xdata_dev, ydata_dev, wdata_dev, (hipStream_t)stream), SPLPAK_E_NODEVICE);
    return 0;
}

int32_t splpak_synth_queries_f64(int32_t ndim, int64_t ndata_before, int64_t first_query, int64_t nq,
                                 double *xq_dev, void *stream)
{
    if (ndim < 1 || ndim > MAXD) return SPLPAK_E_UNSUPPORTED;
    if (int r = device_ready()) return r;
    const long long skip = (long long)ndata_before * (ndim + 2) + (long long)first_query * ndim;
    SPLPAK_HIP_TRY(launch_synth_queries(ndim, skip, nq, xq_dev, (hipStream_t)stream), SPLPAK_E_NODEVICE);
    return 0;
}

int32_t splpak_debug_spd_band_solve_f64(int32_t n, int32_t halfbw, const double *a_lower,
                                        const double *bvec, double *x)
{
    if (n < 1 || halfbw < 0 || !a_lower || !bvec || !x) { set_error("bad argument"); return SPLPAK_E_BADARG; }
    if (int r = device_ready()) return r;
    if (splpak::opt_get("SPLPAK_DEBUG_TWOEND")) {          // the two-ended factorisation (twoend.hip) on the same input
        int hinfo = 0;
        const int rc = twoend_debug_solve(n, halfbw, a_lower, bvec, x, &hinfo);
        if (rc == 0) return hinfo != 0 ? 107 : 0;
        if (rc != 1) { set_error("two-ended band solve failed"); return rc; }
    }
    splpak_plan holder;
    Band b;
    band_bytes(n, halfbw, &b);
    double *dsmall = nullptr, *dx = nullptr, *dtmp = nullptr;
    int *dinfo = nullptr;
    bool ok = dev_alloc(&holder, &b.ab, b.bytes / sizeof(double)) &&
              dev_alloc(&holder, &b.dinv, (size_t)b.nblk * NBLK * NBLK) &&
              dev_alloc(&holder, &b.dinvt, (size_t)b.nblk * NBLK * NBLK) &&
              dev_alloc(&holder, &b.inv64, (size_t)b.nblk * 4 * 64 * 64) &&
              dev_alloc(&holder, &b.mfwd, (size_t)b.nblk * NBLK * NBLK) &&
              dev_alloc(&holder, &b.mbwd, (size_t)b.nblk * NBLK * NBLK) &&
              dev_alloc(&holder, &dsmall, 8) && dev_alloc(&holder, &dx, (size_t)b.npad) &&
              dev_alloc(&holder, &dtmp, (size_t)b.npad) && dev_alloc(&holder, &dinfo, 2);
    int rc = 0;
    if (!ok) rc = SPLPAK_E_NOMEM;
    if (ok) {
        std::vector<double> hb(b.bytes / sizeof(double), 0.0), hx((size_t)b.npad, 0.0);
        for (int j = 0; j < n; ++j)
            for (int i = j; i < n && i - j <= halfbw; ++i)
                hb[(size_t)i + (size_t)j * b.lda] = a_lower[(size_t)i + (size_t)j * n];
        for (int i = n; i < b.npad; ++i) hb[(size_t)i + (size_t)i * b.lda] = 1.0;
        for (int i = 0; i < n; ++i) hx[(size_t)i] = bvec[i];
        const double inf = std::numeric_limits<double>::infinity();
        hipError_t e = hipMemcpy(b.ab, hb.data(), b.bytes, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(dx, hx.data(), sizeof(double) * (size_t)b.npad, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(dsmall + 2, &inf, sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemset(dinfo, 0, 2 * sizeof(int));
        if (e == hipSuccess) e = band_cholesky(b, dinfo, dsmall + 2, nullptr, nullptr);
        if (e == hipSuccess) e = band_solve(b, dx, dtmp, nullptr);
        int hinfo = 0;
        if (e == hipSuccess) e = hipMemcpy(&hinfo, dinfo, sizeof(int), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(hx.data(), dx, sizeof(double) * (size_t)b.npad, hipMemcpyDeviceToHost);
        if (!hip_ok(e, "band solve")) rc = SPLPAK_E_NODEVICE;
        else {
            for (int i = 0; i < n; ++i) x[i] = hx[(size_t)i];
            if (hinfo != 0) rc = 107;
        }
    }
    band_pipeline_destroy(b.pipe);
    for (void *q : holder.owned) (void)hipFree(q);
    return rc;
}

void splpak_shutdown(void)
{
    {
        std::lock_guard<std::mutex> lock(g_hostfit.mu);
        g_hostfit.release();
    }
    eval_scratch_shutdown();
}

int32_t splpak_set_eval_mode(int32_t mode, int64_t chunk)
{
    if (mode < 0 || mode > 2 || chunk < 0) { set_error("bad evaluation mode"); return SPLPAK_E_BADARG; }
    set_eval_mode(mode, chunk);
    return 0;
}

int32_t splpak_last_error_message(char *buf, int32_t buflen)
{
    if (!buf || buflen <= 0) return (int32_t)g_err.size();
    std::strncpy(buf, g_err.c_str(), (size_t)buflen - 1);
    buf[buflen - 1] = '\0';
    return (int32_t)g_err.size();
}

int32_t splpak_device_name(char *buf, int32_t buflen)
{
    if (int r = device_ready()) return r;
    hipDeviceProp_t prop;
    int dev = 0;
    SPLPAK_HIP_TRY(hipGetDevice(&dev), SPLPAK_E_NODEVICE);
    SPLPAK_HIP_TRY(hipGetDeviceProperties(&prop, dev), SPLPAK_E_NODEVICE);
    if (buf && buflen > 0) {
        std::strncpy(buf, prop.gcnArchName, (size_t)buflen - 1);
        buf[buflen - 1] = '\0';
    }
    return 0;
}

}  // extern "C"
