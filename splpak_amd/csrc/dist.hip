// The fit on several GPUs of one node, driven from ONE process.
//
// What is distributed
//   points        every GPU bins and assembles its own shard (assemble.hip); the histogram, the
//                 normal equations and every refinement residual are summed over the GPUs
//   band factor   the block columns of the band are dealt to the GPUs in chunks (DistMap): a GPU stores
//                 and updates only its own block columns -- 1/R of the 26.9 GB at 64^3, and the only way
//                 to hold the 852 GB of the 4-D 32^4 grid (BASELINE config 5) at all
//   sweeps        forward / backward substitution walk the block columns in order; the window of the
//                 right-hand side that is still being updated travels from owner to owner
// Right-looking factorisation, per block step k (panel k = the solved block column k):
//   owner(k)      packs the panel and posts it (event);
//   every rank    copies the panel over xGMI (hipMemcpyPeerAsync: point-to-point, no collective library),
//                 then updates the block columns it owns by it (syrk64d_kernel, f64 MFMA);
//   owner(k+1)    updates block column k+1 FIRST, on its own high-priority stream, factors its diagonal
//                 block and solves its panel (potrf_block_kernel, trsm_kernel) while everybody's bulk
//                 update by panel k is still running: one block column of look-ahead, as on one GPU.
// One host thread per GPU enqueues that GPU's work; the only cross-thread dependencies are "panel k is
// posted", "rank q has panel k" (buffer reuse) and the sweep hand-offs, each an event plus a progress
// counter that tells the waiting thread the event has been recorded.  The reductions are host-synchronous
// (gather to rank 0 in rank order, add, hand back): a handful per fit, and bitwise reproducible.
//
// Virtual GPUs: every rank may be given the SAME device (SPLPAK_VIRTUAL_GPUS=1 or an explicit device
// list): the whole protocol -- ownership, peer copies, events, sweeps -- then runs on one GPU, which is
// how the 1-GPU test tier exercises it.
#include "plan.hpp"

#include <atomic>
#include <chrono>
#include <functional>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

using namespace splpak;

struct splpak_mplan;

namespace {

constexpr int NPB = 3;      // receive buffers per rank (panels in flight)
constexpr int NSB = 4;      // send buffers on the owner side

using Barrier = splpak::HostBarrier;

struct MRank {
    int r = 0, dev = 0;
    splpak_plan *p = nullptr;
    ::splpak_mplan *mp = nullptr;
    hipStream_t st = nullptr, sChain = nullptr, sBulk = nullptr, sCopy = nullptr;
    double *pbuf[NPB] = {nullptr, nullptr, nullptr};
    double *sbuf[NSB] = {nullptr, nullptr, nullptr, nullptr};
    double *stage = nullptr;        // staging of the reductions: a slice of the largest reduced buffer, twice (accumulator | incoming)
    long long stage_slice = 0;      // doubles per half
    size_t extra_bytes = 0;         // device bytes this rank holds beside its plan (panel / staging buffers)
    double *red_ptr = nullptr;      // the buffer this rank brings to the current reduction (published before the first barrier)
    double *part = nullptr;         // backward sweep partial sums
    double *xs = nullptr;           // backward sweep solution / window
    double *coef = nullptr;         // this rank's copy of the coefficients (ranks > 0)
    void *nccl = nullptr;           // this rank's RCCL communicator (SPLPAK_MPLAN_RCCL=1), else NULL
    std::vector<hipEvent_t> evReady, evArr, evBulk, evCol, evF, evB;
    hipEvent_t evTmp = nullptr;
    std::atomic<int> arr_prog{-1};
    int npacked = 0, ncopied = 0;               // panels packed (owner side) / copied in so far: buffer slots go round-robin
    int sslot_panel[NSB] = {-1, -1, -1, -1};    // panel held by each send buffer
    int pslot_panel[NPB] = {-1, -1, -1};        // panel held by each receive buffer
    int h_info = 0;
    double h_minpiv = 0.0;
    // arguments / result of the current fit
    const double *x = nullptr, *y = nullptr, *w = nullptr;
    long long ndata = 0;
    int l1 = 0;
    int rc = 0;
    std::string err;
    double info[10];
};

}  // namespace

struct splpak_mplan {
    int R = 1, chunk = 1;
    splpak::NdGroup *ndgrp = nullptr;       // the grid takes the nested-dissection factorisation (ndchol.hip), distributed over the ranks
    std::vector<MRank *> ranks;
    Barrier bar;
    std::atomic<int> ready_prog{-1};        // last panel whose "posted" event has been recorded
    std::atomic<int> f_prog{-1}, b_prog{1 << 30};
    std::atomic<int> abort{0};
    double *coef0 = nullptr;
    std::vector<int> panel_slot;            // send-buffer slot of panel k on its owner (written before ready_prog)
};

namespace {

bool wait_until(splpak_mplan *mp, const std::function<bool()> &ready)
{
    while (!ready()) {
        if (mp->abort.load(std::memory_order_relaxed)) return false;
        std::this_thread::yield();
    }
    return true;
}

#define DTRY(expr)                                            \
    do {                                                      \
        hipError_t e_ = (expr);                               \
        if (e_ != hipSuccess) { me->mp->abort.store(1); return e_; } \
    } while (0)

// ---- reductions (host-synchronous; deterministic: every element is summed in rank order) -----------------------
// Sum `count` doubles of the buffer every rank brings (any device buffer: the pointers are published); result on all ranks.
// Reduce-scatter + all-gather over point-to-point copies: rank q sums slice q of the buffer -- rank 0's values, then rank
// 1's, ... (the same order whoever owns the slice: bitwise reproducible and identical on every rank) -- and every rank then
// fetches the finished slices from their owners.  Per rank (R-1)/R of the buffer comes in twice, spread over its R-1 links,
// instead of everything passing through rank 0 (round 4: the 363 MB normal equations of a 64^3 grid, 10 GB at 32^4).
hipError_t reduce_all(MRank *me, double *buf, long long count)
{
    splpak_mplan *mp = me->mp;
    const int R = mp->R;
    if (me->nccl) {
        // SPLPAK_MPLAN_RCCL=1 (round 5): "RCCL-reducing the normal equations over xGMI" on the route a Fortran caller reaches.
        // Every rank thread enqueues its ncclAllReduce on its own stream (the calls of one collective come from R threads at
        // once, as RCCL requires of a multi-communicator process); the sum's order is RCCL's, not rank order: the result is
        // the same on every rank but not bitwise the peer-copy form's.
        // Every thread enqueues first and the threads meet on the ENQUEUE status before anybody waits for the stream: a rank
        // whose enqueue failed never joins the collective, and a thread that waited on its stream for a collective that cannot
        // complete would hang instead of returning SPLPAK_E_COMM (round-5 advice).  On failure nobody synchronises.
        const int rc = rccl_allreduce_sum(me->nccl, buf, count, me->st);
        if (rc != 0) mp->abort.store(1);
        if (!mp->bar.wait(mp->abort)) return hipErrorUnknown;
        if (mp->abort.load()) return hipErrorUnknown;
        hipError_t e = hipStreamSynchronize(me->st);
        if (e != hipSuccess) mp->abort.store(1);
        if (!mp->bar.wait(mp->abort)) return hipErrorUnknown;
        return mp->abort.load() ? (e != hipSuccess ? e : hipErrorUnknown) : hipSuccess;
    }
    me->red_ptr = buf;
    hipError_t err = hipStreamSynchronize(me->st);
    if (!mp->bar.wait(mp->abort)) return hipErrorUnknown;
    long long per = (count + R - 1) / R;
    per = (per + 511) / 512 * 512;
    const long long lo = (long long)me->r * per, hi = lo + per < count ? lo + per : count;
    if (err == hipSuccess && hi - lo > me->stage_slice) err = hipErrorInvalidValue;       // (sized for the largest reduction of the plan)
    if (err == hipSuccess && lo < hi) {
        // acc = b_0 + b_1 + ... + b_{R-1} on my slice; the other ranks still read my unreduced values, so the sum is built
        // beside them and put in place after the barrier below
        const long long n = hi - lo;
        double *acc = me->stage, *in = me->stage + me->stage_slice;
        MRank *r0 = mp->ranks[0];
        err = hipMemcpyPeerAsync(acc, me->dev, r0->red_ptr + lo, r0->dev, sizeof(double) * (size_t)n, me->st);
        for (int q = 1; q < R && err == hipSuccess; ++q) {
            MRank *rq = mp->ranks[(size_t)q];
            if (q == me->r) err = launch_vec_add(n, acc, buf + lo, me->st);
            else {
                err = hipMemcpyPeerAsync(in, me->dev, rq->red_ptr + lo, rq->dev, sizeof(double) * (size_t)n, me->st);
                if (err == hipSuccess) err = launch_vec_add(n, acc, in, me->st);
            }
        }
        if (err == hipSuccess) err = hipStreamSynchronize(me->st);
    }
    if (err != hipSuccess) mp->abort.store(1);
    if (!mp->bar.wait(mp->abort)) return hipErrorUnknown;          // every slice is summed, nobody reads the inputs any more
    if (lo < hi) err = hipMemcpyAsync(buf + lo, me->stage, sizeof(double) * (size_t)(hi - lo), hipMemcpyDeviceToDevice, me->st);
    if (err == hipSuccess) err = hipStreamSynchronize(me->st);
    if (err != hipSuccess) mp->abort.store(1);
    if (!mp->bar.wait(mp->abort)) return hipErrorUnknown;          // the finished slices are in place on their owners
    for (int d = 1; d < R && err == hipSuccess; ++d) {               // all-gather, every rank starting at a different peer
        const int q = (me->r + d) % R;
        const long long qlo = (long long)q * per, qhi = qlo + per < count ? qlo + per : count;
        if (qlo >= qhi) continue;
        MRank *rq = mp->ranks[(size_t)q];
        err = hipMemcpyPeerAsync(buf + qlo, me->dev, rq->red_ptr + qlo, rq->dev, sizeof(double) * (size_t)(qhi - qlo), me->st);
    }
    if (err == hipSuccess) err = hipStreamSynchronize(me->st);
    if (err != hipSuccess) mp->abort.store(1);
    if (!mp->bar.wait(mp->abort)) return hipErrorUnknown;          // nobody reads a neighbour's buffer any more
    return mp->abort.load() ? (err != hipSuccess ? err : hipErrorUnknown) : hipSuccess;
}

int32_t ar_callback(void *dev_buf, int64_t count, void *stream, void *user)
{
    (void)stream;       // the rank's own stream: reduce_all synchronises it
    MRank *me = static_cast<MRank *>(user);
    return reduce_all(me, static_cast<double *>(dev_buf), count) == hipSuccess ? 0 : 1;
}

// ---- distributed factorisation ----------------------------------------------------------------------
hipError_t dist_factor(splpak_plan *p, int *info_dev, double *minpiv_dev, hipStream_t st, void *user)
{
    MRank *me = static_cast<MRank *>(user);
    splpak_mplan *mp = me->mp;
    const Band &b = p->band;
    const DistMap &dm = p->dm;
    const int nblk = b.nblk;
    auto tb_of = [&](int k) { int t = nblk - 1 - k; return t > b.bw ? b.bw : (t < 0 ? 0 : t); };
    auto colbase = [&](int J) { return b.ab + dm_shift(dm, J); };      // dense view of block column J

    // everything queued on the caller's stream (expand) comes first
    DTRY(hipEventRecord(me->evTmp, st));
    DTRY(hipStreamWaitEvent(me->sChain, me->evTmp, 0));
    DTRY(hipStreamWaitEvent(me->sBulk, me->evTmp, 0));
    DTRY(hipStreamWaitEvent(me->sCopy, me->evTmp, 0));
    if (!mp->bar.wait(mp->abort)) return hipErrorUnknown;                     // progress counters of the previous factorisation are reset by rank 0 below
    if (me->r == 0) mp->ready_prog.store(-1);
    me->arr_prog.store(-1);
    me->npacked = me->ncopied = 0;
    for (int &v : me->sslot_panel) v = -1;
    for (int &v : me->pslot_panel) v = -1;
    if (!mp->bar.wait(mp->abort)) return hipErrorUnknown;

    // factor block column J (its updates are complete on sChain), solve its panel, pack and post it
    auto factor_column = [&](int J) -> hipError_t {
        const int k0 = J * NBLK, slot = dm_slot(dm, J), nrows = tb_of(J) * NBLK;
        double *A = colbase(J);
        double *inv16 = b.inv64 + (long long)slot * 4 * 64 * 64;
        hipError_t e = launch_potrf_block(A, b.lda, k0, info_dev, minpiv_dev, inv16, me->sChain);
        if (e != hipSuccess) return e;
        e = launch_trsm_panel(A + (long long)k0 + (long long)k0 * b.lda, A + (long long)(k0 + NBLK) + (long long)k0 * b.lda,
                              b.lda, inv16, nrows, me->sChain);
        if (e != hipSuccess) return e;
        const int ss = me->npacked % NSB;
        ++me->npacked;
        if (nrows > 0) {
            // the send buffer was last read by the ranks that copied the panel it held before
            const int Jprev = me->sslot_panel[ss];
            if (Jprev >= 0)
                for (int q = 0; q < mp->R; ++q) {
                    MRank *rq = mp->ranks[q];
                    if (q == me->r) continue;
                    if (!wait_until(mp, [&] { return rq->arr_prog.load(std::memory_order_acquire) >= Jprev; })) return hipErrorUnknown;
                    e = hipStreamWaitEvent(me->sChain, rq->evArr[Jprev], 0);
                    if (e != hipSuccess) return e;
                }
            e = launch_pack_panel(A + (long long)(k0 + NBLK) + (long long)k0 * b.lda, b.lda, me->sbuf[ss], nrows, me->sChain);
            if (e != hipSuccess) return e;
        }
        me->sslot_panel[ss] = J;
        mp->panel_slot[(size_t)J] = ss;
        e = hipEventRecord(me->evReady[J], me->sChain);
        mp->ready_prog.store(J, std::memory_order_release);
        return e;
    };

    if (dm_owned(dm, 0)) DTRY(factor_column(0));
    for (int k = 0; k + 1 < nblk; ++k) {
        const int tb = tb_of(k);
        const int nrows = tb * NBLK;               // rows of panel k = rows k+1 .. k+tb (blocks)
        const int o = dm_owner(dm, k);
        MRank *ro = mp->ranks[o];
        // ---- panel k: in this rank's hands as P (leading dimension nrows)
        const double *P;
        if (o == me->r) {
            P = me->sbuf[mp->panel_slot[(size_t)k]];            // own panels need no copy
            DTRY(hipStreamWaitEvent(me->sBulk, me->evReady[k], 0));
        } else {
            if (!wait_until(mp, [&] { return mp->ready_prog.load(std::memory_order_acquire) >= k; })) return hipErrorUnknown;
            DTRY(hipStreamWaitEvent(me->sCopy, ro->evReady[k], 0));
            const int ps = me->ncopied % NPB;
            ++me->ncopied;
            const int kprev = me->pslot_panel[ps];   // the receive buffer was last read by the updates by that panel
            if (kprev >= 0) {
                DTRY(hipStreamWaitEvent(me->sCopy, me->evBulk[kprev], 0));
                DTRY(hipStreamWaitEvent(me->sCopy, me->evCol[kprev], 0));
            }
            me->pslot_panel[ps] = k;
            if (nrows > 0)
                DTRY(hipMemcpyPeerAsync(me->pbuf[ps], me->dev, ro->sbuf[mp->panel_slot[(size_t)k]], ro->dev,
                                        sizeof(double) * (size_t)nrows * NBLK, me->sCopy));
            DTRY(hipEventRecord(me->evArr[k], me->sCopy));
            me->arr_prog.store(k, std::memory_order_release);
            DTRY(hipStreamWaitEvent(me->sBulk, me->evArr[k], 0));
            P = me->pbuf[ps];
        }
        const int row0 = (k + 1) * NBLK;
        const int re = nrows / 64;
        // ---- chain: the owner of block column k+1 updates it first, then factors it
        const bool chain = dm_owned(dm, k + 1);
        if (chain) {
            DTRY(hipStreamWaitEvent(me->sChain, o == me->r ? me->evReady[k] : me->evArr[k], 0));
            if (k >= 1) DTRY(hipStreamWaitEvent(me->sChain, me->evBulk[k - 1], 0));     // earlier updates of column k+1
            if (nrows > 0) DTRY(launch_syrk64d(b.ab, b.lda, dm, P, nrows, row0, k + 1, k + 2, re, me->sChain));
            DTRY(hipEventRecord(me->evCol[k], me->sChain));
            DTRY(factor_column(k + 1));
        } else {
            DTRY(hipEventRecord(me->evCol[k], me->sBulk));
        }
        // ---- bulk: the other block columns this rank owns in the window of panel k
        if (nrows > 0) DTRY(launch_syrk64d(b.ab, b.lda, dm, P, nrows, row0, k + 2, k + 1 + tb, re, me->sBulk));
        DTRY(hipEventRecord(me->evBulk[k], me->sBulk));
    }
    // join: the caller's stream continues after the pipeline
    DTRY(hipEventRecord(me->evTmp, me->sChain));
    DTRY(hipStreamWaitEvent(st, me->evTmp, 0));
    if (nblk >= 2) DTRY(hipStreamWaitEvent(st, me->evBulk[nblk - 2], 0));
    DTRY(hipStreamSynchronize(me->sCopy));          // peer copies INTO this rank are complete
    DTRY(launch_trtri_owned(b.ab, b.lda, dm, p->own_blocks, p->nown, b.inv64, b.dinv, b.dinvt, st));
    // the send buffers may still be read by slower ranks: nobody leaves before everybody has every panel
    DTRY(hipStreamSynchronize(st));
    if (!mp->bar.wait(mp->abort)) return hipErrorUnknown;
    // pivots: a failure anywhere is a failure everywhere (every rank must take the same way out of the fit)
    DTRY(hipMemcpy(&me->h_info, info_dev, sizeof(int), hipMemcpyDeviceToHost));
    DTRY(hipMemcpy(&me->h_minpiv, minpiv_dev, sizeof(double), hipMemcpyDeviceToHost));
    if (!mp->bar.wait(mp->abort)) return hipErrorUnknown;
    int hinfo = 0;
    double piv = me->h_minpiv;
    for (MRank *rq : mp->ranks) {
        if (rq->h_info != 0 && (hinfo == 0 || rq->h_info < hinfo)) hinfo = rq->h_info;
        if (rq->h_minpiv < piv || !(rq->h_minpiv == rq->h_minpiv)) piv = rq->h_minpiv;
    }
    if (!mp->bar.wait(mp->abort)) return hipErrorUnknown;
    DTRY(hipMemcpy(info_dev, &hinfo, sizeof(int), hipMemcpyHostToDevice));
    DTRY(hipMemcpy(minpiv_dev, &piv, sizeof(double), hipMemcpyHostToDevice));
    return hipSuccess;
}

// ---- distributed sweeps -----------------------------------------------------------------------------
hipError_t dist_solve(splpak_plan *p, double *x, double *tmp, hipStream_t st, void *user)
{
    MRank *me = static_cast<MRank *>(user);
    splpak_mplan *mp = me->mp;
    const Band &b = p->band;
    const DistMap &dm = p->dm;
    const int nblk = b.nblk;
    auto tb_of = [&](int k) { int t = nblk - 1 - k; return t > b.bw ? b.bw : (t < 0 ? 0 : t); };
    const long long nb2 = (long long)NBLK * NBLK;
    const long long off_x = x - p->comm;            // x is p->xvec or p->rho (communication buffer or not)
    const bool x_in_comm = off_x >= 0 && off_x < p->comm_len;
    auto vec_of = [&](MRank *rq) { return x_in_comm ? rq->p->comm + off_x : rq->p->xvec; };

    if (!mp->bar.wait(mp->abort)) return hipErrorUnknown;
    if (me->r == 0) { mp->f_prog.store(-1); mp->b_prog.store(1 << 30); }
    DTRY(hipStreamSynchronize(st));                 // every rank's right-hand side is in place
    if (!mp->bar.wait(mp->abort)) return hipErrorUnknown;
    // forward: L y = b, y block k left in tmp on owner(k); x is updated in place
    for (int k = 0; k < nblk; ++k) {
        if (!dm_owned(dm, k)) continue;
        const int k0 = k * NBLK, nrows = tb_of(k) * NBLK, slot = dm_slot(dm, k);
        if (k > 0 && dm_owner(dm, k - 1) != me->r) {
            MRank *rp = mp->ranks[dm_owner(dm, k - 1)];
            if (!wait_until(mp, [&] { return mp->f_prog.load(std::memory_order_acquire) >= k - 1; })) return hipErrorUnknown;
            DTRY(hipStreamWaitEvent(st, rp->evF[k - 1], 0));
            const int wrows = tb_of(k - 1) * NBLK;          // rows step k-1 has updated: block k .. k-1+tb
            if (wrows > 0)
                DTRY(hipMemcpyPeerAsync(x + k0, me->dev, vec_of(rp) + k0, rp->dev, sizeof(double) * (size_t)wrows, st));
        }
        const double *A = b.ab + dm_shift(dm, k);
        DTRY(launch_blockmv(b.dinv + slot * nb2, x + k0, tmp + k0, st));
        DTRY(launch_fwd_update(A + (long long)(k0 + NBLK) + (long long)k0 * b.lda, b.lda, tmp + k0, x + k0 + NBLK, nrows, st));
        DTRY(hipEventRecord(me->evF[k], st));
        mp->f_prog.store(k, std::memory_order_release);
    }
    // backward: L^T x = y, column oriented: x_k = Linv_k^T (y_k - L[below, k]^T x[below])
    for (int k = nblk - 1; k >= 0; --k) {
        if (!dm_owned(dm, k)) continue;
        const int k0 = k * NBLK, nrows = tb_of(k) * NBLK, slot = dm_slot(dm, k);
        if (k < nblk - 1 && dm_owner(dm, k + 1) != me->r) {
            MRank *rn = mp->ranks[dm_owner(dm, k + 1)];
            if (!wait_until(mp, [&] { return mp->b_prog.load(std::memory_order_acquire) <= k + 1; })) return hipErrorUnknown;
            DTRY(hipStreamWaitEvent(st, rn->evB[k + 1], 0));
            if (nrows > 0)
                DTRY(hipMemcpyPeerAsync(me->xs + k0 + NBLK, me->dev, rn->xs + k0 + NBLK, rn->dev, sizeof(double) * (size_t)nrows, st));
        }
        const double *A = b.ab + dm_shift(dm, k);
        DTRY(launch_bwd_column(A + (long long)(k0 + NBLK) + (long long)k0 * b.lda, b.lda, me->xs + k0 + NBLK, nrows,
                               b.dinvt + slot * nb2, tmp + k0, me->part, me->xs + k0, st));
        DTRY(hipEventRecord(me->evB[k], st));
        mp->b_prog.store(k, std::memory_order_release);
    }
    // every block of the solution sits on its owner: mask the rest, sum over the ranks
    DTRY(hipStreamSynchronize(st));
    if (!mp->bar.wait(mp->abort)) return hipErrorUnknown;                                 // nobody reads a neighbour's window any more
    DTRY(launch_mask_owned(b.npad, dm, me->xs, st));
    DTRY(reduce_all(me, me->xs, b.npad));
    DTRY(hipMemcpyAsync(x, me->xs, sizeof(double) * (size_t)b.npad, hipMemcpyDeviceToDevice, st));
    return hipSuccess;
}

void rank_main(MRank *me)
{
    (void)hipSetDevice(me->dev);
    double *coef = me->r == 0 ? me->mp->coef0 : me->coef;
    me->rc = splpak_plan_fit_dev(me->p, me->x, me->l1, me->y, me->w, me->ndata, coef, me->st, me->info);
    if (me->rc < 0) {
        char buf[400];
        splpak_last_error_message(buf, sizeof buf);
        me->err = buf;
        me->mp->abort.store(1);
    }
}

void free_rank(MRank *m)
{
    if (!m) return;
    (void)hipSetDevice(m->dev);
    for (auto *v : {&m->evReady, &m->evArr, &m->evBulk, &m->evCol, &m->evF, &m->evB})
        for (hipEvent_t e : *v) if (e) (void)hipEventDestroy(e);
    if (m->evTmp) (void)hipEventDestroy(m->evTmp);
    for (double *q : m->pbuf) if (q) (void)hipFree(q);
    for (double *q : m->sbuf) if (q) (void)hipFree(q);
    for (double *q : {m->stage, m->part, m->xs, m->coef}) if (q) (void)hipFree(q);
    for (hipStream_t s : {m->st, m->sChain, m->sBulk, m->sCopy}) if (s) (void)hipStreamDestroy(s);
    if (m->nccl) rccl_comm_free(m->nccl);
    if (m->p) splpak_plan_destroy(m->p);
    delete m;
}

}  // namespace

extern "C" {

int32_t splpak_mplan_create(int32_t ngpus, const int32_t *devices, int32_t chunk, int32_t ndim, const int32_t *nodes,
                            const double *xmin, const double *xmax, double xtrap, int64_t max_ndata_per_gpu,
                            splpak_mplan **mplan)
{
    if (!mplan || !nodes || !xmin || !xmax) { set_error("null argument"); return SPLPAK_E_BADARG; }
    *mplan = nullptr;
    if (ngpus < 1 || ngpus > 64) { set_error("ngpus must be 1..64"); return SPLPAK_E_BADARG; }
    if (int r = device_ready()) return r;
    int ndev = 0, cur = 0;
    (void)hipGetDeviceCount(&ndev);
    (void)hipGetDevice(&cur);
    const bool virt = splpak::opt_get("SPLPAK_VIRTUAL_GPUS") != nullptr;
    splpak_mplan *mp = new splpak_mplan();
    mp->R = ngpus;
    if (chunk < 1) {
        // automatic: about one chunk per rank inside the band window, at most 8 blocks.  Consecutive block
        // columns on the same GPU keep the panel chain (column update -> potrf -> panel solve) local for
        // chunk-1 of every chunk steps -- the 25.7 MB panel then crosses xGMI on the critical path only once
        // per chunk -- while the window (bw blocks) still spreads over all the ranks.
        Grid g0;
        long long nc = 0;
        Band b0{};
        chunk = 1;
        if (build_grid(ndim, nodes, xmin, xmax, g0, &nc, splpak::opt_get("SPLPAK_NO_REORDER") == nullptr) == 0) {
            band_bytes(g0.ncol, g0.halfbw, &b0);
            chunk = b0.bw / ngpus;
            if (chunk > 8) chunk = 8;
            if (chunk < 1) chunk = 1;
        }
    }
    mp->chunk = chunk;
    mp->bar.n = ngpus;
    // Grids that take the nested-dissection factorisation on one GPU take it here too, distributed: subtrees per GPU, the
    // fronts above them by block columns (round 4; SPLPAK_MPLAN_BAND=1 keeps the distributed band of round 2).
    bool want_nd = ngpus > 1 && !splpak::opt_get("SPLPAK_MPLAN_BAND") && nd_wanted_for(ndim, nodes, xmin, xmax);
    if (want_nd) {
        // The distributed nested dissection READS the other GPUs' memory from kernels and needs peer access between every pair of
        // distinct devices; the distributed band only copies.  Probed BEFORE anything large is allocated (round-5 advice): without
        // peer access the plan takes the band form where a rank's share of the band fits its device, and is refused otherwise.
        bool peers = splpak::opt_get("SPLPAK_DEBUG_NO_PEER") == nullptr || virt;
        int pa = -1, pb = -1;
        for (int a = 0; a < ngpus && !virt; ++a)
            for (int b2 = 0; b2 < ngpus; ++b2) {
                const int da = devices ? devices[a] : a, db = devices ? devices[b2] : b2;
                if (da == db || da < 0 || db < 0 || da >= ndev || db >= ndev) continue;
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, da, db) != hipSuccess || !can || splpak::opt_get("SPLPAK_DEBUG_NO_PEER")) { peers = false; pa = da; pb = db; }
                (void)hipGetLastError();
            }
        if (!peers) {
            Grid g0;
            long long nc = 0;
            Band b0{};
            size_t fr = 0, tot = 0;
            bool band_fits = false;
            if (build_grid(ndim, nodes, xmin, xmax, g0, &nc, splpak::opt_get("SPLPAK_NO_REORDER") == nullptr) == 0 &&
                hipMemGetInfo(&fr, &tot) == hipSuccess) {
                band_bytes(g0.ncol, g0.halfbw, &b0);
                band_fits = (double)b0.bytes / ngpus < 0.8 * (double)tot;
            }
            (void)hipGetLastError();
            if (band_fits) {
                want_nd = false;
                if (splpak::opt_get("SPLPAK_DEBUG"))
                    fprintf(stderr, "[splpak] multi-GPU plan: no peer access between devices %d and %d: the distributed band instead of the distributed nested dissection\n", pa, pb);
            } else {
                char buf[256];
                snprintf(buf, sizeof buf, "multi-GPU plan: device %d cannot map device %d's memory (no peer access), which the distributed nested "
                                          "dissection needs, and the distributed band does not fit the devices either", pa, pb);
                set_error(buf);
                delete mp;
                return SPLPAK_E_UNSUPPORTED;
            }
        }
    }
    if (want_nd) {
        const char *ck = splpak::opt_get("SPLPAK_ND_CHUNK");
        mp->ndgrp = nd_group_create(ngpus, ck ? atoi(ck) : 1, &mp->abort);
    }
    int rc = 0;
    for (int r = 0; r < ngpus && rc == 0; ++r) {
        MRank *m = new MRank();
        mp->ranks.push_back(m);
        m->r = r;
        m->mp = mp;
        m->dev = devices ? devices[r] : (virt ? cur : r);
        if (m->dev < 0 || m->dev >= ndev) {
            set_error("not enough GPUs for ngpus (set SPLPAK_VIRTUAL_GPUS=1 to rehearse on one device)");
            rc = SPLPAK_E_BADARG;
            break;
        }
        (void)hipSetDevice(m->dev);
        rc = plan_create_dist(ndim, nodes, xmin, xmax, xtrap, max_ndata_per_gpu, nullptr, 0, ngpus, r, mp->chunk, &m->p, mp->ndgrp != nullptr,
                              mp->ndgrp);
        if (rc != 0) break;
        splpak_plan *p = m->p;
        const bool nd = p->fn_code == 5;    // nested dissection, distributed: the plan carries its own factorisation and solve
        if (mp->ndgrp && !nd) { set_error("multi-GPU plan: a rank did not take the nested-dissection factorisation"); rc = SPLPAK_E_UNSUPPORTED; break; }
        if (!nd) twoend_detach(p);        // (a one-rank plan may have chosen the two-ended single-GPU factorisation)
        p->ar = ar_callback;
        p->ar_user = m;
        p->rank = r;
        p->world = ngpus;
        if (!nd) {
            p->factor_fn = dist_factor;
            p->solve_fn = dist_solve;
            p->fn_user = m;
            p->fn_name = "band Cholesky distributed over several GPUs by block columns (csrc/dist.hip)";
            p->fn_code = 3;
        }
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        bool ok = hipStreamCreateWithFlags(&m->st, hipStreamNonBlocking) == hipSuccess &&
                  hipStreamCreateWithPriority(&m->sChain, hipStreamNonBlocking, hi) == hipSuccess &&
                  hipStreamCreateWithFlags(&m->sBulk, hipStreamNonBlocking) == hipSuccess &&
                  hipStreamCreateWithPriority(&m->sCopy, hipStreamNonBlocking, hi) == hipSuccess &&
                  hipEventCreateWithFlags(&m->evTmp, hipEventDisableTiming) == hipSuccess;
        const Band &b = p->band;
        if (!nd) {                          // panel buffers, sweep scratch and step events of the distributed band
            const size_t panel = (size_t)(b.bw > 0 ? b.bw : 1) * NBLK * NBLK;
            for (int i = 0; i < NPB && ok; ++i) ok = hipMalloc((void **)&m->pbuf[i], sizeof(double) * panel) == hipSuccess;
            for (int i = 0; i < NSB && ok; ++i) ok = hipMalloc((void **)&m->sbuf[i], sizeof(double) * panel) == hipSuccess;
            const size_t nsplit = (size_t)(b.bw * NBLK) / (4 * NBLK) + 2;
            ok = ok && hipMalloc((void **)&m->part, sizeof(double) * nsplit * NBLK) == hipSuccess;
            ok = ok && hipMalloc((void **)&m->xs, sizeof(double) * (size_t)(b.npad + NBLK)) == hipSuccess;
            m->extra_bytes += sizeof(double) * ((NPB + NSB) * panel + nsplit * NBLK + (size_t)(b.npad + NBLK));
            if (ok) ok = hipMemset(m->xs, 0, sizeof(double) * (size_t)(b.npad + NBLK)) == hipSuccess;
            for (auto *v : {&m->evReady, &m->evArr, &m->evBulk, &m->evCol, &m->evF, &m->evB}) {
                v->assign((size_t)b.nblk + 1, nullptr);
                for (auto &e : *v) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
            }
        }
        {   // staging of the reductions: one slice of the largest reduced buffer, twice (reduce_all)
            const long long big = p->lenG > (long long)b.npad + NBLK ? p->lenG : (long long)b.npad + NBLK;
            m->stage_slice = ((big + ngpus - 1) / ngpus + 511) / 512 * 512 + 512;
            ok = ok && hipMalloc((void **)&m->stage, sizeof(double) * 2 * (size_t)m->stage_slice) == hipSuccess;
            m->extra_bytes += sizeof(double) * 2 * (size_t)m->stage_slice;
        }
        if (r != 0) ok = ok && hipMalloc((void **)&m->coef, sizeof(double) * (size_t)p->g.ncol) == hipSuccess;
        if (!ok) { set_error("device allocation of the multi-GPU plan failed"); (void)hipGetLastError(); rc = SPLPAK_E_NOMEM; }
    }
    // peer access where the ranks sit on different devices.  The distributed band only COPIES between devices
    // (hipMemcpyPeerAsync stages through the host without it); the distributed nested dissection READS the other GPUs' memory
    // from kernels (nd_pull_add_kernel: the children's Schur complements through the peer mapping) and needs it -- without it
    // the fit would take a memory fault, so the plan is refused instead (round-4 advice)
    if (rc == 0) {
        bool all_peers = true;
        int pa = -1, pb = -1;
        for (MRank *a : mp->ranks)
            for (MRank *b2 : mp->ranks)
                if (a->dev != b2->dev) {
                    (void)hipSetDevice(a->dev);
                    int can = 0;
                    bool on = false;
                    if (hipDeviceCanAccessPeer(&can, a->dev, b2->dev) == hipSuccess && can) {
                        const hipError_t e = hipDeviceEnablePeerAccess(b2->dev, 0);
                        on = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
                    }
                    (void)hipGetLastError();
                    if (!on || splpak::opt_get("SPLPAK_DEBUG_NO_PEER")) { all_peers = false; pa = a->dev; pb = b2->dev; }
                }
        if (mp->ndgrp && !all_peers) {
            char buf[256];
            snprintf(buf, sizeof buf, "multi-GPU plan: device %d cannot map device %d's memory (no peer access), which the distributed nested "
                                      "dissection needs; SPLPAK_MPLAN_BAND=1 selects the distributed band, which only copies", pa, pb);
            set_error(buf);
            rc = SPLPAK_E_UNSUPPORTED;
        }
    }
    // the sums over the ranks as RCCL all-reduces instead of peer copies (behind a switch: the peer-copy form is bitwise
    // reproducible in rank order and needs no library)
    if (rc == 0 && ngpus > 1 && splpak::opt_get("SPLPAK_MPLAN_RCCL") && atoi(splpak::opt_get("SPLPAK_MPLAN_RCCL")) != 0) {
        std::vector<int> devs;
        std::vector<void *> comms((size_t)ngpus, nullptr);
        for (MRank *m : mp->ranks) devs.push_back(m->dev);
        rc = rccl_comms_for_devices(ngpus, devs.data(), comms.data());
        if (rc == 0)
            for (int r = 0; r < ngpus; ++r) mp->ranks[(size_t)r]->nccl = comms[(size_t)r];
    }
    (void)hipSetDevice(cur);
    if (rc != 0) {
        for (MRank *m : mp->ranks) free_rank(m);
        nd_group_destroy(mp->ndgrp);
        delete mp;
        return rc;
    }
    if (rc == 0 && mp->ndgrp) rc = nd_group_finalize(mp->ndgrp);          // (the tables that hold the peers' addresses)
    (void)hipSetDevice(cur);
    if (rc != 0) {
        for (MRank *m : mp->ranks) free_rank(m);
        nd_group_destroy(mp->ndgrp);
        delete mp;
        return rc;
    }
    mp->panel_slot.assign((size_t)mp->ranks[0]->p->band.nblk + 1, 0);
    *mplan = mp;
    return 0;
}

void splpak_mplan_destroy(splpak_mplan *mp)
{
    if (!mp) return;
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (MRank *m : mp->ranks) free_rank(m);
    nd_group_destroy(mp->ndgrp);
    (void)hipSetDevice(cur);
    delete mp;
}

int32_t splpak_mplan_factorisation(const splpak_mplan *mp, char *buf, int32_t buflen)
{
    if (!mp || mp->ranks.empty()) return SPLPAK_E_BADARG;
    return splpak_plan_factorisation(mp->ranks[0]->p, buf, buflen);
}

int64_t splpak_mplan_rank_bytes(const splpak_mplan *mp, int32_t rank)
{
    if (!mp || rank < 0 || rank >= mp->R) return -1;
    const MRank *m = mp->ranks[(size_t)rank];
    return splpak_plan_device_bytes(m->p) + (int64_t)m->extra_bytes;
}

int32_t splpak_mplan_device(const splpak_mplan *mp, int32_t rank)
{
    return (mp && rank >= 0 && rank < mp->R) ? mp->ranks[(size_t)rank]->dev : -1;
}

int32_t splpak_mplan_fit_dev(splpak_mplan *mp, const double *const *xdata_dev, int32_t l1xdat, const double *const *ydata_dev,
                             const double *const *wdata_dev, const int64_t *ndata, double *coef_dev, double *info)
{
    if (!mp || !xdata_dev || !ydata_dev || !ndata || !coef_dev) { set_error("null argument"); return SPLPAK_E_BADARG; }
    long long total = 0;
    for (int r = 0; r < mp->R; ++r) total += ndata[r] > 0 ? ndata[r] : 0;
    if (total < 1) return 105;
    int cur = 0;
    (void)hipGetDevice(&cur);
    mp->abort.store(0);
    {   // a fit that was abandoned (a rank failed while others stood in the barrier) leaves arrivals counted: every
        // rank thread of the previous call has been joined, so the barrier can simply start afresh (round-2 advice)
        std::lock_guard<std::mutex> lk(mp->bar.mu);
        mp->bar.count = 0;
        ++mp->bar.phase;
    }
    nd_group_reset(mp->ndgrp);
    mp->coef0 = coef_dev;
    std::vector<std::thread> th;
    for (int r = 0; r < mp->R; ++r) {
        MRank *m = mp->ranks[(size_t)r];
        m->x = xdata_dev[r];
        m->y = ydata_dev[r];
        m->w = wdata_dev ? wdata_dev[r] : nullptr;
        m->ndata = ndata[r];
        m->l1 = l1xdat;
        m->rc = 0;
        m->err.clear();
        th.emplace_back(rank_main, m);
    }
    for (auto &t : th) t.join();
    (void)hipSetDevice(cur);
    int rc = 0;
    for (MRank *m : mp->ranks) {
        if (m->rc < 0 && (rc >= 0 || !m->err.empty())) { rc = m->rc; if (!m->err.empty()) set_error(m->err); }
        else if (rc == 0 && m->rc > 0) rc = m->rc;
    }
    if (info) std::memcpy(info, mp->ranks[0]->info, sizeof(double) * 10);
    return rc;
}

int32_t splpak_fit_multi_f64(int32_t ngpus, int32_t ndim, const double *xdata, int32_t l1xdat, const double *ydata,
                             const double *wdata, int64_t ndata, const double *xmin, const double *xmax,
                             const int32_t *nodes, double xtrap, double *coef, int64_t ncf, int64_t nwrk,
                             double *hist_out, double *info)
{
    if (ngpus <= 1)
        return splpak_fit_f64(ndim, xdata, l1xdat, ydata, wdata, ndata, xmin, xmax, nodes, xtrap, coef, ncf, nwrk, hist_out, info);
    if (!nodes || !xmin || !xmax) { set_error("null argument"); return SPLPAK_E_BADARG; }
    Grid g;
    long long ncol = 0;
    const int v = build_grid(ndim, nodes, xmin, xmax, g, &ncol);       // 101, 102, 103
    if (v != 0) return v;
    if (ncol > ncf) return 104;
    if (ndata < 1) return 105;
    if (nwrk >= 0) {
        const long long nwrk1 = (xtrap != 0.0) ? ncol + 1 : 1;
        if (nwrk - nwrk1 + 1 < 1) return 106;
    }
    if (!xdata || !ydata || !coef) { set_error("null argument"); return SPLPAK_E_BADARG; }
    if (l1xdat < ndim) { set_error("l1xdat < ndim"); return SPLPAK_E_BADARG; }
    if (wdata && wdata[0] < 0.0) wdata = nullptr;
    const long long per = (ndata + ngpus - 1) / ngpus;
    splpak_mplan *mp = nullptr;
    const char *ck = splpak::opt_get("SPLPAK_DIST_CHUNK");
    int rc = splpak_mplan_create(ngpus, nullptr, ck ? atoi(ck) : 0, ndim, nodes, xmin, xmax, xtrap, per, &mp);
    if (rc != 0) return rc;
    int cur = 0;
    (void)hipGetDevice(&cur);
    std::vector<double *> dx((size_t)ngpus, nullptr), dy((size_t)ngpus, nullptr), dw((size_t)ngpus, nullptr);
    std::vector<const double *> cx((size_t)ngpus), cy((size_t)ngpus), cw((size_t)ngpus);
    std::vector<int64_t> cnt((size_t)ngpus, 0);
    double *dcoef = nullptr;
    bool ok = true;
    for (int r = 0; r < ngpus && ok; ++r) {
        const long long first = (long long)r * per;
        const long long n = first >= ndata ? 0 : (ndata - first < per ? ndata - first : per);
        cnt[(size_t)r] = n;
        (void)hipSetDevice(mp->ranks[(size_t)r]->dev);
        const size_t nn = (size_t)(n > 0 ? n : 1);
        ok = hipMalloc((void **)&dx[(size_t)r], sizeof(double) * nn * l1xdat) == hipSuccess &&
             hipMalloc((void **)&dy[(size_t)r], sizeof(double) * nn) == hipSuccess &&
             (!wdata || hipMalloc((void **)&dw[(size_t)r], sizeof(double) * nn) == hipSuccess);
        if (ok && n > 0) {
            ok = hipMemcpy(dx[(size_t)r], xdata + first * l1xdat, sizeof(double) * (size_t)n * l1xdat, hipMemcpyHostToDevice) == hipSuccess &&
                 hipMemcpy(dy[(size_t)r], ydata + first, sizeof(double) * (size_t)n, hipMemcpyHostToDevice) == hipSuccess &&
                 (!wdata || hipMemcpy(dw[(size_t)r], wdata + first, sizeof(double) * (size_t)n, hipMemcpyHostToDevice) == hipSuccess);
        }
        cx[(size_t)r] = dx[(size_t)r];
        cy[(size_t)r] = dy[(size_t)r];
        cw[(size_t)r] = dw[(size_t)r];
    }
    (void)hipSetDevice(mp->ranks[0]->dev);
    ok = ok && hipMalloc((void **)&dcoef, sizeof(double) * (size_t)ncol) == hipSuccess;
    if (!ok) { set_error("device allocation / upload of the point shards failed"); (void)hipGetLastError(); rc = SPLPAK_E_NOMEM; }
    if (rc == 0) rc = splpak_mplan_fit_dev(mp, cx.data(), l1xdat, cy.data(), wdata ? cw.data() : nullptr, cnt.data(), dcoef, info);
    if (rc == 0 || rc == 107) {
        (void)hipSetDevice(mp->ranks[0]->dev);
        hipError_t e = hipMemcpy(coef, dcoef, sizeof(double) * (size_t)ncol, hipMemcpyDeviceToHost);
        if (e == hipSuccess && hist_out && xtrap != 0.0)
            e = hipMemcpy(hist_out, mp->ranks[0]->p->hist, sizeof(double) * (size_t)ncol, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { set_error("hipMemcpy D2H failed"); rc = SPLPAK_E_NODEVICE; }
    }
    for (int r = 0; r < ngpus; ++r) {
        (void)hipSetDevice(mp->ranks[(size_t)r]->dev);
        for (double *q : {dx[(size_t)r], dy[(size_t)r], dw[(size_t)r]}) if (q) (void)hipFree(q);
    }
    (void)hipSetDevice(mp->ranks[0]->dev);
    if (dcoef) (void)hipFree(dcoef);
    (void)hipSetDevice(cur);
    splpak_mplan_destroy(mp);
    return rc;
}

}  // extern "C"
