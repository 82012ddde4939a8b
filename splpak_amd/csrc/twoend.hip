// Two-ended ("twisted") band Cholesky for narrow bands.
//
// A narrow-band factorisation is bound by its chain of dependent steps -- potrf(k) -> panel solve -> update of
// the next diagonal block, ~0.22 ms per 256 columns, with the chip mostly idle (2-D 64x64 nodes: 16 steps,
// 32^3 nodes: 128 steps).  The normal equations N (replacing the row-streaming
// Householder triangularisation of suprls, src/splpak.F90:1375-1695) are banded, so the elimination can start
// at BOTH ends: with w = block half-bandwidth and the block columns split into
//
//        top  = [0, m)        S = [m, m + w)        bottom = [m + w, nblk)
//
// no entry couples top and bottom, and in the ordering (top ascending, bottom DESCENDING, S last) the two
// eliminations are independent chains that only meet in the Schur complement of S:
//
//     band "top"  = block columns 0 .. m+w-1 of N in the natural order          (its last w blocks = S)
//     band "bot"  = block columns nblk-1 .. m of N in REVERSED order            (its last w blocks = S reversed,
//                   position q = npad - 1 - i                                     stored as ZERO: only updates)
//
//     1. band_cholesky(top, columns [0, m))  ||  band_cholesky(bot, columns [0, nb)),  nb = nblk - m - w
//     2. S(top) += flip(S(bot))                       the bottom chain's Schur complement, index-reversed
//     3. band_cholesky(top, columns [m, m + w))       factor S
//
// and the solves follow the same pattern (forward sweeps of both chains, sum into S, forward through S and
// back, the solved S handed to both backward sweeps).  Both bands are ordinary lower bands, so every kernel of
// bandchol.hip is used as it is; the critical path is max(m, nb) + w steps instead of nblk.
#include "plan.hpp"
#include <cstdlib>
#include <new>

namespace splpak {

namespace {

struct TwoEnd {
    Band top{}, bot{};
    int m = 0, w = 0, nb = 0;
    long long npad = 0;                 // order of the whole (padded) system
    double *x2 = nullptr, *tmp2 = nullptr;
    hipStream_t s2 = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
    std::vector<void *> owned;
};

// half stencil -> the two bands.  Entry (i, j), j <= i (natural internal order):
//   i <  cS1 (top or S row, so j < cS1 too)            -> top(i, j)
//   i >= cS1 (bottom row; j >= cS0 because i - j <= bw) -> bot(q(j), q(i)), q = npad - 1 - .   (q(j) >= q(i))
template <int D>
__global__ void __launch_bounds__(256)
expand2_kernel(Grid g, const double *__restrict__ nst, double *__restrict__ ab1, double *__restrict__ ab2,
               long long lda, int cS1, int npad)
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
        if (i < cS1) ab1[(long long)i + (long long)j * lda] = nst[t];
        else ab2[(long long)(npad - 1 - j) + (long long)(npad - 1 - i) * lda] = nst[t];
    }
}

// identity on the padding columns [n, npad) of the natural order = the first columns of the reversed band
__global__ void pad2_kernel(double *ab2, long long lda, int n, int npad)
{
    const int i = n + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npad) {
        const long long q = npad - 1 - i;
        ab2[q + q * lda] = 1.0;
    }
}

// S(top)(i, j) += S(bot)(q(j), q(i)) over the lower triangle of S x S; a, c = row / column inside S
__global__ void __launch_bounds__(256)
combine_kernel(double *__restrict__ ab1, const double *__restrict__ ab2, long long lda, int cS0, int W, int n2)
{
    const int a = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (a >= W || a < c) return;
    // natural i = cS0 + a has reversed position n2 - 1 - a inside the bottom band (its S part is the last W positions)
    const long long qi = n2 - 1 - a, qj = n2 - 1 - c;
    ab1[(long long)(cS0 + a) + (long long)(cS0 + c) * lda] += ab2[qj + qi * lda];
}

// x2 = reversed bottom part of x, zero on S
__global__ void split_kernel(const double *__restrict__ x, double *__restrict__ x2, int nbot, int n2, int npad)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q < n2) x2[q] = q < nbot ? x[npad - 1 - q] : 0.0;
}
// x(S) += flip(x2(S))
__global__ void addS_kernel(double *__restrict__ x, const double *__restrict__ x2, int cS0, int W, int n2)
{
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a < W) x[cS0 + a] += x2[n2 - 1 - a];
}
// x2(S) = flip(x(S))
__global__ void giveS_kernel(const double *__restrict__ x, double *__restrict__ x2, int cS0, int W, int n2)
{
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a < W) x2[n2 - 1 - a] = x[cS0 + a];
}
// bottom part of x = reversed x2
__global__ void merge_kernel(double *__restrict__ x, const double *__restrict__ x2, int nbot, int npad)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q < nbot) x[npad - 1 - q] = x2[q];
}

template <typename T>
bool te_alloc(TwoEnd *t, T **ptr, size_t count)
{
    void *q = nullptr;
    if (hipMalloc(&q, (count ? count : 1) * sizeof(T)) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    t->owned.push_back(q);
    *ptr = static_cast<T *>(q);
    return true;
}

void te_destroy(void *user)
{
    TwoEnd *t = static_cast<TwoEnd *>(user);
    if (!t) return;
    band_pipeline_destroy(t->top.pipe);
    band_pipeline_destroy(t->bot.pipe);
    for (hipEvent_t e : {t->e0, t->e1, t->e2}) if (e) (void)hipEventDestroy(e);
    if (t->s2) (void)hipStreamDestroy(t->s2);
    for (void *q : t->owned) (void)hipFree(q);
    delete t;
}

inline unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256 > 0 ? (n + 255) / 256 : 1); }

hipError_t te_factor_bands(TwoEnd *t, int *info_dev, double *minpiv_dev, hipStream_t st, CholStats *stats)
{
    (void)hipEventRecord(t->e0, st);
    (void)hipStreamWaitEvent(t->s2, t->e0, 0);
    hipError_t e = band_cholesky_narrow(t->top, info_dev, minpiv_dev, st, 0, t->m, 0);
    if (e != hipSuccess) return e;
    e = band_cholesky_narrow(t->bot, info_dev, minpiv_dev, t->s2, 0, t->nb, t->nb);
    if (e != hipSuccess) return e;
    (void)hipEventRecord(t->e1, t->s2);
    (void)hipStreamWaitEvent(st, t->e1, 0);
    const int W = t->w * NBLK;
    hipLaunchKernelGGL(combine_kernel, dim3(blocks_for(W), W), dim3(256), 0, st, t->top.ab, (const double *)t->bot.ab,
                       t->top.lda, t->m * NBLK, W, t->bot.npad);
    (void)stats;
    return band_cholesky_narrow(t->top, info_dev, minpiv_dev, st, t->m, t->m + t->w, t->m + t->w);
}

hipError_t te_solve_bands(TwoEnd *t, double *x, double *tmp, hipStream_t st)
{
    const int W = t->w * NBLK, cS0 = t->m * NBLK, nbot = t->nb * NBLK, n2 = t->bot.npad, npad = (int)t->npad;
    const int nblk1 = t->top.nblk;
    hipLaunchKernelGGL(split_kernel, dim3(blocks_for(n2)), dim3(256), 0, st, (const double *)x, t->x2, nbot, n2, npad);
    (void)hipEventRecord(t->e0, st);
    (void)hipStreamWaitEvent(t->s2, t->e0, 0);
    // forward: both chains, then S
    hipError_t e = band_forward(t->top, x, tmp, 0, t->m, st);
    if (e != hipSuccess) return e;
    e = band_forward(t->bot, t->x2, t->tmp2, 0, t->nb, t->s2);
    if (e != hipSuccess) return e;
    (void)hipEventRecord(t->e1, t->s2);
    (void)hipStreamWaitEvent(st, t->e1, 0);
    hipLaunchKernelGGL(addS_kernel, dim3(blocks_for(W)), dim3(256), 0, st, x, (const double *)t->x2, cS0, W, n2);
    e = band_forward(t->top, x, tmp, t->m, nblk1, st);
    if (e != hipSuccess) return e;
    // backward: S, which both chains then start from
    e = band_backward(t->top, x, tmp, nblk1, st, t->m, false);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(giveS_kernel, dim3(blocks_for(W)), dim3(256), 0, st, (const double *)x, t->x2, cS0, W, n2);
    (void)hipEventRecord(t->e2, st);
    (void)hipStreamWaitEvent(t->s2, t->e2, 0);
    e = band_backward(t->top, x, tmp, t->m, st, 0, true);
    if (e != hipSuccess) return e;
    e = band_backward(t->bot, t->x2, t->tmp2, t->nb, t->s2, 0, false);
    if (e != hipSuccess) return e;
    (void)hipEventRecord(t->e1, t->s2);
    (void)hipStreamWaitEvent(st, t->e1, 0);
    hipLaunchKernelGGL(merge_kernel, dim3(blocks_for(nbot)), dim3(256), 0, st, x, (const double *)t->x2, nbot, npad);
    return hipGetLastError();
}

// the split: m blocks from the top, nb from the bottom, w in the middle; false = not worth it / not possible
bool te_split(const Band &full, int *m, int *w, int *nb)
{
    *w = full.bw;
    if (*w < 1 || full.nblk - *w < 2) return false;
    *m = (full.nblk - *w + 1) / 2;
    *nb = full.nblk - *w - *m;
    return *m >= 1 && *nb >= 1;
}

// descriptors + storage of the two bands; `full` = descriptor of the whole band (band_bytes), whose ab (if any)
// becomes the top band's storage
TwoEnd *te_create(const Band &full)
{
    int m, w, nb;
    if (!te_split(full, &m, &w, &nb)) return nullptr;
    TwoEnd *t = new (std::nothrow) TwoEnd();
    if (!t) return nullptr;
    t->m = m; t->w = w; t->nb = nb; t->npad = full.npad;
    t->top = full;
    t->top.pipe = nullptr;
    t->top.nblk = m + w;
    t->top.npad = t->top.n = (m + w) * NBLK;
    t->bot = full;
    t->bot.pipe = nullptr;
    t->bot.nblk = nb + w;
    t->bot.npad = t->bot.n = (nb + w) * NBLK;
    const long long ld = full.lda + 1;
    t->bot.bytes = (size_t)ld * (size_t)t->bot.npad * sizeof(double) + 4096;
    const size_t nb2 = (size_t)NBLK * NBLK, k2 = (size_t)t->bot.nblk;
    bool ok = te_alloc(t, &t->bot.ab, t->bot.bytes / sizeof(double)) && te_alloc(t, &t->bot.dinv, k2 * nb2) &&
              te_alloc(t, &t->bot.dinvt, k2 * nb2) && te_alloc(t, &t->bot.mfwd, k2 * nb2) &&
              te_alloc(t, &t->bot.mbwd, k2 * nb2) && te_alloc(t, &t->bot.inv64, k2 * 4 * 64 * 64) &&
              te_alloc(t, &t->x2, (size_t)t->bot.npad) && te_alloc(t, &t->tmp2, (size_t)t->bot.npad);
    // the blocks of S are never inverted in the bottom band, but its last sweep pair reads them (unused products)
    ok = ok && hipMemset(t->bot.dinv, 0, k2 * nb2 * sizeof(double)) == hipSuccess &&
         hipMemset(t->bot.dinvt, 0, k2 * nb2 * sizeof(double)) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&t->s2, hipStreamNonBlocking) == hipSuccess;
    for (hipEvent_t *e : {&t->e0, &t->e1, &t->e2}) ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        te_destroy(t);
        return nullptr;
    }
    return t;
}

hipError_t te_expand(splpak_plan *p, hipStream_t st, void *user)
{
    TwoEnd *t = static_cast<TwoEnd *>(user);
    const Grid &g = p->g;
    // the top band lives in the plan's band storage (its first m + w block columns)
    hipError_t e = hipMemsetAsync(t->top.ab, 0, (size_t)(t->top.lda + 1) * (size_t)t->top.npad * sizeof(double), st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(t->bot.ab, 0, t->bot.bytes, st);
    if (e != hipSuccess) return e;
    const long long total = (long long)g.ncol * g.hstencil;
    long long nblocks = (total + 255) / 256;
    if (nblocks > 256LL * 64) nblocks = 256LL * 64;
    if (nblocks < 1) nblocks = 1;
    const dim3 gr((unsigned)nblocks), bl(256);
    const int cS1 = (t->m + t->w) * NBLK, npad = (int)t->npad;
    switch (g.ndim) {
    case 1: hipLaunchKernelGGL(expand2_kernel<1>, gr, bl, 0, st, g, (const double *)p->nst, t->top.ab, t->bot.ab, t->top.lda, cS1, npad); break;
    case 2: hipLaunchKernelGGL(expand2_kernel<2>, gr, bl, 0, st, g, (const double *)p->nst, t->top.ab, t->bot.ab, t->top.lda, cS1, npad); break;
    case 3: hipLaunchKernelGGL(expand2_kernel<3>, gr, bl, 0, st, g, (const double *)p->nst, t->top.ab, t->bot.ab, t->top.lda, cS1, npad); break;
    default: hipLaunchKernelGGL(expand2_kernel<4>, gr, bl, 0, st, g, (const double *)p->nst, t->top.ab, t->bot.ab, t->top.lda, cS1, npad); break;
    }
    if (p->band.npad > p->band.n)
        hipLaunchKernelGGL(pad2_kernel, dim3(blocks_for(p->band.npad - p->band.n)), dim3(256), 0, st, t->bot.ab, t->bot.lda,
                           p->band.n, p->band.npad);
    return hipGetLastError();
}

hipError_t te_factor(splpak_plan *p, int *info_dev, double *minpiv_dev, hipStream_t st, void *user)
{
    p->stats = CholStats{p->stats.enabled};        // no per-kernel accounting on this path (the chains, not a kernel, bound it)
    return te_factor_bands(static_cast<TwoEnd *>(user), info_dev, minpiv_dev, st, nullptr);
}

hipError_t te_solve(splpak_plan *, double *x, double *tmp, hipStream_t st, void *user)
{
    return te_solve_bands(static_cast<TwoEnd *>(user), x, tmp, st);
}

}  // namespace

// Narrow bands on one GPU: install the two-ended factorisation in the plan (plan.hip calls this once the band
// storage exists).  Not for bands of >= narrow_band_limit() blocks (28): there the trailing update, not the
// chain, bounds the factorisation and two chains would only share the matrix cores.
void twoend_attach(splpak_plan *p)
{
    if (p->dm.R != 1 || p->factor_fn || splpak::opt_get("SPLPAK_NO_TWOEND")) return;
    const int lim = narrow_band_limit();
    if (p->band.bw >= lim) return;
    TwoEnd *t = te_create(p->band);
    if (!t) return;                                // too few blocks (or no memory for the second band): one chain
    t->top.ab = p->band.ab;
    t->top.dinv = p->band.dinv; t->top.dinvt = p->band.dinvt;
    t->top.mfwd = p->band.mfwd; t->top.mbwd = p->band.mbwd; t->top.inv64 = p->band.inv64;
    p->expand_fn = te_expand;
    p->factor_fn = te_factor;
    p->solve_fn = te_solve;
    p->fn_user = t;
    p->fn_destroy = te_destroy;
    p->fn_name = "two-ended band Cholesky (both ends eliminated concurrently, csrc/twoend.hip)";
    p->fn_code = 2;
}

// another factorisation takes over the plan (the distributed band of dist.hip, also with one rank): release the
// two-ended state and its hooks
void twoend_detach(splpak_plan *p)
{
    if (p->fn_destroy != te_destroy) return;
    te_destroy(p->fn_user);
    p->expand_fn = nullptr;
    p->factor_fn = nullptr;
    p->solve_fn = nullptr;
    p->fn_user = nullptr;
    p->fn_destroy = nullptr;
    p->fn_name = nullptr;
    p->fn_code = 0;
}

// Dense-input debugging entry (splpak_debug_spd_band_solve_f64 with two ends): factor and solve an SPD band
// matrix given as a dense lower triangle on the host.  1 = the matrix has too few blocks for two ends.
int twoend_debug_solve(int n, int halfbw, const double *a_lower, const double *bvec, double *x_out, int *hinfo_out)
{
    Band full{};
    band_bytes(n, halfbw, &full);
    TwoEnd *t = te_create(full);
    if (!t) return 1;
    const size_t nb2 = (size_t)NBLK * NBLK, k1 = (size_t)t->top.nblk;
    const size_t top_doubles = (size_t)(full.lda + 1) * (size_t)t->top.npad + 512;
    double *x = nullptr, *tmp = nullptr, *small = nullptr;
    int *dinfo = nullptr;
    bool ok = te_alloc(t, &t->top.ab, top_doubles) && te_alloc(t, &t->top.dinv, k1 * nb2) &&
              te_alloc(t, &t->top.dinvt, k1 * nb2) && te_alloc(t, &t->top.mfwd, k1 * nb2) &&
              te_alloc(t, &t->top.mbwd, k1 * nb2) && te_alloc(t, &t->top.inv64, k1 * 4 * 64 * 64) &&
              te_alloc(t, &x, (size_t)full.npad) && te_alloc(t, &tmp, (size_t)full.npad) && te_alloc(t, &small, 8) &&
              te_alloc(t, &dinfo, 2);
    int rc = ok ? 0 : SPLPAK_E_NOMEM;
    if (ok) {
        const int cS0 = t->m * NBLK, cS1 = (t->m + t->w) * NBLK, npad = full.npad;
        const long long lda = full.lda;
        std::vector<double> h1(top_doubles, 0.0), h2(t->bot.bytes / sizeof(double), 0.0), hx((size_t)npad, 0.0);
        for (int j = 0; j < n; ++j)
            for (int i = j; i < n && i - j <= halfbw; ++i) {
                const double v = a_lower[(size_t)i + (size_t)j * n];
                if (i < cS1) h1[(size_t)i + (size_t)j * lda] = v;
                else if (j >= cS0) h2[(size_t)(npad - 1 - j) + (size_t)(npad - 1 - i) * lda] = v;
                else if (v != 0.0) rc = SPLPAK_E_BADARG;          // an entry outside the stated band
            }
        for (int i = n; i < npad; ++i) { const size_t q = (size_t)(npad - 1 - i); h2[q + q * lda] = 1.0; }
        for (int i = 0; i < n; ++i) hx[(size_t)i] = bvec[i];
        const double inf = 1.0 / 0.0;
        hipError_t e = hipMemcpy(t->top.ab, h1.data(), top_doubles * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(t->bot.ab, h2.data(), h2.size() * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(x, hx.data(), sizeof(double) * (size_t)npad, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(small + 2, &inf, sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemset(dinfo, 0, 2 * sizeof(int));
        if (e == hipSuccess) e = te_factor_bands(t, dinfo, small + 2, nullptr, nullptr);
        if (e == hipSuccess) e = te_solve_bands(t, x, tmp, nullptr);
        int hinfo = 0;
        if (e == hipSuccess) e = hipMemcpy(&hinfo, dinfo, sizeof(int), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(hx.data(), x, sizeof(double) * (size_t)npad, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = SPLPAK_E_NODEVICE;
        if (hinfo_out) *hinfo_out = hinfo;
        if (rc == 0) for (int i = 0; i < n; ++i) x_out[i] = hx[(size_t)i];
    }
    te_destroy(t);
    return rc;
}

}  // namespace splpak
