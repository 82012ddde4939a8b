// Device pieces of the blocked Cholesky shared by the band factorisation (bandchol.hip) and the nested-dissection
// multifrontal factorisation (ndchol.hip): the 256x256 diagonal-block factorisation in strip form, the panel solve
// on the f64 matrix cores, small helpers.  Everything lives in an anonymous namespace: each translation unit gets
// its own copy, the kernels that wrap these bodies are defined where they are launched.
#pragma once
#include "kernels.hpp"

namespace splpak {
namespace {

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));


constexpr int IB = 16;                 // inner panel width
constexpr int PTB = 4;                 // 16x16 tiles of the in-block update per wave and round
constexpr int XLD = NBLK + 16;         // LDS row of the panel image (bank-half alternation, as in syrk)

__device__ inline double readlane_f64(double v, int srclane)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, srclane);
    hi = __builtin_amdgcn_readlane(hi, srclane);
    return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------------------
// Cholesky of one 256x256 diagonal block, strip form (round 2).  The kernel above sends the whole
// trailing part of the block through global memory (L2) once per 16-column panel: 16 passes, ~2.7 MB per
// block, which is where its time goes (47 % "in-block update", mostly waiting for C tiles).  Here the
// block is processed in four 64-column STRIPS that live in LDS (rows below the strip's first column x 64
// columns, 139 KB): the sixteen 16-column steps -- leaf factorisation by wave 0, row solves, update of the
// rest of the strip -- never leave the CU, and the part of the block right of the strip is updated ONCE
// per strip with K = 64 from the LDS-resident strip: 4 passes, ~0.6 MB of global traffic per block.  Same
// arithmetic per element as the panel form (the sums over the 16-column panels are taken in the same
// order), rolled loops (cold instruction cache, see above).  It has a CU's LDS to itself, which is what
// the reserved CU of the look-ahead pipeline gives it anyway.
constexpr int SPW = 64;                // strip width
constexpr int SLD = NBLK + 16;         // LDS column stride of the strip image S[c*SLD + r]

// The part of the diagonal block right of a finished 64-column strip: C -= S S^T with K = 64, C tiles in global
// memory (lower part), operands from the LDS strip image S.  32x32 pieces per wave (2x2 MFMA tiles: four independent
// accumulator chains, one LDS operand read per MFMA); the next piece's C values are in flight while the current one
// is computed.  (16x16 pieces with one dependent chain of 16 MFMAs each took 44 us per block, 5x their MFMA time.)
// Not inlined: its register allocation and scheduling stay apart from the latency-critical leaf / row-solve code.
typedef const __attribute__((address_space(3))) double *lds_cptr;      // LDS pointer that survives a function boundary as ds_read
__device__ __noinline__ void strip_trailing_update(double *__restrict__ A, long long lda, lds_cptr S,
                                                    int c0, int wave, int l15, int q, int nw)
{
    const int base = c0 + SPW;
    const int nt = (NBLK - base) / 32;
    const int ntiles = nt * (nt + 1) / 2;
    auto decode = [&](int t, int &roff, int &coff) {        // tile t -> column-major over the lower triangle
        int ct = 0, rem = t;
        while (rem >= nt - ct) { rem -= nt - ct; ++ct; }
        coff = base + 32 * ct;
        roff = base + 32 * (ct + rem);
    };
    auto cload = [&](int roff, int coff, d4_t (&cc)[2][2]) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    cc[mi][ni][v] = A[(roff + 16 * ni + l15) + (long long)(coff + 16 * mi + q + 4 * v) * lda];
    };
    d4_t cur[2][2], nxt[2][2];
    int t = wave, roff = 0, coff = 0;
    if (t < ntiles) {
        decode(t, roff, coff);
        cload(roff, coff, cur);
    }
    while (t < ntiles) {
        const int tn = t + nw;
        int rn = 0, cn = 0;
        if (tn < ntiles) {
            decode(tn, rn, cn);
            cload(rn, cn, nxt);
        }
#pragma unroll 4
        for (int s4 = 0; s4 < SPW / 4; ++s4) {
            lds_cptr Sk = S + (4 * s4 + q) * SLD + l15;
            const double a0 = -Sk[coff], a1 = -Sk[coff + 16];
            const double b0 = Sk[roff], b1 = Sk[roff + 16];
            cur[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, cur[0][0], 0, 0, 0);
            cur[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, cur[0][1], 0, 0, 0);
            cur[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, cur[1][0], 0, 0, 0);
            cur[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, cur[1][1], 0, 0, 0);
        }
        const bool diag = roff == coff;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int rr = 16 * ni + l15, cc = 16 * mi + q + 4 * v;      // inside the 32x32 piece
                    if (!diag || rr >= cc) A[(roff + rr) + (long long)(coff + cc) * lda] = cur[mi][ni][v];
                }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) cur[mi][ni] = nxt[mi][ni];
        t = tn;
        roff = rn;
        coff = cn;
    }
}

#ifdef SPLPAK_POTRF_STAMPS        // tools/potrf_probe.hip: cycles of wave 0 per phase, summed over the kernel
__device__ unsigned long long g_strip_cycles[8];
#define STRIP_T0() unsigned long long st_last = __builtin_amdgcn_s_memtime()
#define STRIP_ACC(i) do { const unsigned long long st_now = __builtin_amdgcn_s_memtime(); if (tid == 0) g_strip_cycles[i] += st_now - st_last; st_last = st_now; } while (0)
#else
#define STRIP_T0() do { } while (0)
#define STRIP_ACC(i) do { } while (0)
#endif
// A = the diagonal block (A(r,c) = A[r + c*lda], r >= c); k0 only labels the pivot index reported through info.
// The minimum pivot is tracked with an integer atomic on the bit pattern (positive doubles order like their
// bits), so that several blocks may be factored by concurrent workgroups (the batched launches of ndchol.hip).
// ncols (1 .. 256): the columns from ncols on are identity padding (a front of the nested-dissection factorisation whose
// own variables do not fill its last block): strips that hold nothing else are skipped -- their part of L is the identity
// the assembly left there, and the leaf inverses at the end read it.
// NW waves per workgroup (4 or 16): the leaf and the row solves are the work of waves 0 .. 3 either way; the strip copies, the
// in-strip updates and the K = 64 update of the rest of the block are shared by all of them (round 3: 16 waves).
template <int NW = 4>
__device__ __forceinline__ void potrf_strip_body(double *__restrict__ A, long long lda, int k0, int *__restrict__ info,
                                                 double *__restrict__ minpiv, double *__restrict__ inv16, int ncols = NBLK)
{
    static_assert(NW == 4 || NW == 8 || NW == 16, "whole groups of 256 threads");
    constexpr int NP = NW / 4;             // groups of 256 threads: the strip copies split the columns among them
    __shared__ double Ls[IB * (IB + 1)];
    __shared__ double Lrd[IB];               // reciprocals of the leaf's diagonal
    __shared__ double S[SPW * SLD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q = lane >> 4;
    __builtin_amdgcn_s_setprio(3);
    STRIP_T0();

    const int cend = ((ncols + SPW - 1) / SPW) * SPW;        // first column of the first all-padding strip
    for (int c0 = 0; c0 < cend; c0 += SPW) {
        // ---- strip -> LDS (whole rectangle rows >= c0; the part above the diagonal is never used):
        // one row per thread, 16 columns in flight at a time
        if (c0 + (tid & 255) < NBLK) {
            const double *__restrict__ rowp = A + (c0 + (tid & 255)) + (long long)c0 * lda;
#pragma unroll 1
            for (int cb = 16 * (tid >> 8); cb < SPW; cb += 16 * NP) {
                double v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = rowp[(long long)(cb + i) * lda];
#pragma unroll
                for (int i = 0; i < 16; ++i) S[(cb + i) * SLD + c0 + (tid & 255)] = v[i];
            }
        }
        __syncthreads();
        STRIP_ACC(0);
        // one 16x16 tile of the in-strip update by the panel at strip column ppc (first row below its leaf: pbase):
        // tile column tc (strip column ppc+16+16 tc), tile row tr >= tc; an MFMA chain of K = 16 on LDS operands
        auto tile_update = [&](int ppc, int pbase, int tc, int tr) {
            const int scol = ppc + IB + 16 * tc;        // strip column of the tile's first column
            const int crow = pbase + 16 * tc;           // block row that corresponds to that column
            const int rrow = pbase + 16 * tr;
            d4_t acc;
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[v] = S[(scol + q + 4 * v) * SLD + rrow + l15];
#pragma unroll
            for (int s4 = 0; s4 < IB / 4; ++s4) {
                const double av = -S[(ppc + 4 * s4 + q) * SLD + crow + l15];
                const double bv = S[(ppc + 4 * s4 + q) * SLD + rrow + l15];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) S[(scol + q + 4 * v) * SLD + rrow + l15] = acc[v];
        };
        for (int pc = 0; pc < SPW; pc += IB) {     // 16-column steps inside the strip
            const int d0 = c0 + pc;                // first row / column of the leaf (block relative)
            const int base = d0 + IB;              // first row below the leaf
            const int mrem = NBLK - base;
            // (i) leaf: wave 0, one row per lane, columns broadcast with v_readlane.  Beside it waves 1..3 apply
            // the PREVIOUS panel to the strip columns right of this panel (look-ahead: only this panel's own
            // columns were updated before the leaf could start)
            if (wave == 0) {
                const int r = l15;
                double a[IB];
#pragma unroll
                for (int c = 0; c < IB; ++c) a[c] = (c <= r) ? S[(pc + c) * SLD + d0 + r] : 0.0;
                double dmin = a[0];
                bool bad = false;
                double rdiag = 0.0;
#pragma unroll
                for (int j = 0; j < IB; ++j) {
                    const double d = readlane_f64(a[j], j);
                    bad = bad || !(d > 0.0);
                    dmin = fmin(dmin, d);
                    // 1/sqrt(d) from the hardware estimate + ONE third-order (Halley) step, sqrt(d) = d * rs corrected
                    // once: the sqrt and the division of the textbook form are ~55 instructions on the critical
                    // path of every column, two Newton steps 8 dependent ones, this is 4 (e = 1 - d rs^2 is ~2^-26
                    // after v_rsq_f64, the step leaves e^3: results within an ulp of the correctly rounded ones)
                    double rs = __builtin_amdgcn_rsq(d);
                    {
                        const double e = fma(-d * rs, rs, 1.0);
                        rs = fma(rs * e, fma(0.375, e, 0.5), rs);
                    }
                    double sd = d * rs;
                    sd = fma(fma(-sd, sd, d), 0.5 * rs, sd);
                    if (r == j) rdiag = rs;
                    a[j] = (r == j) ? sd : a[j] * rs;
#pragma unroll
                    for (int c = j + 1; c < IB; ++c) a[c] -= a[j] * readlane_f64(a[j], c);
                }
                if (lane < IB) {
                    Lrd[r] = rdiag;
#pragma unroll
                    for (int c = 0; c < IB; ++c) {
                        Ls[r * (IB + 1) + c] = a[c];
                        if (c <= r) S[(pc + c) * SLD + d0 + r] = a[c];
                    }
                }
                if (lane == 0) {
                    if (bad) atomicCAS(info, 0, k0 + d0 + 1);
                    if (dmin > 0.0) atomicMin(reinterpret_cast<unsigned long long *>(minpiv), (unsigned long long)__double_as_longlong(dmin));
                    else *minpiv = dmin;         // non-positive or NaN: the factorisation has failed anyway (info)
                }
            } else if (pc > 0) {
                const int ppc = pc - IB, pbase = d0;           // the previous panel; the first row below its leaf is this leaf's
                const int ntc = (SPW - ppc - IB) / 16, ntr = (NBLK - pbase) / 16;
                int ntl = 0;                                    // tiles of the tile columns 1 .. ntc-1
                for (int tc = 1; tc < ntc; ++tc) ntl += ntr - tc;
                for (int t = wave - 1; t < ntl; t += NW - 1) {
                    int tc = 1, rem = t;
                    while (rem >= ntr - tc) { rem -= ntr - tc; ++tc; }
                    tile_update(ppc, pbase, tc, tc + rem);
                }
            }
            __syncthreads();
            STRIP_ACC(1);
            // (ii) rows below the leaf: x = a L^{-T}, one row per thread, in place in the strip
            if (tid < mrem) {
                const int row = base + tid;
                double x[IB];
#pragma unroll
                for (int c = 0; c < IB; ++c) x[c] = S[(pc + c) * SLD + row];
                // (bound by the 136 broadcasts of leaf elements, not by the FMAs: the same time whether they are
                // LDS broadcast reads as here, or v_readlane from a register copy of the leaf, column by column or
                // right-looking -- measured; without the arithmetic the phase is 10x shorter)
#pragma unroll
                for (int c = 0; c < IB; ++c) {
#pragma unroll
                    for (int k = 0; k < c; ++k) x[c] -= x[k] * Ls[c * (IB + 1) + k];
                    x[c] *= Lrd[c];
                }
#pragma unroll
                for (int c = 0; c < IB; ++c) S[(pc + c) * SLD + row] = x[c];
            }
            __syncthreads();
            STRIP_ACC(2);
            // (iii) the next panel's 16 columns (tile column 0) now, by all waves; the columns beyond them wait
            // for the next leaf (above).  Every tile still receives its panels in the same order as before.
            if (pc + IB < SPW) {
                const int ntr = mrem / 16;
                for (int t = wave; t < ntr; t += NW) tile_update(pc, base, 0, t);
                __syncthreads();
            }
            STRIP_ACC(3);
        }
        // ---- the finished strip (64 columns of L) back to global: lower part only, one row per thread
        if (c0 + (tid & 255) < NBLK) {
            const int row = tid & 255;
            double *__restrict__ rowp = A + (c0 + row) + (long long)c0 * lda;
            const int cmax = row < SPW - 1 ? row : SPW - 1;     // row c0+row holds columns c0 .. c0+min(row, 63)
#pragma unroll 4
            for (int c = tid >> 8; c <= cmax; c += NP) rowp[(long long)c * lda] = S[c * SLD + c0 + row];
        }
        STRIP_ACC(4);
        // ---- the block right of the strip: C -= S S^T with K = 64 (strip_trailing_update above)
        if (c0 + SPW < cend) strip_trailing_update(A, lda, (lds_cptr)S, c0, wave, l15, q, NW);     // (right of it: padding only)
        __syncthreads();        // everybody is done with the strip (and its stores are issued) before it is replaced
        __threadfence_block();
        STRIP_ACC(5);
    }
    // the next strip's loads read what this workgroup stored: make the stores visible to the whole workgroup
    __syncthreads();
    // inverses of the sixteen 16x16 diagonal leaves, as in potrf_block_kernel
    if (wave < 4) {
        const int leaf = wave * 4 + q, r = l15;
        const int d0 = leaf * IB;
        double a[IB];
#pragma unroll
        for (int c = 0; c < IB; ++c) a[c] = (c <= r) ? A[(d0 + r) + (long long)(d0 + c) * lda] : 0.0;
        double xi[IB];
#pragma unroll
        for (int rr = 0; rr < IB; ++rr) {
            double sacc = (rr == r) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < rr; ++k) sacc -= __shfl(a[k], rr, 16) * xi[k];
            xi[rr] = sacc / __shfl(a[rr], rr, 16);
        }
        double *out = inv16 + leaf * (IB * IB) + r * IB;
#pragma unroll
        for (int rr = 0; rr < IB; ++rr) out[rr] = (rr >= r) ? xi[rr] : 0.0;
    }
}

// ---------------------------------------------------------------------------
// X = A * L^{-T} for rows below the diagonal block, on the f64 matrix cores.
//
// One wave owns 16 rows for the whole solve and walks the 256 columns in 16-column blocks:
//     X_c = (A_c - sum_{k<c} X_k L_ck^T) Inv_cc^T ,   Inv_cc = (16x16 diagonal leaf of L)^{-1}
// (the leaf inverses come out of potrf_block_kernel; -X_k is what is parked, so the sum is
// accumulated into A_c directly).  Both products are computed transposed,
// D[col][row] = sum_k Aop[col][k] * Bop[k][row]: the lane that holds row = lane&15 of an
// accumulator tile holds, in register s, exactly the B operand of k-step s, so T = A_c - ...
// feeds the Inv product straight from registers.  Finished blocks X_k are parked in LDS
// ([column][row], 30 KB) so the block-row loop stays rolled: the kernel runs once per step on
// every CU with a cold instruction cache.  Single-wave workgroups of < 200 registers: a wave
// takes the place of one retiring trailing-update wave (252 registers, two per SIMD) beside the
// other one.  Only L / Inv elements (L2 resident, shared by all waves) and the wave's own rows
// are loaded.
// The L operands of block row cb+1 and its right-hand side are fetched while block row cb is
// being computed (lb[] is refilled slot by slot as soon as the MFMA that read the slot has been
// issued): one exposed L2 round trip per block row instead of one per two k blocks took the
// in-pipeline kernel from 162 to 103 us.  Measured alternatives that lost: 32 rows per wave
// (257 registers: no longer fits beside a trailing-update wave, 380 us), unguarded refills
// (exactly counted waits, but 76 % more loads through the L1 that the update waves stream
// their operands through: 127 us).
// EYE: the right-hand side is the identity (rows r0.. of it, never read from memory), X has its own leading
// dimension ldx and is also stored transposed into Xt: X = L^{-T}, i.e. the inverse of the diagonal block
// (trinv_kernel below).
// KREG (round 5): the first KREG finished blocks are parked in registers, not in LDS -- the operand a later product needs from
// a parked block is the value the SAME lane produced (accumulator register v = k-step s), so LDS is only storage here, and at
// 30 KB per wave it held the panel solves to 5 waves per CU; with 6 blocks in registers xs is (NBLK - 16 - 16 KREG) x 16
// doubles = 18 KB and 8 waves fit.
template <bool EYE, int KREG = 0>
__device__ __forceinline__ void trsm_rows(const double *__restrict__ L, double *__restrict__ Xbase, long long lda,
                                          long long ldx, const double *__restrict__ inv16, double *__restrict__ Xt,
                                          int r0, double *__restrict__ xs, int ncb = NBLK / 16)
{   // ncb: 16-column blocks to solve (the columns beyond are identity padding of a front: their X is the zero already there)
    const int lane = threadIdx.x & 63, l15 = lane & 15, q = lane >> 4;
    double *__restrict__ Xr = Xbase + r0 + l15;          // Xr[c*ldx] = X(row, c)
    constexpr int NCB = NBLK / 16;

    double lb[NCB - 1][4];        // lb[kb][s] = L(16 cb + l15, 16 kb + 4 s + q), block row cb (then cb+1)
    double xp[KREG > 0 ? KREG : 1][4];       // parked blocks 0 .. KREG-1 (negated), this lane's four values of each
    d4_t Tn;
#pragma unroll
    for (int v = 0; v < 4; ++v) Tn[v] = EYE ? (r0 + l15 == q + 4 * v ? 1.0 : 0.0) : Xr[(long long)(q + 4 * v) * ldx];
#pragma unroll 1
    for (int cb = 0; cb < ncb; ++cb) {          // stays rolled: the code must stay small (cold I-cache)
        d4_t T = Tn;
        const bool more = cb + 1 < ncb;
        const double *__restrict__ Inv = inv16 + cb * 256;           // Inv[row + 16*col]
        double iv[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) iv[s] = Inv[l15 + 16 * (4 * s + q)];
        if (more) {
#pragma unroll
            for (int v = 0; v < 4; ++v)
                Tn[v] = EYE ? (r0 + l15 == 16 * (cb + 1) + q + 4 * v ? 1.0 : 0.0) : Xr[(long long)(16 * (cb + 1) + q + 4 * v) * ldx];
        }
        const double *__restrict__ Ln = L + (16 * (cb + 1) + l15) + (long long)q * lda;   // block row cb+1
#pragma unroll
        for (int kb = 0; kb < NCB - 1; ++kb) {
            if (kb < cb) {
                double bq[4];
                if (kb < KREG) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) bq[s] = xp[kb < KREG ? kb : 0][s];
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) bq[s] = xs[(16 * (kb - KREG) + 4 * s + q) * 16 + l15];
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) T = __builtin_amdgcn_mfma_f64_16x16x4f64(lb[kb][s], bq[s], T, 0, 0, 0);
            }
            if (kb <= cb && more) {                    // slot kb is free: tile (cb+1, kb)
#pragma unroll
                for (int s = 0; s < 4; ++s) lb[kb][s] = Ln[(long long)(16 * kb + 4 * s) * lda];
            }
        }
        d4_t X = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) X = __builtin_amdgcn_mfma_f64_16x16x4f64(iv[s], T[s], X, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int c = 16 * cb + q + 4 * v;
            Xr[(long long)c * ldx] = X[v];
            if (EYE) Xt[(long long)(r0 + l15) * ldx + c] = X[v];
            if (more && cb >= KREG) xs[(c - 16 * KREG) * 16 + l15] = -X[v];
        }
        if (more && cb < KREG) {
#pragma unroll
            for (int kk = 0; kk < KREG; ++kk)
                if (cb == kk) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) xp[kk][v] = -X[v];
                }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------
__device__ inline double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace
}  // namespace splpak
