// Experimental / earlier kernels of the band factorisation, kept OUT of libsplpak_hip.so:
//   potrf_block_kernel   round 1's panel-form Cholesky of a 256x256 diagonal block (with cycle stamps for tools/potrf_probe.hip)
//   syrk_kernel          the LDS-tiled 128x128 trailing update
//   syrk64_abl_kernel    the register-streaming trailing update WITH its ablation switches (ABL bits) and K = 512 / 1024 forms
// Included by tools/syrk_bench.hip and tools/potrf_probe.hip after ../splpak_amd/csrc/bandchol.hip, whose constants, typedefs
// and helpers (my_cu_id, d4_t, ...) they use.  The findings these kernels produced are in DESIGN.md section 4.
#pragma once
namespace splpak {
namespace {

// ---------------------------------------------------------------------------
// Cholesky of one 256x256 diagonal block: one workgroup of 4 waves, 16-column panels.
// (The inverses of the sixteen 16x16 leaves, needed by the panel solve, are computed together at the
// end: every 16-lane group takes one leaf -- 290 -> 250 us for the kernel, tools/potrf_probe.hip.)
// Per panel: (i) the 16x16 diagonal block is factored by wave 0 alone, one row per lane in
// registers, columns broadcast with v_readlane (no barriers); (ii) the rows below are solved
// one row per thread against that factor (broadcast LDS reads); (iii) the rest of the block
// is updated on the f64 matrix cores from the LDS-resident panel, four 16x16 tiles per wave
// in flight.  All global loads of a panel are issued before phase (i) starts, so a panel
// costs about one memory round trip.
//
// The kernel is on the critical path of the look-ahead, runs ONCE per step on whatever CU the
// dispatcher finds, i.e. with a cold instruction cache beside MFMA-saturating update waves:
// measured cost of cold code was ~0.5 us per 64-B line, which made a 60 KB fully unrolled
// version take 700 us against 215 us alone.  Hence 16-wide panels and rolled loops: the
// whole kernel is ~14 KB.  LDS 37 KB, <= 200 VGPRs: it fits beside trailing-update waves.

#ifdef SPLPAK_POTRF_STAMPS        // tools/potrf_probe.hip: cycle stamps of wave 0 at the phase boundaries
__device__ unsigned long long g_potrf_stamps[16 * 8];
#define POTRF_STAMP(i) do { if (tid == 0) g_potrf_stamps[(c0 / IB) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define POTRF_STAMP(i) do { } while (0)
#endif
__global__ void __launch_bounds__(256)
potrf_block_kernel(double *__restrict__ ab, long long lda, int k0, int *__restrict__ info,
                   double *__restrict__ minpiv, double *__restrict__ inv16)
{
    __shared__ double Ls[IB * (IB + 1)];     // factor of the current diagonal 16x16: Ls[c*(IB+1) + k]
    __shared__ double Xs[IB * XLD];          // panel image Xs[k*XLD + r], r = row inside the 256 block
    double *A = ab + (long long)k0 + (long long)k0 * lda;    // A(r,c) = A[r + c*lda], r >= c
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: tile bookkeeping stays scalar
    const int l15 = lane & 15, q = lane >> 4;
    __builtin_amdgcn_s_setprio(3);           // critical path of the look-ahead: win issue arbitration

    for (int c0 = 0; c0 < NBLK; c0 += IB) {
        const int base = c0 + IB;            // first row / column of the trailing part
        const int mrem = NBLK - base;        // rows below the diagonal block
        const int nt = mrem / 16;
        const int ntiles = nt * (nt + 1) / 2;

        POTRF_STAMP(0);
        // ---- every load of this panel is issued here
        double x[IB];                        // (ii) this thread's row of the panel
        const int row = base + tid;
        if (tid < mrem) {
#pragma unroll
            for (int c = 0; c < IB; ++c) x[c] = A[row + (long long)(c0 + c) * lda];
        }
        // (i) diagonal block: wave 0, one row per lane (the four 16-lane groups mirror each other)
        if (wave == 0) {
            const int r = l15;
            double a[IB];
#pragma unroll
            for (int c = 0; c < IB; ++c) a[c] = (c <= r) ? A[(c0 + r) + (long long)(c0 + c) * lda] : 0.0;
            POTRF_STAMP(1);
            double dmin = a[0];
            bool bad = false;
#pragma unroll
            for (int j = 0; j < IB; ++j) {
                const double d = readlane_f64(a[j], j);
                bad = bad || !(d > 0.0);
                dmin = fmin(dmin, d);
                const double sd = sqrt(d);
                a[j] = (r == j) ? sd : a[j] / sd;
#pragma unroll
                for (int c = j + 1; c < IB; ++c) a[c] -= a[j] * readlane_f64(a[j], c);
            }
            POTRF_STAMP(2);
            if (lane < IB) {
#pragma unroll
                for (int c = 0; c < IB; ++c) {
                    Ls[r * (IB + 1) + c] = a[c];
                    if (c <= r) A[(c0 + r) + (long long)(c0 + c) * lda] = a[c];
                }
            }
            if (lane == 0) {
                if (bad) atomicCAS(info, 0, k0 + c0 + 1);
                if (dmin < *minpiv || !(dmin == dmin)) *minpiv = dmin;
            }
        }
        POTRF_STAMP(3);
        __syncthreads();
        POTRF_STAMP(4);
        // (ii) rows below: x = a L^{-T}, one row per thread
        if (tid < mrem) {
#pragma unroll
            for (int c = 0; c < IB; ++c) {
#pragma unroll
                for (int k = 0; k < c; ++k) x[c] -= x[k] * Ls[c * (IB + 1) + k];
                x[c] /= Ls[c * (IB + 1) + c];
            }
#pragma unroll
            for (int c = 0; c < IB; ++c) {
                A[row + (long long)(c0 + c) * lda] = x[c];
                Xs[c * XLD + row] = x[c];
            }
        }
        POTRF_STAMP(5);
        __syncthreads();
        POTRF_STAMP(6);
        // (iii) trailing update inside the block on the matrix cores: 16x16 tiles, rt >= ct,
        // PTB tiles per wave and round with all their C loads in flight together.  The kernel has a
        // SIMD per wave to itself, so every instruction is exposed: tile addresses are wave-uniform
        // (scalar) bases plus one 32-bit per-lane offset, the tile index advances incrementally and
        // only diagonal tiles take the predicated store path.
        {
            const unsigned voff = (unsigned)(l15 + (long long)q * lda) * 8u;      // bytes; q*lda*8 < 2^20
            int ct = 0, rem = wave * PTB;                                        // tile t -> (ct, ct + rem)
            while (rem >= nt - ct && ct < nt) { rem -= nt - ct; ++ct; }
            // a round = PTB tiles of this wave; the C tiles of the next round are loaded before the
            // current one is computed (two register sets, loop unrolled by two)
            auto load_round = [&](int t0, d4_t (&cc)[PTB], int (&roff)[PTB], int (&coff)[PTB]) {
                int ctu = ct, remu = rem;
#pragma unroll
                for (int u = 0; u < PTB; ++u) {
                    roff[u] = -1;
                    coff[u] = 0;
                    if (t0 + u < ntiles) {
                        roff[u] = base + 16 * (ctu + remu);
                        coff[u] = base + 16 * ctu;
                        const char *tile = reinterpret_cast<const char *>(A + roff[u] + (long long)coff[u] * lda);
#pragma unroll
                        for (int v = 0; v < 4; ++v)
                            cc[u][v] = *reinterpret_cast<const double *>(tile + (long long)(4 * v) * lda * 8 + voff);
                    }
                    if (++remu >= nt - ctu) { remu = 0; ++ctu; }
                }
                rem += 4 * PTB;                      // this wave's next round
                while (ct < nt && rem >= nt - ct) { rem -= nt - ct; ++ct; }
            };
            auto compute_round = [&](d4_t (&cc)[PTB], int (&roff)[PTB], int (&coff)[PTB]) {
#pragma unroll
                for (int u = 0; u < PTB; ++u) {
                    if (roff[u] >= 0) {
                        d4_t acc = cc[u];                 // C - X X^T accumulated in place (negated operand)
#pragma unroll
                        for (int s4 = 0; s4 < IB / 4; ++s4) {
                            const double av = -Xs[(4 * s4 + q) * XLD + coff[u] + l15];
                            const double bv = Xs[(4 * s4 + q) * XLD + roff[u] + l15];
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                        }
                        char *tile = reinterpret_cast<char *>(A + roff[u] + (long long)coff[u] * lda);
                        if (roff[u] != coff[u]) {
#pragma unroll
                            for (int v = 0; v < 4; ++v)
                                *reinterpret_cast<double *>(tile + (long long)(4 * v) * lda * 8 + voff) = acc[v];
                        } else {
#pragma unroll
                            for (int v = 0; v < 4; ++v)
                                if (l15 >= q + 4 * v)
                                    *reinterpret_cast<double *>(tile + (long long)(4 * v) * lda * 8 + voff) = acc[v];
                        }
                    }
                }
            };
            d4_t ccA[PTB], ccB[PTB];
            int roA[PTB], coA[PTB], roB[PTB], coB[PTB];
            int tA = wave * PTB;
            if (tA < ntiles) load_round(tA, ccA, roA, coA);
            while (tA < ntiles) {
                const int tB = tA + 4 * PTB;
                if (tB < ntiles) load_round(tB, ccB, roB, coB);
                compute_round(ccA, roA, coA);
                if (tB >= ntiles) break;
                tA = tB + 4 * PTB;
                if (tA < ntiles) load_round(tA, ccA, roA, coA);
                compute_round(ccB, roB, coB);
            }
        }
        POTRF_STAMP(7);
        __syncthreads();
    }
    // Inverses of the sixteen 16x16 diagonal leaves (the panel solve's MFMA operands), all at once:
    // every 16-lane group of every wave takes one leaf; lane c of a group solves L x = e_c by
    // forward substitution, L(rr,k) comes from the lane of the group that holds row rr.
    {
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
        double *out = inv16 + leaf * (IB * IB) + r * IB;    // column r: out[row]
#pragma unroll
        for (int rr = 0; rr < IB; ++rr) out[rr] = (rr >= r) ? xi[rr] : 0.0;
    }
}




// ---------------------------------------------------------------------------
// Trailing update on the f64 matrix cores.
constexpr int TS = 128;            // C tile edge per workgroup
constexpr int KC = 16;             // K chunk staged per LDS buffer
constexpr int LDT = TS + 16;       // padded LDS row: k-rows land on alternating bank halves

__global__ void __launch_bounds__(256, 2)
syrk_kernel(double *__restrict__ ab, long long lda, int k0, int row0, int nt, int tj_begin)
{
    __shared__ double sI[2][KC * LDT];   // panel rows of the C-row block  (MFMA B operand)
    __shared__ double sJ[2][KC * LDT];   // panel rows of the C-col block  (MFMA A operand)

    // block -> lower-triangular tile (ti >= tj) of tile columns >= tj_begin, column-major
    int b = blockIdx.x, tj = tj_begin;
    while (b >= nt - tj) { b -= nt - tj; ++tj; }
    const int ti = tj + b;
    const bool diag = (ti == tj);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave & 1, wj = wave >> 1;
    const int l15 = lane & 15, kq = lane >> 4;
    const bool wave_on = !(diag && wi < wj);     // strictly-upper quarter of a diagonal tile

    const double *__restrict__ panI = ab + (long long)(row0 + ti * TS) + (long long)k0 * lda;
    const double *__restrict__ panJ = ab + (long long)(row0 + tj * TS) + (long long)k0 * lda;

    d4_t acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = (d4_t){0.0, 0.0, 0.0, 0.0};

    // staging: 4 x 16-byte loads per thread and operand per chunk; one
    // wave-instruction reads 1 KiB contiguous (one k column, 128 rows)
    const int skk = tid >> 6, srp = tid & 63;
    d2_t rI[4], rJ[4];
    auto gload = [&](int kc) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long off = (long long)(kc + u * 4 + skk) * lda + 2 * srp;
            rI[u] = *reinterpret_cast<const d2_t *>(panI + off);
            rJ[u] = *reinterpret_cast<const d2_t *>(panJ + off);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int o = (u * 4 + skk) * LDT + 2 * srp;
            *reinterpret_cast<d2_t *>(&sI[buf][o]) = rI[u];
            *reinterpret_cast<d2_t *>(&sJ[buf][o]) = rJ[u];
        }
    };

    gload(0);
    lstore(0);
    __syncthreads();
    int buf = 0;
    for (int kc = 0; kc < NBLK; kc += KC) {
        const bool more = (kc + KC < NBLK);
        if (more) gload(kc + KC);
        if (wave_on) {
#pragma unroll
            for (int ks = 0; ks < KC / 4; ++ks) {
                double a[4], bb[4];
                const int krow = (ks * 4 + kq) * LDT + l15;
#pragma unroll
                for (int m = 0; m < 4; ++m) a[m] = sJ[buf][krow + wj * 64 + m * 16];
#pragma unroll
                for (int n = 0; n < 4; ++n) bb[n] = sI[buf][krow + wi * 64 + n * 16];
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], bb[n], acc[m][n], 0, 0, 0);
            }
        }
        if (more) lstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    if (!wave_on) return;
    // D[i][j]: lane holds j = lane&15 (C row), i = (lane>>4) + 4*v (C column).
    // Read-modify-write of the C tile in batches of 16 independent loads (one m slice),
    // the next batch in flight while the current one is stored.  Loads are unconditional
    // (the strictly-upper part of a diagonal tile aliases valid band storage of earlier
    // columns); only the stores are masked.
    double *__restrict__ C = ab + (long long)(row0 + ti * TS) + (long long)(row0 + tj * TS) * lda;
    d4_t cold[2][4];
    auto cload = [&](int m, d4_t (&dst)[4]) {
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int v = 0; v < 4; ++v)
                dst[n][v] = __builtin_nontemporal_load(&C[(wi * 64 + n * 16 + l15) + (long long)(wj * 64 + m * 16 + kq + 4 * v) * lda]);
    };
    auto cstore = [&](int m, const d4_t (&src)[4]) {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int r = wi * 64 + n * 16 + l15;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int c = wj * 64 + m * 16 + kq + 4 * v;
                const double val = src[n][v] - acc[m][n][v];
                if (!diag || r >= c) __builtin_nontemporal_store(val, &C[r + (long long)c * lda]);
            }
        }
    };
    cload(0, cold[0]);
    cload(1, cold[1]);
    cstore(0, cold[0]);
    cload(2, cold[0]);
    cstore(1, cold[1]);
    cload(3, cold[1]);
    cstore(2, cold[0]);
    cstore(3, cold[1]);
}


// ---------------------------------------------------------------------------
// Trailing update, register-streaming form: one wave = one 64x64 piece of C, no LDS, no
// barriers.  The MFMA operands are loaded straight from the panel (L2 / Infinity-Cache
// resident) in fragment shape -- lane (l15, q) reads P[row0 + 16m + l15][k + q], 16 rows
// = one 128-B line per k -- D k-steps ahead into a rotating register queue; every loaded
// operand feeds 4 MFMAs, so the load path carries only 16 B/clk/CU.  v_mfma_f64_16x16x4
// occupies the pipe for 64 cycles, which leaves ample time for 8 loads per 16 MFMAs;
// what the LDS version lost to per-chunk workgroup barriers and to the coupling of the
// two co-resident workgroups is gone.  Two waves per SIMD (<= 256 VGPRs) overlap one
// wave's C read-modify-write with the other's main loop.
// SD = k-steps of look-ahead in the operand queue; WPS = waves per SIMD the register budget
// allows; ABL = ablation switches for tools/syrk_bench (0 = the product kernel)
// queue[0]: next item, queue[1]: waves that stepped aside.  A wave that finds itself on the
// CU reserved for the panel factorisation (`reserved`, ~0u = none) steps aside without taking
// an item -- the grid carries `margin` spare waves for that -- unless the margin is used up.
__device__ unsigned long long *g_syrk_clock_probe = nullptr;   // tools/syrk_bench only (ABL & 128)
template <int SD, int WPS, int ABL, int KTOT = NBLK>
__global__ void __launch_bounds__((ABL & 256) ? 256 : 64, WPS)
syrk64_abl_kernel(double *__restrict__ ab, long long lda, int k0, int row0, int cb, int ce, int rb, int re,
              int nitems, int margin, unsigned reserved, int *__restrict__ queue)
{
    // ABL&256: workgroups of 4 waves take 4 consecutive items (same tile column: the waves stream the
    // same 64 panel rows of the column operand, which the L1 can then serve three times out of four)
    int it = (ABL & 256) ? (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6) : (int)blockIdx.x;
    if ((ABL & 256) && !queue && it >= nitems) return;
    if (queue && (ABL & 256)) {
        __shared__ int s_base;
        if (threadIdx.x == 0) {
            int base = -1;
            bool aside = false;
            if (reserved != ~0u && my_cu_id() == reserved) aside = atomicAdd(&queue[1], 1) < margin;
            if (aside) {
                __builtin_amdgcn_s_sleep(127);       // ~8k cycles: do not drain the grid through this CU
                __builtin_amdgcn_s_sleep(127);
            } else {
                base = atomicAdd(&queue[0], 4);
            }
            s_base = base;
        }
        __syncthreads();
        const int base = s_base;
        if (base < 0) return;
        it = base + (int)(threadIdx.x >> 6);
        if (it >= nitems) return;
    } else if (queue) {
        if (reserved != ~0u && my_cu_id() == reserved) {
            int e = 0;
            if (threadIdx.x == 0) e = atomicAdd(&queue[1], 1);
            e = __builtin_amdgcn_readfirstlane(e);
            if (e < margin) {
                __builtin_amdgcn_s_sleep(127);       // ~8k cycles: do not drain the grid through this CU
                __builtin_amdgcn_s_sleep(127);
                return;
            }
        }
        if (threadIdx.x == 0) it = atomicAdd(&queue[0], 1);
        it = __builtin_amdgcn_readfirstlane(it);
        if (it >= nitems) return;
    }
    unsigned long long pt0 = 0, pr0 = 0;
    if (ABL & 128) { pt0 = __builtin_amdgcn_s_memtime(); pr0 = __builtin_amdgcn_s_memrealtime(); }
    // item -> (tj, ti) in 64-row units: columns [cb, ce), rows [max(tj, rb), re)
    int tj = cb, ti;
    if (ABL & 16) {
        // experiment: 16x16 super-blocks of items dealt to the XCDs (item index = workgroup id,
        // workgroup b runs on XCD b%8): the ~256 waves an XCD has in flight share 16+16 panel
        // row blocks (4 MB = its L2)
        const int x = it & 7, l = it >> 3;
        int s = (l >> 8) * 8 + x;
        const int w = l & 255;
        const int ns = (re - cb + 15) >> 4;
        if (s >= ns * (ns + 1) / 2) return;
        int sj = 0;
        while (s >= ns - sj) { s -= ns - sj; ++sj; }
        ti = cb + 16 * (sj + s) + (w & 15);
        tj = cb + 16 * sj + (w >> 4);
        if (ti >= re || tj >= ce || ti < tj) return;
    } else {
        for (;;) {
            const int lo = tj > rb ? tj : rb;
            const int cnt = re - lo;
            if (it < cnt) { it += lo; break; }
            it -= cnt;
            ++tj;
        }
        ti = it;
    }
    const bool diag = (ti == tj);
    const int lane = threadIdx.x & 63, l15 = lane & 15, q = lane >> 4;

    // per-lane operand streams: element (m, step) at base + 16*m + step*4*lda
    const double *__restrict__ pJ = ab + (long long)(row0 + tj * 64 + l15) + (long long)(k0 + q) * lda;
    const double *__restrict__ pI = ab + (long long)(row0 + ti * 64 + l15) + (long long)(k0 + q) * lda;

    // ABL&4 (product default): the accumulators START as the C tile and the products are
    // subtracted (negated A operand), so the tile is read at the very beginning -- together with
    // the first operands, one exposed latency -- and the epilogue is stores only.
    constexpr bool CINIT = (ABL & 4) != 0;
    // ABL&512 (the small launches of the panel chain, which have a SIMD per wave): accumulators start as -C and
    // collect +P P^T, the epilogue stores their negatives (bit for bit C - P P^T), and the refills are pinned SD
    // k-steps ahead of their use with scheduling barriers.  With the operand negated at load time the compiler
    // waits for every refill right behind its issue (s_waitcnt vmcnt(2) after the loads, then the v_xor), which a
    // second wave on the SIMD hides in the bulk launch (measured there: 0.695 ms this way, 0.730 ms pinned with two
    // waves and SD 4, 0.838 ms un-negated without the barriers because the loads are sunk to their uses) but a
    // lone wave pays as one memory round trip per k-step.
    constexpr bool NEGEND = CINIT && (ABL & 512) != 0;
    double *__restrict__ C = ab + (long long)(row0 + ti * 64) + (long long)(row0 + tj * 64) * lda;
    d4_t acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            if (CINIT) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const double c0v = __builtin_nontemporal_load(&C[(n * 16 + l15) + (long long)(m * 16 + q + 4 * v) * lda]);
                    acc[m][n][v] = NEGEND ? -c0v : c0v;
                }
            } else {
                acc[m][n] = (d4_t){0.0, 0.0, 0.0, 0.0};
            }
        }

    double qa[SD][4], qb[SD][4];
    auto fetch = [&](int slot, int step) {
        const long long off = (long long)(4 * step) * lda;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            qa[slot][m] = (CINIT && !NEGEND) ? -pJ[off + 16 * m] : pJ[off + 16 * m];
            qb[slot][m] = pI[off + 16 * m];
        }
    };
#pragma unroll
    for (int d = 0; d < SD; ++d) fetch(d, d);
    constexpr int NSTEP = KTOT / 4;             // KTOT = panel width applied per pass
    static_assert(NSTEP % SD == 0, "queue depth must divide the k-steps");
    if (ABL & 64) {
        // rolled form: refills are unconditional in the main loop, the last SD steps are peeled
#pragma unroll 1
        for (int ks = 0; ks < NSTEP - SD; ks += SD) {
#pragma unroll
            for (int d = 0; d < SD; ++d) {
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[d][m], qb[d][n], acc[m][n], 0, 0, 0);
                fetch(d, ks + d + SD);
            }
        }
#pragma unroll
        for (int d = 0; d < SD; ++d)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[d][m], qb[d][n], acc[m][n], 0, 0, 0);
    } else if (ABL & 1024) {
        // KTOT = several panels per pass: the unrolled 64-step body of one panel, repeated (code size of K = 256)
        constexpr int NH = KTOT / NBLK, HS = NBLK / 4;
#pragma unroll 1
        for (int h = 0; h < NH; ++h) {
#pragma unroll
            for (int ks = 0; ks < HS; ks += SD) {
#pragma unroll
                for (int d = 0; d < SD; ++d) {
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int n = 0; n < 4; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[d][m], qb[d][n], acc[m][n], 0, 0, 0);
                    if (ks + d + SD < HS) fetch(d, h * HS + ks + d + SD);
                    else {                                   // the first steps of the next panel (clamped: re-reads in the last one)
                        const int nx = h * HS + ks + d + SD;
                        fetch(d, nx < NSTEP ? nx : NSTEP - 1);
                    }
                }
            }
        }
    } else
    for (int ks = 0; ks < NSTEP; ks += SD) {
#pragma unroll
        for (int d = 0; d < SD; ++d) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[d][m], qb[d][n], acc[m][n], 0, 0, 0);
            // NEGEND: the refill of slot d is issued HERE, SD k-steps ahead of its use, and stays here (without the
            // scheduling barriers the compiler sinks the side-effect free loads down to the MFMA that consumes them)
            if (NEGEND) __builtin_amdgcn_sched_barrier(0);
            if ((ABL & 1) == 0 && ks + d + SD < NSTEP) fetch(d, ks + d + SD);   // ABL&1: no operand refills
            if (NEGEND) __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (ABL & 2) {                                 // ABL&2: no C read-modify-write
        double s = 0.0;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
        if (s == 123.456) ab[0] = s;
        return;
    }

    // C(r, c) -= acc: lane holds r = l15 (+16n), c = q + 4v (+16m)
    if (CINIT) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int r = n * 16 + l15;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int c = m * 16 + q + 4 * v;
                    if (!diag || r >= c) __builtin_nontemporal_store(NEGEND ? -acc[m][n][v] : acc[m][n][v], &C[r + (long long)c * lda]);
                }
            }
        if ((ABL & 128) && threadIdx.x == 0 && g_syrk_clock_probe) {     // shader cycles and 100 MHz ticks of this wave
            g_syrk_clock_probe[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - pt0;
            g_syrk_clock_probe[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - pr0;
        }
        return;
    }
    // batches of 16 loads
    d4_t cold[2][4];
    auto cload = [&](int m, d4_t (&dst)[4]) {
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int v = 0; v < 4; ++v)
                dst[n][v] = __builtin_nontemporal_load(&C[(n * 16 + l15) + (long long)(m * 16 + q + 4 * v) * lda]);
    };
    auto cstore = [&](int m, const d4_t (&src)[4]) {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int r = n * 16 + l15;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int c = m * 16 + q + 4 * v;
                const double val = src[n][v] - acc[m][n][v];
                if (!diag || r >= c) __builtin_nontemporal_store(val, &C[r + (long long)c * lda]);
            }
        }
    };
    cload(0, cold[0]);
    cload(1, cold[1]);
    cstore(0, cold[0]);
    cload(2, cold[0]);
    cstore(1, cold[1]);
    cload(3, cold[1]);
    cstore(2, cold[0]);
    cstore(3, cold[1]);
}


}  // namespace
}  // namespace splpak
