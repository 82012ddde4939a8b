// Standalone timing of trailing-update kernel variants on a C3-shaped panel
// (12544 trailing rows, K = 256), several variants in one process (interleaved rounds).
#include "../splpak_amd/csrc/bandchol.hip"
#include <cstdio>
#include <vector>
using namespace splpak;
namespace splpak { void set_error(const std::string &) {} bool hip_ok(hipError_t e, const char *) { return e == hipSuccess; } }

static hipStream_t g_stream = nullptr;
template <int SD, int WPS, int ABL, int KTOT = NBLK>
static float run64(double *ab, long long lda, int n64)
{
    long long items = 0;
    for (int c = 4; c < n64; ++c) items += n64 - c;
    if (ABL & 16) { const int ns = (n64 - 4 + 15) / 16, nsb = ns * (ns + 1) / 2; items = (long long)((nsb + 7) / 8) * 8 * 256; }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, g_stream);
    hipLaunchKernelGGL((syrk64_kernel<SD, WPS, ABL, KTOT>), dim3((unsigned)items), dim3(64), 0, g_stream, ab, lda, 0, KTOT, 4, n64, 0, n64, (int)items, 0, ~0u, (int *)nullptr);
    (void)hipEventRecord(e1, g_stream); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
static float runlds(double *ab, long long lda, int nt)
{
    const int ntiles = nt * (nt + 1) / 2 - (nt + nt - 1);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, g_stream);
    hipLaunchKernelGGL(syrk_kernel, dim3(ntiles), dim3(256), 0, g_stream, ab, lda, 0, NBLK, nt, 2);
    (void)hipEventRecord(e1, g_stream); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main()
{
    const int tb = 49, nrows = tb * NBLK, nt = nrows / 128, n64 = nrows / 64;
    const long long lda = (long long)(tb + 1) * NBLK + 16;
    const size_t ncols = (size_t)2 * NBLK + nrows;
    const size_t elems = (size_t)(lda + 1) * ncols + 4096;
    double *ab; (void)hipMalloc(&ab, elems * sizeof(double));
    std::vector<double> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 1e-3 * (double)((i * 2654435761u) % 1000) - 0.5;
    for (size_t off = 0; off < elems; off += h.size())
        (void)hipMemcpy(ab + off, h.data(), sizeof(double) * std::min(h.size(), elems - off), hipMemcpyHostToDevice);
    double flop64 = 0; for (int c = 4; c < n64; ++c) flop64 += (double)(n64 - c) * 2.0 * 64 * 64 * NBLK;
    double floplds = (double)(nt * (nt + 1) / 2 - (2 * nt - 1)) * 2.0 * 128 * 128 * NBLK;
    const int rounds = 5;
    struct V { const char *name; double flop; std::vector<float> t; } v[] = {
        {"lds 128x128 (2 WG/CU)", floplds, {}}, {"s64 SD4 2w/SIMD", flop64, {}}, {"s64 SD8 1w/SIMD", flop64, {}},
        {"s64 SD4 2w no-refill", flop64, {}}, {"s64 SD4 2w no-epilogue", flop64, {}}, {"s64 SD4 2w neither", flop64, {}},
        {"s64 SD2 2w", flop64, {}}, {"s64 SD8 1w no-epilogue", flop64, {}}, {"s64 SD16 1w", flop64, {}},
        {"s64 SD16 1w no-epilogue", flop64, {}}, {"s64 SD16 1w no-refill", flop64, {}}, {"s64 SD8 1w", flop64, {}}, {"s64 SD8 2w", flop64, {}}, {"s64 SD16 2w", flop64, {}}, {"s64 SD16 2w no-epilogue", flop64, {}}, {"s64 SD16 1w C-init", flop64, {}}, {"s64 SD16 2w C-init", flop64, {}}, {"s64 SD4 2w C-init", flop64, {}}, {"s64 SD8 1w C-init", flop64, {}}, {"s64 SD16 1w C-init K=512", 2 * flop64, {}}, {"s64 SD16 1w K=512", 2 * flop64, {}}, {"s64 SD16 1w xcd-blocked", flop64, {}}, {"s64 SD16 1w K=512 xcd-blocked", 2 * flop64, {}}};
    for (int r = 0; r < rounds; ++r) {
        v[0].t.push_back(runlds(ab, lda, nt));
        v[1].t.push_back(run64<4, 2, 0>(ab, lda, n64));
        v[2].t.push_back(run64<8, 1, 0>(ab, lda, n64));
        v[3].t.push_back(run64<4, 2, 1>(ab, lda, n64));
        v[4].t.push_back(run64<4, 2, 2>(ab, lda, n64));
        v[5].t.push_back(run64<4, 2, 3>(ab, lda, n64));
        v[6].t.push_back(run64<2, 2, 0>(ab, lda, n64));
        v[7].t.push_back(run64<8, 1, 2>(ab, lda, n64));
        v[8].t.push_back(run64<16, 1, 0>(ab, lda, n64));
        v[9].t.push_back(run64<16, 1, 2>(ab, lda, n64));
        v[10].t.push_back(run64<16, 1, 1>(ab, lda, n64));
        v[11].t.push_back(run64<8, 1, 0>(ab, lda, n64));
        v[12].t.push_back(run64<8, 2, 0>(ab, lda, n64));
        v[13].t.push_back(run64<16, 2, 0>(ab, lda, n64));
        v[14].t.push_back(run64<16, 2, 2>(ab, lda, n64));
        v[15].t.push_back(run64<16, 1, 4>(ab, lda, n64));
        v[16].t.push_back(run64<16, 2, 4>(ab, lda, n64));
        v[17].t.push_back(run64<4, 2, 4>(ab, lda, n64));
        v[18].t.push_back(run64<8, 1, 4>(ab, lda, n64));
        v[19].t.push_back(run64<16, 1, 4, 512>(ab, lda, n64));
        v[20].t.push_back(run64<16, 1, 0, 512>(ab, lda, n64));
        v[21].t.push_back(run64<16, 1, 16>(ab, lda, n64));
        v[22].t.push_back(run64<16, 1, 16, 512>(ab, lda, n64));
    }
    // the same two kernels on a CU-masked stream (one CU left out) and on a plain created stream
    for (int variant = 0; variant < 2; ++variant) {
        hipStream_t st;
        if (variant == 0) { uint32_t mask[8]; for (auto &m : mask) m = 0xffffffffu; mask[0] &= ~1u; (void)hipExtStreamCreateWithCUMask(&st, 8, mask); }
        else (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        g_stream = st;
        std::vector<float> a, b;
        for (int r = 0; r < rounds; ++r) { a.push_back(runlds(ab, lda, nt)); b.push_back(run64<16, 1, 0>(ab, lda, n64)); }
        std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
        printf("%s stream: lds %.3f ms (%.1f TF)   s64 SD16 %.3f ms (%.1f TF)\n", variant == 0 ? "CU-masked (255 CUs)" : "plain non-blocking",
               a[rounds / 2], floplds / a[rounds / 2] / 1e9, b[rounds / 2], flop64 / b[rounds / 2] / 1e9);
        // steady state: 10 launches back to back on that stream
        {
            long long items = 0; for (int c = 4; c < n64; ++c) items += n64 - c;
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0, st);
            for (int r = 0; r < 10; ++r)
                hipLaunchKernelGGL((syrk64_kernel<16, 1, 0>), dim3((unsigned)items), dim3(64), 0, st, ab, lda, 0, NBLK, 4, n64, 0, n64, (int)items, 0, ~0u, (int *)nullptr);
            (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("   10 back-to-back s64 SD16 launches: %.3f ms each (%.1f TF)\n", ms / 10, flop64 / (ms / 10) / 1e9);
        }
        g_stream = nullptr;
    }
    {
        long long items = 0; for (int c = 4; c < n64; ++c) items += n64 - c;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, 0);
        for (int r = 0; r < 10; ++r)
            hipLaunchKernelGGL((syrk64_kernel<16, 1, 0>), dim3((unsigned)items), dim3(64), 0, 0, ab, lda, 0, NBLK, 4, n64, 0, n64, (int)items, 0, ~0u, (int *)nullptr);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("null stream, 10 back-to-back s64 SD16 launches: %.3f ms each (%.1f TF)\n", ms / 10, flop64 / (ms / 10) / 1e9);
    }
    for (auto &x : v) {
        std::sort(x.t.begin(), x.t.end());
        printf("%-28s median %.3f ms  min %.3f ms  -> %.1f TFLOP/s (median)\n", x.name, x.t[rounds / 2], x.t[0], x.flop / x.t[rounds / 2] / 1e9);
    }
    return 0;
}
