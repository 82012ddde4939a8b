// Standalone timing of trailing-update kernel variants on a C3-shaped panel
// (12544 trailing rows, K = 256), several variants in one process (interleaved rounds).
#include "../splpak_amd/csrc/bandchol.hip"
#include "ablation_kernels.hpp"
#include <cstdio>
#include <hip/hip_ext.h>
#include <vector>
using namespace splpak;
namespace splpak { void set_error(const std::string &) {} bool hip_ok(hipError_t e, const char *) { return e == hipSuccess; } }

static hipStream_t g_stream = nullptr;
__global__ void __launch_bounds__(64) empty_kernel(int *p) { if (p && threadIdx.x == 9999) p[0] = 1; }
__global__ void __launch_bounds__(64) sleep_kernel(int n) { for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127); }
template <int SD, int WPS, int ABL, int KTOT = NBLK>
static float run64(double *ab, long long lda, int n64)
{
    long long items = 0;
    for (int c = 4; c < n64; ++c) items += n64 - c;
    if (ABL & 16) { const int ns = (n64 - 4 + 15) / 16, nsb = ns * (ns + 1) / 2; items = (long long)((nsb + 7) / 8) * 8 * 256; }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, g_stream);
    hipLaunchKernelGGL((syrk64_abl_kernel<SD, WPS, ABL, KTOT>), dim3((unsigned)items), dim3(64), 0, g_stream, ab, lda, 0, KTOT, 4, n64, 0, n64, (int)items, 0, ~0u, (int *)nullptr);
    (void)hipEventRecord(e1, g_stream); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
template <int SD, int ABL>
static float run64x4(double *ab, long long lda, int n64)
{
    long long items = 0;
    for (int c = 4; c < n64; ++c) items += n64 - c;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, g_stream);
    hipLaunchKernelGGL((syrk64_abl_kernel<SD, 1, ABL | 256>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, g_stream, ab, lda, 0, NBLK, 4, n64, 0, n64, (int)items, 0, ~0u, (int *)nullptr);
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
    const size_t ncols = (size_t)4 * NBLK + nrows;
    const size_t elems = (size_t)(lda + 1) * ncols + 4096;
    double *ab; (void)hipMalloc(&ab, elems * sizeof(double));
    std::vector<double> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 1e-3 * (double)((i * 2654435761u) % 1000) - 0.5;
    for (size_t off = 0; off < elems; off += h.size())
        (void)hipMemcpy(ab + off, h.data(), sizeof(double) * std::min(h.size(), elems - off), hipMemcpyHostToDevice);
    double flop64 = 0; for (int c = 4; c < n64; ++c) flop64 += (double)(n64 - c) * 2.0 * 64 * 64 * NBLK;
    double floplds = (double)(nt * (nt + 1) / 2 - (2 * nt - 1)) * 2.0 * 128 * 128 * NBLK;
    const int rounds = 7;
    struct V { const char *name; double flop; std::vector<float> t; } v[] = {
        {"s64 SD16 C-init (product, unrolled)", flop64, {}}, {"s64 SD16 C-init rolled K=256", flop64, {}},
        {"s64 SD16 C-init rolled K=512", 2 * flop64, {}}, {"s64 SD8 C-init rolled K=512", 2 * flop64, {}},
        {"s64 SD16 C-init rolled K=1024", 4 * flop64, {}}, {"s64 SD8 C-init rolled K=256", flop64, {}},
        {"s64 SD16 C-init 4-wave WG", flop64, {}}, {"s64 SD16 C-init unrolled K=512", 2 * flop64, {}},
        {"s64 SD16 C-init K=512 as 2 unrolled halves", 2 * flop64, {}}, {"s64 SD16 C-init K=1024 as 4 unrolled quarters", 4 * flop64, {}}};
    for (int r = 0; r < rounds; ++r) {
        v[0].t.push_back(run64<16, 1, 4>(ab, lda, n64));
        v[1].t.push_back(run64<16, 1, 4 | 64>(ab, lda, n64));
        v[2].t.push_back(run64<16, 1, 4 | 64, 512>(ab, lda, n64));
        v[3].t.push_back(run64<8, 1, 4 | 64, 512>(ab, lda, n64));
        v[4].t.push_back(run64<16, 1, 4 | 64, 1024>(ab, lda, n64));
        v[5].t.push_back(run64<8, 1, 4 | 64>(ab, lda, n64));
        v[6].t.push_back(run64x4<16, 4>(ab, lda, n64));
        v[7].t.push_back(run64<16, 1, 4, 512>(ab, lda, n64));
        v[8].t.push_back(run64<16, 1, 4 | 1024, 512>(ab, lda, n64));
        v[9].t.push_back(run64<16, 1, 4 | 1024, 1024>(ab, lda, n64));
    }
    // steady state on a plain created stream: back-to-back launches (includes the launch-to-launch gap)
    {
        hipStream_t st;
        (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        long long items = 0; for (int c = 4; c < n64; ++c) items += n64 - c;
        for (int variant = 0; variant < 4; ++variant) {
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            const int reps = variant >= 2 ? 10 : 20;
            (void)hipEventRecord(e0, st);
            for (int r = 0; r < reps; ++r) {
                if (variant == 0)
                    hipLaunchKernelGGL((syrk64_abl_kernel<16, 1, 4>), dim3((unsigned)items), dim3(64), 0, st, ab, lda, 0, NBLK, 4, n64, 0, n64, (int)items, 0, ~0u, (int *)nullptr);
                else if (variant == 1)
                    hipLaunchKernelGGL((syrk64_abl_kernel<16, 1, 4 | 64>), dim3((unsigned)items), dim3(64), 0, st, ab, lda, 0, NBLK, 4, n64, 0, n64, (int)items, 0, ~0u, (int *)nullptr);
                else if (variant == 2)
                    hipLaunchKernelGGL((syrk64_abl_kernel<16, 1, 4 | 64, 512>), dim3((unsigned)items), dim3(64), 0, st, ab, lda, 0, 512, 4, n64, 0, n64, (int)items, 0, ~0u, (int *)nullptr);
                else
                    hipLaunchKernelGGL((syrk64_abl_kernel<16, 1, 4 | 1024, 512>), dim3((unsigned)items), dim3(64), 0, st, ab, lda, 0, 512, 4, n64, 0, n64, (int)items, 0, ~0u, (int *)nullptr);
            }
            (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double fl = (variant >= 2 ? 2.0 : 1.0) * flop64;
            printf("back-to-back %-28s: %.3f ms per launch (%.1f TF incl. gaps)\n",
                   variant == 0 ? "product K=256 unrolled" : variant == 1 ? "rolled K=256" : variant == 2 ? "rolled K=512" : "K=512 in 2 unrolled halves", ms / reps, fl / (ms / reps) / 1e9);
        }
    }
    {
        hipStream_t st;
        (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        long long items = 0; for (int c = 4; c < n64; ++c) items += n64 - c;
        for (int variant = 0; variant < 4; ++variant) {
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            const int reps = 20;
            (void)hipEventRecord(e0, st);
            for (int r = 0; r < reps; ++r) {
                if (variant == 0) hipLaunchKernelGGL(empty_kernel, dim3((unsigned)items), dim3(64), 0, st, (int *)nullptr);
                else if (variant == 1) hipLaunchKernelGGL(empty_kernel, dim3(2048), dim3(64), 0, st, (int *)nullptr);
                else if (variant == 2) hipLaunchKernelGGL(sleep_kernel, dim3((unsigned)items), dim3(64), 0, st, 20);   // ~76 us per wave
                else hipLaunchKernelGGL(sleep_kernel, dim3(2048), dim3(64), 0, st, 191);                               // same total sleep, one wave per slot
            }
            (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("back-to-back %-44s: %.1f us per launch\n", variant == 0 ? "empty kernel, 19094 one-wave workgroups" : variant == 1 ? "empty kernel, 2048 workgroups" :
                   variant == 2 ? "sleep 20x127 (~68 us/wave), 19094 workgroups" : "sleep 191x127, 2048 workgroups", 1e3 * ms / reps);
        }
    }
    {   // the same back-to-back sequence with start/stop events carried by each dispatch
        hipStream_t st;
        (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        long long items = 0; for (int c = 4; c < n64; ++c) items += n64 - c;
        const int reps = 12;
        hipEvent_t a[reps], c[reps];
        for (int r = 0; r < reps; ++r) { (void)hipEventCreate(&a[r]); (void)hipEventCreate(&c[r]); }
        for (int r = 0; r < reps; ++r)
            hipExtLaunchKernelGGL((syrk64_abl_kernel<16, 1, 4>), dim3((unsigned)items), dim3(64), 0, st, a[r], c[r], 0, ab, lda, 0, NBLK, 4, n64, 0, n64, (int)items, 0, ~0u, (int *)nullptr);
        (void)hipStreamSynchronize(st);
        for (int r = 0; r < reps; ++r) {
            float d = 0, g = 0;
            (void)hipEventElapsedTime(&d, a[r], c[r]);
            if (r + 1 < reps) (void)hipEventElapsedTime(&g, c[r], a[r + 1]);
            printf("   launch %2d: kernel %.1f us, gap to next %.1f us\n", r, 1e3 * d, 1e3 * g);
        }
    }
    {   // effective shader clock inside the kernel (s_memtime / s_memrealtime per wave), sustained launches
        hipStream_t st;
        (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        long long items = 0; for (int c = 4; c < n64; ++c) items += n64 - c;
        const int reps = 12;
        unsigned long long *buf; (void)hipMalloc(&buf, sizeof(unsigned long long) * 2 * items * reps);
        (void)hipMemset(buf, 0, sizeof(unsigned long long) * 2 * items * reps);
        hipEvent_t a[reps], c[reps];
        for (int r = 0; r < reps; ++r) { (void)hipEventCreate(&a[r]); (void)hipEventCreate(&c[r]); }
        for (int r = 0; r < reps; ++r) {
            unsigned long long *pb = buf + 2 * items * r;
            (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_syrk_clock_probe), &pb, sizeof(pb), 0, hipMemcpyHostToDevice, st);
            hipExtLaunchKernelGGL((syrk64_abl_kernel<16, 1, 4 | 128>), dim3((unsigned)items), dim3(64), 0, st, a[r], c[r], 0, ab, lda, 0, NBLK, 4, n64, 0, n64, (int)items, 0, ~0u, (int *)nullptr);
        }
        (void)hipStreamSynchronize(st);
        std::vector<unsigned long long> h(2 * items * reps);
        (void)hipMemcpy(h.data(), buf, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
        for (int r = 0; r < reps; ++r) {
            std::vector<double> ghz, us;
            for (long long i = 0; i < items; ++i) {
                const double cyc = (double)h[2 * (items * r + i)], rt = (double)h[2 * (items * r + i) + 1];
                if (rt > 0) { ghz.push_back(cyc / rt * 0.1); us.push_back(rt * 0.01); }
            }
            std::sort(ghz.begin(), ghz.end()); std::sort(us.begin(), us.end());
            float d = 0; (void)hipEventElapsedTime(&d, a[r], c[r]);
            printf("   clock probe launch %2d: kernel %.1f us, shader clock median %.3f GHz (p10 %.3f, p90 %.3f), wave time median %.1f us\n",
                   r, 1e3 * d, ghz[ghz.size() / 2], ghz[ghz.size() / 10], ghz[ghz.size() * 9 / 10], us[us.size() / 2]);
        }
    }
    {   // sustained back-to-back: 4-wave workgroups vs single-wave workgroups
        hipStream_t st;
        (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        long long items = 0; for (int c = 4; c < n64; ++c) items += n64 - c;
        for (int variant = 0; variant < 4; ++variant) {
            const int reps = 30;
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0, st);
            for (int r = 0; r < reps; ++r) {
                if (variant == 0)
                    hipLaunchKernelGGL((syrk64_abl_kernel<16, 1, 4>), dim3((unsigned)items), dim3(64), 0, st, ab, lda, 0, NBLK, 4, n64, 0, n64, (int)items, 0, ~0u, (int *)nullptr);
                else if (variant == 1)
                    hipLaunchKernelGGL((syrk64_abl_kernel<16, 1, 4 | 256>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, ab, lda, 0, NBLK, 4, n64, 0, n64, (int)items, 0, ~0u, (int *)nullptr);
                else
                    hipLaunchKernelGGL(syrk_kernel, dim3(nt * (nt + 1) / 2 - (nt + nt - 1)), dim3(256), 0, st, ab, lda, 0, NBLK, nt, 2);
            }
            (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("sustained 30 launches, %s: %.3f ms per launch (%.1f TF)\n", variant == 0 ? "single-wave workgroups" : variant == 1 ? "4-wave workgroups    " : "LDS 128x128 tiles     ", ms / reps, (variant == 2 ? floplds : flop64) / (ms / reps) / 1e9);
        }
    }
    for (auto &x : v) {
        std::sort(x.t.begin(), x.t.end());
        printf("%-28s median %.3f ms  min %.3f ms  -> %.1f TFLOP/s (median)\n", x.name, x.t[rounds / 2], x.t[0], x.flop / x.t[rounds / 2] / 1e9);
    }
    return 0;
}
