// Phase timing of potrf_block_kernel (cycle stamps of wave 0): which part of a 16-column panel costs what.
#define SPLPAK_POTRF_STAMPS 1
#include "../splpak_amd/csrc/bandchol.hip"
#include "ablation_kernels.hpp"
#include <cstdio>
#include <vector>
using namespace splpak;
namespace splpak { void set_error(const std::string &) {} bool hip_ok(hipError_t e, const char *) { return e == hipSuccess; } }
int main()
{
    const long long lda = 2 * NBLK + 16;
    std::vector<double> h((size_t)lda * NBLK + NBLK, 0.0);
    unsigned long long x = 88172645463325252ull;
    for (int c = 0; c < NBLK; ++c)
        for (int r = c; r < NBLK; ++r) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            h[r + (size_t)c * lda] = r == c ? 300.0 : (double)(x >> 11) / 9007199254740992.0 - 0.5;
        }
    double *ab, *inv, *minp; int *info;
    (void)hipMalloc(&ab, h.size() * 8); (void)hipMalloc(&inv, 8 * 4 * 64 * 64); (void)hipMalloc(&minp, 8); (void)hipMalloc(&info, 4);
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipMemcpy(ab, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        (void)hipMemset(info, 0, 4);
        double big = 1e300; (void)hipMemcpy(minp, &big, 8, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(potrf_block_kernel, dim3(1), dim3(256), 0, 0, ab, lda, 0, info, minp, inv);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long st[128];
        (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_potrf_stamps), sizeof(st));
        int hinfo; (void)hipMemcpy(&hinfo, info, 4, hipMemcpyDeviceToHost);
        printf("rep %d: %.1f us, info %d\n", rep, 1e3 * ms, hinfo);
        if (rep == 3) {
            const char *names[7] = {"issue loads", "leaf factor", "leaf inverse", "sync", "row solve + stores", "sync", "trailing update"};
            for (int p : {0, 4, 8, 12, 15}) {
                printf("  panel %2d:", p);
                for (int i = 0; i < 7; ++i) printf("  %s %.2f us;", names[i], (double)(st[p * 8 + i + 1] - st[p * 8 + i]) * 0.01);
                if (p < 15) printf("  to next panel %.2f us", (double)(st[(p + 1) * 8] - st[p * 8 + 7]) * 0.01);
                printf("\n");
            }
        }
    }
    // strip form: cycles of wave 0 per phase, summed over the kernel (s_memtime ticks of 10 ns)
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipMemcpy(ab, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        (void)hipMemset(info, 0, 4);
        double big = 1e300; (void)hipMemcpy(minp, &big, 8, hipMemcpyHostToDevice);
        unsigned long long zero[8] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_strip_cycles), zero, sizeof(zero));
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(potrf_strip_kernel, dim3(1), dim3(256), 0, 0, ab, lda, 0, info, minp, inv);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long st[8];
        (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_strip_cycles), sizeof(st));
        int hinfo; (void)hipMemcpy(&hinfo, info, 4, hipMemcpyDeviceToHost);
        printf("strip rep %d: %.1f us, info %d:", rep, 1e3 * ms, hinfo);
        const char *nm[6] = {"strip load", "leaf", "row solve", "in-strip update", "strip store", "trailing K=64"};
        for (int i = 0; i < 6; ++i) printf("  %s %.1f us;", nm[i], (double)st[i] * 0.01);
        printf("\n");
    }
    return 0;
}
