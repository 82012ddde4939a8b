// Sustained v_mfma_f64_16x16x4_f64 rate with operands that CHANGE every instruction (random
// per-lane values cycling through 16+16 registers), several tens of ms per run: reports TFLOP/s and
// the effective shader clock (s_memtime / s_memrealtime), i.e. what the socket power limit leaves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256, 2) probe(double *out, unsigned long long *clk, int iters, const double *seed, int mode)
{
    d4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a[16], b[16];
    for (int i = 0; i < 16; ++i) {
        a[i] = mode == 0 ? 1.0 : seed[(threadIdx.x * 16 + i) & 4095];
        b[i] = mode == 0 ? 1.0 : seed[(threadIdx.x * 16 + i + 2048) & 4095];
    }
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[(i + (mode == 2 ? it : 0)) & 15], acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
int main()
{
    const int nb = 256 * 2, iters = 120000;     // ~50 ms per launch
    double *out, *seed; unsigned long long *clk;
    (void)hipMalloc(&out, sizeof(double) * nb * 256); (void)hipMalloc(&clk, 16 * nb); (void)hipMalloc(&seed, 8 * 4096);
    std::vector<double> hs(4096);
    unsigned long long x = 88172645463325252ull;
    for (auto &v : hs) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (double)(x >> 11) / 9007199254740992.0 - 0.5; }
    (void)hipMemcpy(seed, hs.data(), 8 * 4096, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 0, 0, out, clk, iters, (const double *)seed, mode);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(2 * nb);
            (void)hipMemcpy(h.data(), clk, 16 * nb, hipMemcpyDeviceToHost);
            std::vector<double> ghz;
            for (int i = 0; i < nb; ++i) ghz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1);
            std::sort(ghz.begin(), ghz.end());
            const double flop = 2.0 * 16 * 16 * 4 * 16.0 * iters * 4.0 * nb;
            printf("%-28s rep %d: %.1f ms  %6.2f TFLOP/s  shader clock median %.3f GHz\n",
                   mode == 0 ? "constant operands (1.0)" : "random per-lane operands", rep, ms, flop / ms / 1e9, ghz[nb / 2]);
        }
    return 0;
}
