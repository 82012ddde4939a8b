// Peak-rate probe for v_mfma_f64_16x16x4_f64 on gfx950: back-to-back MFMAs, independent
// accumulators, operands in registers; reports TFLOP/s, shader clock and cycles per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(256, 2) probe(double *out, unsigned long long *clk, int iters, double a0, double b0, double eps)
{
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = a0 + threadIdx.x * eps, b = b0 - threadIdx.x * eps;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int NACC>
static void run(int blocks_per_cu, int iters, double a0, double eps, const char *tag)
{
    int ncu = 256, nb = ncu * blocks_per_cu;
    double *out; (void)hipMalloc(&out, sizeof(double) * nb * 256);
    unsigned long long *clk; (void)hipMalloc(&clk, 16 * nb);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(probe<NACC>, dim3(nb), dim3(256), 0, 0, out, clk, iters, a0, a0, eps);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(probe<NACC>, dim3(nb), dim3(256), 0, 0, out, clk, iters, a0, a0, eps);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * nb);
    (void)hipMemcpy(h.data(), clk, 16 * nb, hipMemcpyDeviceToHost);
    std::vector<double> ghz, cyc;
    for (int i = 0; i < nb; ++i) { ghz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1); cyc.push_back((double)h[2 * i] / ((double)NACC * iters)); }
    std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
    double flop = 2.0 * 16 * 16 * 4 * (double)NACC * iters * 4.0 * nb;
    printf("%-8s NACC=%2d waves/SIMD=%d: %.3f ms  %6.2f TFLOP/s  clock(median)=%.2f GHz  cycles/MFMA/wave(median)=%.1f\n", tag, NACC,
           blocks_per_cu, ms, flop / ms / 1e9, ghz[nb / 2], cyc[nb / 2]);
    (void)hipFree(out); (void)hipFree(clk);
}
int main()
{
    run<16>(1, 20000, 1.0, 1e-9, "random"); run<16>(2, 20000, 1.0, 1e-9, "random");
    run<16>(1, 20000, 0.0, 0.0, "zeros");  run<16>(2, 20000, 0.0, 0.0, "zeros");
    run<16>(1, 20000, 1.2345678901234567, 3.3e-7, "dense");
    return 0;
}
