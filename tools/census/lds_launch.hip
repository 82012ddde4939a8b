// How long do (nearly) empty workgroups take as a function of their static LDS size?  tools/census/lds_launch.bin
// (round 5: sp_scatter_kernel, 147 KB of LDS, took 0.8 ms for 1 221 workgroups that returned after a few loads)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS_BYTES, int NT>
__global__ void __launch_bounds__(NT) k_empty(const int *in, int *out)
{
    __shared__ int buf[LDS_BYTES / 4];
    if (threadIdx.x == 0) buf[0] = in[blockIdx.x & 1023];
    __syncthreads();
    if (buf[0] == 12345 && threadIdx.x == 1) out[blockIdx.x] = buf[threadIdx.x];
}
// the same with ~120 live vector registers per thread (a full CU per 1024-thread workgroup)
template <int LDS_BYTES, int NT>
__global__ void __launch_bounds__(NT) k_regs(const int *in, int *out, const double *src)
{
    __shared__ int buf[LDS_BYTES / 4];
    double v[56];
#pragma unroll
    for (int i = 0; i < 56; ++i) v[i] = src[(blockIdx.x & 7) * 64 + i];
    if (threadIdx.x == 0) buf[0] = in[blockIdx.x & 1023];
    __syncthreads();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 56; ++i) s += v[i] * (double)(i + buf[0]);
    if (s == 12345.0) out[blockIdx.x] = (int)s;
}
template <int LDS_BYTES, int NT>
static void run_regs(const char *name, int grid, const int *in, int *out, const double *src)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_regs<LDS_BYTES, NT>), dim3(grid), dim3(NT), 0, 0, in, out, src);
    hipEventRecord(a, 0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k_regs<LDS_BYTES, NT>), dim3(grid), dim3(NT), 0, 0, in, out, src);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    printf("%-28s grid %5d: %8.1f us per launch (many registers)\n", name, grid, 100.0 * ms);
}
template <int LDS_BYTES, int NT>
static void run(const char *name, int grid, const int *in, int *out)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_empty<LDS_BYTES, NT>), dim3(grid), dim3(NT), 0, 0, in, out);
    hipEventRecord(a, 0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k_empty<LDS_BYTES, NT>), dim3(grid), dim3(NT), 0, 0, in, out);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    printf("%-28s grid %5d: %8.1f us per launch\n", name, grid, 100.0 * ms);
}
int main()
{
    int *in, *out;
    hipMalloc(&in, 4096); hipMemset(in, 0, 4096); hipMalloc(&out, 1 << 20);
    double *src; hipMalloc(&src, 8 * 1024); hipMemset(src, 0, 8 * 1024);
    for (int grid : {256, 1221, 4884}) {
        run<1024, 1024>("LDS   1 KB, 1024 threads", grid, in, out);
        run<16384, 1024>("LDS  16 KB, 1024 threads", grid, in, out);
        run<65536, 1024>("LDS  64 KB, 1024 threads", grid, in, out);
        run<81920, 1024>("LDS  80 KB, 1024 threads", grid, in, out);
        run<147456, 1024>("LDS 144 KB, 1024 threads", grid, in, out);
        run<147456, 256>("LDS 144 KB,  256 threads", grid, in, out);
        run<16384, 256>("LDS  16 KB,  256 threads", grid, in, out);
        run_regs<16384, 1024>("LDS  16 KB, 1024 threads", grid, in, out, src);
        run_regs<147456, 1024>("LDS 144 KB, 1024 threads", grid, in, out, src);
    }
    return 0;
}
