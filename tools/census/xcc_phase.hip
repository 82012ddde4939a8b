// Do the workgroups of ONE kernel run on all eight XCDs at once?  tools/census/xcc_phase.bin
// (round 5: sp_scatter_kernel -- 144 KB of LDS, 128 VGPRs, 1024 threads = one workgroup per CU -- ran its 1 221 workgroups
//  in two phases, the even XCDs first and the odd ones 450 us later.)  Every workgroup spins for `ticks` of the 100 MHz
// clock and records start, end and XCC_ID; the host prints per-XCD first start / last end.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int LDS_BYTES, int NT, bool REGS>
__global__ void __launch_bounds__(NT) k_busy(unsigned long long *st, int ticks, const int *in)
{
    __shared__ int buf[LDS_BYTES / 4];
    if (REGS) asm volatile("" ::: "v127");
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    buf[threadIdx.x] = in[threadIdx.x];
    int acc = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) acc += in[(acc + threadIdx.x) & 1023];
    __syncthreads();
    if (threadIdx.x == 0) {
        st[blockIdx.x * 4 + 0] = t0;
        st[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
        st[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg(63508) & 0xf;
        st[blockIdx.x * 4 + 3] = (unsigned long long)(acc + buf[0]);
    }
}
template <int LDS_BYTES, int NT, bool REGS>
static void run(const char *name, int grid, int ticks, unsigned long long *st, const int *in)
{
    std::vector<unsigned long long> h((size_t)4 * grid);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k_busy<LDS_BYTES, NT, REGS>), dim3(grid), dim3(NT), 0, 0, st, ticks, in);
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int q = 0; q < grid; ++q) t0 = std::min(t0, h[4 * q]), t1 = std::max(t1, h[4 * q + 1]);
    printf("%-44s grid %5d x %3d us: span %7.1f us;  per XCD first start / last end:", name, grid, ticks / 100, (t1 - t0) / 100.0);
    for (int x = 0; x < 8; ++x) {
        unsigned long long a = ~0ull, b = 0;
        for (int q = 0; q < grid; ++q)
            if ((int)h[4 * q + 2] == x) a = std::min(a, h[4 * q]), b = std::max(b, h[4 * q + 1]);
        printf(" %d:%.0f/%.0f", x, (a - t0) / 100.0, (b - t0) / 100.0);
    }
    printf("\n");
}
int main()
{
    int *in;
    unsigned long long *st;
    hipMalloc(&in, 4096); hipMemset(in, 0, 4096); hipMalloc(&st, 8 * 4 * 8192);
    for (int grid : {256, 1221}) {
        run<1024, 1024, false>("LDS   4 KB, 1024 threads", grid, 5000, st, in);
        run<65536, 1024, false>("LDS  64 KB, 1024 threads", grid, 5000, st, in);
        run<81920, 1024, false>("LDS  80 KB, 1024 threads", grid, 5000, st, in);
        run<147456, 1024, false>("LDS 144 KB, 1024 threads", grid, 5000, st, in);
        run<147456, 1024, true>("LDS 144 KB, 1024 threads, 128 VGPRs", grid, 5000, st, in);
        run<16384, 1024, true>("LDS  16 KB, 1024 threads, 128 VGPRs", grid, 5000, st, in);
        run<147456, 256, false>("LDS 144 KB,  256 threads", grid, 5000, st, in);
    }
    return 0;
}
