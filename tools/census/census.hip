// Census of HW_REG_HW_ID / HW_REG_XCC_ID values seen by workgroups (gfx950 field layout probe).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>
__global__ void census(unsigned *out)
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
    // keep the block alive a little so that blocks spread
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 20000) {}
}
int main()
{
    const int nb = 8192;
    unsigned *d; hipMalloc(&d, nb * 8);
    hipLaunchKernelGGL(census, dim3(nb), dim3(256), 0, 0, d);
    std::vector<unsigned> h(2 * nb);
    hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
    std::set<unsigned> xs; std::map<unsigned, int> bits;
    unsigned orv = 0, andv = ~0u;
    for (int i = 0; i < nb; ++i) { orv |= h[2 * i]; andv &= h[2 * i]; xs.insert(h[2 * i + 1]); }
    printf("HW_ID or=%08x and=%08x\n", orv, andv);
    printf("XCC_ID values:"); for (auto v : xs) printf(" %08x", v); printf("\n");
    // candidate fields: cu_id [11:8], sh_id [12], se_id [15:13]
    std::set<unsigned> cus;
    for (int i = 0; i < nb; ++i) cus.insert(((h[2 * i + 1] & 0xf) << 16) | (h[2 * i] & 0xff00));
    printf("distinct (xcc, hw_id[15:8]) = %zu\n", cus.size());
    std::map<unsigned,int> cnt;
    for (int i = 0; i < nb; ++i) cnt[(h[2*i] >> 8) & 0xff]++;
    for (auto &kv : cnt) printf("  hw[15:8]=%02x n=%d\n", kv.first, kv.second);
    return 0;
}
