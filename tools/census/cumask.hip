// Does hipExtStreamCreateWithCUMask restrict placement on MI355X?  Census per mask.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void census(unsigned *out)
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 20000) {}
}
static void run(const char *name, hipStream_t s)
{
    const int nb = 8192;
    unsigned *d; (void)hipMalloc(&d, nb * 8);
    hipLaunchKernelGGL(census, dim3(nb), dim3(256), 0, s, d);
    (void)hipStreamSynchronize(s);
    std::vector<unsigned> h(2 * nb);
    (void)hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
    std::set<unsigned> cus;
    for (int i = 0; i < nb; ++i) cus.insert(((h[2 * i + 1] & 0xf) << 16) | (h[2 * i] & 0xff00));
    printf("%s: distinct CUs used = %zu; missing:", name, cus.size());
    // list which (xcc,se,cu) are missing relative to the unmasked census is left to the caller
    printf("\n");
    for (auto v : cus) if (0) printf("%x ", v);
    (void)hipFree(d);
}
int main()
{
    hipStream_t s0; (void)hipStreamCreate(&s0);
    run("unmasked", s0);
    for (int drop = 1; drop <= 3; ++drop) {
        std::vector<uint32_t> mask(8, 0xffffffffu);       // 256 bits
        if (drop == 1) mask[0] = 0xfffffffeu;             // clear bit 0
        if (drop == 2) mask[0] = 0xffffff00u;             // clear bits 0..7
        if (drop == 3) { mask[0] = 0; }                   // clear bits 0..31
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask.data());
        printf("mask variant %d create: %s\n", drop, hipGetErrorString(e));
        if (e == hipSuccess) { char nm[32]; snprintf(nm, 32, "variant %d", drop); run(nm, s); }
    }
    return 0;
}
