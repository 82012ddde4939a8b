// Device-side generator of the synthetic inputs of SURVEY.md section 8d
// (Park-Miller LCG, seed 42; same stream as splpak_amd/synth.py).  Every thread
// jumps to its point with s_k = s_0 * 48271^k mod (2^31-1), so any shard of the
// stream can be produced independently (multi-GPU ranks generate their own slice).
#include "kernels.hpp"

namespace splpak {

namespace {
constexpr unsigned long long LCG_A = 48271ULL;
constexpr unsigned long long LCG_M = 2147483647ULL;
constexpr unsigned long long LCG_SEED = 42ULL;

__device__ inline unsigned long long lcg_jump(unsigned long long k)
{   // state after k steps from the seed
    unsigned long long r = 1, a = LCG_A;
    while (k) {
        if (k & 1ULL) r = (r * a) % LCG_M;
        a = (a * a) % LCG_M;
        k >>= 1;
    }
    return (LCG_SEED * r) % LCG_M;
}

template <int D>
__global__ void __launch_bounds__(256)
synth_points_kernel(long long first, long long n, double *__restrict__ x, double *__restrict__ y,
                    double *__restrict__ w)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        unsigned long long s = lcg_jump((unsigned long long)(first + i) * (D + 2));
        double xv[D];
        double ysum = 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            s = (s * LCG_A) % LCG_M;
            xv[d] = (double)s / (double)LCG_M;
            ysum += sin(3.0 * xv[d] + (double)(d + 1));
        }
        s = (s * LCG_A) % LCG_M;
        ysum += 0.01 * ((double)s / (double)LCG_M - 0.5);
        s = (s * LCG_A) % LCG_M;
        const double wv = 0.5 + (double)s / (double)LCG_M;
        if (x) {
#pragma unroll
            for (int d = 0; d < D; ++d) x[i * D + d] = xv[d];
        }
        if (y) y[i] = ysum;
        if (w) w[i] = wv;
    }
}

__global__ void __launch_bounds__(256)
synth_uniform_kernel(long long skip, long long n, double *__restrict__ out)
{
    // 8 consecutive draws per thread
    const long long nchunk = (n + 7) / 8;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < nchunk; c += stride) {
        unsigned long long s = lcg_jump((unsigned long long)(skip + c * 8));
        for (int j = 0; j < 8; ++j) {
            const long long i = c * 8 + j;
            if (i >= n) break;
            s = (s * LCG_A) % LCG_M;
            out[i] = (double)s / (double)LCG_M;
        }
    }
}
}  // namespace

hipError_t launch_synth_points(int ndim, long long first, long long n, double *x, double *y,
                               double *w, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    long long blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    dim3 gr((unsigned)blocks), bl(256);
    switch (ndim) {
    case 1: hipLaunchKernelGGL(synth_points_kernel<1>, gr, bl, 0, st, first, n, x, y, w); break;
    case 2: hipLaunchKernelGGL(synth_points_kernel<2>, gr, bl, 0, st, first, n, x, y, w); break;
    case 3: hipLaunchKernelGGL(synth_points_kernel<3>, gr, bl, 0, st, first, n, x, y, w); break;
    default: hipLaunchKernelGGL(synth_points_kernel<4>, gr, bl, 0, st, first, n, x, y, w); break;
    }
    return hipGetLastError();
}

hipError_t launch_synth_queries(int ndim, long long skip_draws, long long nq, double *xq,
                                hipStream_t st)
{
    const long long n = nq * ndim;
    if (n <= 0) return hipSuccess;
    long long blocks = ((n + 7) / 8 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(synth_uniform_kernel, dim3((unsigned)blocks), dim3(256), 0, st, skip_draws, n, xq);
    return hipGetLastError();
}

}  // namespace splpak
