// Batched spline / partial-derivative evaluation: one thread per query.
//
// Replaces a loop of scalar splde/splfe calls (src/splpak.F90:1089-1240,
// :1258-1275).  Per query the reference calls bascmp 4^ndim times and recomputes
// every 1-D factor 4^(ndim-1) times; here each thread builds the separable
// 4 x ndim table once (basis.hpp) and walks the 4^ndim window with the same
// summation order as the reference's odometer (dimension 1 fastest, :1228-1232),
// so results differ from the reference only by FMA contraction.
//
// HBM roofline: 8*(ndim+1) algorithmic bytes per query (ndim coordinates in, one
// value out); the coefficient gathers (2 MB at 64^3) are L2 / Infinity-Cache
// resident.
#include "basis.hpp"
#include "kernels.hpp"

namespace splpak {

struct NDeriv { int v[MAXD]; };

template <int D, typename T>
__global__ void __launch_bounds__(256)
eval_kernel(Grid g, long long nq, const T *__restrict__ xq, int ldxq, NDeriv nd,
            const T *__restrict__ coef, T *__restrict__ out)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += stride) {
        double b[D][4];
        int base = 0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const double x = (double)xq[i * ldxq + d];
            const int ws = window_table(g, d, x, nd.v[d], b[d]);
            base += ws * g.colstride[d];
        }
        // The 4 coefficients of a window row (k0 = 0..3) are contiguous: they are fetched with two
        // 16-byte loads instead of four 8-byte ones (the kernel is bound by gather instructions,
        // ~64 cycles of address processing per wave-load whatever its width), and summed in the
        // reference's order (dimension 1 fastest).
        double sum = 0.0;
        auto row4 = [&](long long idx, double scale) {
            double c0, c1, c2, c3;
            if constexpr (sizeof(T) == 8) {
                typedef double d2v __attribute__((ext_vector_type(2), aligned(8)));
                const d2v lo = *reinterpret_cast<const d2v *>(coef + idx);
                const d2v hi = *reinterpret_cast<const d2v *>(coef + idx + 2);
                c0 = lo[0]; c1 = lo[1]; c2 = hi[0]; c3 = hi[1];
            } else {
                typedef float f4v __attribute__((ext_vector_type(4), aligned(4)));
                const f4v v = *reinterpret_cast<const f4v *>(coef + idx);
                c0 = v[0]; c1 = v[1]; c2 = v[2]; c3 = v[3];
            }
            sum += c0 * (b[0][0] * scale);
            sum += c1 * (b[0][1] * scale);
            sum += c2 * (b[0][2] * scale);
            sum += c3 * (b[0][3] * scale);
        };
        if constexpr (D == 1) {
#pragma unroll
            for (int k0 = 0; k0 < 4; ++k0) sum += (double)coef[base + k0] * b[0][k0];
        } else if constexpr (D == 2) {
            const int s1 = g.colstride[1];
#pragma unroll
            for (int k1 = 0; k1 < 4; ++k1) row4(base + k1 * s1, b[1][k1]);
        } else if constexpr (D == 3) {
            const int s1 = g.colstride[1], s2 = g.colstride[2];
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2)
#pragma unroll
                for (int k1 = 0; k1 < 4; ++k1) row4(base + k1 * s1 + k2 * s2, b[1][k1] * b[2][k2]);
        } else {
            const int s1 = g.colstride[1], s2 = g.colstride[2], s3 = g.colstride[3];
            for (int k3 = 0; k3 < 4; ++k3)
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2)
#pragma unroll
                    for (int k1 = 0; k1 < 4; ++k1)
                        row4(base + k1 * s1 + k2 * s2 + k3 * s3, (b[1][k1] * b[2][k2]) * b[3][k3]);
        }
        out[i] = (T)sum;
    }
}

template <typename T>
static hipError_t launch_eval_t(const Grid &g, long long nq, const T *xq, int ldxq,
                                const int *nderiv, const T *coef, T *out, hipStream_t st)
{
    if (nq <= 0) return hipSuccess;
    NDeriv nd;
    for (int d = 0; d < MAXD; ++d) {
        int v = (nderiv && d < g.ndim) ? nderiv[d] : 0;
        nd.v[d] = v < 0 ? 0 : (v > 2 ? 2 : v);
    }
    const int threads = 256;
    long long blocks = (nq + threads - 1) / threads;
    if (blocks > 256LL * 32) blocks = 256LL * 32;   // grid-stride the rest
    dim3 gr((unsigned)blocks), bl(threads);
    switch (g.ndim) {
    case 1: hipLaunchKernelGGL((eval_kernel<1, T>), gr, bl, 0, st, g, nq, xq, ldxq, nd, coef, out); break;
    case 2: hipLaunchKernelGGL((eval_kernel<2, T>), gr, bl, 0, st, g, nq, xq, ldxq, nd, coef, out); break;
    case 3: hipLaunchKernelGGL((eval_kernel<3, T>), gr, bl, 0, st, g, nq, xq, ldxq, nd, coef, out); break;
    default: hipLaunchKernelGGL((eval_kernel<4, T>), gr, bl, 0, st, g, nq, xq, ldxq, nd, coef, out); break;
    }
    return hipGetLastError();
}

hipError_t launch_eval(const Grid &g, long long nq, const double *xq, int ldxq, const int *nderiv,
                       const double *coef, double *out, hipStream_t st)
{
    return launch_eval_t<double>(g, nq, xq, ldxq, nderiv, coef, out, st);
}

hipError_t launch_eval_f32(const Grid &g, long long nq, const float *xq, int ldxq, const int *nderiv,
                           const float *coef, float *out, hipStream_t st)
{
    return launch_eval_t<float>(g, nq, xq, ldxq, nderiv, coef, out, st);
}

}  // namespace splpak
