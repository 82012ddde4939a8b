// Shared host/device definitions of the gfx950 SPLPAK hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>
#include "options.hpp"

namespace splpak {

constexpr int MAXD = 4;

// Node grid and everything derived from it; passed to kernels by value.
// Mirrors the scratch state the reference keeps in splpak_type
// (dx, dxin: src/splpak.F90:95-100, :747-748).
struct Grid {
    int ndim;
    int nodes[MAXD];
    double xmin[MAXD];
    double dx[MAXD];
    double dxin[MAXD];
    int colstride[MAXD];   // column index = sum ib_d * colstride_d (leftmost fastest, :227-228)
    int cells[MAXD];       // distinct 4-wide windows per dim = nodes-3
    int cellstride[MAXD];
    int ncol;              // product of nodes
    int ncell;             // product of cells
    int nb;                // 4^ndim basis functions per window
    int hstencil;          // (7^ndim+1)/2 stored entries per row of the normal equations
    int halfbw;            // 3*sum colstride_d : half bandwidth of the normal equations
    // A fit plan orders the dimensions by ascending node count (largest slowest): the half bandwidth
    // 3 (1 + n_a + n_a n_b) then involves the two SMALLEST dimensions.  Everything above is in that
    // internal order; the caller's (reference) order only matters where data enters or leaves:
    int perm[MAXD];        // internal dimension d = reference dimension perm[d]
    int refstride[MAXD];   // reference column stride of internal dimension d (coef / histogram address)
    int ref_nodes[MAXD];   // reference-order copies for the nearest-node histogram address (:894-902)
    double ref_xmin[MAXD];
    double ref_dxin[MAXD];
};

// half-stencil slot of the column offset o_d in [-3,3] (dim 0 fastest); valid
// (lower triangle, column <= row) iff the returned code <= centre.
__host__ __device__ inline int stencil_code(const int *o, int ndim)
{
    int e = 0, m = 1;
    for (int d = 0; d < ndim; ++d) { e += (o[d] + 3) * m; m *= 7; }
    return e;
}

void set_error(const std::string &msg);
bool hip_ok(hipError_t e, const char *what);

#define SPLPAK_HIP_TRY(expr, ret)                      \
    do {                                               \
        if (!::splpak::hip_ok((expr), #expr)) return (ret); \
    } while (0)

}  // namespace splpak
