// 1-D natural-spline basis functions and the per-point 4-wide window table.
//
// What the reference computes one (point, basis tuple) at a time in bascmp
// (src/splpak.F90:206-389) is restructured here as a separable table: for every
// dimension the (at most) four 1-D factors that can be non-zero at x are
// evaluated once, and the 4^ndim tensor products are formed by the caller.
// Closed forms follow SURVEY.md appendix A (derived from :231-381); the strict
// inequalities are kept exactly as in the reference so values at nodes agree in
// sign and zero-ness.
#pragma once
#include "common.hpp"

namespace splpak {

// kind 1: left-linear (ib <= 1), 2: chapeau, 3: right-linear (ib >= nodes-2)  (:231-240)
__host__ __device__ inline int basis_kind(int ib, int nod)
{
    return ib <= 1 ? 1 : (ib >= nod - 2 ? 3 : 2);
}

// value (deriv 0), d/dx (1) or d2/dx2 (2) of the 1-D basis function centred at xb.
// s = 1/dx.
__host__ __device__ inline double basis_1d(int kind, int deriv, double x, double xb, double s)
{
#pragma clang fp contract(off)
    double b = 0.0;
    if (kind == 2) {
        if (deriv == 0) {                       // :253-270
            const double z = fabs(s * (x - xb)) - 2.0;
            if (z < 0.0) {
                b = -0.25 * (z * z * z);
                const double z1 = z + 1.0;
                if (z1 < 0.0) b += z1 * z1 * z1;
            }
        } else if (deriv == 1) {                // :272-286
            const double u = x - xb;
            const double f = (u < 0.0) ? -s : s;
            const double z = f * u - 2.0;
            if (z < 0.0) {
                b = -0.75 * (z * z);
                const double z1 = z + 1.0;
                if (z1 < 0.0) b += 3.0 * (z1 * z1);
                b *= f;
            }
        } else {                                // :288-300
            const double z = s * fabs(x - xb) - 2.0;
            if (z < 0.0) {
                b = -1.5 * z;
                const double z1 = z + 1.0;
                if (z1 < 0.0) b += 6.0 * z1;
                b *= s * s;
            }
        }
        return b;
    }
    // end functions: zero for z <= 0, cubic on (0,2), straight line 3z-3 beyond
    // (natural boundary + linear extrapolation, :358-379).  kind 1 mirrors kind 3.
    const double f = (kind == 1) ? -s : s;
    if (deriv == 0) {                           // :345-379
        const double z = (kind == 1) ? s * (xb - x) + 2.0 : s * (x - xb) + 2.0;
        if (z > 0.0) {
            if (z < 2.0) {
                b = 0.5 * (z * z * z);
                const double z1 = z - 1.0;
                if (z1 > 0.0) b -= z1 * z1 * z1;
            } else {
                b = 3.0 * z - 3.0;
            }
        }
    } else if (deriv == 1) {                    // :302-322
        const double z = f * (x - xb) + 2.0;
        if (z > 0.0) {
            if (z < 2.0) {
                b = 1.5 * (z * z);
                const double z1 = z - 1.0;
                if (z1 > 0.0) b -= 3.0 * (z1 * z1);
                b *= f;
            } else {
                b = 3.0 * f;
            }
        }
    } else {                                    // :324-340
        const double z = f * (x - xb) + 2.0;
        const double z1 = z - 1.0;
        if (fabs(z1) < 1.0) {
            b = 3.0 * z;
            if (z1 > 0.0) b -= 6.0 * z1;
            b *= f * f;
        }
    }
    return b;
}

// Window rule of the fit (:821-827) and of the evaluation (:1201-1209):
//   it = trunc(dxin*(x-xmin)); ibmn = min(max(it-1,0),nod-2); ibmx = max(min(it+2,nod-1),1).
// The reference visits [ibmn, ibmx] (2..4 nodes).  Here every point gets a fixed
// 4-wide window [ws, ws+3] that contains it; entries outside [ibmn, ibmx] are set
// to exactly 0, which is what the reference's row holds there.
__host__ __device__ inline int window_start(const Grid &g, int d, double x, int &lo, int &hi)
{
#pragma clang fp contract(off)
    const int nod = g.nodes[d];
    const double t = g.dxin[d] * (x - g.xmin[d]);
    // saturating truncation toward zero (keeps far-outside points defined)
    int it = (t >= 2.0e9) ? 2000000000 : (t <= -2.0e9 ? -2000000000 : (int)t);
    int a = it - 1;
    if (a < 0) a = 0;
    lo = a < nod - 2 ? a : nod - 2;
    int h = it + 2;
    if (h > nod - 1) h = nod - 1;
    hi = h > 1 ? h : 1;
    return a < nod - 4 ? a : nod - 4;
}

__host__ __device__ inline int window_table(const Grid &g, int d, double x, int deriv, double b[4])
{
#pragma clang fp contract(off)
    int lo, hi;
    const int ws = window_start(g, d, x, lo, hi);
    const int nod = g.nodes[d];
    const double s = g.dxin[d];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ib = ws + k;
        double v = 0.0;
        if (ib >= lo && ib <= hi) {
            const double xb = g.xmin[d] + (double)ib * g.dx[d];   // :246
            v = basis_1d(basis_kind(ib, nod), deriv, x, xb, s);
        }
        b[k] = v;
    }
    return ws;
}

// The same table for the VALUE (no derivative), written for the evaluation kernels, which spent most
// of their instructions here (round 2 counters: ~550 of 650 VALU instructions per query).  The
// piecewise definitions are applied with min/max instead of compare-and-select -- a cube of
// min(z, 0) IS "z^3 if z < 0 else 0" -- which keeps the reference's strict inequalities and, on the
// piece that applies, exactly basis_1d's operations in the same order: both tables return identical
// bits.  `interior`: the caller knows that all four window functions are interior ("chapeau")
// functions and all four lie inside [ibmn, ibmx] (2 <= ws <= nod-6 with the point inside the grid), so
// the end-function form and the window clipping are skipped.
__host__ __device__ inline double chapeau_value(double u)
{
#pragma clang fp contract(off)
    const double z = fabs(u) - 2.0;                    // :253-270
    const double zc = fmin(z, 0.0);
    double b = -0.25 * (zc * zc * zc);
    const double z1 = fmin(zc + 1.0, 0.0);
    b += z1 * z1 * z1;
    return b;
}

__host__ __device__ inline double endfn_value(double zarg)
{
#pragma clang fp contract(off)
    // :345-379 with z = zarg: 0 for z <= 0, z^3/2 - (z-1)^3 [z > 1] on (0,2), 3z - 3 from 2 on
    const double zp = fmax(zarg, 0.0);
    double b = 0.5 * (zp * zp * zp);
    const double z1 = fmax(zp - 1.0, 0.0);
    b -= z1 * z1 * z1;
    return (zarg < 2.0) ? b : 3.0 * zarg - 3.0;
}

// window start and whether the window is an interior one (see above); it = trunc(dxin (x - xmin))
__host__ __device__ inline int window_start_value(const Grid &g, int d, double x, int &lo, int &hi, bool &interior)
{
#pragma clang fp contract(off)
    const int nod = g.nodes[d];
    const double t = g.dxin[d] * (x - g.xmin[d]);
    int it = (t >= 2.0e9) ? 2000000000 : (t <= -2.0e9 ? -2000000000 : (int)t);
    int a = it - 1;
    if (a < 0) a = 0;
    lo = a < nod - 2 ? a : nod - 2;
    int h = it + 2;
    if (h > nod - 1) h = nod - 1;
    hi = h > 1 ? h : 1;
    // it in [3, nod-5]: ws = it-1 in [2, nod-6], entries ws..ws+3 = it-1..it+2 <= nod-3 are chapeau
    // functions and [lo, hi] = [ws, ws+3]
    interior = it >= 3 && it <= nod - 5;
    return a < nod - 4 ? a : nod - 4;
}

// The four values of an INTERIOR window in closed form (round 4).  With t = dxin (x - xmin), it = trunc(t) and u = t - it in
// [0, 1) the window functions are those of the nodes it-1 .. it+2 at distances 1+u, u, 1-u, 2-u (in units of dx), and the
// chapeau function (:253-270) is (2-z)^3/4 on [1, 2) and (2-z)^3/4 - (1-z)^3 on [0, 1):
//     b0 = (1-u)^3/4,   b1 = (2-u)^3/4 - (1-u)^3,   b2 = (1+u)^3/4 - u^3,   b3 = u^3/4
// 16 operations instead of the 56 of four separate evaluations (each with its own node coordinate, distance, two clamps and
// two cubes).  The values differ from those by rounding only (both carry the ~1e-14 that dxin (x - x_node) loses at 64 nodes);
// which form a query gets depends on the query alone (interior window in this dimension or not), never on the wave it is
// evaluated in, so every evaluation path still returns the same bits for the same query.
__host__ __device__ inline void window_values_interior(double u, double b[4])
{
    const double v = 1.0 - u, p = 2.0 - u, q = 1.0 + u;
    const double v3 = v * v * v, u3 = u * u * u, p3 = p * p * p, q3 = q * q * q;
    b[0] = 0.25 * v3;
    b[1] = fma(0.25, p3, -v3);
    b[2] = fma(0.25, q3, -u3);
    b[3] = 0.25 * u3;
}

// window start, whether the window is an interior one, and the fractional position u of x in its cell (interior windows)
__host__ __device__ inline int window_start_frac(const Grid &g, int d, double x, int &lo, int &hi, bool &interior, double &u, double &t, int &it)
{
#pragma clang fp contract(off)
    const int nod = g.nodes[d];
    t = g.dxin[d] * (x - g.xmin[d]);
    it = (t >= 2.0e9) ? 2000000000 : (t <= -2.0e9 ? -2000000000 : (int)t);
    u = t - (double)it;
    int a = it - 1;
    if (a < 0) a = 0;
    lo = a < nod - 2 ? a : nod - 2;
    int h = it + 2;
    if (h > nod - 1) h = nod - 1;
    hi = h > 1 ? h : 1;
    interior = it >= 3 && it <= nod - 5;
    return a < nod - 4 ? a : nod - 4;
}

// The four values of a window NEXT TO AN END of the grid (round 4): the point lies inside the grid (0 <= t, it <= nod - 2) in
// one of the first three or last three cells, and the grid has at least 8 nodes, so the window meets end functions of ONE end
// only.  With t = it + u as above, the entries are the interior window's values c[0..3] (nodes it-1 .. it+2, computed by
// window_values_interior for every lane anyway) with the end functions (:345-379) put in for the two end nodes, shifted by one
// place in the first and the last cell, where the window start is clamped (ws = 0 / nod - 4 instead of it - 1) and the entry
// that falls outside [lo, hi] is zero:
//     it = 0        nodes 0..3        [e0, e1, c3, 0 ]      e0 = endfn(2 - t)   (node 0: s (xb - x) + 2 = 0 - t + 2)
//     it = 1        nodes 0..3        [e0, e1, c2, c3]      e1 = endfn(3 - t)   (node 1)
//     it = 2        nodes 1..4        [e1, c1, c2, c3]
//     it = nod-4    nodes nod-5..     [c0, c1, c2, f0]      f0 = endfn(t - nod + 4)   (node nod-2: s (x - xb) + 2)
//     it = nod-3    nodes nod-4..     [c0, c1, f0, f1]      f1 = endfn(t - nod + 3)   (node nod-1)
//     it = nod-2    nodes nod-4..     [0,  c0, f0, f1]
// ~50 operations beside the closed form instead of the ~170 of four separate evaluations with both forms each (the boundary
// runs cost 3.3 times the interior ones per query: 18 % of the queries of a uniform batch at 64 nodes, 43 % of the vector
// instructions of the evaluation pass).  Differs from the general form by rounding only (the end function's argument is taken
// from t instead of s (xb - x) + 2); which form a query gets depends on the query alone.
__host__ __device__ inline void window_values_near(double t, int it, int nod, const double c[4], double b[4])
{
#pragma clang fp contract(off)
    const bool left = it <= 2;
    const double w = left ? 2.0 - t : t - (double)(nod - 4);      // argument of the end function of node 0 / node nod-2
    const double ea = endfn_value(w);                               // node 0      / node nod-2
    const double eb = endfn_value(left ? w + 1.0 : w - 1.0);        // node 1      / node nod-1
    const bool l0 = it == 0, l2 = it == 2, r4 = it == nod - 4, r2 = it == nod - 2;
    // entry 0: e1 (it = 2), e0 (it = 0, 1), 0 (it = nod-2), c0 (it = nod-4, nod-3)
    b[0] = l2 ? eb : (left ? ea : (r2 ? 0.0 : c[0]));
    // entry 1: e1 (it = 0, 1), c0 (it = nod-2), c1 (it = 2, nod-4, nod-3)
    b[1] = (left && !l2) ? eb : (r2 ? c[0] : c[1]);
    // entry 2: c3 (it = 0), c2 (it = 1, 2, nod-4), f0 (it = nod-3, nod-2)
    b[2] = l0 ? c[3] : ((left || r4) ? c[2] : ea);
    // entry 3: 0 (it = 0), c3 (it = 1, 2), f0 (it = nod-4), f1 (it = nod-3, nod-2)
    b[3] = l0 ? 0.0 : (left ? c[3] : (r4 ? ea : eb));
}

template <bool INTERIOR>
__host__ __device__ inline void window_values(const Grid &g, int d, double x, int ws, int lo, int hi, double b[4])
{
#pragma clang fp contract(off)
    const int nod = g.nodes[d];
    const double s = g.dxin[d];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ib = ws + k;
        const double xb = g.xmin[d] + (double)ib * g.dx[d];      // :246
        const double bc = chapeau_value(s * (x - xb));
        if (INTERIOR) {
            b[k] = bc;
        } else {
            // entries 0,1 can only be left-end or interior functions, entries 2,3 interior or right-end
            const double be = endfn_value(((k < 2) ? s * (xb - x) : s * (x - xb)) + 2.0);
            const bool end = (k < 2) ? (ib <= 1) : (ib >= nod - 2);
            const double v = end ? be : bc;
            b[k] = (ib >= lo && ib <= hi) ? v : 0.0;
        }
    }
}

// The value table as the evaluation kernels compute it, per query (eval.hip's eval_table does the same with wave-uniform
// branches around the rarer forms): form 0 = interior closed form, 1 = next to an end of the grid, 2 = general.
__host__ __device__ inline int window_table_selected(const Grid &g, int d, double x, double b[4], int &form)
{
    int lo, hi, it;
    bool interior;
    double u, t;
    const int ws = window_start_frac(g, d, x, lo, hi, interior, u, t, it);
    const int nod = g.nodes[d];
    double c[4];
    window_values_interior(u, c);
    if (interior) {
        form = 0;
        for (int k = 0; k < 4; ++k) b[k] = c[k];
    } else if (nod >= 8 && t >= 0.0 && it <= nod - 2) {
        form = 1;
        window_values_near(t, it, nod, c, b);
    } else {
        form = 2;
        window_values<false>(g, d, x, ws, lo, hi, b);
    }
    return ws;
}

__host__ __device__ inline int window_table_value(const Grid &g, int d, double x, double b[4])
{
    int lo, hi;
    bool interior;
    const int ws = window_start_value(g, d, x, lo, hi, interior);
    window_values<false>(g, d, x, ws, lo, hi, b);
    return ws;
}

// nearest-node address of the sparse-area histogram (:894-902), including the
// reference's quirk: an out-of-range coordinate only skips ITS dimension in the
// Horner address; the point is still counted (:899).
__host__ __device__ inline int nearest_node_address(const Grid &g, const double *x)
{
#pragma clang fp contract(off)
    // x and the Horner order are the caller's (reference) dimension order
    int iin = 0;
    for (int dc = 0; dc < g.ndim; ++dc) {
        const int d = g.ndim - 1 - dc;
        const double t = g.ref_dxin[d] * (x[d] - g.ref_xmin[d]) + 0.5;
        const int inidim = (t >= 2.0e9) ? 2000000000 : (t <= -2.0e9 ? -2000000000 : (int)t);
        if (inidim < 0 || inidim > g.ref_nodes[d] - 1) continue;
        iin = g.ref_nodes[d] * iin + inidim;
    }
    return iin;
}

}  // namespace splpak
