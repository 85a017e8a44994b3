// vigo_grid.hpp — device-side access to the packed voxel snapshot (GridView) and the uniform
// B-spline evaluation shared by the map/gate kernels.
#pragma once

#include "vigo_internal.hpp"

namespace vigo {

// 3-bit voxel value (bit0 inflated-occupied, bit1 unknown, bit2 occupied) at integer index,
// or 7 (everything set) outside the box: mapManager::occMap treats out-of-map as occupied and
// unknown (contract in include/vigo.h).
__device__ __forceinline__ unsigned grid_bits_at(const GridView& g, int ix, int iy, int iz) {
    if (ix < 0 || iy < 0 || iz < 0 || ix >= g.nx || iy >= g.ny || iz >= g.nz) return 7u;
    const size_t w = ((size_t)ix * g.ny + iy) * g.nzw + (iz >> 5);
    const unsigned sh = (unsigned)iz & 31u;
    const unsigned b0 = (g.planes[w] >> sh) & 1u;
    const unsigned b1 = (g.planes[g.plane_words + w] >> sh) & 1u;
    const unsigned b2 = (g.planes[2 * g.plane_words + w] >> sh) & 1u;
    return b0 | (b1 << 1) | (b2 << 2);
}

__device__ __forceinline__ unsigned grid_plane_at(const GridView& g, int plane, int ix, int iy, int iz) {
    if (ix < 0 || iy < 0 || iz < 0 || ix >= g.nx || iy >= g.ny || iz >= g.nz) return 1u;
    const size_t w = ((size_t)ix * g.ny + iy) * g.nzw + (iz >> 5);
    return (g.planes[(size_t)plane * g.plane_words + w] >> ((unsigned)iz & 31u)) & 1u;
}

// posToIndex: floor((p - origin) / res).  The range test is made on the double, so a NaN coordinate (which the
// conversion would turn into index 0) and anything beyond int range are outside the map like on the CPU.
__device__ __forceinline__ int grid_index(double p, double origin, double res, int n) {
    const double f = floor((p - origin) / res);
    return (f >= 0.0 && f < (double)n) ? (int)f : -1;
}

__device__ __forceinline__ unsigned grid_plane_pos(const GridView& g, int plane, double x, double y, double z) {
    const int ix = grid_index(x, g.origin[0], g.res, g.nx);
    const int iy = grid_index(y, g.origin[1], g.res, g.ny);
    const int iz = grid_index(z, g.origin[2], g.res, g.nz);
    return grid_plane_at(g, plane, ix, iy, iz);
}

// bspline::at (BS.cpp:32-58) for knots (i - degree) * ts, control point i = get(i)
template <int DEGREE, typename Get>
__device__ __forceinline__ void deboor(int ncp, double ts, double t, Get get, double (&out)[3]) {
    auto knot = [ts](int i) -> double { return (i - DEGREE) * ts; };
    const int knotsNum = ncp - 1 + DEGREE + 1 + 1;
    const double duration = knot(knotsNum - DEGREE - 1);
    const double tb = fmin(fmax(0.0, t), duration);
    int k = DEGREE;
    while (!(knot(k + 1) >= tb)) ++k;  // the lower span at exact knots (BS.cpp:37-42)
    double d[DEGREE + 1][3];
#pragma unroll
    for (int i = 0; i <= DEGREE; ++i) get(k - DEGREE + i, d[i]);
#pragma unroll
    for (int r = 1; r <= DEGREE; ++r) {
#pragma unroll
        for (int i = DEGREE; i >= r; --i) {
            const double alpha = (tb - knot(i + k - DEGREE)) / (knot(i + 1 + k - r) - knot(i + k - DEGREE));
#pragma unroll
            for (int a = 0; a < 3; ++a) d[i][a] = (1 - alpha) * d[i - 1][a] + alpha * d[i][a];
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) out[a] = d[DEGREE][a];
}

// value (deriv 0) / velocity (1) / acceleration (2) of the cubic spline over ctrl[N][3]
// (bspline(3, ctrl, ts) and its getDerivative() chain, BS.cpp:64-72)
__device__ __forceinline__ void traj_eval(const double* ctrl, int N, double ts, int deriv, double t, double (&out)[3]) {
    auto c0 = [ctrl](int i, double (&o)[3]) {
        o[0] = ctrl[3 * i]; o[1] = ctrl[3 * i + 1]; o[2] = ctrl[3 * i + 2];
    };
    // first derivative control point i: 3 * (c[i+1] - c[i]) / (knots(i+4) - knots(i+1)), knots(j) = (j-3)*ts
    auto c1 = [ctrl, ts](int i, double (&o)[3]) {
        const double den = ((i + 4) - 3) * ts - ((i + 1) - 3) * ts;
#pragma unroll
        for (int a = 0; a < 3; ++a) o[a] = 3 * (ctrl[3 * (i + 1) + a] - ctrl[3 * i + a]) / den;
    };
    // second derivative: 2 * (v[i+1] - v[i]) / (knots2(i+3) - knots2(i+1)), knots2(j) = (j-2)*ts
    auto c2 = [c1, ts](int i, double (&o)[3]) {
        double va[3], vb[3];
        c1(i, va);
        c1(i + 1, vb);
        const double den = ((i + 3) - 2) * ts - ((i + 1) - 2) * ts;
#pragma unroll
        for (int a = 0; a < 3; ++a) o[a] = 2 * (vb[a] - va[a]) / den;
    };
    if (deriv == 0) deboor<3>(N, ts, t, c0, out);
    else if (deriv == 1) deboor<2>(N - 1, ts, t, c1, out);
    else deboor<1>(N - 2, ts, t, c2, out);
}

}  // namespace vigo
