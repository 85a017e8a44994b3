// vigo_fit.hip — batched bspline::parameterizeToBspline (bspline.cpp:74-138): waypoints + start/end
// velocity/acceleration -> cubic B-spline control points, the step right before the optimizer
// (bsplineTraj::updatePath, bsplineTraj.cpp:290-323).
//
// The reference builds the (K+4) x (K+2) system A (rows [1 4 1]/6 per waypoint, two velocity rows
// [-1 0 1]/(2 ts), two acceleration rows [1 -2 1]/ts^2) and solves min |A x - b| three times with
// Eigen's colPivHouseholderQr.  A depends on (K, ts) only, never on the data, so the batch shares
// one factorisation:
//   k_fit_setup   (once per (K, ts), one workgroup, all on the device) Householder QR of [A | I]
//                 -> R, Q'; back-substitution of every column of Q' gives the least-squares
//                 operator A+ = R^-1 Q'[:K+2] stored TRANSPOSED, pinvT[j][row], so that lanes
//                 (rows) read consecutive addresses;
//   k_bspline_fit_mfma  the batch as ONE dense product A+ * [rhs of all paths] on the fp64 matrix
//                 cores, 5 paths per wave and tile column group (see below); HBM-bound,
//                 (2K + 6) * 24 algorithmic bytes per path: 4.1 TB/s (51 % of peak) at 65 536
//                 paths x 30 waypoints, 2.0 TB/s at 62 waypoints;
//   k_bspline_fit_gen   more than 62 waypoints: VALU kernel, A+ streamed from L2.
// fp64, explicit fused multiply-adds in index order: deterministic, batch-invariant.
#include "vigo_internal.hpp"

namespace vigo {
namespace {

constexpr int kSetupThreads = 256;
constexpr int kMaxRows = VIGO_MAX_CTRL_POINTS + 2;  // R = K + 4 <= N_max + 2

__device__ __forceinline__ double fit_entry(int r, int c, int K, double ts) {
    if (r < K) {  // BS.cpp:102-104
        const int d = c - r;
        return d == 0 ? 1 / 6.0 : (d == 1 ? 4 / 6.0 : (d == 2 ? 1 / 6.0 : 0.0));
    }
    const int base = (r == K || r == K + 2) ? 0 : K - 1;  // BS.cpp:106-109
    const int d = c - base;
    if (d < 0 || d > 2) return 0.0;
    if (r < K + 2) return d == 0 ? -1 / 2.0 / ts : (d == 2 ? 1 / 2.0 / ts : 0.0);
    return d == 1 ? -2 / ts / ts : 1 / ts / ts;
}

// deterministic block sum (fixed tree)
__device__ __forceinline__ double block_sum(double v, double* red) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int s = kSetupThreads / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    const double out = red[0];
    __syncthreads();
    return out;
}

// W: R x (C + R) row-major work matrix in global memory, [A | I] -> [R | Q'].
__global__ void __launch_bounds__(kSetupThreads) k_fit_setup(int K, double ts, double* __restrict__ W,
                                                             double* __restrict__ pinvT) {
    const int R = K + 4, C = K + 2, Wc = C + R;
    const int tid = threadIdx.x;
    __shared__ double v[kMaxRows];
    __shared__ double red[kSetupThreads];
    for (int idx = tid; idx < R * Wc; idx += kSetupThreads) {
        const int r = idx / Wc, c = idx % Wc;
        W[idx] = c < C ? fit_entry(r, c, K, ts) : ((c - C == r) ? 1.0 : 0.0);
    }
    __syncthreads();
    for (int c = 0; c < C; ++c) {
        double part = 0.0;
        for (int r = c + tid; r < R; r += kSetupThreads) {
            const double a = W[(size_t)r * Wc + c];
            v[r - c] = a;
            part += a * a;
        }
        const double nrm = sqrt(block_sum(part, red));
        if (nrm == 0.0) continue;  // cannot happen for K >= 4 (A has full column rank)
        const double head = v[0];
        const double alpha = head > 0 ? -nrm : nrm;
        __syncthreads();
        if (tid == 0) v[0] = head - alpha;
        __syncthreads();
        part = 0.0;
        for (int r = tid; r < R - c; r += kSetupThreads) part += v[r] * v[r];
        const double vn = block_sum(part, red);
        if (vn != 0.0) {
            for (int cc = c + tid; cc < Wc; cc += kSetupThreads) {  // column cc <- (I - 2 v v'/v'v) column cc
                double s = 0.0;
                for (int r = c; r < R; ++r) s += v[r - c] * W[(size_t)r * Wc + cc];
                s = 2.0 * s / vn;
                for (int r = c; r < R; ++r) W[(size_t)r * Wc + cc] -= s * v[r - c];
            }
        }
        __syncthreads();
    }
    // x_j = R^-1 Q'[:C, j] for every column j of Q'; pinvT[j][row] = x_j[row]
    for (int j = tid; j < R; j += kSetupThreads) {
        double* x = pinvT + (size_t)j * C;
        for (int r = C - 1; r >= 0; --r) {
            double s = W[(size_t)r * Wc + C + j];
            for (int cc = r + 1; cc < C; ++cc) s -= W[(size_t)r * Wc + cc] * x[cc];
            x[r] = s / W[(size_t)r * Wc + r];
        }
    }
}

// rhs row j of trajectory b: waypoint j (j < K) or boundary condition j - K (zero when conds == NULL)
__device__ __forceinline__ double fit_rhs(const double* __restrict__ points, const double* __restrict__ conds,
                                          size_t b, int K, int i) {
    if (i < 3 * K) return points[b * 3 * (size_t)K + i];
    return conds ? conds[b * 12 + (i - 3 * K)] : 0.0;
}

// ---- the fit as what it is, a dense product ---------------------------------------------------
// ctrl[b] = A+ * rhs[b] applies ONE (K+2) x (K+4) operator to every path: out = A+ * [rhs_0 | rhs_1 | ...]
// with 3 columns (x, y, z) per path — a genuine shared-operand contraction, so it runs on the fp64
// matrix cores (v_mfma_f64_16x16x4_f64).  One wave takes 5 paths = 15 of the 16 tile columns:
//   A operand  A+ rows (16 per tile, TILES tiles), one double per lane and k-step, loaded once;
//   B operand  rhs(j, column): lane (k = lane >> 4, n = lane & 15) loads path n/3, row 4*step + k,
//              axis n%3 — every byte of the 5 paths is read exactly once, all loads of a group are
//              independent and in flight together (earlier VALU versions — A+ row in registers, the
//              rhs through LDS or the scalar cache — walked one path at a time behind its own load
//              latency and stayed at 1.7 TB/s whatever the LDS, FMA or store variant);
//   C/D        4 doubles per lane and tile: row (lane >> 4) + 4 r of the tile, column n.
// K + 4 is padded to a multiple of 4 with zero coefficients.  HBM-bound: (2K + 6) * 24 B per path.
typedef double v4f64 __attribute__((ext_vector_type(4)));

// SPLIT waves share a group of paths, each owning TILES of the SPLIT * TILES row tiles (the second
// wave's right-hand-side loads hit the cache lines the first one fetched): halves the A+ registers
// per wave at 62 waypoints, doubling the waves in flight.
template <int TILES, int KSTEPS, int SPLIT>
__global__ void __launch_bounds__(256) k_bspline_fit_mfma(int B, int K, const double* __restrict__ pinvT,
                                                          const double* __restrict__ points,
                                                          const double* __restrict__ conds, double* __restrict__ out) {
    const int R = K + 4, C = K + 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, kk = lane >> 4;
    const int bl = n / 3, ax = n - 3 * bl;       // path within the group, axis
    const int tile0 = (wave % SPLIT) * TILES;    // first row tile of this wave
    double Areg[TILES][KSTEPS];
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            const int row = 16 * (tile0 + t) + n, j = 4 * s + kk;
            Areg[t][s] = (row < C && j < R) ? pinvT[(size_t)j * C + row] : 0.0;
        }
    const int groups = (B + 4) / 5;
    constexpr int GPB = 4 / SPLIT;               // groups per workgroup and trip
    for (int g = blockIdx.x * GPB + wave / SPLIT; g < groups; g += gridDim.x * GPB) {
        const int b = 5 * g + bl;
        const bool live = (n < 15) && (b < B);
        double Bv[KSTEPS];
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            const int j = 4 * s + kk;
            double v = 0.0;
            if (live && j < K) v = points[((size_t)b * K + j) * 3 + ax];
            else if (live && j < R && conds) v = conds[(size_t)b * 12 + (j - K) * 3 + ax];
            Bv[s] = v;
        }
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            v4f64 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Areg[t][s], Bv[s], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * (tile0 + t) + kk + 4 * r;
                if (live && row < C) out[((size_t)b * C + row) * 3 + ax] = acc[r];
            }
        }
    }
}

// any K: one 256-thread workgroup per trajectory, rows strided, A+ read from L2 (same sums, same order)
__global__ void __launch_bounds__(256) k_bspline_fit_gen(int B, int K, const double* __restrict__ pinvT,
                                                         const double* __restrict__ points,
                                                         const double* __restrict__ conds, double* __restrict__ out) {
    const int R = K + 4, C = K + 2;
    __shared__ double in[kMaxRows * 3];
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        for (int i = threadIdx.x; i < 3 * R; i += 256) in[i] = fit_rhs(points, conds, (size_t)b, K, i);
        __syncthreads();
        for (int row = threadIdx.x; row < C; row += 256) {
            double ax = 0.0, ay = 0.0, az = 0.0;
            for (int j = 0; j < R; ++j) {
                const double p = pinvT[(size_t)j * C + row];
                ax = __builtin_fma(p, in[3 * j], ax);
                ay = __builtin_fma(p, in[3 * j + 1], ay);
                az = __builtin_fma(p, in[3 * j + 2], az);
            }
            double* dst = out + ((size_t)b * C + row) * 3;
            dst[0] = ax; dst[1] = ay; dst[2] = az;
        }
        __syncthreads();
    }
}

}  // namespace

size_t fit_work_doubles(int K) { return (size_t)(K + 4) * (size_t)(2 * K + 6); }
size_t fit_pinv_doubles(int K) { return (size_t)(K + 4) * (size_t)(K + 2); }

int launch_fit_setup(hipStream_t s, int K, double ts, double* work, double* pinvT) {
    hipLaunchKernelGGL(k_fit_setup, dim3(1), dim3(kSetupThreads), 0, s, K, ts, work, pinvT);
    return (int)hipGetLastError();
}

int launch_bspline_fit(hipStream_t s, int B, int K, const double* pinvT, const double* points, const double* conds,
                       double* out) {
    if (B <= 0) return hipSuccess;
    const int grid = B < 4096 ? B : 4096;  // 16 waves per CU worth of workgroups, grid-stride beyond
    const int groups = (B + 4) / 5;
    auto mgrid = [groups](int gpb) { const int need = (groups + gpb - 1) / gpb; return need < 2048 ? need : 2048; };
    if (K + 2 <= 32)        // 2 row tiles, K + 4 <= 36 = 9 k-steps, one wave per group
        hipLaunchKernelGGL((k_bspline_fit_mfma<2, 9, 1>), dim3(mgrid(4)), dim3(256), 0, s, B, K, pinvT, points, conds, out);
    else if (K + 2 <= 64)   // 4 row tiles, K + 4 <= 68 = 17 k-steps, two waves per group
        hipLaunchKernelGGL((k_bspline_fit_mfma<2, 17, 2>), dim3(mgrid(2)), dim3(256), 0, s, B, K, pinvT, points, conds, out);
    else
        hipLaunchKernelGGL(k_bspline_fit_gen, dim3(grid), dim3(256), 0, s, B, K, pinvT, points, conds, out);
    return (int)hipGetLastError();
}

}  // namespace vigo
