// vigo_minsnap.hip — batched min-snap QP: polyTrajSolver::solve (polyTrajSolver.cpp:849-904) with
// its problem construction (avgTimeAllocation :125-138, constructP :241-307, constructA :314-584,
// constructBound :587-846, updateCorridorParam :985-1012) and the rescale to un-normalised local
// time (:874-878), for many waypoint paths per launch.  The reference hands the three per-axis QPs
// to OSQP (ADMM, eps 1e-3); here each path is solved EXACTLY by one wavefront:
//   1. equality rows (waypoints, continuity up to `cont`, end conditions; 6 per segment at
//      cont = 4) are eliminated with an orthonormal null-space basis — Householder QR of A_eq'
//      held in LDS, lanes <-> columns;
//   2. the reduced Hessian H = Z'PZ (2 free coefficients per segment) is factorised once and
//      shared by the three axes;
//   3. the corridor boxes (two inequalities per box and axis) are handled by the Goldfarb-Idnani
//      dual active-set iteration on the reduced problem; an infeasible corridor is reported.
// One 64-lane workgroup per path, all matrices in LDS (59 KB at 7 segments, 114 KB at 10), no HBM traffic
// besides the waypoints in and the coefficients out.  This is dense fp64 linear algebra on
// 56..80-dimensional systems — latency-bound small-matrix work, no MFMA-sized contraction.
// The host restatement of the same algorithm is trajectory_planner_amd/host/src/polyTrajSolver.cpp.
#include "vigo_internal.hpp"

namespace vigo {
namespace {

constexpr int kD = 8;          // coefficients per segment (degree 7)
constexpr int kMaxSeg = 10;    // segments per path supported on the device
constexpr int kMaxBox = 1024;  // corridor boxes per path
constexpr int kMaxFree = 40;   // free coefficients after the elimination (2 per segment at cont = 4)
constexpr int kLanes = 64;

struct MinsnapArgs {
    int T, W, deg, diff, cont;
    double vel, corridor_res;
    const double* wp;        // [T][W][3]
    const double* corridor;  // [T][W-1] or NULL
    const double* conds;     // [T][4][3] or NULL
    double* out_coeffs;      // [T][W-1][3][deg+1]
    double* out_knots;       // [T][W]
    int32_t* out_status;     // [T]
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 1; m < kLanes; m <<= 1) v += __shfl_xor(v, m, kLanes);
    return v;
}

// d/dt^order of t^d at t (t is 0 or 1 here, or a corridor time for order 0)
__device__ __forceinline__ double deriv_coef(int d, int order, double t) {
    if (d < order) return 0.0;
    double f = 1.0;
    for (int k = 0; k < order; ++k) f *= (double)(d - k);
    double p = 1.0;
    for (int k = 0; k < d - order; ++k) p *= t;
    return f * p;
}
__device__ __forceinline__ double ipow(double x, int e) {
    double p = 1.0;
    for (int k = 0; k < e; ++k) p *= x;
    return p;
}
// PS.cpp:257-272: integral over normalised time of the squared diff-th derivative
__device__ __forceinline__ double snap_coef(int i, int j, int diff) {
    if (i < diff || j < diff) return 0.0;
    double f = 1.0;
    for (int d = 0; d < diff; ++d) f *= (double)(i - d) * (double)(j - d);
    return f / (double)(i + j - diff * 2 + 1);
}

// in-place Cholesky (lower) of a q x q matrix with leading dimension ld, by one lane
__device__ bool chol_serial(double* M, int q, int ld) {
    for (int j = 0; j < q; ++j) {
        double d = M[j * ld + j];
        for (int k = 0; k < j; ++k) d -= M[j * ld + k] * M[j * ld + k];
        if (!(d > 0)) return false;
        d = sqrt(d);
        M[j * ld + j] = d;
        for (int i = j + 1; i < q; ++i) {
            double s = M[i * ld + j];
            for (int k = 0; k < j; ++k) s -= M[i * ld + k] * M[j * ld + k];
            M[i * ld + j] = s / d;
        }
    }
    return true;
}
__device__ void chol_solve_serial(const double* L, int q, int ld, double* b) {
    for (int i = 0; i < q; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[i * ld + k] * b[k];
        b[i] = s / L[i * ld + i];
    }
    for (int i = q - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < q; ++k) s -= L[k * ld + i] * b[k];
        b[i] = s / L[i * ld + i];
    }
}

struct Layout {
    int n, me, nf;
    // offsets in doubles
    int knots, wpl, Q, M, beq, x0, PZ, H, Hinv, cvec, vbuf, xcur, boxSeg, boxT;
    int w, z, Hn, r, u, HN, Mq, rhs, npv, Nact, act, inAct, total;
};

__host__ __device__ inline Layout make_layout(int K, int cont) {
    Layout L;
    L.n = K * kD;
    L.me = (2 + (K - 1) + (K - 1)) + 2 * (2 + (K - 1)) + (K - 1) * (cont - 2);
    L.nf = L.n - L.me;
    int o = 0;
    auto take = [&](int cnt) { int at = o; o += (cnt + 1) & ~1; return at; };
    L.knots = take(K + 1);
    L.wpl = take((K + 1) * 3);
    L.Q = take(L.n * (L.me + L.n));   // W = [A_eq' | Q'] row by row, see the kernel
    L.M = L.Q;
    L.beq = take(L.me * 3);
    L.x0 = take(L.n * 3);
    L.PZ = take(L.n * L.nf);
    L.H = take(L.nf * L.nf);
    L.Hinv = take(L.nf * L.nf);
    L.cvec = take(L.nf * 3);
    L.vbuf = take(L.n);
    L.xcur = take(L.n);
    L.boxSeg = take(kMaxBox / 2 + 1);   // ints, two per double
    L.boxT = take(kMaxBox);
    L.w = take(L.nf);
    L.z = take(L.nf);
    L.Hn = take(L.nf);
    L.r = take(L.nf);
    L.u = take(L.nf + 1);
    L.HN = take(L.nf * L.nf);
    L.Mq = take(L.nf * L.nf);
    L.rhs = take(L.nf);
    L.npv = take(L.nf);
    L.Nact = take(L.nf * L.nf);         // normals of the working set, [a * nf + i]
    L.act = take(L.nf / 2 + 2);         // ints
    L.inAct = take((2 * kMaxBox) / 8 + 1);  // bytes
    L.total = o;
    return L;
}

__global__ void __launch_bounds__(kLanes) k_minsnap(MinsnapArgs A) {
    extern __shared__ __align__(16) double S[];
    __shared__ int s_nb, s_status, s_ip, s_q, s_drop, s_flag;
    __shared__ double s_t, s_sip, s_uq;
    const int t = blockIdx.x;
    if (t >= A.T) return;
    const int lane = threadIdx.x;
    const int W = A.W, K = W - 1, D = A.deg + 1;
    const Layout L = make_layout(K, A.cont);
    const int n = L.n, me = L.me, nf = L.nf;
    double* knots = S + L.knots;
    // One work matrix W, n rows of ld = me + n doubles: columns [0, me) hold A_eq' (-> R), columns
    // [me, me + n) hold Q TRANSPOSED (W(i, me + r) = Q(r, i)).  A Householder step then applies the
    // same "column -= 2 v (v'column) / v'v" to every column of W — one uniform task per lane, and
    // consecutive lanes hit consecutive LDS words (row-major Q at a 56-double stride put 64 rows on
    // two banks).
    double* Wm = S + L.Q;
    const int ld = me + n;
#define MAT(i, c) Wm[(i) * ld + (c)]
#define QAT(r, c) Wm[(c) * ld + me + (r)]
    double* beq = S + L.beq;  // [r * 3 + axis]
    double* x0 = S + L.x0;    // [i * 3 + axis]
    double* PZ = S + L.PZ;    // [i * nf + k]
    double* H = S + L.H;
    double* Hinv = S + L.Hinv;
    double* cvec = S + L.cvec;  // [k * 3 + axis]
    double* vbuf = S + L.vbuf;
    double* xcur = S + L.xcur;   // x0 + Z w of the axis being solved (normalised-time coefficients)
    double* wpl = S + L.wpl;
    double* npv = S + L.npv;
    double* Nact = S + L.Nact;
    int* boxSeg = reinterpret_cast<int*>(S + L.boxSeg);
    double* boxT = S + L.boxT;
    double* w = S + L.w;
    double* z = S + L.z;
    double* Hn = S + L.Hn;
    double* rr = S + L.r;
    double* u = S + L.u;
    double* HN = S + L.HN;     // [i * nf + j]
    double* Mq = S + L.Mq;     // [a * nf + b]
    double* rhs = S + L.rhs;
    int* act = reinterpret_cast<int*>(S + L.act);
    unsigned char* inAct = reinterpret_cast<unsigned char*>(S + L.inAct);

    const double* wp = A.wp + (size_t)t * W * 3;
    const double* cor = A.corridor ? A.corridor + (size_t)t * K : nullptr;
    const double* cond = A.conds ? A.conds + (size_t)t * 12 : nullptr;

    for (int i = lane; i < W * 3; i += kLanes) wpl[i] = wp[i];
    // ---- time knots, PS.cpp:125-138 (sequential accumulation, like the reference) ----
    if (lane == 0) {
        double total = 0.0;
        knots[0] = 0.0;
        for (int i = 1; i < W; ++i) {
            const double dx = wp[3 * i] - wp[3 * (i - 1)], dy = wp[3 * i + 1] - wp[3 * (i - 1) + 1], dz = wp[3 * i + 2] - wp[3 * (i - 1) + 2];
            total += sqrt(dx * dx + dy * dy + dz * dz) / A.vel;
            knots[i] = total;
        }
        s_status = 0;
        s_nb = 0;
    }
    for (int idx = lane; idx < n * ld; idx += kLanes) {
        const int i = idx / ld, c = idx % ld;
        Wm[idx] = (c >= me && c - me == i) ? 1.0 : 0.0;
    }
    __syncthreads();

    // ---- equality rows (PS.cpp:314-560 order), one lane per row, written as columns of M ----
    if (lane < me) {
        const int r = lane;
        const int last = (K - 1) * D;
        double b[3] = {0.0, 0.0, 0.0};
        auto put = [&](int col, double v) { MAT(col, r) += v; };
        int row = r;
        bool done = false;
        // position: start, end, K-1 interior waypoints, K-1 continuity
        if (row == 0) { for (int d = 0; d < D; ++d) put(d, deriv_coef(d, 0, 0.0)); for (int a = 0; a < 3; ++a) b[a] = wp[a]; done = true; }
        else if (row == 1) { for (int d = 0; d < D; ++d) put(last + d, deriv_coef(d, 0, 1.0)); for (int a = 0; a < 3; ++a) b[a] = wp[3 * (W - 1) + a]; done = true; }
        row -= 2;
        if (!done && row < K - 1) { for (int d = 0; d < D; ++d) put(row * D + d, deriv_coef(d, 0, 1.0)); for (int a = 0; a < 3; ++a) b[a] = wp[3 * (row + 1) + a]; done = true; }
        row -= K - 1;
        if (!done && row >= 0 && row < K - 1) {
            for (int d = 0; d < D; ++d) { put(row * D + d, deriv_coef(d, 0, 1.0)); put((row + 1) * D + d, -deriv_coef(d, 0, 0.0)); }
            done = true;
        }
        row -= K - 1;
        // velocity, acceleration: endpoints (normalised-time derivative, as the reference) + continuity
        for (int order = 1; order <= 2 && !done; ++order) {
            if (row == 0) { for (int d = 0; d < D; ++d) put(d, deriv_coef(d, order, 0.0)); if (cond) for (int a = 0; a < 3; ++a) b[a] = cond[3 * (order == 1 ? 0 : 2) + a]; done = true; break; }
            if (row == 1) { for (int d = 0; d < D; ++d) put(last + d, deriv_coef(d, order, 1.0)); if (cond) for (int a = 0; a < 3; ++a) b[a] = cond[3 * (order == 1 ? 1 : 3) + a]; done = true; break; }
            row -= 2;
            if (row < K - 1) {
                const double dtL = knots[row + 1] - knots[row], dtR = knots[row + 2] - knots[row + 1];
                for (int d = 0; d < D; ++d) {
                    put(row * D + d, deriv_coef(d, order, 1.0) * ipow(dtR, order));
                    put((row + 1) * D + d, -deriv_coef(d, order, 0.0) * ipow(dtL, order));
                }
                done = true;
                break;
            }
            row -= K - 1;
        }
        // jerk, snap, ... continuity
        for (int order = 3; order <= A.cont && !done; ++order) {
            if (row < K - 1) {
                const double dtL = knots[row + 1] - knots[row], dtR = knots[row + 2] - knots[row + 1];
                for (int d = 0; d < D; ++d) {
                    put(row * D + d, deriv_coef(d, order, 1.0) * ipow(dtR, order));
                    put((row + 1) * D + d, -deriv_coef(d, order, 0.0) * ipow(dtL, order));
                }
                done = true;
                break;
            }
            row -= K - 1;
        }
        // unit infinity norm per row (the continuity rows carry dt^order factors)
        double mx = 0.0;
        for (int i = 0; i < n; ++i) mx = fmax(mx, fabs(MAT(i, r)));
        if (!(mx > 0)) s_status = -1;
        else {
            for (int i = 0; i < n; ++i) MAT(i, r) /= mx;
            for (int a = 0; a < 3; ++a) beq[r * 3 + a] = b[a] / mx;
        }
    }
    __syncthreads();

    // ---- Householder QR of M = A_eq' : M -> R (upper me x me), Q accumulated ----
    for (int j = 0; j < me && s_status == 0; ++j) {
        double part = 0.0;
        for (int i = j + lane; i < n; i += kLanes) {
            const double a = MAT(i, j);
            vbuf[i] = a;
            part += a * a;
        }
        const double nrm = sqrt(wave_sum(part));
        __syncthreads();
        if (!(nrm > 1e-10)) { if (lane == 0) s_status = -1; __syncthreads(); break; }
        const double head = vbuf[j];
        const double alpha = head > 0 ? -nrm : nrm;
        __syncthreads();
        if (lane == 0) vbuf[j] = head - alpha;
        __syncthreads();
        part = 0.0;
        for (int i = j + lane; i < n; i += kLanes) part += vbuf[i] * vbuf[i];
        const double vv = wave_sum(part);
        if (vv > 0) {
            // tasks: columns j .. me + n - 1 of W, all alike
            const double* __restrict__ vb = vbuf;
            const double scale = 2.0 / vv;
            for (int c = j + lane; c < ld; c += kLanes) {
                double* __restrict__ col = Wm + c;
                double s = 0.0;
#pragma unroll 8
                for (int i = j; i < n; ++i) s += vb[i] * col[i * ld];
                s *= scale;
#pragma unroll 8
                for (int i = j; i < n; ++i) col[i * ld] -= s * vb[i];
            }
        }
        __syncthreads();
    }
    if (s_status != 0) {
        if (lane == 0) A.out_status[t] = s_status;
        return;
    }

    // ---- particular solution x0 = Y R^-T beq (per axis) ----
    // R'y = beq by column-oriented forward substitution: y_i is final once rows < i have been
    // eliminated; all lanes then remove its contribution from the rows below (y overwrites beq)
    for (int i = 0; i < me; ++i) {
        if (lane < 3) beq[i * 3 + lane] /= MAT(i, i);
        __syncthreads();
        for (int idx = lane; idx < (me - i - 1) * 3; idx += kLanes) {
            const int r = i + 1 + idx / 3, a = idx % 3;
            beq[r * 3 + a] -= MAT(i, r) * beq[i * 3 + a];
        }
        __syncthreads();
    }
    for (int idx = lane; idx < n * 3; idx += kLanes) {
        const int i = idx / 3, a = idx % 3;
        double s = 0.0;
        for (int k = 0; k < me; ++k) s += QAT(i, k) * beq[k * 3 + a];
        x0[idx] = s;
    }
    __syncthreads();

    if (nf > 0) {
        // ---- reduced problem: PZ = P Z, H = Z'PZ, c = Z'(P x0) ----
        for (int idx = lane; idx < n * nf; idx += kLanes) {
            const int k = idx / n, i = idx % n;   // i fastest: conflict-free reads of Qt
            const int seg = i / D, di = i % D;
            double s = 0.0;
            for (int dj = A.diff; dj < D; ++dj) s += snap_coef(di, dj, A.diff) * QAT(seg * D + dj, me + k);
            PZ[i * nf + k] = s;
        }
        __syncthreads();
        for (int idx = lane; idx < nf * nf; idx += kLanes) {
            const int a = idx / nf, b = idx % nf;
            double s = 0.0;
            for (int i = 0; i < n; ++i) s += QAT(i, me + a) * PZ[i * nf + b];
            H[idx] = s;
        }
        for (int idx = lane; idx < nf * 3; idx += kLanes) {
            const int k = idx / 3, a = idx % 3;
            // c_k = sum_i x0_i (P Z)_ik   (P symmetric)
            double s = 0.0;
            for (int i = 0; i < n; ++i) s += x0[i * 3 + a] * PZ[i * nf + k];
            cvec[idx] = s;
        }
        __syncthreads();
        for (int idx = lane; idx < nf * nf; idx += kLanes) {   // symmetrise into Hinv (scratch), then factor there
            const int a = idx / nf, b = idx % nf;
            Hinv[idx] = 0.5 * (H[a * nf + b] + H[b * nf + a]);
        }
        __syncthreads();
        // right-looking Cholesky across the lanes (Hinv holds L for now): column j is scaled, then
        // every lane updates its share of the trailing triangle
        for (int j = 0; j < nf && s_status == 0; ++j) {
            const double djj = Hinv[j * nf + j];
            if (!(djj > 0)) { if (lane == 0) s_status = -1; }
            const double dj = sqrt(djj);
            __syncthreads();
            if (s_status != 0) break;
            for (int i = j + lane; i < nf; i += kLanes) Hinv[i * nf + j] = (i == j) ? dj : Hinv[i * nf + j] / dj;
            __syncthreads();
            const int rem = nf - j - 1;
            for (int idx = lane; idx < rem * rem; idx += kLanes) {
                const int i = j + 1 + idx / rem, k = j + 1 + idx % rem;
                if (k <= i) Hinv[i * nf + k] -= Hinv[i * nf + j] * Hinv[k * nf + j];
            }
            __syncthreads();
        }
        __syncthreads();
        if (s_status != 0) {
            if (lane == 0) A.out_status[t] = s_status;
            return;
        }
        // H <- H^-1: lane k solves L L' x = e_k into column k of H
        for (int k = lane; k < nf; k += kLanes) {
            double col[kMaxFree];
            for (int i = 0; i < nf; ++i) col[i] = (i == k) ? 1.0 : 0.0;
            chol_solve_serial(Hinv, nf, nf, col);
            for (int i = 0; i < nf; ++i) H[i * nf + k] = col[i];
        }
        __syncthreads();
        for (int idx = lane; idx < nf * nf; idx += kLanes) Hinv[idx] = H[idx];
        __syncthreads();

        // ---- corridor boxes, PS.cpp:985-1012: centres on the straight leg at normalised times 0, dt, 2dt, .. <= 1 ----
        if (cor && lane == 0) {
            int nb = 0;
            bool overflow = false;
            for (int i = 0; i < K && !overflow; ++i) {
                if (cor[i] == 0.0) continue;
                const double duration = knots[i + 1] - knots[i];
                const int num = (int)ceil(duration * A.corridor_res);
                const double dt = 1.0 / num;
                for (double tt = 0; tt <= 1.0; tt += dt) {
                    if (nb >= kMaxBox) { overflow = true; break; }
                    boxSeg[nb] = i;
                    boxT[nb] = tt;
                    ++nb;
                }
            }
            s_nb = nb;
            if (overflow) s_status = -1;
        }
        __syncthreads();
        const int nb = s_nb;
        if (s_status != 0) {
            if (lane == 0) A.out_status[t] = s_status;
            return;
        }
        // ---- per axis: Goldfarb-Idnani dual active set on  min 1/2 w'Hw + c'w,  a_k'w >= b_k ----
        // constraint k: box k>>1, side k&1 (0: lower bound, 1: upper bound)
        const int nc = 2 * nb;
        for (int axis = 0; axis < 3; ++axis) {
            const double* corL = cor;
            // value side of constraint k at the current iterate: (polynomial value - lower bound) or
            // (upper bound - polynomial value) at the box time; bound magnitude for the tolerance
            auto slack_of = [&](int k, double& bound_mag) {
                const int b = k >> 1, seg = boxSeg[b];
                const double tt = boxT[b];
                double val = 0.0;
                for (int d = D - 1; d >= 0; --d) val = val * tt + xcur[seg * D + d];
                const double cen = wpl[3 * seg + axis] + (wpl[3 * (seg + 1) + axis] - wpl[3 * seg + axis]) * tt;
                const double rad = corL[seg];
                bound_mag = fabs((k & 1) ? cen + rad : cen - rad);
                return (k & 1) ? (cen + rad) - val : val - (cen - rad);
            };
            // normal of constraint k in the reduced space: +-(row of the box) * Z
            auto normal_of = [&](int k, int i) {
                const int b = k >> 1, seg = boxSeg[b];
                const double tt = boxT[b];
                double s = 0.0, pw = 1.0;
                for (int d = 0; d < D; ++d) { s += pw * QAT(seg * D + d, me + i); pw *= tt; }
                return (k & 1) ? -s : s;
            };
            // xcur <- x0 + Z w (normalised-time coefficients of this axis)
            auto refresh_xcur = [&]() {
                for (int i = lane; i < n; i += kLanes) {
                    double s = 0.0;
                    for (int k = 0; k < nf; ++k) s += QAT(i, me + k) * w[k];
                    xcur[i] = x0[i * 3 + axis] + s;
                }
            };
            for (int i = lane; i < nf; i += kLanes) {
                double s = 0.0;
                for (int j = 0; j < nf; ++j) s += Hinv[i * nf + j] * cvec[j * 3 + axis];
                w[i] = -s;
            }
            for (int k = lane; k < nc; k += kLanes) inAct[k] = 0;
            if (lane == 0) { s_q = 0; s_flag = 0; }
            __syncthreads();
            refresh_xcur();
            __syncthreads();
            const int maxIter = 50 * (nc + nf) + 100;
            int iter = 0;
            int polished = 0;
            // Once nothing is violated any more the primal point and the multipliers are recomputed from the
            // working set itself (u = M^-1 (b_A + N H^-1 c), w = H^-1 (N'u - c)): the increments t z that carried
            // w there leave it up to 1e-5 (relative) off the working set's face on long corridors (dozens of nearly
            // degenerate partial steps) — feasible but measurably sub-optimal.  Same formulas as the host solver;
            // should the polished point violate a box, the loop resumes from it.  Returns false when it keeps the
            // carried point (empty or ill-conditioned working set, a negative multiplier).
            auto polish = [&]() -> bool {
                const int q = s_q;
                if (q == 0) return false;
                // Hn <- H^-1 c ; HN <- H^-1 N ; rhs <- b_A + N H^-1 c with b_A = N w - slack (w is the carried point)
                for (int i = lane; i < nf; i += kLanes) {
                    double hs = 0.0;
                    for (int j = 0; j < nf; ++j) hs += Hinv[i * nf + j] * cvec[j * 3 + axis];
                    Hn[i] = hs;
                    for (int a = 0; a < q; ++a) {
                        double h = 0.0;
                        for (int j = 0; j < nf; ++j) h += Hinv[i * nf + j] * Nact[a * nf + j];
                        HN[i * nf + a] = h;
                    }
                }
                __syncthreads();
                for (int idx = lane; idx < q * q; idx += kLanes) {
                    const int a = idx / q, b = idx % q;
                    double ms = 0.0;
                    for (int i = 0; i < nf; ++i) ms += Nact[a * nf + i] * HN[i * nf + b];
                    Mq[a * nf + b] = ms;
                }
                for (int a = lane; a < q; a += kLanes) {
                    double mag;
                    double rs = -slack_of(act[a], mag);
                    for (int i = 0; i < nf; ++i) rs += Nact[a * nf + i] * (w[i] + Hn[i]);
                    rhs[a] = rs;
                }
                __syncthreads();
                if (lane == 0) {
                    bool ok = chol_serial(Mq, q, nf);
                    if (ok) {
                        chol_solve_serial(Mq, q, nf, rhs);
                        for (int a = 0; a < q; ++a) ok = ok && (rhs[a] >= 0.0);
                    }
                    s_drop = ok ? 1 : 0;
                    if (ok) for (int a = 0; a < q; ++a) u[a] = rhs[a];
                }
                __syncthreads();
                if (!s_drop) return false;
                for (int i = lane; i < nf; i += kLanes) {
                    double ws = -Hn[i];
                    for (int a = 0; a < q; ++a) ws += HN[i * nf + a] * rhs[a];
                    w[i] = ws;
                }
                __syncthreads();
                refresh_xcur();
                __syncthreads();
                return true;
            };
            while (nc > 0) {
                // most violated constraint outside the working set
                double worst = 0.0;
                int ip = -1;
                for (int k = lane; k < nc; k += kLanes) {
                    if (inAct[k]) continue;
                    double mag;
                    const double sk = slack_of(k, mag), tol = 1e-9 * (1.0 + mag);
                    if (sk < -tol && sk < worst) { worst = sk; ip = k; }
                }
#pragma unroll
                for (int m = 1; m < kLanes; m <<= 1) {
                    const double ow = __shfl_xor(worst, m, kLanes);
                    const int oi = __shfl_xor(ip, m, kLanes);
                    if (oi >= 0 && (ip < 0 || ow < worst || (ow == worst && oi < ip))) { worst = ow; ip = oi; }
                }
                if (ip < 0) {
                    if (polished < 2 && polish()) { ++polished; continue; }   // re-test the boxes at the polished point
                    break;
                }
                if (lane == 0) { s_ip = ip; s_sip = worst; s_uq = 0.0; }
                for (int i = lane; i < nf; i += kLanes) npv[i] = normal_of(ip, i);
                __syncthreads();
                bool failed = false;
                for (;;) {   // steps until the constraint ip is active (or the problem proves infeasible)
                    if (++iter > maxIter) { failed = true; if (lane == 0) s_flag = -1; break; }
                    const int q = s_q;
                    const int ipk = s_ip;
                    for (int i = lane; i < nf; i += kLanes) {
                        double hs = 0.0;
                        for (int j = 0; j < nf; ++j) hs += Hinv[i * nf + j] * npv[j];
                        Hn[i] = hs;
                        for (int a = 0; a < q; ++a) {
                            double h = 0.0;
                            for (int j = 0; j < nf; ++j) h += Hinv[i * nf + j] * Nact[a * nf + j];
                            HN[i * nf + a] = h;
                        }
                    }
                    __syncthreads();
                    for (int idx = lane; idx < q * q; idx += kLanes) {
                        const int a = idx / q, b = idx % q;
                        double ms = 0.0;
                        for (int i = 0; i < nf; ++i) ms += Nact[a * nf + i] * HN[i * nf + b];
                        Mq[a * nf + b] = ms;
                    }
                    for (int a = lane; a < q; a += kLanes) {
                        double rs = 0.0;
                        for (int i = 0; i < nf; ++i) rs += Nact[a * nf + i] * Hn[i];
                        rhs[a] = rs;
                    }
                    __syncthreads();
                    if (lane == 0) {
                        bool ok = true;
                        if (q > 0) {
                            ok = chol_serial(Mq, q, nf);
                            if (ok) chol_solve_serial(Mq, q, nf, rhs);
                        }
                        for (int a = 0; a < q; ++a) rr[a] = rhs[a];
                        if (!ok) s_flag = -1;
                    }
                    __syncthreads();
                    if (s_flag != 0) { failed = true; break; }
                    double npHn_p = 0.0, zn_p = 0.0;
                    for (int i = lane; i < nf; i += kLanes) {
                        double zi = Hn[i];
                        for (int a = 0; a < q; ++a) zi -= HN[i * nf + a] * rr[a];
                        z[i] = zi;
                        npHn_p += npv[i] * Hn[i];
                        zn_p += zi * npv[i];
                    }
                    const double npHn = wave_sum(npHn_p), zn = wave_sum(zn_p);
                    __syncthreads();
                    if (lane == 0) {
                        double t1 = INFINITY, t2 = INFINITY;
                        int drop = -1;
                        for (int a = 0; a < q; ++a)
                            if (rr[a] > 0 && u[a] / rr[a] < t1) { t1 = u[a] / rr[a]; drop = a; }
                        if (zn > 1e-11 * npHn) t2 = -s_sip / zn;
                        const double tt = t1 < t2 ? t1 : t2;
                        if (!(tt < INFINITY)) { s_flag = -2; }
                        else {
                            for (int a = 0; a < q; ++a) u[a] -= tt * rr[a];
                            s_uq += tt;
                            s_t = (t2 < INFINITY) ? tt : 0.0;   // primal step length
                            if (tt == t2) {           // full step: ip joins the working set
                                act[q] = ipk; u[q] = s_uq; inAct[ipk] = 1; s_q = q + 1; s_drop = -2;
                                for (int i = 0; i < nf; ++i) Nact[q * nf + i] = npv[i];
                            } else {                  // partial step: the blocking constraint leaves
                                inAct[act[drop]] = 0;
                                for (int a = drop; a + 1 < q; ++a) {
                                    act[a] = act[a + 1]; u[a] = u[a + 1];
                                    for (int i = 0; i < nf; ++i) Nact[a * nf + i] = Nact[(a + 1) * nf + i];
                                }
                                s_q = q - 1; s_drop = drop;
                            }
                        }
                    }
                    __syncthreads();
                    if (s_flag != 0) { failed = true; break; }
                    const double step = s_t;
                    for (int i = lane; i < nf; i += kLanes) w[i] += step * z[i];
                    __syncthreads();
                    refresh_xcur();
                    __syncthreads();
                    if (s_drop == -2) break;
                    if (lane == 0) { double mag; s_sip = slack_of(ipk, mag); }
                    __syncthreads();
                }
                if (failed) break;
            }
            __syncthreads();
            if (s_flag == 0 && nc > 0) {
                // the loop above only tests constraints outside its working set; on an infeasible corridor
                // rounding can hide the vanishing step direction and let it "finish" on an ill-conditioned
                // working set: every box is verified at the end (as the host solver does)
                int viol = 0;
                for (int k = lane; k < nc; k += kLanes) {
                    double mag;
                    if (slack_of(k, mag) < -1e-7 * (1.0 + mag)) viol = 1;
                }
                if (__any(viol) && lane == 0) s_flag = -2;
            }
            __syncthreads();
            if (s_flag != 0) {
                if (lane == 0) A.out_status[t] = s_flag;
                return;
            }
            // x = x0 + Z w, rescaled to un-normalised local time (PS.cpp:874-878), straight to HBM
            for (int i = lane; i < n; i += kLanes) {
                const int seg = i / D, d = i % D;
                const double dtS = knots[seg + 1] - knots[seg];
                A.out_coeffs[(((size_t)t * K + seg) * 3 + axis) * D + d] = xcur[i] / ipow(dtS, d);
            }
            __syncthreads();
        }
    } else {
        for (int idx = lane; idx < n * 3; idx += kLanes) {
            const int i = idx / 3, axis = idx % 3;
            const int seg = i / D, d = i % D;
            const double dtS = knots[seg + 1] - knots[seg];
            A.out_coeffs[(((size_t)t * K + seg) * 3 + axis) * D + d] = x0[idx] / ipow(dtS, d);
        }
    }
    for (int i = lane; i < W; i += kLanes) A.out_knots[(size_t)t * W + i] = knots[i];
    if (lane == 0) A.out_status[t] = 0;
}

#undef QAT
#undef MAT

}  // namespace

size_t minsnap_lds_bytes(int W, int cont) { return (size_t)make_layout(W - 1, cont).total * sizeof(double) + 1024; }
int minsnap_max_waypoints() { return kMaxSeg + 1; }

int launch_minsnap(hipStream_t s, int T, int W, int deg, int diff, int cont, double vel, double corridor_res,
                   const double* wp, const double* corridor, const double* conds, double* out_coeffs, double* out_knots,
                   int32_t* out_status, LaunchState& L) {
    if (T <= 0) return hipSuccess;
    MinsnapArgs a{T, W, deg, diff, cont, vel, corridor_res, wp, corridor, conds, out_coeffs, out_knots, out_status};
    const size_t lds = minsnap_lds_bytes(W, cont) - 1024;
    if (!L.minsnap_attr_set) {   // per handle = per device (LaunchState), not a function static
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_minsnap), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);  // minus the static __shared__ scalars
        if (e != hipSuccess) return (int)e;
        L.minsnap_attr_set = true;
    }
    hipLaunchKernelGGL(k_minsnap, dim3(T), dim3(kLanes), lds, s, a);
    return (int)hipGetLastError();
}

}  // namespace vigo
