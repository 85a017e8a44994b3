// vigo_solver.hip — batched ViGO cost/gradient and the whole-solve L-BFGS kernel for gfx950.
//
// Mapping (MI355X-first, no MFMA: there is no dense contraction on this path):
//   * one 64-lane wavefront per workgroup; a trajectory owns a lane GROUP of 32 (N <= 32,
//     two trajectories per wave) or 64 lanes (N <= 64); lane p <-> control point p, so the
//     4-point jerk stencil and the 2/3-point vel/acc stencils are DPP wave shifts (no LDS).
//   * x, g, xp, gp, d live in VGPRs (3 scalars per lane each); the L-BFGS history
//     (m x {s,y}) lives in LDS, one column per interior control point: HBM sees the initial
//     control points and the result only.
//   * every scalar of the More-Thuente search is replicated across the group's lanes; the
//     two groups of a wave diverge freely (exec masking), all cross-lane traffic stays
//     inside a group.
//   * per-trajectory sums (cost terms, dot products) are butterfly all-reduces inside the
//     group: v += lane[i ^ m], m = 1,2,..,GROUP/2 — a fixed tree, so results are
//     deterministic and reproducible bit-for-bit by oracle/vigo_oracle.c's emulation mode.
//
// Arithmetic follows the reference expression by expression (bsplineTraj.cpp:802-1064 and
// solver/lbfgs.hpp:295-1349; see the citations on each block); the file is built with
// -ffp-contract=off so no FMA contraction changes a rounding.  Differences to the CPU
// reference are confined to: summation order of the reductions above, x*x / x*x*x instead
// of glibc pow(x,2|3), sqrt instead of pow(.,0.5).
#include "vigo_internal.hpp"

namespace vigo {
namespace {

constexpr int kWave = 64;
constexpr int kMaxMem = VIGO_MAX_MEM_SIZE;

// reference status codes, LB:20-80
enum : int {
    LB_CONVERGENCE = 0,
    LB_STOP = 1,
    LB_ALREADY_MINIMIZED = 2,
    LBERR_UNKNOWN = -1024,
    LBERR_LOGIC,
    LBERR_CANCELED,
    LBERR_INVALID_N,
    LBERR_INVALID_MEMSIZE,
    LBERR_INVALID_GEPSILON,
    LBERR_INVALID_TESTPERIOD,
    LBERR_INVALID_DELTA,
    LBERR_INVALID_MINSTEP,
    LBERR_INVALID_MAXSTEP,
    LBERR_INVALID_FDECCOEFF,
    LBERR_INVALID_SCURVCOEFF,
    LBERR_INVALID_XTOL,
    LBERR_INVALID_MAXLINESEARCH,
    LBERR_OUTOFINTERVAL,
    LBERR_INCORRECT_TMINMAX,
    LBERR_ROUNDING_ERROR,
    LBERR_MINIMUMSTEP,
    LBERR_MAXIMUMSTEP,
    LBERR_MAXIMUMLINESEARCH,
    LBERR_MAXIMUMITERATION,
    LBERR_WIDTHTOOSMALL,
    LBERR_INVALIDPARAMETERS,
    LBERR_INCREASEGRADIENT
};

// ---- cross-lane primitives -----------------------------------------------------------

// DPP wave shifts (GFX9 family): lane i receives lane i-1 (wave_shr:1) / i+1 (wave_shl:1).
__device__ __forceinline__ int dpp_prev_i32(int v) {
    return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, false);
}
__device__ __forceinline__ int dpp_next_i32(int v) {
    return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xF, 0xF, false);
}
__device__ __forceinline__ double from_prev(double v) {
    return __hiloint2double(dpp_prev_i32(__double2hiint(v)), dpp_prev_i32(__double2loint(v)));
}
__device__ __forceinline__ double from_next(double v) {
    return __hiloint2double(dpp_next_i32(__double2hiint(v)), dpp_next_i32(__double2loint(v)));
}
__device__ __forceinline__ float from_prev(float v) {
    return __int_as_float(dpp_prev_i32(__float_as_int(v)));
}
__device__ __forceinline__ float from_next(float v) {
    return __int_as_float(dpp_next_i32(__float_as_int(v)));
}

// butterfly all-reduce of K independent values inside a GROUP-lane group
template <int GROUP, int K>
__device__ __forceinline__ void group_sum(double (&v)[K]) {
#pragma unroll
    for (int m = 1; m < GROUP; m <<= 1) {
#pragma unroll
        for (int q = 0; q < K; ++q) v[q] += __shfl_xor(v[q], m, kWave);
    }
}
template <int GROUP>
__device__ __forceinline__ double group_sum1(double v) {
    double a[1] = {v};
    group_sum<GROUP, 1>(a);
    return a[0];
}

__device__ __forceinline__ double sum3(double a, double b, double c) { return (a + b) + c; }

// per-lane view of one trajectory's inputs
struct LaneProblem {
    int N;
    int p;            // control point of this lane
    bool has_pt;      // p < N
    bool interior;    // 3 <= p <= N-4 (a free control point)
    int g_begin, g_end;  // this control point's guide pairs
    const double* gpv;
    const uint8_t* gunk;
    int o_begin, o_end;  // this trajectory's obstacles
    const double* obs;
    double w[4];
};

// ---- cost + gradient at the point held in c (BT.cpp:802-821) ---------------------------
// T is the element type of points/gradients; sums are fp64.  Returns the weighted total
// (group-uniform); g receives the weighted gradient on interior lanes, 0 elsewhere.
template <typename T, int GROUP>
__device__ __forceinline__ double eval_cost_grad(const DevConst& K, const LaneProblem& Q,
                                                 const T (&c)[3], T (&g)[3], double (&terms)[4]) {
    const int N = Q.N, p = Q.p;
    // 7-point window c[p-3..p+3] by chained DPP shifts
    T m1[3], m2[3], m3[3], p1[3], p2[3], p3[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        m1[a] = from_prev(c[a]);
        m2[a] = from_prev(m1[a]);
        m3[a] = from_prev(m2[a]);
        p1[a] = from_next(c[a]);
        p2[a] = from_next(p1[a]);
        p3[a] = from_next(p2[a]);
    }

    T Gd[3] = {0, 0, 0}, Gs[3] = {0, 0, 0}, Gf[3] = {0, 0, 0}, Go[3] = {0, 0, 0};
    double part[4] = {0.0, 0.0, 0.0, 0.0};  // distance, smoothness, feasibility, dynamic

    // ---- smoothness, BT.cpp:934-950 (gather form of the scatter-add) ----
    {
        T J0[3], J1[3], J2[3], J3[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            J0[a] = ((p3[a] - 3 * p2[a]) + 3 * p1[a]) - c[a];   // i = p
            J1[a] = ((p2[a] - 3 * p1[a]) + 3 * c[a]) - m1[a];   // i = p-1
            J2[a] = ((p1[a] - 3 * c[a]) + 3 * m1[a]) - m2[a];   // i = p-2
            J3[a] = ((c[a] - 3 * m1[a]) + 3 * m2[a]) - m3[a];   // i = p-3
        }
        if (Q.has_pt && p <= N - 4)
            part[1] = sum3((double)(J0[0] * J0[0]), (double)(J0[1] * J0[1]), (double)(J0[2] * J0[2]));
        if (Q.interior) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                T acc = T(2.0) * J3[a];             // i=p-3: col(i+3) += gradTemp
                acc += T(-3.0) * (T(2.0) * J2[a]);  // i=p-2: col(i+2) += -3*gradTemp
                acc += T(3.0) * (T(2.0) * J1[a]);   // i=p-1: col(i+1) += 3*gradTemp
                acc += -(T(2.0) * J0[a]);           // i=p  : col(i)   += -gradTemp
                Gs[a] = acc;
            }
        }
    }

    // ---- feasibility, BT.cpp:952-999 (limits hard-coded to 1.0, :955-956) ----
    {
        const T ts = (T)K.ts_ctrl, tis = (T)K.ts_inv_sqr;
        auto excess = [](T v) -> T { return v > T(1.0) ? v - T(1.0) : (v < T(-1.0) ? v + T(1.0) : T(0.0)); };
        double cf = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            T evP = excess((p1[a] - c[a]) / ts);    // velocity i = p
            T evM = excess((c[a] - m1[a]) / ts);    // velocity i = p-1
            T eaP = excess(((p2[a] - 2 * p1[a]) + c[a]) * tis);    // acc i = p
            T eaM1 = excess(((p1[a] - 2 * c[a]) + m1[a]) * tis);   // acc i = p-1
            T eaM2 = excess(((c[a] - 2 * m1[a]) + m2[a]) * tis);   // acc i = p-2
            if (Q.interior) {
                T acc = (T(2) * evM) / ts * tis;    // i=p-1: gradient(j,i+1)
                acc += (T(-2) * evP) / ts * tis;    // i=p  : gradient(j,i)
                acc += (T(2) * eaM2) * tis;         // i=p-2: gradient(j,i+2)
                acc += (T(-4) * eaM1) * tis;        // i=p-1: gradient(j,i+1)
                acc += (T(2) * eaP) * tis;          // i=p  : gradient(j,i)
                Gf[a] = acc;
            }
        }
        // cost partial of lane p: velocity i=p (x,y,z) then acceleration i=p (x,y,z)
        if (Q.has_pt && p <= N - 2) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                T ev = excess((p1[a] - c[a]) / ts);
                cf += (double)((ev * ev) * tis);
            }
        }
        if (Q.has_pt && p <= N - 3) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                T ea = excess(((p2[a] - 2 * p1[a]) + c[a]) * tis);
                cf += (double)(ea * ea);
            }
        }
        part[2] = cf;
    }

    // ---- guide-point distance, BT.cpp:823-932 ----
    if (Q.interior) {
        const T dth = (T)K.dth, da = (T)K.da, db = (T)K.db, dcc = (T)K.dc, uf = (T)K.unc_factor;
        double cd = 0.0;
        for (int j = Q.g_begin; j < Q.g_end; ++j) {
            const double* pv = Q.gpv + 6 * (size_t)j;
            const T px = (T)pv[0], py = (T)pv[1], pz = (T)pv[2];
            const T vx = (T)pv[3], vy = (T)pv[4], vz = (T)pv[5];
            const bool unk = Q.gunk ? (Q.gunk[j] != 0) : false;
            T dist = ((c[0] - px) * vx + (c[1] - py) * vy) + (c[2] - pz) * vz;
            T e = dth - dist;
            T ct, k;
            bool hit = true, scale = false;
            if (e <= -dth) {                       // too far: never scaled by the unknown factor
                T ne = -e;
                ct = (ne * ne) * ne;
                k = T(3.0) * (ne * ne);
            } else if (e > T(0) && e <= dth) {     // e == dth lands here
                ct = (e * e) * e;
                k = T(-3.0) * (e * e);
                scale = unk;
            } else if (e >= dth) {
                ct = (da * (e * e) + db * e) + dcc;
                k = -((T(2) * da) * e + db);
                scale = unk;
            } else {
                hit = false; ct = 0; k = 0;
            }
            if (hit) {
                T gx = k * vx, gy = k * vy, gz = k * vz;
                if (scale) { ct *= uf; gx *= uf; gy *= uf; gz *= uf; }
                if (!K.plan_in_z) gz = T(0.0);
                cd += (double)ct;
                Gd[0] += gx; Gd[1] += gy; Gd[2] += gz;
            }
        }
        if (K.plan_in_z) {
            // BT.cpp:897-930, reproduced with its x-row gradient and heightDistMax band test
            const T hth = (T)K.hth, ha = (T)K.ha, hb = (T)K.hb, hc = (T)K.hc;
            T hmin = c[2] - (T)K.min_h, hmax = c[2] - (T)K.max_h;
            if (hmin < T(0)) {
                T e = hth - hmin;
                cd += (double)((ha * (e * e) + hb * e) + hc);
                Gd[0] += -((T(2) * ha) * e + hb) * T(-1.0);
            } else if (hmin >= T(0) && hmax < hth) {
                T e = hth - hmin;
                cd += (double)((e * e) * e);
                Gd[0] += T(-3.0) * (e * e) * T(-1.0);
            }
            if (hmax > T(0)) {
                T e = hth + hmax;
                cd += (double)((ha * (e * e) + hb * e) + hc);
                Gd[0] += -((T(2) * ha) * e + hb) * T(1.0);
            } else if (hmax <= T(0) && hmax >= -hth) {
                T e = hth + hmax;
                cd += (double)((e * e) * e);
                Gd[0] += T(-3.0) * (e * e) * T(1.0);
            }
        }
        part[0] = cd;
    }

    // ---- dynamic obstacles, BT.cpp:1001-1064 ----
    if (Q.interior && Q.o_end > Q.o_begin) {
        const T thr0 = (T)K.thr_dyn, oa = (T)K.oa, ob = (T)K.ob, oc = (T)K.oc;
        double co = 0.0;
        for (int j = Q.o_begin; j < Q.o_end; ++j) {
            const double* o = Q.obs + 9 * (size_t)j;
            const T opx = (T)o[0], opy = (T)o[1], ovx = (T)o[3], ovy = (T)o[4];
            const T hx = (T)o[6] / 2, hy = (T)o[7] / 2;
            const T size = sqrt(hx * hx + hy * hy);
            for (int n = 0; n <= K.pred_num; n += 2) {  // skipFactor = 2, BT.cpp:1006
                const T tn = (T)((double)n * K.ts);
                const T px = opx + tn * ovx, py = opy + tn * ovy;
                // integer division n/predictionNum, BT.cpp:1020
                const T thr = (T(1) - (T)(n / K.pred_num) * T(0.2)) * thr0;
                const T dx = c[0] - px, dy = c[1] - py, dz = T(0.0);
                const T nrm = sqrt((dx * dx + dy * dy) + dz * dz);
                const T e = thr - (nrm - size);
                const T gx = dx / nrm, gy = dy / nrm, gz = dz / nrm;
                if (e <= T(0)) {
                    // no punishment
                } else if (e > T(0) && e <= thr) {
                    co += (double)((e * e) * e);
                    T k = T(-3.0) * (e * e);
                    Go[0] += k * gx; Go[1] += k * gy; Go[2] += k * gz;
                } else if (e >= thr) {
                    co += (double)((oa * (e * e) + ob * e) + oc);
                    T k = -((T(2) * oa) * e + ob);
                    Go[0] += k * gx; Go[1] += k * gy; Go[2] += k * gz;
                }
            }
        }
        part[3] = co;
    }

    group_sum<GROUP, 4>(part);
#pragma unroll
    for (int q = 0; q < 4; ++q) terms[q] = part[q];
    const T w0 = (T)Q.w[0], w1 = (T)Q.w[1], w2 = (T)Q.w[2], w3 = (T)Q.w[3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
        g[a] = Q.interior ? (((w0 * Gd[a] + w1 * Gs[a]) + w2 * Gf[a]) + w3 * Go[a]) : T(0);
    return ((Q.w[0] * part[0] + Q.w[1] * part[1]) + Q.w[2] * part[2]) + Q.w[3] * part[3];
}

template <typename T>
__device__ __forceinline__ double dot3(const T (&a)[3], const T (&b)[3]) {
    return sum3((double)a[0] * (double)b[0], (double)a[1] * (double)b[1], (double)a[2] * (double)b[2]);
}

template <int GROUP>
__device__ __forceinline__ void load_problem(const SolveArgs& A, const DevConst& K, int b, int p,
                                             LaneProblem& Q) {
    const int N = A.N;
    Q.N = N;
    Q.p = p;
    Q.has_pt = p < N;
    Q.interior = (p >= 3) && (p <= N - 4);
    Q.gpv = A.guide_pv;
    Q.gunk = A.guide_unk;
    Q.g_begin = Q.g_end = 0;
    if (Q.interior && A.guide_off) {
        Q.g_begin = A.guide_off[(size_t)b * N + p];
        Q.g_end = A.guide_off[(size_t)b * N + p + 1];
    }
    Q.obs = A.obs;
    if (A.obs_off) {
        Q.o_begin = A.obs_off[b];
        Q.o_end = A.obs_off[b + 1];
    } else {
        Q.o_begin = 0;
        Q.o_end = A.obs ? A.n_obs_shared : 0;
    }
    if (A.weights) {
#pragma unroll
        for (int q = 0; q < 4; ++q) Q.w[q] = A.weights[4 * (size_t)b + q];
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) Q.w[q] = K.w[q];
    }
}

// ---- standalone cost/gradient kernel (vigo_cost_grad) ----------------------------------
template <typename T, int GROUP>
__global__ void __launch_bounds__(kWave) k_cost_grad(SolveArgs A, DevConst K) {
    constexpr int TPB = kWave / GROUP;
    const int lane = threadIdx.x;
    const int p = lane % GROUP;
    const int b = blockIdx.x * TPB + lane / GROUP;
    if (b >= A.B) return;
    LaneProblem Q;
    load_problem<GROUP>(A, K, b, p, Q);
    T c[3] = {0, 0, 0};
    if (Q.has_pt) {
        const double* src = A.ctrl + ((size_t)b * A.N + p) * 3;
        c[0] = (T)src[0]; c[1] = (T)src[1]; c[2] = (T)src[2];
    }
    T g[3];
    double terms[4];
    double f = eval_cost_grad<T, GROUP>(K, Q, c, g, terms);
    if (Q.interior && A.out_grad) {
        double* dst = A.out_grad + ((size_t)b * (A.N - 6) + (p - 3)) * 3;
        dst[0] = (double)g[0]; dst[1] = (double)g[1]; dst[2] = (double)g[2];
    }
    if (p == 0) {
        if (A.out_cost) A.out_cost[b] = f;
        if (A.out_terms) {
#pragma unroll
            for (int q = 0; q < 4; ++q) A.out_terms[4 * (size_t)b + q] = terms[q];
        }
    }
}

// ---- More-Thuente helpers (per-lane scalar code, group-uniform values) -------------------
struct LsPoint { double t, f, d; };

// LB:308-324
__device__ __forceinline__ double cubic_min(double u, double fu, double du, double v, double fv, double dv) {
    double d = v - u;
    double theta = (fu - fv) * 3 / d + du + dv;
    double p = fabs(theta), q = fabs(du), r = fabs(dv);
    double s = p >= q ? p : q;
    s = s >= r ? s : r;
    double a = theta / s;
    double gamm = s * sqrt(a * a - (du / s) * (dv / s));
    if (v < u) gamm = -gamm;
    p = gamm - du + theta;
    q = gamm - du + gamm + dv;
    r = p / q;
    return u + r * d;
}
// LB:338-366
__device__ __forceinline__ double cubic_min_bounded(double u, double fu, double du, double v, double fv,
                                                    double dv, double xmin, double xmax) {
    double d = v - u;
    double theta = (fu - fv) * 3 / d + du + dv;
    double p = fabs(theta), q = fabs(du), r = fabs(dv);
    double s = p >= q ? p : q;
    s = s >= r ? s : r;
    double a = theta / s;
    double gamm = a * a - (du / s) * (dv / s);
    gamm = gamm > 0 ? s * sqrt(gamm) : 0;
    if (u < v) gamm = -gamm;
    p = gamm - dv + theta;
    q = gamm - dv + gamm + du;
    r = p / q;
    if (r < 0. && gamm != 0.) return v - r * d;
    if (a < 0) return xmax;
    return xmin;
}
// LB:377-379
__device__ __forceinline__ double quad_min(double u, double fu, double du, double v, double fv) {
    double a = v - u;
    return u + du / ((fu - fv) / a + du) / 2 * a;
}
// LB:389-391
__device__ __forceinline__ double quad_min_secant(double u, double du, double v, double dv) {
    double a = u - v;
    return v + dv / (dv - du) * a;
}

// LB:506-714
__device__ __forceinline__ int trial_interval(LsPoint& X, LsPoint& Y, LsPoint& Tr, double tmin,
                                              double tmax, int& brackt) {
    int bound;
    const int dsign = Tr.d * (X.d / fabs(X.d)) < 0.;
    double mc, mq, newt;
    if (brackt) {
        const double lo = X.t <= Y.t ? X.t : Y.t;
        const double hi = X.t >= Y.t ? X.t : Y.t;
        if (Tr.t <= lo || hi <= Tr.t) return LBERR_OUTOFINTERVAL;
        if (0. <= X.d * (Tr.t - X.t)) return LBERR_INCREASEGRADIENT;
        if (tmax < tmin) return LBERR_INCORRECT_TMINMAX;
    }
    if (X.f < Tr.f) {
        brackt = 1;
        bound = 1;
        mc = cubic_min(X.t, X.f, X.d, Tr.t, Tr.f, Tr.d);
        mq = quad_min(X.t, X.f, X.d, Tr.t, Tr.f);
        newt = (fabs(mc - X.t) < fabs(mq - X.t)) ? mc : mc + 0.5 * (mq - mc);
    } else if (dsign) {
        brackt = 1;
        bound = 0;
        mc = cubic_min(X.t, X.f, X.d, Tr.t, Tr.f, Tr.d);
        mq = quad_min_secant(X.t, X.d, Tr.t, Tr.d);
        newt = (fabs(mc - Tr.t) > fabs(mq - Tr.t)) ? mc : mq;
    } else if (fabs(Tr.d) < fabs(X.d)) {
        bound = 1;
        mc = cubic_min_bounded(X.t, X.f, X.d, Tr.t, Tr.f, Tr.d, tmin, tmax);
        mq = quad_min_secant(X.t, X.d, Tr.t, Tr.d);
        if (brackt) newt = (fabs(Tr.t - mc) < fabs(Tr.t - mq)) ? mc : mq;
        else        newt = (fabs(Tr.t - mc) > fabs(Tr.t - mq)) ? mc : mq;
    } else {
        bound = 0;
        if (brackt)          newt = cubic_min(Tr.t, Tr.f, Tr.d, Y.t, Y.f, Y.d);
        else if (X.t < Tr.t) newt = tmax;
        else                 newt = tmin;
    }
    if (X.f < Tr.f) {
        Y = Tr;
    } else {
        if (dsign) Y = X;
        X = Tr;
    }
    if (tmax < newt) newt = tmax;
    if (newt < tmin) newt = tmin;
    if (brackt && bound) {
        mq = X.t + 0.66 * (Y.t - X.t);
        if (X.t < Y.t) { if (mq < newt) newt = mq; }
        else           { if (newt < mq) newt = mq; }
    }
    Tr.t = newt;
    return 0;
}

// ---- whole-solve kernel (vigo_optimize): BT.cpp:687-718 + LB:1024-1349 -------------------
// LDS: hist[(slot*2 + {0:s,1:y})*3 + axis][ROW] of T with ROW = TPB*(N-6) columns, then
//      ys[slot][TPB] doubles.
template <typename T, int GROUP>
__global__ void __launch_bounds__(kWave, 1) k_optimize(SolveArgs A, DevConst K) {
    constexpr int TPB = kWave / GROUP;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int N = A.N, NI = N - 6;
    const int ROW = TPB * NI;
    const int m = K.mem_size;
    T* hist = reinterpret_cast<T*>(lds_raw);
    double* ys_tab = reinterpret_cast<double*>(lds_raw + (((size_t)m * 6 * ROW * sizeof(T) + 15) & ~(size_t)15));

    const int lane = threadIdx.x;
    const int grp = lane / GROUP;
    const int p = lane % GROUP;
    const int b = blockIdx.x * TPB + grp;
    if (b >= A.B) return;

    LaneProblem Q;
    load_problem<GROUP>(A, K, b, p, Q);
    const int col = grp * NI + (p - 3);  // valid on interior lanes only

    // x holds this lane's control point: a free variable on interior lanes, a fixed boundary
    // point elsewhere (its g, d, s, y are identically zero so it never moves).
    T x[3] = {0, 0, 0};
    if (Q.has_pt) {
        const double* src = A.ctrl + ((size_t)b * N + p) * 3;
        x[0] = (T)src[0]; x[1] = (T)src[1]; x[2] = (T)src[2];
    }
    T g[3], xp[3], gp[3], d[3];
    double terms[4];
    int evals = 0;
    int ret = LBERR_UNKNOWN;
    int k = 0;

    double fx = eval_cost_grad<T, GROUP>(K, Q, x, g, terms);  // LB:1132
    ++evals;
#pragma unroll
    for (int a = 0; a < 3; ++a) d[a] = -g[a];  // LB:1144

    double xnorm, gnorm;
    {
        double r[2] = {Q.interior ? dot3(x, x) : 0.0, dot3(g, g)};
        group_sum<GROUP, 2>(r);
        xnorm = sqrt(r[0]);
        gnorm = sqrt(r[1]);
    }
    if (xnorm < 1.0) xnorm = 1.0;
    if (gnorm / xnorm <= K.g_epsilon) {
        ret = LB_ALREADY_MINIMIZED;  // LB:1154-1157
    } else {
        double step = 1.0 / sqrt(group_sum1<GROUP>(dot3(d, d)));  // LB:1163
        int end = 0;
        k = 1;
        for (;;) {
#pragma unroll
            for (int a = 0; a < 3; ++a) { xp[a] = x[a]; gp[a] = g[a]; }  // LB:1172-1173

            // ---------------- line_search_morethuente, LB:716-937 ----------------
            int ls;
            {
                const double stpmin = K.min_step, stpmax = K.max_step;
                int count = 0, brackt = 0, stage1 = 1, uinfo = 0;
                double dginit = 0.0;
                if (step <= 0.) {
                    ls = LBERR_INVALIDPARAMETERS;
                } else if ((dginit = group_sum1<GROUP>(dot3(g, d))) > 0) {
                    ls = LBERR_INCREASEGRADIENT;
                } else {
                    const double finit = fx;
                    const double dgtest = K.ftol * dginit;
                    double width = stpmax - stpmin;
                    double prev_width = 2.0 * width;
                    LsPoint X = {0., finit, dginit}, Y = {0., finit, dginit};
                    double stmin, stmax;
                    for (;;) {
                        if (brackt) {
                            stmin = X.t <= Y.t ? X.t : Y.t;
                            stmax = X.t >= Y.t ? X.t : Y.t;
                        } else {
                            stmin = X.t;
                            stmax = step + 4.0 * (step - X.t);
                        }
                        if (step < stpmin) step = stpmin;
                        if (stpmax < step) step = stpmax;
                        if ((brackt && ((step <= stmin || stmax <= step) || K.max_linesearch <= count + 1 || uinfo != 0)) ||
                            (brackt && (stmax - stmin <= K.xtol * stmax))) {
                            step = X.t;
                        }
                        // x <- xp + step * d  (LB:824-825)
#pragma unroll
                        for (int a = 0; a < 3; ++a) x[a] = xp[a] + (T)step * d[a];

                        fx = eval_cost_grad<T, GROUP>(K, Q, x, g, terms);  // LB:828
                        ++evals;
                        const double dg = group_sum1<GROUP>(dot3(g, d));
                        const double ftest1 = finit + step * dgtest;
                        ++count;

                        if (brackt && ((step <= stmin || stmax <= step) || uinfo != 0)) { ls = LBERR_ROUNDING_ERROR; break; }
                        if (step == stpmax && fx <= ftest1 && dg <= dgtest) { ls = LBERR_MAXIMUMSTEP; break; }
                        if (step == stpmin && (ftest1 < fx || dgtest <= dg)) { ls = LBERR_MINIMUMSTEP; break; }
                        if (brackt && (stmax - stmin) <= K.xtol * stmax) { ls = LBERR_WIDTHTOOSMALL; break; }
                        if (K.max_linesearch <= count) { ls = LBERR_MAXIMUMLINESEARCH; break; }
                        if (fx <= ftest1 && fabs(dg) <= K.gtol * (-dginit)) { ls = count; break; }

                        const double cmin = K.ftol <= K.gtol ? K.ftol : K.gtol;
                        if (stage1 && fx <= ftest1 && cmin * dginit <= dg) stage1 = 0;

                        LsPoint Tr;
                        if (stage1 && ftest1 < fx && fx <= X.f) {
                            LsPoint Xm = {X.t, X.f - X.t * dgtest, X.d - dgtest};
                            LsPoint Ym = {Y.t, Y.f - Y.t * dgtest, Y.d - dgtest};
                            Tr.t = step; Tr.f = fx - step * dgtest; Tr.d = dg - dgtest;
                            uinfo = trial_interval(Xm, Ym, Tr, stmin, stmax, brackt);
                            X.t = Xm.t; Y.t = Ym.t;
                            X.f = Xm.f + Xm.t * dgtest;
                            Y.f = Ym.f + Ym.t * dgtest;
                            X.d = Xm.d + dgtest;
                            Y.d = Ym.d + dgtest;
                            step = Tr.t;
                        } else {
                            Tr.t = step; Tr.f = fx; Tr.d = dg;
                            uinfo = trial_interval(X, Y, Tr, stmin, stmax, brackt);
                            step = Tr.t;
                        }
                        if (brackt) {
                            if (0.66 * prev_width <= fabs(Y.t - X.t)) step = X.t + 0.5 * (Y.t - X.t);
                            prev_width = width;
                            width = fabs(Y.t - X.t);
                        }
                    }
                }
            }

            if (ls < 0) {
                // LB:1189-1197.  optData_.controlPoints keeps the last trial (BT.cpp:803): write
                // it out now, then revert x like the reference does.
                if (Q.has_pt) {
                    double* dst = A.ctrl + ((size_t)b * N + p) * 3;
                    dst[0] = (double)x[0]; dst[1] = (double)x[1]; dst[2] = (double)x[2];
                }
#pragma unroll
                for (int a = 0; a < 3; ++a) { x[a] = xp[a]; g[a] = gp[a]; }
                ret = ls;
                break;
            }

            // convergence test, LB:1200-1225
            {
                double r[2] = {Q.interior ? dot3(x, x) : 0.0, dot3(g, g)};
                group_sum<GROUP, 2>(r);
                xnorm = sqrt(r[0]);
                gnorm = sqrt(r[1]);
            }
            if (xnorm < 1.0) xnorm = 1.0;
            if (gnorm / xnorm <= K.g_epsilon) { ret = LB_CONVERGENCE; break; }
            if (K.max_iterations != 0 && K.max_iterations < k + 1) { ret = LBERR_MAXIMUMITERATION; break; }

            // s, y, ys, yy — LB:1264-1276
            T sv[3], yv[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) { sv[a] = x[a] - xp[a]; yv[a] = g[a] - gp[a]; }
            if (Q.interior) {
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    hist[((size_t)(end * 2 + 0) * 3 + a) * ROW + col] = sv[a];
                    hist[((size_t)(end * 2 + 1) * 3 + a) * ROW + col] = yv[a];
                }
            }
            double ysyy[2] = {dot3(yv, sv), dot3(yv, yv)};
            group_sum<GROUP, 2>(ysyy);
            const double ys = ysyy[0], yy = ysyy[1];
            ys_tab[end * TPB + grp] = ys;

            // two-loop recursion, LB:1286-1316
            const int bound = (m <= k) ? m : k;
            ++k;
            end = (end + 1 == m) ? 0 : end + 1;
#pragma unroll
            for (int a = 0; a < 3; ++a) d[a] = -g[a];

            double alpha[kMaxMem];
#pragma unroll
            for (int age = 0; age < kMaxMem; ++age) {  // newest -> oldest
                if (age < bound) {
                    int j = end - 1 - age;  // ring slot of that age
                    if (j < 0) j += m;
                    T sj[3] = {0, 0, 0}, yj[3] = {0, 0, 0};
                    if (Q.interior) {
#pragma unroll
                        for (int a = 0; a < 3; ++a) {
                            sj[a] = hist[((size_t)(j * 2 + 0) * 3 + a) * ROW + col];
                            yj[a] = hist[((size_t)(j * 2 + 1) * 3 + a) * ROW + col];
                        }
                    }
                    double al = group_sum1<GROUP>(dot3(sj, d));
                    al /= ys_tab[j * TPB + grp];
                    alpha[age] = al;
                    const T na = (T)(-al);
#pragma unroll
                    for (int a = 0; a < 3; ++a) d[a] += na * yj[a];
                }
            }
            {
                const T sc = (T)(ys / yy);
#pragma unroll
                for (int a = 0; a < 3; ++a) d[a] *= sc;
            }
#pragma unroll
            for (int age = kMaxMem - 1; age >= 0; --age) {  // oldest -> newest
                if (age < bound) {
                    int j = end - 1 - age;  // ring slot of that age
                    if (j < 0) j += m;
                    T sj[3] = {0, 0, 0}, yj[3] = {0, 0, 0};
                    if (Q.interior) {
#pragma unroll
                        for (int a = 0; a < 3; ++a) {
                            sj[a] = hist[((size_t)(j * 2 + 0) * 3 + a) * ROW + col];
                            yj[a] = hist[((size_t)(j * 2 + 1) * 3 + a) * ROW + col];
                        }
                    }
                    double beta = group_sum1<GROUP>(dot3(yj, d));
                    beta /= ys_tab[j * TPB + grp];
                    const T co = (T)(alpha[age] - beta);
#pragma unroll
                    for (int a = 0; a < 3; ++a) d[a] += co * sj[a];
                }
            }
            step = 1.0;  // LB:1321
        }
    }

    // results.  On success/convergence/iteration cap the last evaluated point is x itself.
    if (ret >= 0 || ret == LBERR_MAXIMUMITERATION) {
        if (Q.has_pt) {
            double* dst = A.ctrl + ((size_t)b * N + p) * 3;
            dst[0] = (double)x[0]; dst[1] = (double)x[1]; dst[2] = (double)x[2];
        }
    }
    if (Q.interior && A.out_x) {
        double* dst = A.out_x + ((size_t)b * NI + (p - 3)) * 3;
        dst[0] = (double)x[0]; dst[1] = (double)x[1]; dst[2] = (double)x[2];
    }
    if (p == 0) {
        if (A.out_status) A.out_status[b] = ret;
        if (A.out_fx) A.out_fx[b] = fx;
        if (A.out_iters) A.out_iters[b] = k;
        if (A.out_evals) A.out_evals[b] = evals;
    }
}

template <typename T, int GROUP>
size_t optimize_lds_bytes(int N, int m) {
    const int TPB = kWave / GROUP;
    size_t h = (size_t)m * 6 * TPB * (N - 6) * sizeof(T);
    h = (h + 15) & ~(size_t)15;
    return h + (size_t)m * TPB * sizeof(double);
}

}  // namespace

DevConst make_dev_const(const vigo_params_t& P) {
    DevConst K{};
    K.dth = P.dthresh;
    K.da = 3.0 * P.dthresh;
    K.db = -3.0 * pow(P.dthresh, 2);
    K.dc = pow(P.dthresh, 3);
    K.unc_factor = P.uncertain_factor;
    K.hth = 0.2;
    K.ha = 3.0 * K.hth;
    K.hb = -3 * pow(K.hth, 2);
    K.hc = pow(K.hth, 3);
    K.min_h = P.min_height;
    K.max_h = P.max_height;
    K.ts_ctrl = P.ts_ctrl;
    K.ts_inv_sqr = 1 / pow(P.ts_ctrl, 2);
    K.ts = P.ts;
    K.thr_dyn = P.dist_thresh_dynamic;
    K.oa = 3.0 * P.dist_thresh_dynamic;
    K.ob = -3 * pow(P.dist_thresh_dynamic, 2);
    K.oc = pow(P.dist_thresh_dynamic, 3);
    K.pred_num = (int)(P.pred_horizon / P.ts);
    K.plan_in_z = P.plan_in_z;
    K.w[0] = P.w_distance; K.w[1] = P.w_smoothness; K.w[2] = P.w_feasibility; K.w[3] = P.w_dynamic;
    K.mem_size = P.mem_size;
    K.max_iterations = P.max_iterations;
    K.max_linesearch = P.max_linesearch;
    K.g_epsilon = P.g_epsilon;
    K.min_step = P.min_step;
    K.max_step = P.max_step;
    K.ftol = P.f_dec_coeff;
    K.gtol = P.s_curv_coeff;
    K.xtol = P.xtol;
    return K;
}

int launch_cost_grad(hipStream_t s, const SolveArgs& a, const DevConst& k, int precision) {
    if (a.B <= 0) return hipSuccess;
    const bool g32 = a.N <= 32;
    const int tpb = g32 ? 2 : 1;
    dim3 grid((a.B + tpb - 1) / tpb), block(kWave);
    if (precision == VIGO_PREC_F32) {
        if (g32) hipLaunchKernelGGL((k_cost_grad<float, 32>), grid, block, 0, s, a, k);
        else     hipLaunchKernelGGL((k_cost_grad<float, 64>), grid, block, 0, s, a, k);
    } else {
        if (g32) hipLaunchKernelGGL((k_cost_grad<double, 32>), grid, block, 0, s, a, k);
        else     hipLaunchKernelGGL((k_cost_grad<double, 64>), grid, block, 0, s, a, k);
    }
    return (int)hipGetLastError();
}

template <typename T, int GROUP>
static int launch_optimize_t(hipStream_t s, const SolveArgs& a, const DevConst& k) {
    const int tpb = kWave / GROUP;
    const size_t lds = optimize_lds_bytes<T, GROUP>(a.N, k.mem_size);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_optimize<T, GROUP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid((a.B + tpb - 1) / tpb), block(kWave);
    hipLaunchKernelGGL((k_optimize<T, GROUP>), grid, block, lds, s, a, k);
    return (int)hipGetLastError();
}

int launch_optimize(hipStream_t s, const SolveArgs& a, const DevConst& k, int precision) {
    if (a.B <= 0) return hipSuccess;
    const bool g32 = a.N <= 32;
    if (precision == VIGO_PREC_F32)
        return g32 ? launch_optimize_t<float, 32>(s, a, k) : launch_optimize_t<float, 64>(s, a, k);
    return g32 ? launch_optimize_t<double, 32>(s, a, k) : launch_optimize_t<double, 64>(s, a, k);
}

}  // namespace vigo
