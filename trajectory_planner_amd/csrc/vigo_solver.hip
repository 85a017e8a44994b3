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
// Lane 0 (63) has no source and keeps an unspecified value: every consumer of a shifted value
// is guarded by the control-point index, so the wave edges (and the seam between the two
// 32-lane groups) are never read.
__device__ __forceinline__ int dpp_prev_i32(int v) {
    return __builtin_amdgcn_mov_dpp(v, 0x138, 0xF, 0xF, false);
}
__device__ __forceinline__ int dpp_next_i32(int v) {
    return __builtin_amdgcn_mov_dpp(v, 0x130, 0xF, 0xF, false);
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

// Butterfly all-reduce of K independent values inside a GROUP-lane group, as VALU-speed DPP
// moves instead of ds_bpermute round trips.  The tree is the xor butterfly
// v += lane[i ^ m], m = 1, 2, 4, ..., GROUP/2:
//   m = 1, 2   quad_perm [1,0,3,2] / [2,3,0,1]
//   m = 4, 8   row_half_mirror / row_mirror: after the quad steps every lane of a quad holds the
//              quad sum, so pairing lane i with 7-i (15-i) adds the same two partial sums as
//              pairing it with i^4 (i^8) — identical bits, fp add being commutative;
//   m = 16     v_permlane16_swap (gfx950): rows 0/1 and 2/3 exchange, sum = even row + odd row;
//   m = 32     v_permlane32_swap: the two 32-lane halves exchange.
// All partners stay inside the group, so an exec-masked sibling group never contributes.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    // every source lane of these patterns lies in the reader's own row: no `old` value needed
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double xor16_sum(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double(h[0], l[0]) + __hiloint2double(h[1], l[1]);
}
__device__ __forceinline__ double xor32_sum(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(h[0], l[0]) + __hiloint2double(h[1], l[1]);
}
template <int GROUP, int K>
__device__ __forceinline__ void group_sum(double (&v)[K]) {
    static_assert(GROUP == 32 || GROUP == 64, "group is half a wave or a wave");
#pragma unroll
    for (int q = 0; q < K; ++q) v[q] += dpp_f64<0xB1>(v[q]);   // quad_perm:[1,0,3,2]
#pragma unroll
    for (int q = 0; q < K; ++q) v[q] += dpp_f64<0x4E>(v[q]);   // quad_perm:[2,3,0,1]
#pragma unroll
    for (int q = 0; q < K; ++q) v[q] += dpp_f64<0x141>(v[q]);  // row_half_mirror
#pragma unroll
    for (int q = 0; q < K; ++q) v[q] += dpp_f64<0x140>(v[q]);  // row_mirror
#pragma unroll
    for (int q = 0; q < K; ++q) v[q] = xor16_sum(v[q]);
    if (GROUP == 64) {
#pragma unroll
        for (int q = 0; q < K; ++q) v[q] = xor32_sum(v[q]);
    }
}
template <int GROUP>
__device__ __forceinline__ double group_sum1(double v) {
    double a[1] = {v};
    group_sum<GROUP, 1>(a);
    return a[0];
}

__device__ __forceinline__ double sum3(double a, double b, double c) { return (a + b) + c; }

// one L-BFGS history pair of one control point as it sits in LDS (48 B in fp64: three
// conflict-free ds_read_b128 per lane, one address register)
template <typename T>
struct alignas(16) HPair {
    T s[3];
    T y[3];
};

// per-lane view of one trajectory's inputs
constexpr int kGuideRegs = 2;  // guide pairs per control point kept in VGPRs (more: re-read from HBM/L2)
template <typename T>
struct LaneProblem {
    int N;
    int p;            // control point of this lane
    bool has_pt;      // p < N
    bool interior;    // 3 <= p <= N-4 (a free control point)
    int g_begin, g_end;  // this control point's guide pairs
    const double* gpv;
    const uint8_t* gunk;
    T gq[kGuideRegs][6];     // the first pairs, loaded once per solve (they never move)
    bool gqu[kGuideRegs];
    int o_begin, o_end;  // this trajectory's obstacles
    const double* obs;
    double w[4];
};

// One guide pair's contribution, BT.cpp:839-895.  e == dthresh takes the cubic branch (first
// else-if wins); the "too far" branch is never scaled by the unknown factor.
template <typename T>
__device__ __forceinline__ void guide_pair_term(const DevConst& K, const T (&c)[3], T px, T py, T pz, T vx,
                                                T vy, T vz, bool unk, double& cd, T (&Gd)[3]) {
    const T dth = (T)K.dth, da = (T)K.da, db = (T)K.db, dcc = (T)K.dc, uf = (T)K.unc_factor;
    const T dist = ((c[0] - px) * vx + (c[1] - py) * vy) + (c[2] - pz) * vz;
    const T e = dth - dist;
    T ct, k;
    bool scale = false;
    if (e <= -dth) {
        const T ne = -e;
        ct = (ne * ne) * ne;
        k = T(3.0) * (ne * ne);
    } else if (e > T(0) && e <= dth) {
        ct = (e * e) * e;
        k = T(-3.0) * (e * e);
        scale = unk;
    } else if (e >= dth) {
        ct = (da * (e * e) + db * e) + dcc;
        k = -((T(2) * da) * e + db);
        scale = unk;
    } else {
        return;  // -dthresh < e <= 0 (or NaN): no penalty
    }
    T gx = k * vx, gy = k * vy, gz = k * vz;
    if (scale) { ct *= uf; gx *= uf; gy *= uf; gz *= uf; }
    if (!K.plan_in_z) gz = T(0.0);
    cd += (double)ct;
    Gd[0] += gx; Gd[1] += gy; Gd[2] += gz;
}

template <typename T>
__device__ __forceinline__ double dot3(const T (&a)[3], const T (&b)[3]) {
    return sum3((double)a[0] * (double)b[0], (double)a[1] * (double)b[1], (double)a[2] * (double)b[2]);
}

// ---- cost + gradient at the point held in c (BT.cpp:802-821) ---------------------------
// T is the element type of points/gradients; sums are fp64.  g receives the weighted gradient
// on interior lanes, 0 elsewhere.  ONE 7-value group reduction returns
//   sums[0..3] = un-weighted distance / smoothness / feasibility / dynamic costs,
//   sums[4] = g.d, sums[5] = x.x over the free points, sums[6] = g.g
// (the line search needs g.d after every evaluation, LB:829, and the norms after the last one,
// LB:1200-1201; fusing them costs three more DPP chains instead of two more reductions).
// Returns the weighted total cost (group-uniform).
template <typename T, int GROUP>
__device__ __forceinline__ double eval_cost_grad(const DevConst& K, const LaneProblem<T>& Q, const T (&c)[3],
                                                 const T (&d)[3], T (&g)[3], double (&sums)[7]) {
    const int N = Q.N, p = Q.p;
    // forward window c[p+1..p+3] by chained DPP shifts
    T p1[3], p2[3], p3[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        p1[a] = from_next(c[a]);
        p2[a] = from_next(p1[a]);
        p3[a] = from_next(p2[a]);
    }

    T Gd[3] = {0, 0, 0}, Gs[3] = {0, 0, 0}, Gf[3] = {0, 0, 0}, Go[3] = {0, 0, 0};
    double part[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};

    // Every lane evaluates the stencil term whose FIRST point it owns (index i = p): jerk_p,
    // velocity_p, acceleration_p and their gradient magnitudes.  A control point's gradient is
    // the reference's scatter-add seen from the receiving column: the terms of i = p-3..p, which
    // are the SAME expressions evaluated by the lanes below, fetched with DPP shifts — identical
    // bits, no recomputation (and half the fp64 divisions by ts).

    // ---- smoothness, BT.cpp:934-950 ----
    {
        T gt0[3];  // gradTemp = 2 * jerk_i, i = p
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const T J0 = ((p3[a] - 3 * p2[a]) + 3 * p1[a]) - c[a];
            gt0[a] = T(2.0) * J0;
            Gs[a] = J0;  // parked for the cost below
        }
        if (Q.has_pt && p <= N - 4)
            part[1] = sum3((double)(Gs[0] * Gs[0]), (double)(Gs[1] * Gs[1]), (double)(Gs[2] * Gs[2]));
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const T gt1 = from_prev(gt0[a]);   // i = p-1
            const T gt2 = from_prev(gt1);      // i = p-2
            const T gt3 = from_prev(gt2);      // i = p-3
            T acc = gt3;                       // i=p-3: col(i+3) += gradTemp
            acc += T(-3.0) * gt2;              // i=p-2: col(i+2) += -3*gradTemp
            acc += T(3.0) * gt1;               // i=p-1: col(i+1) += 3*gradTemp
            acc += -gt0[a];                    // i=p  : col(i)   += -gradTemp
            Gs[a] = acc;
        }
    }

    // ---- feasibility, BT.cpp:952-999 (limits hard-coded to 1.0, :955-956) ----
    {
        const T ts = (T)K.ts_ctrl, tis = (T)K.ts_inv_sqr;
        auto excess = [](T v) -> T { return v > T(1.0) ? v - T(1.0) : (v < T(-1.0) ? v + T(1.0) : T(0.0)); };
        double cfv = 0.0, ea2[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const T evP = excess((p1[a] - c[a]) / ts);                   // velocity i = p
            const T eaP = excess(((p2[a] - 2 * p1[a]) + c[a]) * tis);    // acceleration i = p
            // gradient(j,i+1) += 2(v-vmax)/ts*tsInvSqr, gradient(j,i) += the negation (exactly)
            const T gv = (T(2) * evP) / ts * tis;
            // gradient(j,i) and (j,i+2) += 2(a-amax)*tsInvSqr, gradient(j,i+1) += -4(...) = -2x that (exactly)
            const T ga = (T(2) * eaP) * tis;
            const T gvM = from_prev(gv);       // velocity i = p-1
            const T gaM1 = from_prev(ga);      // acceleration i = p-1
            const T gaM2 = from_prev(gaM1);    // acceleration i = p-2
            T acc = gvM;                       // i=p-1: gradient(j,i+1)
            acc += -gv;                        // i=p  : gradient(j,i)
            acc += gaM2;                       // i=p-2: gradient(j,i+2)
            acc += -(T(2) * gaM1);             // i=p-1: gradient(j,i+1)
            acc += ga;                         // i=p  : gradient(j,i)
            Gf[a] = acc;
            cfv += (double)((evP * evP) * tis);
            ea2[a] = (double)(eaP * eaP);
        }
        // cost partial of lane p: velocity i=p (x,y,z) then acceleration i=p (x,y,z)
        double cf = 0.0;
        if (Q.has_pt && p <= N - 2) cf = cfv;
        if (Q.has_pt && p <= N - 3) { cf += ea2[0]; cf += ea2[1]; cf += ea2[2]; }
        part[2] = cf;
    }

    // ---- guide-point distance, BT.cpp:823-932 ----
    if (Q.interior) {
        double cd = 0.0;
        const int cnt = Q.g_end - Q.g_begin;
#pragma unroll
        for (int j = 0; j < kGuideRegs; ++j) {
            if (j < cnt)
                guide_pair_term<T>(K, c, Q.gq[j][0], Q.gq[j][1], Q.gq[j][2], Q.gq[j][3], Q.gq[j][4], Q.gq[j][5],
                                   Q.gqu[j], cd, Gd);
        }
        for (int j = Q.g_begin + kGuideRegs; j < Q.g_end; ++j) {
            const double* pv = Q.gpv + 6 * (size_t)j;
            guide_pair_term<T>(K, c, (T)pv[0], (T)pv[1], (T)pv[2], (T)pv[3], (T)pv[4], (T)pv[5],
                               Q.gunk ? (Q.gunk[j] != 0) : false, cd, Gd);
        }
        if (K.plan_in_z) {
            // BT.cpp:897-930, reproduced with its x-row gradient and heightDistMax band test
            const T hth = (T)K.hth, ha = (T)K.ha, hb = (T)K.hb, hc = (T)K.hc;
            const T hmin = c[2] - (T)K.min_h, hmax = c[2] - (T)K.max_h;
            if (hmin < T(0)) {
                const T e = hth - hmin;
                cd += (double)((ha * (e * e) + hb * e) + hc);
                Gd[0] += -((T(2) * ha) * e + hb) * T(-1.0);
            } else if (hmin >= T(0) && hmax < hth) {
                const T e = hth - hmin;
                cd += (double)((e * e) * e);
                Gd[0] += T(-3.0) * (e * e) * T(-1.0);
            }
            if (hmax > T(0)) {
                const T e = hth + hmax;
                cd += (double)((ha * (e * e) + hb * e) + hc);
                Gd[0] += -((T(2) * ha) * e + hb) * T(1.0);
            } else if (hmax <= T(0) && hmax >= -hth) {
                const T e = hth + hmax;
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
                    const T k = T(-3.0) * (e * e);
                    Go[0] += k * gx; Go[1] += k * gy; Go[2] += k * gz;
                } else if (e >= thr) {
                    co += (double)((oa * (e * e) + ob * e) + oc);
                    const T k = -((T(2) * oa) * e + ob);
                    Go[0] += k * gx; Go[1] += k * gy; Go[2] += k * gz;
                }
            }
        }
        part[3] = co;
    }

    const T w0 = (T)Q.w[0], w1 = (T)Q.w[1], w2 = (T)Q.w[2], w3 = (T)Q.w[3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
        g[a] = Q.interior ? (((w0 * Gd[a] + w1 * Gs[a]) + w2 * Gf[a]) + w3 * Go[a]) : T(0);
    part[4] = dot3(g, d);
    part[5] = Q.interior ? dot3(c, c) : 0.0;
    part[6] = dot3(g, g);
    group_sum<GROUP, 7>(part);
#pragma unroll
    for (int q = 0; q < 7; ++q) sums[q] = part[q];
    return ((Q.w[0] * part[0] + Q.w[1] * part[1]) + Q.w[2] * part[2]) + Q.w[3] * part[3];
}

template <typename T, int GROUP>
__device__ __forceinline__ void load_problem(const SolveArgs& A, const DevConst& K, int b, int p,
                                             LaneProblem<T>& Q) {
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
#pragma unroll
    for (int j = 0; j < kGuideRegs; ++j) {
        Q.gqu[j] = false;
#pragma unroll
        for (int q = 0; q < 6; ++q) Q.gq[j][q] = T(0);
        if (Q.g_begin + j < Q.g_end) {
            const double* pv = A.guide_pv + 6 * (size_t)(Q.g_begin + j);
#pragma unroll
            for (int q = 0; q < 6; ++q) Q.gq[j][q] = (T)pv[q];
            Q.gqu[j] = A.guide_unk ? (A.guide_unk[Q.g_begin + j] != 0) : false;
        }
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
    LaneProblem<T> Q;
    load_problem<T, GROUP>(A, K, b, p, Q);
    T c[3] = {0, 0, 0};
    if (Q.has_pt) {
        const double* src = A.ctrl + ((size_t)b * A.N + p) * 3;
        c[0] = (T)src[0]; c[1] = (T)src[1]; c[2] = (T)src[2];
    }
    T g[3];
    const T zero[3] = {0, 0, 0};
    double sums[7];
    const double f = eval_cost_grad<T, GROUP>(K, Q, c, zero, g, sums);
    if (Q.interior && A.out_grad) {
        double* dst = A.out_grad + ((size_t)b * (A.N - 6) + (p - 3)) * 3;
        dst[0] = (double)g[0]; dst[1] = (double)g[1]; dst[2] = (double)g[2];
    }
    if (p == 0) {
        if (A.out_cost) A.out_cost[b] = f;
        if (A.out_terms) {
#pragma unroll
            for (int q = 0; q < 4; ++q) A.out_terms[4 * (size_t)b + q] = sums[q];
        }
    }
}

// ---- More-Thuente helpers (per-lane scalar code, group-uniform values) -------------------

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

// LB:506-714 on scalars held in registers: (xt,xf,xd) best point, (yt,yf,yd) other end,
// (tt,tf,td) trial; tt receives the new trial step.
__device__ __forceinline__ int trial_interval(double& xt, double& xf, double& xd, double& yt, double& yf,
                                              double& yd, double& tt, const double tf, const double td,
                                              const double tmin, const double tmax, int& brackt) {
    int bound;
    const int dsign = td * (xd / fabs(xd)) < 0.;
    double mc, mq, newt;
    if (brackt) {
        const double lo = xt <= yt ? xt : yt;
        const double hi = xt >= yt ? xt : yt;
        if (tt <= lo || hi <= tt) return LBERR_OUTOFINTERVAL;
        if (0. <= xd * (tt - xt)) return LBERR_INCREASEGRADIENT;
        if (tmax < tmin) return LBERR_INCORRECT_TMINMAX;
    }
    if (xf < tf) {
        brackt = 1;
        bound = 1;
        mc = cubic_min(xt, xf, xd, tt, tf, td);
        mq = quad_min(xt, xf, xd, tt, tf);
        newt = (fabs(mc - xt) < fabs(mq - xt)) ? mc : mc + 0.5 * (mq - mc);
    } else if (dsign) {
        brackt = 1;
        bound = 0;
        mc = cubic_min(xt, xf, xd, tt, tf, td);
        mq = quad_min_secant(xt, xd, tt, td);
        newt = (fabs(mc - tt) > fabs(mq - tt)) ? mc : mq;
    } else if (fabs(td) < fabs(xd)) {
        bound = 1;
        mc = cubic_min_bounded(xt, xf, xd, tt, tf, td, tmin, tmax);
        mq = quad_min_secant(xt, xd, tt, td);
        if (brackt) newt = (fabs(tt - mc) < fabs(tt - mq)) ? mc : mq;
        else        newt = (fabs(tt - mc) > fabs(tt - mq)) ? mc : mq;
    } else {
        bound = 0;
        if (brackt)       newt = cubic_min(tt, tf, td, yt, yf, yd);
        else if (xt < tt) newt = tmax;
        else              newt = tmin;
    }
    {
        // LB:664-684 as value selects (pointer-style conditional copies end up in scratch)
        const bool higher = xf < tf;
        const bool y_from_x = !higher && dsign;
        const double nyt = higher ? tt : (y_from_x ? xt : yt);
        const double nyf = higher ? tf : (y_from_x ? xf : yf);
        const double nyd = higher ? td : (y_from_x ? xd : yd);
        const double nxt = higher ? xt : tt;
        const double nxf = higher ? xf : tf;
        const double nxd = higher ? xd : td;
        xt = nxt; xf = nxf; xd = nxd;
        yt = nyt; yf = nyf; yd = nyd;
    }
    if (tmax < newt) newt = tmax;
    if (newt < tmin) newt = tmin;
    if (brackt && bound) {
        mq = xt + 0.66 * (yt - xt);
        if (xt < yt) { if (mq < newt) newt = mq; }
        else         { if (newt < mq) newt = mq; }
    }
    tt = newt;
    return 0;
}

// ---- whole-solve kernel (vigo_optimize): BT.cpp:687-718 + LB:1024-1349 -------------------
// LDS: hist[(slot*2 + {0:s,1:y})*3 + axis][ROW] of T with ROW = TPB*(N-6) columns (each lane
//      reads and writes only its own column: LDS is a per-lane register extension here, no
//      cross-lane traffic and no barriers), then ys[slot][TPB] and alpha[age][TPB] doubles.
// Control flow: an outer trip per L-BFGS iteration (trip 0 = the initial evaluation) with ONE
// evaluation site inside the line-search loop, so the two groups of a wave re-converge at every
// iteration boundary and run the (dominant) two-loop recursion together.
template <typename T, int GROUP>
__global__ void __launch_bounds__(kWave, 1) k_optimize(SolveArgs A, DevConst K) {
    constexpr int TPB = kWave / GROUP;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int N = A.N, NI = N - 6;
    const int ROW = TPB * NI;
    const int m = K.mem_size;
    HPair<T>* hist = reinterpret_cast<HPair<T>*>(lds_raw);
    double* ys_tab = reinterpret_cast<double*>(lds_raw + (((size_t)m * ROW * sizeof(HPair<T>) + 15) & ~(size_t)15));

    const int lane = threadIdx.x;
    const int grp = lane / GROUP;
    const int p = lane % GROUP;
    const int b = blockIdx.x * TPB + grp;
    if (b >= A.B) return;

    LaneProblem<T> Q;
    load_problem<T, GROUP>(A, K, b, p, Q);
    // this lane's history column.  Lanes that own no free point (p < 3, p > N-4) read a
    // neighbour's column — finite data their zero d/g wipes out — and never write.
    const int pc = (p < 3) ? 3 : ((p > N - 4) ? N - 4 : p);
    HPair<T>* hl = hist + (grp * NI + (pc - 3));
    double* ys_l = ys_tab + grp;
    double* al_l = ys_tab + (size_t)m * TPB + grp;

    // x holds this lane's control point: a free variable on interior lanes, a fixed boundary
    // point elsewhere (its g, d, s, y are identically zero so it never moves).
    T x[3] = {0, 0, 0};
    if (Q.has_pt) {
        const double* src = A.ctrl + ((size_t)b * N + p) * 3;
        x[0] = (T)src[0]; x[1] = (T)src[1]; x[2] = (T)src[2];
    }
    T g[3] = {0, 0, 0}, xp[3] = {0, 0, 0}, gp[3] = {0, 0, 0}, d[3] = {0, 0, 0};
    double sums[7];
    int evals = 0;
    int ret = LBERR_UNKNOWN;
    int k = 0, end = 0;
    double fx = 0.0, step = 0.0;
    bool first = true;

    for (;;) {  // one trip per L-BFGS iteration; trip 0 only evaluates the start point (LB:1132)
        // ---------------- line_search_morethuente, LB:716-937 ----------------
        int ls = 0;
        int count = 0, brackt = 0, stage1 = 1, uinfo = 0;
        double dginit = 0.0, finit = 0.0, dgtest = 0.0, width = 0.0, prev_width = 0.0;
        double xt = 0., xf = 0., xd = 0., yt = 0., yf = 0., yd = 0.;
        const double stpmin = K.min_step, stpmax = K.max_step;
        bool run = true;
        if (!first) {
#pragma unroll
            for (int a = 0; a < 3; ++a) { xp[a] = x[a]; gp[a] = g[a]; }  // LB:1172-1173
            dginit = sums[4];  // g.d for the d just built: reduced below with the two-loop's last step
            if (step <= 0.) { ls = LBERR_INVALIDPARAMETERS; run = false; }
            else if (0 < dginit) { ls = LBERR_INCREASEGRADIENT; run = false; }
            finit = fx;
            dgtest = K.ftol * dginit;
            width = stpmax - stpmin;
            prev_width = 2.0 * width;
            xt = yt = 0.;
            xf = yf = finit;
            xd = yd = dginit;
        }
        while (run) {
            double stmin = 0., stmax = 0.;
            if (!first) {
                if (brackt) {
                    stmin = xt <= yt ? xt : yt;
                    stmax = xt >= yt ? xt : yt;
                } else {
                    stmin = xt;
                    stmax = step + 4.0 * (step - xt);
                }
                if (step < stpmin) step = stpmin;
                if (stpmax < step) step = stpmax;
                if ((brackt && ((step <= stmin || stmax <= step) || K.max_linesearch <= count + 1 || uinfo != 0)) ||
                    (brackt && (stmax - stmin <= K.xtol * stmax))) {
                    step = xt;
                }
                // x <- xp + step * d  (LB:824-825)
#pragma unroll
                for (int a = 0; a < 3; ++a) x[a] = xp[a] + (T)step * d[a];
            }

            fx = eval_cost_grad<T, GROUP>(K, Q, x, d, g, sums);  // the only evaluation site (LB:828, :1132)
            ++evals;
            if (first) break;

            const double dg = sums[4];
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

            // LB:883-920: the interval update runs on the modified function while stage1 holds and
            // the decrease is insufficient; one call site, operands selected here.
            const bool mod = stage1 && ftest1 < fx && fx <= xf;
            double axf = mod ? xf - xt * dgtest : xf, axd = mod ? xd - dgtest : xd;
            double ayf = mod ? yf - yt * dgtest : yf, ayd = mod ? yd - dgtest : yd;
            const double atf = mod ? fx - step * dgtest : fx, atd = mod ? dg - dgtest : dg;
            uinfo = trial_interval(xt, axf, axd, yt, ayf, ayd, step, atf, atd, stmin, stmax, brackt);
            xf = mod ? axf + xt * dgtest : axf;
            yf = mod ? ayf + yt * dgtest : ayf;
            xd = mod ? axd + dgtest : axd;
            yd = mod ? ayd + dgtest : ayd;

            if (brackt) {
                if (0.66 * prev_width <= fabs(yt - xt)) step = xt + 0.5 * (yt - xt);
                prev_width = width;
                width = fabs(yt - xt);
            }
        }

        double xnorm = sqrt(sums[5]), gnorm = sqrt(sums[6]);
        if (first) {
            first = false;
#pragma unroll
            for (int a = 0; a < 3; ++a) d[a] = -g[a];  // LB:1144
            if (xnorm < 1.0) xnorm = 1.0;
            if (gnorm / xnorm <= K.g_epsilon) { ret = LB_ALREADY_MINIMIZED; break; }  // LB:1154-1157
            // d = -g: d.d = g.g and g.d = -(g.g) exactly (negation commutes with every rounding)
            step = 1.0 / sqrt(sums[6]);  // LB:1163
            sums[4] = -sums[6];
            k = 1;
            end = 0;
            continue;
        }

        if (ls < 0) {
            // LB:1189-1197.  optData_.controlPoints keeps the last trial (BT.cpp:803): write it
            // out now, then revert x like the reference does.
            if (Q.has_pt) {
                double* dst = A.ctrl + ((size_t)b * N + p) * 3;
                dst[0] = (double)x[0]; dst[1] = (double)x[1]; dst[2] = (double)x[2];
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) { x[a] = xp[a]; g[a] = gp[a]; }
            ret = ls;
            break;
        }

        // convergence test, LB:1200-1225 (norms came with the last evaluation)
        if (xnorm < 1.0) xnorm = 1.0;
        if (gnorm / xnorm <= K.g_epsilon) { ret = LB_CONVERGENCE; break; }
        if (K.max_iterations != 0 && K.max_iterations < k + 1) { ret = LBERR_MAXIMUMITERATION; break; }

        // s, y, ys, yy — LB:1264-1276
        T sv[3], yv[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) { sv[a] = x[a] - xp[a]; yv[a] = g[a] - gp[a]; }
        if (Q.interior) {
            HPair<T> hp;
#pragma unroll
            for (int a = 0; a < 3; ++a) { hp.s[a] = sv[a]; hp.y[a] = yv[a]; }
            hl[end * ROW] = hp;
        }
        double ysyy[2] = {dot3(yv, sv), dot3(yv, yv)};
        group_sum<GROUP, 2>(ysyy);
        const double ys = ysyy[0], yy = ysyy[1];
        ys_l[end * TPB] = ys;

        // two-loop recursion, LB:1286-1316, fully unrolled over the pair's age with a register
        // window of kWin pairs (static index age % kWin): the pair needed kWin steps ahead is
        // fetched from LDS into the window slot the current step has just consumed, so the
        // dependent chain never waits for LDS and does no address arithmetic or copies.
        const int bound = (m <= k) ? m : k;
        ++k;
        const int newest = end;                        // slot of the pair just stored (age 0)
        end = (end + 1 == m) ? 0 : end + 1;
#pragma unroll
        for (int a = 0; a < 3; ++a) d[a] = -g[a];

#ifndef VIGO_TWOLOOP_WIN
#define VIGO_TWOLOOP_WIN 2
#endif
        constexpr int kWin = VIGO_TWOLOOP_WIN;
        T Ps[kWin][3], Py[kWin][3];
        double Pys[kWin];
        auto fetch = [&](int age, T (&s_)[3], T (&y_)[3], double& ys_) {
            int slot = newest - age;
            if (slot < 0) slot += m;
            const HPair<T> h = hl[slot * ROW];
#pragma unroll
            for (int a = 0; a < 3; ++a) { s_[a] = h.s[a]; y_[a] = h.y[a]; }
            ys_ = ys_l[slot * TPB];
        };
#pragma unroll
        for (int a = 0; a < 3; ++a) { Ps[0][a] = sv[a]; Py[0][a] = yv[a]; }
        Pys[0] = ys;
#pragma unroll
        for (int age = 1; age < kWin; ++age)
            if (age < bound) fetch(age, Ps[age], Py[age], Pys[age]);
#pragma unroll
        for (int age = 0; age < kMaxMem; ++age) {      // newest -> oldest, LB:1294-1303
            if (age < bound) {
                const int w = age % kWin;
                double al = group_sum1<GROUP>(dot3(Ps[w], d));
                al /= Pys[w];
                al_l[age * TPB] = al;      // alpha_j parks in LDS at a static offset
                const T na = Q.interior ? (T)(-al) : T(0);
#pragma unroll
                for (int a = 0; a < 3; ++a) d[a] += na * Py[w][a];
                if (age + kWin < kMaxMem && age + kWin < bound) fetch(age + kWin, Ps[w], Py[w], Pys[w]);
            }
        }
        {
            const T sc = (T)(ys / yy);  // LB:1305
#pragma unroll
            for (int a = 0; a < 3; ++a) d[a] *= sc;
        }
        // the window now holds the ages [max(0, bound - kWin), bound)
#pragma unroll
        for (int age = kMaxMem - 1; age >= 0; --age) {  // oldest -> newest, LB:1307-1316
            if (age < bound) {
                const int w = age % kWin;
                double beta = group_sum1<GROUP>(dot3(Py[w], d));
                beta /= Pys[w];
                const T co = Q.interior ? (T)(al_l[age * TPB] - beta) : T(0);
#pragma unroll
                for (int a = 0; a < 3; ++a) d[a] += co * Ps[w][a];
                if (age - kWin >= 0) fetch(age - kWin, Ps[w], Py[w], Pys[w]);
            }
        }
        sums[4] = group_sum1<GROUP>(dot3(g, d));  // dginit of the next line search (LB:746)
        step = 1.0;  // LB:1321
    }

    // results.  On success / convergence / iteration cap the last evaluated point is x itself.
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
    size_t h = (size_t)m * TPB * (N - 6) * sizeof(HPair<T>);
    h = (h + 15) & ~(size_t)15;
    return h + 2 * (size_t)m * TPB * sizeof(double);
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
