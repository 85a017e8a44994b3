// vigo_solver.hip — batched ViGO cost/gradient and the whole-solve L-BFGS kernel for gfx950.
//
// Mapping (MI355X-first, no MFMA: there is no dense contraction on this path):
//   * one 64-lane wavefront per workgroup; a trajectory owns a lane GROUP of 32 (N <= 32, two
//     trajectories per wave) or 64 lanes; a lane owns PPL consecutive control points (1 for
//     N <= 64, 2 for N <= 128, 4 beyond), so the 4-point jerk stencil and the 2/3-point vel/acc
//     stencils are register renames plus DPP wave shifts (no LDS).
//   * x, g, xp, gp, d live in VGPRs (3 scalars per owned point each); the L-BFGS history
//     (m x {s,y}) lives in LDS, one 48-byte record per free control point and slot: HBM sees the
//     initial control points and the result only.
//   * every scalar of the More-Thuente search is replicated across the group's lanes; the two
//     groups of a wave diverge freely (exec masking), all cross-lane traffic stays inside a group.
//   * a trajectory whose control points all lie at one height (and no z planning) cannot move in z by more than
//     rounding noise: the LEVEL RULE (include/vigo.h) holds its z fixed, and waves of such trajectories run the
//     D = 2 instantiation of the solve kernel, which carries x and y only — 2/3 of the history, dots and stencils.
//   * per-trajectory sums (cost terms, dot products) are: a per-point partial, the lane's
//     points added in index order, then a butterfly all-reduce inside the group,
//     v += lane[i ^ m], m = 1,2,..,GROUP/2 — a fixed tree, so results are deterministic and
//     reproducible bit-for-bit by oracle/vigo_oracle.c's emulation mode.
//
// Arithmetic follows the reference expression by expression (bsplineTraj.cpp:802-1064 and
// solver/lbfgs.hpp:295-1349; see the citations on each block); the file is built with
// -ffp-contract=off so no FMA contraction changes a rounding.  Differences to the CPU
// reference are confined to: summation order of the reductions above, x*x / x*x*x instead
// of glibc pow(x,2|3), sqrt instead of pow(.,0.5).
#include <stdlib.h>

#include <type_traits>

#include "vigo_internal.hpp"

#ifndef VIGO_DOUBLE_ADD
#define VIGO_DOUBLE_ADD 1
#endif

namespace vigo {
namespace {

constexpr int kWave = 64;
constexpr int kMaxMem = VIGO_MAX_MEM_SIZE;
// Dynamic-obstacle table of the solve kernel: per trajectory, {predicted x, predicted y, threshold}
// of up to kObsTabEntries (obstacle, predicted step) pairs plus the obstacles' sizes, staged in LDS
// once per solve (they do not depend on the control points).  Sized so that the N = 32 and N = 64
// shapes keep four waves per CU (<= 40 KiB per wave, history slots with their zero column included):
// 8 obstacles x 11 predicted steps for two trajectories per wave, 3 x 11 for one.
constexpr int kObsTabObs = 16;
template <int GROUP> constexpr int kObsTabEntries = GROUP == 32 ? 88 : 33;
template <int GROUP> constexpr int kObsTabDoubles = 3 * kObsTabEntries<GROUP> + kObsTabObs;

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

// Shift of a per-lane run of PPL consecutive control-point values by one point along the
// trajectory: o[q] = value of point (own + q + 1) resp. (own + q - 1).  Inside the lane that is
// a register rename, across lanes one DPP shift.
template <typename T, int PPL>
__device__ __forceinline__ void seq_next(const T (&a)[PPL], T (&o)[PPL]) {
    const T edge = from_next(a[0]);
#pragma unroll
    for (int q = 0; q + 1 < PPL; ++q) o[q] = a[q + 1];
    o[PPL - 1] = edge;
}
template <typename T, int PPL>
__device__ __forceinline__ void seq_prev(const T (&a)[PPL], T (&o)[PPL]) {
    const T edge = from_prev(a[PPL - 1]);
#pragma unroll
    for (int q = PPL - 1; q > 0; --q) o[q] = a[q - 1];
    o[0] = edge;
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
// The xor-16 level needs the value twice (v_permlane16_swap exchanges rows between TWO registers and clobbers
// both).  Instead of copying the result of the row_mirror level (two v_mov_b32), that level's add is issued twice:
// one VALU slot instead of two.  The second add is an asm statement so that it is not merged with the first (the
// compiler still sees its register definition and keeps the VALU-write -> permlane-read wait states after it).
__device__ __forceinline__ double add_f64_again(double a, double b) {
    double r;
    asm volatile("v_add_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (a == b bitwise, in two registers; the two halves of the result are returned so that a 64-lane group can issue the
// final add twice as well, for its xor-32 level)
__device__ __forceinline__ void xor16_parts(double a, double b, double& p0, double& p1) {
    const int alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    auto l = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    auto h = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    p0 = __hiloint2double(h[0], l[0]);
    p1 = __hiloint2double(h[1], l[1]);
}
__device__ __forceinline__ double xor32_sum2(double a, double b) {
    const int alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    auto l = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    auto h = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    return __hiloint2double(h[0], l[0]) + __hiloint2double(h[1], l[1]);
}
template <int GROUP, int K>
__device__ __forceinline__ void group_sum(double (&v)[K]) {
    static_assert(GROUP == 16 || GROUP == 32 || GROUP == 64, "group is a DPP row, half a wave or a wave");
#pragma unroll
    for (int q = 0; q < K; ++q) v[q] += dpp_f64<0xB1>(v[q]);   // quad_perm:[1,0,3,2]
#pragma unroll
    for (int q = 0; q < K; ++q) v[q] += dpp_f64<0x4E>(v[q]);   // quad_perm:[2,3,0,1]
#pragma unroll
    for (int q = 0; q < K; ++q) v[q] += dpp_f64<0x141>(v[q]);  // row_half_mirror
    if (GROUP >= 32 && VIGO_DOUBLE_ADD) {
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const double t = dpp_f64<0x140>(v[q]);             // row_mirror
            const double a = v[q] + t;
            const double b = add_f64_again(v[q], t);
            double p0, p1;
            xor16_parts(a, b, p0, p1);
            if (GROUP == 64) v[q] = xor32_sum2(p0 + p1, add_f64_again(p0, p1));
            else v[q] = p0 + p1;
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < K; ++q) v[q] += dpp_f64<0x140>(v[q]);  // row_mirror
    if (GROUP >= 32) {
#pragma unroll
        for (int q = 0; q < K; ++q) v[q] = xor16_sum(v[q]);
    }
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

// min and max of a value over the lanes of a group (the same xor butterfly; order is irrelevant for min / max)
template <int GROUP>
__device__ __forceinline__ void group_minmax(double& mn, double& mx) {
    static_assert(GROUP == 32 || GROUP == 64, "half a wave or a wave");
    auto step = [&](double omn, double omx) { mn = fmin(mn, omn); mx = fmax(mx, omx); };
    step(dpp_f64<0xB1>(mn), dpp_f64<0xB1>(mx));
    step(dpp_f64<0x4E>(mn), dpp_f64<0x4E>(mx));
    step(dpp_f64<0x141>(mn), dpp_f64<0x141>(mx));
    step(dpp_f64<0x140>(mn), dpp_f64<0x140>(mx));
    {
        double a0, a1, b0, b1;
        xor16_parts(mn, mn, a0, a1);
        xor16_parts(mx, mx, b0, b1);
        mn = fmin(a0, a1);
        mx = fmax(b0, b1);
    }
    if (GROUP == 64) {
        const int nlo = __double2loint(mn), nhi = __double2hiint(mn), xlo = __double2loint(mx), xhi = __double2hiint(mx);
        auto l = __builtin_amdgcn_permlane32_swap(nlo, nlo, false, false);
        auto h = __builtin_amdgcn_permlane32_swap(nhi, nhi, false, false);
        mn = fmin(__hiloint2double(h[0], l[0]), __hiloint2double(h[1], l[1]));
        auto l2 = __builtin_amdgcn_permlane32_swap(xlo, xlo, false, false);
        auto h2 = __builtin_amdgcn_permlane32_swap(xhi, xhi, false, false);
        mx = fmax(__hiloint2double(h2[0], l2[0]), __hiloint2double(h2[1], l2[1]));
    }
}

__device__ __forceinline__ double sum3(double a, double b, double c) { return (a + b) + c; }

// FAST (VIGO_PREC_F64_FAST): explicit fused multiply-adds in the dot products, axpys, stencils
// and the weight combination — what GCC's default -ffp-contract=fast does to the reference on
// its ARM targets — plus one reciprocal per history pair instead of a division per two-loop
// step.  Every fma below is mirrored by oracle/vigo_oracle.c's fast emulation, bit for bit.
__device__ __forceinline__ double fmaT(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fmaT(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// D = the number of leading coordinates that MOVE: 3, or 2 for a level trajectory (see k_optimize): the z components of
// g, d, s, y are then exactly zero, their products contribute +-0 to every sum and are left out
template <bool FAST, typename T, int D = 3>
__device__ __forceinline__ double dot3(const T (&a)[3], const T (&b)[3]) {
    if (D == 2) {
        if (FAST) return fmaT((double)a[1], (double)b[1], (double)a[0] * (double)b[0]);
        return (double)a[0] * (double)b[0] + (double)a[1] * (double)b[1];
    }
    if (FAST)
        return fmaT((double)a[2], (double)b[2], fmaT((double)a[1], (double)b[1], (double)a[0] * (double)b[0]));
    return sum3((double)a[0] * (double)b[0], (double)a[1] * (double)b[1], (double)a[2] * (double)b[2]);
}
// lane partial of a dot product: the lane's first point, then its other points in index order
template <bool FAST, typename T, int PPL, int D = 3>
__device__ __forceinline__ double dot_lane(const T (&a)[PPL][3], const T (&b)[PPL][3]) {
    double s = dot3<FAST, T, D>(a[0], b[0]);
#pragma unroll
    for (int q = 1; q < PPL; ++q) s += dot3<FAST, T, D>(a[q], b[q]);
    return s;
}

// ys of a history pair as the two-loop uses it (LB:1300, :1312 divide by it).  FAST keeps only the
// reciprocal in v.  The reference-order mode keeps ys in v and its correctly rounded reciprocal in
// r (YSv<false>): the steady-state two-loop divides with Markstein's sequence q = a*r, e = fma(-q, ys, a),
// q' = fma(e, r, q) — the correctly rounded a/ys, the same bits as the 13-instruction fp64
// division, whenever r = RN(1/ys) and nothing over/underflows (Markstein 1990; checked against
// 4e8 random divisions incl. all-ones significands).  r is NaN for a ys outside 2^+-500, the
// dividends' magnitudes are tracked with one max and one min per step, and a two-loop that saw a
// dividend outside 2^+-500 (or produced a NaN) is repeated with true divisions.
template <bool FAST>
struct YSv {            // reference-order mode
    double v, r;
};
template <>
struct YSv<true> {      // FAST: the reciprocal only
    double v;
};
__device__ __forceinline__ bool exp_in_safe_range(double a) {
    const unsigned e = ((unsigned)__double2hiint(a) >> 20) & 0x7ffu;
    return (e - 523u) < 1000u;
}
__device__ __forceinline__ YSv<true> make_ys(double ys, YSv<true>*) { return YSv<true>{1.0 / ys}; }
__device__ __forceinline__ YSv<false> make_ys(double ys, YSv<false>*) {
    return YSv<false>{ys, exp_in_safe_range(ys) ? 1.0 / ys : __builtin_nan("")};
}
template <bool MARK>
__device__ __forceinline__ double over_ys(double a, const YSv<true>& y, double&, double&) { return a * y.v; }
template <bool MARK>
__device__ __forceinline__ double over_ys(double a, const YSv<false>& y, double& amin, double& amax) {
    if (!MARK) return a / y.v;
    const double q = a * y.r;
    const double e = __builtin_fma(-q, y.v, a);
    amin = fmin(amin, fabs(a));
    amax = fmax(amax, fabs(a));
    return __builtin_fma(e, y.r, q);
}

// one L-BFGS history pair of one control point as it sits in LDS (48 B in fp64: three
// conflict-free ds_read_b128 per lane, one address register)
template <typename T, int D = 3>
struct alignas(16) HPair {
    T s[D];
    T y[D];
};

// per-lane view of one trajectory's inputs
template <typename T, int PPL>
struct LaneProblem {
    // guide pairs per control point kept in VGPRs (more: re-read from HBM/L2 every evaluation)
    static constexpr int kGuideRegs = (PPL == 1) ? 2 : 0;
    static constexpr int kGuideDim = kGuideRegs > 0 ? kGuideRegs : 1;
    int N;
    // LEVEL RULE (include/vigo.h): plan_in_z off and all N control points at one height to 2^-40 relative — the z
    // coordinate then feels only the smoothness and feasibility terms of values that differ by rounding noise; those
    // z terms are taken as exactly zero (cost and gradient), so z never moves.  Group-uniform; set by set_level().
    bool level;
    int p0;                   // first control point of this lane
    bool has_pt[PPL];         // p0 + q < N
    bool interior[PPL];       // 3 <= p0 + q <= N-4 (a free control point)
    int g_begin[PPL], g_end[PPL];
    const double* gpv;
    const uint8_t* gunk;
    T gq[kGuideDim][6];       // (PPL == 1) the first pairs, loaded once per solve: they never move
    bool gqu[kGuideDim];
    int o_begin, o_end;       // this trajectory's obstacles
    const double* obs;
    // (whole-solve kernel) the first o_tab obstacles of this trajectory as a table in LDS:
    // obs_tab[3 * (j * o_steps + s)] = {x, y, threshold} at predicted step n = 2 s, then the sizes
    // at obs_tab[3 * entries + j].  Staged once per solve: nothing in it depends on the control
    // points, and a global load per obstacle per evaluation is a ~2 us round trip on the critical path
    const double* obs_tab;
    const double* obs_size;
    int o_tab, o_steps;
    double w[4];
};

// One guide pair's contribution, BT.cpp:839-895.  e == dthresh takes the cubic branch (first
// else-if wins); the "too far" branch is never scaled by the unknown factor.
template <bool FAST, typename T>
__device__ __forceinline__ void guide_pair_term(const DevConst& K, const T (&c)[3], T px, T py, T pz, T vx,
                                                T vy, T vz, bool unk, double& cd, T (&Gd)[3]) {
    const T dth = (T)K.dth, da = (T)K.da, db = (T)K.db, dcc = (T)K.dc, uf = (T)K.unc_factor;
    const T dist = FAST ? fmaT(c[2] - pz, vz, fmaT(c[1] - py, vy, (c[0] - px) * vx))
                        : ((c[0] - px) * vx + (c[1] - py) * vy) + (c[2] - pz) * vz;
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

// One dynamic obstacle's contribution to one control point, BT.cpp:1011-1059: the obstacle at its
// predicted positions n = 0, 2, .., predictionNum (skipFactor = 2, BT.cpp:1006).
template <typename T>
__device__ __forceinline__ void obstacle_term(const DevConst& K, const T (&c)[3], T opx, T opy, T ovx, T ovy, T size,
                                              double& co, T (&Go)[3]) {
    const T thr0 = (T)K.thr_dyn, oa = (T)K.oa, ob = (T)K.ob, oc = (T)K.oc;
    for (int n = 0; n <= K.pred_num; n += 2) {
        const T tn = (T)((double)n * K.ts);
        const T px = opx + tn * ovx, py = opy + tn * ovy;
        // integer division n/predictionNum, BT.cpp:1020
        const T thr = (T(1) - (T)(n / K.pred_num) * T(0.2)) * thr0;
        const T dx = c[0] - px, dy = c[1] - py, dz = T(0.0);
        const T nrm = sqrt((dx * dx + dy * dy) + dz * dz);
        const T e = thr - (nrm - size);
        // BT.cpp:1030-1058: e <= 0 no punishment; 0 < e <= thr cubic; e >= thr quadratic.  The two
        // penalty branches share their tail — grad = diff / |diff| (BT.cpp:1025) and the
        // accumulation — so a wave whose lanes split between them pays the three fp64 divisions
        // once, and a far step pays none.
        if (e > T(0)) {
            const bool cubic = e <= thr;
            const T ct = cubic ? (e * e) * e : (oa * (e * e) + ob * e) + oc;
            const T k = cubic ? T(-3.0) * (e * e) : -((T(2) * oa) * e + ob);
            co += (double)ct;
            Go[0] += k * (dx / nrm); Go[1] += k * (dy / nrm);
            // diff.z = 0 (BT.cpp:1022): for a finite positive |diff| the z term is k * (+0) = +-0 and
            // leaves the +0 (or NaN) accumulator as it is, so its fp64 division is only issued in
            // the degenerate cases (control point exactly on the predicted centre, NaN/inf input)
            if (__builtin_expect(!(nrm > T(0) && nrm < (T)INFINITY), 0)) Go[2] += k * (dz / nrm);
        }
    }
}

// The same term with the per-step operands {px, py, thr} read from the solve kernel's LDS table
// (they were computed there by the expressions above, so the bits are the same).
template <typename T>
__device__ __forceinline__ void obstacle_term_tab(const DevConst& K, const T (&c)[3], const double* tab, int steps,
                                                  T size, double& co, T (&Go)[3]) {
    const T oa = (T)K.oa, ob = (T)K.ob, oc = (T)K.oc;
    for (int s = 0; s < steps; ++s) {
        const T px = (T)tab[3 * s + 0], py = (T)tab[3 * s + 1], thr = (T)tab[3 * s + 2];
        const T dx = c[0] - px, dy = c[1] - py, dz = T(0.0);
        const T nrm = sqrt((dx * dx + dy * dy) + dz * dz);
        const T e = thr - (nrm - size);
        if (e > T(0)) {
            const bool cubic = e <= thr;
            const T ct = cubic ? (e * e) * e : (oa * (e * e) + ob * e) + oc;
            const T k = cubic ? T(-3.0) * (e * e) : -((T(2) * oa) * e + ob);
            co += (double)ct;
            Go[0] += k * (dx / nrm); Go[1] += k * (dy / nrm);
            if (__builtin_expect(!(nrm > T(0) && nrm < (T)INFINITY), 0)) Go[2] += k * (dz / nrm);
        }
    }
}

// ---- cost + gradient at the points held in c (BT.cpp:802-821) ---------------------------
// T is the element type of points/gradients; sums are fp64.  g receives the weighted gradient
// of the free points, 0 elsewhere.  ONE 7-value group reduction returns
//   sums[0..3] = un-weighted distance / smoothness / feasibility / dynamic costs,
//   sums[4] = g.d, sums[5] = x.x over the free points, sums[6] = g.g
// (the line search needs g.d after every evaluation, LB:829, and the norms after the last one,
// LB:1200-1201; fusing them costs three more DPP chains instead of two more reductions).
// Returns the weighted total cost (group-uniform).
template <typename T, int GROUP, int PPL, bool FAST, bool OBS = true, int D = 3>
__device__ __forceinline__ double eval_cost_grad(const DevConst& K, const LaneProblem<T, PPL>& Q,
                                                 const T (&c)[PPL][3], const T (&d)[PPL][3], T (&g)[PPL][3],
                                                 double (&sums)[7]) {
    using LP = LaneProblem<T, PPL>;
    const int N = Q.N;
    T Gd[PPL][3], Gs[PPL][3], Gf[PPL][3], Go[PPL][3];
    double pt_s[PPL], pt_f[PPL], pt_d[PPL], pt_o[PPL];  // per-point cost partials
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
        pt_s[q] = pt_f[q] = pt_d[q] = pt_o[q] = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) Gd[q][a] = Go[q][a] = T(0);
    }

    // Every point evaluates the stencil term whose FIRST point it is (index i = its own): jerk_i,
    // velocity_i, acceleration_i and their gradient magnitudes.  A control point's gradient is the
    // reference's scatter-add seen from the receiving column: the terms of i-3..i, which are the
    // SAME expressions evaluated at the points below, fetched with register renames / DPP shifts —
    // identical bits, no recomputation (and half the fp64 divisions by ts).
    {
        const T ts = (T)K.ts_ctrl, tis = (T)K.ts_inv_sqr;
        auto excess = [](T v) -> T { return v > T(1.0) ? v - T(1.0) : (v < T(-1.0) ? v + T(1.0) : T(0.0)); };
        double jj[PPL][3], vv[PPL][3], aa[PPL][3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (a >= D) {   // a level trajectory: these stencil terms and their gradients are exactly zero (checked at entry)
#pragma unroll
                for (int q = 0; q < PPL; ++q) { jj[q][a] = vv[q][a] = aa[q][a] = 0.0; Gs[q][a] = Gf[q][a] = T(0); }
                continue;
            }
            T C[PPL], P1[PPL], P2[PPL], P3[PPL];
#pragma unroll
            for (int q = 0; q < PPL; ++q) C[q] = c[q][a];
            seq_next<T, PPL>(C, P1);
            seq_next<T, PPL>(P1, P2);
            seq_next<T, PPL>(P2, P3);
            T gt0[PPL], gv[PPL], ga[PPL];
#pragma unroll
            for (int q = 0; q < PPL; ++q) {
                // smoothness, BT.cpp:934-950
                T J0 = FAST ? fmaT(T(3), P1[q], fmaT(T(-3), P2[q], P3[q])) - C[q]
                            : ((P3[q] - 3 * P2[q]) + 3 * P1[q]) - C[q];
                if (a == 2 && Q.level) J0 = T(0);                              // level rule: no z terms
                gt0[q] = T(2.0) * J0;                                          // gradTemp
                jj[q][a] = (double)(J0 * J0);
                // feasibility, BT.cpp:952-999 (limits hard-coded to 1.0, :955-956)
                T evP = excess((P1[q] - C[q]) / ts);                           // velocity i
                T eaP = excess((FAST ? fmaT(T(-2), P1[q], P2[q]) + C[q]
                                     : (P2[q] - 2 * P1[q]) + C[q]) * tis);     // acceleration i
                if (a == 2 && Q.level) { evP = T(0); eaP = T(0); }
                // gradient(j,i+1) += 2(v-vmax)/ts*tsInvSqr, gradient(j,i) += the negation (exactly)
                gv[q] = (T(2) * evP) / ts * tis;
                // gradient(j,i), (j,i+2) += 2(a-amax)*tsInvSqr; gradient(j,i+1) += -4(..) = -2x that (exactly)
                ga[q] = (T(2) * eaP) * tis;
                vv[q][a] = (double)((evP * evP) * tis);
                aa[q][a] = (double)(eaP * eaP);
            }
            T gt1[PPL], gt2[PPL], gt3[PPL], gvM[PPL], gaM1[PPL], gaM2[PPL];
            seq_prev<T, PPL>(gt0, gt1);
            seq_prev<T, PPL>(gt1, gt2);
            seq_prev<T, PPL>(gt2, gt3);
            seq_prev<T, PPL>(gv, gvM);
            seq_prev<T, PPL>(ga, gaM1);
            seq_prev<T, PPL>(gaM1, gaM2);
#pragma unroll
            for (int q = 0; q < PPL; ++q) {
                T acc = gt3[q];                    // i-3: col(i+3) += gradTemp
                if (FAST) {
                    acc = fmaT(T(-3.0), gt2[q], acc);
                    acc = fmaT(T(3.0), gt1[q], acc);
                } else {
                    acc += T(-3.0) * gt2[q];       // i-2: col(i+2) += -3*gradTemp
                    acc += T(3.0) * gt1[q];        // i-1: col(i+1) += 3*gradTemp
                }
                acc += -gt0[q];                    // i  : col(i)   += -gradTemp
                Gs[q][a] = acc;
                T fcc = gvM[q];                    // i-1: gradient(j,i+1)
                fcc += -gv[q];                     // i  : gradient(j,i)
                fcc += gaM2[q];                    // i-2: gradient(j,i+2)
                fcc += -(T(2) * gaM1[q]);          // i-1: gradient(j,i+1)
                fcc += ga[q];                      // i  : gradient(j,i)
                Gf[q][a] = fcc;
            }
        }
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            const int p = Q.p0 + q;
            if (Q.has_pt[q] && p <= N - 4) pt_s[q] = sum3(jj[q][0], jj[q][1], jj[q][2]);
            // velocity i (x,y,z) then acceleration i (x,y,z)
            double cf = 0.0;
            if (Q.has_pt[q] && p <= N - 2) cf = (vv[q][0] + vv[q][1]) + vv[q][2];
            if (Q.has_pt[q] && p <= N - 3) { cf += aa[q][0]; cf += aa[q][1]; cf += aa[q][2]; }
            pt_f[q] = cf;
        }
    }

    // ---- guide-point distance, BT.cpp:823-932 ----
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
        if (Q.interior[q]) {
            double cd = 0.0;
            const int cnt = Q.g_end[q] - Q.g_begin[q];
#pragma unroll
            for (int j = 0; j < LP::kGuideRegs; ++j) {
                if (j < cnt)
                    guide_pair_term<FAST, T>(K, c[q], Q.gq[j][0], Q.gq[j][1], Q.gq[j][2], Q.gq[j][3], Q.gq[j][4], Q.gq[j][5],
                                       Q.gqu[j], cd, Gd[q]);
            }
            for (int j = Q.g_begin[q] + LP::kGuideRegs; j < Q.g_end[q]; ++j) {
                const double* pv = Q.gpv + 6 * (size_t)j;
                guide_pair_term<FAST, T>(K, c[q], (T)pv[0], (T)pv[1], (T)pv[2], (T)pv[3], (T)pv[4], (T)pv[5],
                                   Q.gunk ? (Q.gunk[j] != 0) : false, cd, Gd[q]);
            }
            if (K.plan_in_z) {
                // BT.cpp:897-930, reproduced with its x-row gradient and heightDistMax band test
                const T hth = (T)K.hth, ha = (T)K.ha, hb = (T)K.hb, hc = (T)K.hc;
                const T hmin = c[q][2] - (T)K.min_h, hmax = c[q][2] - (T)K.max_h;
                if (hmin < T(0)) {
                    const T e = hth - hmin;
                    cd += (double)((ha * (e * e) + hb * e) + hc);
                    Gd[q][0] += -((T(2) * ha) * e + hb) * T(-1.0);
                } else if (hmin >= T(0) && hmax < hth) {
                    const T e = hth - hmin;
                    cd += (double)((e * e) * e);
                    Gd[q][0] += T(-3.0) * (e * e) * T(-1.0);
                }
                if (hmax > T(0)) {
                    const T e = hth + hmax;
                    cd += (double)((ha * (e * e) + hb * e) + hc);
                    Gd[q][0] += -((T(2) * ha) * e + hb) * T(1.0);
                } else if (hmax <= T(0) && hmax >= -hth) {
                    const T e = hth + hmax;
                    cd += (double)((e * e) * e);
                    Gd[q][0] += T(-3.0) * (e * e) * T(1.0);
                }
            }
            pt_d[q] = cd;
        }
    }

    // ---- dynamic obstacles, BT.cpp:1001-1064 (OBS == false: an instantiation for calls without an obstacle list) ----
    if (OBS && Q.o_end > Q.o_begin) {
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            if (!Q.interior[q]) continue;
            double co = 0.0;
            // the obstacles staged in LDS by the solve kernel first (same order as the list) ...
            for (int j = 0; j < Q.o_tab; ++j)
                obstacle_term_tab<T>(K, c[q], Q.obs_tab + 3 * j * Q.o_steps, Q.o_steps, (T)Q.obs_size[j], co, Go[q]);
            // ... then the rest (all of them for the standalone cost/gradient kernel) from HBM/L2
            for (int j = Q.o_begin + Q.o_tab; j < Q.o_end; ++j) {
                const double* o = Q.obs + 9 * (size_t)j;
                const T hx = (T)o[6] / 2, hy = (T)o[7] / 2;
                obstacle_term<T>(K, c[q], (T)o[0], (T)o[1], (T)o[3], (T)o[4], sqrt(hx * hx + hy * hy), co, Go[q]);
            }
            pt_o[q] = co;
        }
    }

    const T w0 = (T)Q.w[0], w1 = (T)Q.w[1], w2 = (T)Q.w[2], w3 = (T)Q.w[3];
    double part[7];  // lane partials: the lane's first point, then its other points in index order
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
            g[q][a] = (!Q.interior[q] || a >= D || (a == 2 && Q.level)) ? T(0)
                      : (FAST ? fmaT(w3, Go[q][a], fmaT(w2, Gf[q][a], fmaT(w1, Gs[q][a], w0 * Gd[q][a])))
                              : ((w0 * Gd[q][a] + w1 * Gs[q][a]) + w2 * Gf[q][a]) + w3 * Go[q][a]);
        const double v4 = dot3<FAST, T, D>(g[q], d[q]);
        const double v5 = Q.interior[q] ? dot3<FAST, T>(c[q], c[q]) : 0.0;     // (x.x: the level coordinate counts)
        const double v6 = dot3<FAST, T, D>(g[q], g[q]);
        if (q == 0) {
            part[0] = pt_d[0]; part[1] = pt_s[0]; part[2] = pt_f[0]; part[3] = pt_o[0];
            part[4] = v4; part[5] = v5; part[6] = v6;
        } else {
            part[0] += pt_d[q]; part[1] += pt_s[q]; part[2] += pt_f[q]; part[3] += pt_o[q];
            part[4] += v4; part[5] += v5; part[6] += v6;
        }
    }
    if (OBS) {
        group_sum<GROUP, 7>(part);
    } else {
        // no obstacles: the dynamic cost is a sum of zeros — 0.0, as the reduction would return — so it stays out of it
        double p6[6] = {part[0], part[1], part[2], part[4], part[5], part[6]};
        group_sum<GROUP, 6>(p6);
        part[0] = p6[0]; part[1] = p6[1]; part[2] = p6[2]; part[3] = 0.0; part[4] = p6[3]; part[5] = p6[4]; part[6] = p6[5];
    }
#pragma unroll
    for (int q = 0; q < 7; ++q) sums[q] = part[q];
    return ((Q.w[0] * part[0] + Q.w[1] * part[1]) + Q.w[2] * part[2]) + Q.w[3] * part[3];
}

template <typename T, int GROUP, int PPL>
__device__ __forceinline__ void load_problem(const SolveArgs& A, const DevConst& K, int b, int lane_in_group,
                                             LaneProblem<T, PPL>& Q) {
    using LP = LaneProblem<T, PPL>;
    const int N = A.N;
    Q.N = N;
    Q.level = false;
    Q.p0 = lane_in_group * PPL;
    Q.gpv = A.guide_pv;
    Q.gunk = A.guide_unk;
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
        const int p = Q.p0 + q;
        Q.has_pt[q] = p < N;
        Q.interior[q] = (p >= 3) && (p <= N - 4);
        Q.g_begin[q] = Q.g_end[q] = 0;
        if (Q.interior[q] && A.guide_off && A.guide_pv) {   // offsets without pairs: no guides
            Q.g_begin[q] = A.guide_off[(size_t)b * N + p];
            Q.g_end[q] = A.guide_off[(size_t)b * N + p + 1];
        }
    }
#pragma unroll
    for (int j = 0; j < LP::kGuideDim; ++j) {
        Q.gqu[j] = false;
#pragma unroll
        for (int q = 0; q < 6; ++q) Q.gq[j][q] = T(0);
        if (j < LP::kGuideRegs && Q.g_begin[0] + j < Q.g_end[0]) {
            const double* pv = A.guide_pv + 6 * (size_t)(Q.g_begin[0] + j);
#pragma unroll
            for (int q = 0; q < 6; ++q) Q.gq[j][q] = (T)pv[q];
            Q.gqu[j] = A.guide_unk ? (A.guide_unk[Q.g_begin[0] + j] != 0) : false;
        }
    }
    Q.obs = A.obs;
    Q.obs_tab = Q.obs_size = nullptr;
    Q.o_tab = Q.o_steps = 0;
    if (!A.obs) {                    // offsets (or a shared count) without the list: no obstacles
        Q.o_begin = Q.o_end = 0;
    } else if (A.obs_off) {
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

template <typename T, int PPL>
__device__ __forceinline__ void load_points(const SolveArgs& A, int b, const LaneProblem<T, PPL>& Q, T (&x)[PPL][3]) {
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
        x[q][0] = x[q][1] = x[q][2] = T(0);
        if (Q.has_pt[q]) {
            const double* src = A.ctrl + ((size_t)b * A.N + Q.p0 + q) * 3;
            x[q][0] = (T)src[0]; x[q][1] = (T)src[1]; x[q][2] = (T)src[2];
        }
    }
}

// the level rule's test on the control points a group holds (every lane of the group gets the same answer)
template <typename T, int GROUP, int PPL>
__device__ __forceinline__ void set_level(const DevConst& K, LaneProblem<T, PPL>& Q, const T (&x)[PPL][3]) {
    double mn = INFINITY, mx = -INFINITY;
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
        if (Q.has_pt[q]) { mn = fmin(mn, (double)x[q][2]); mx = fmax(mx, (double)x[q][2]); }
    }
    group_minmax<GROUP>(mn, mx);
    Q.level = !K.plan_in_z && !K.strict_z && (mx - mn) <= 0x1p-40 * fmax(1.0, fmax(fabs(mn), fabs(mx)));
}

template <typename T, int PPL>
__device__ __forceinline__ void store_points(const SolveArgs& A, int b, const LaneProblem<T, PPL>& Q, const T (&x)[PPL][3]) {
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
        if (Q.has_pt[q]) {
            double* dst = A.ctrl + ((size_t)b * A.N + Q.p0 + q) * 3;
            dst[0] = (double)x[q][0]; dst[1] = (double)x[q][1]; dst[2] = (double)x[q][2];
        }
    }
}

// ---- standalone cost/gradient kernel (vigo_cost_grad) ----------------------------------
template <typename T, int GROUP, int PPL, bool FAST>
__global__ void __launch_bounds__(kWave) k_cost_grad(SolveArgs A, const DevConst* __restrict__ Kp) {
    const DevConst& K = *Kp;  // uniform address: scalar loads at the use sites, not 100+ live SGPRs
    constexpr int TPB = kWave / GROUP;
    const int lane = threadIdx.x;
    const int b = blockIdx.x * TPB + lane / GROUP;
    if (b >= A.B) return;
    LaneProblem<T, PPL> Q;
    load_problem<T, GROUP, PPL>(A, K, b, lane % GROUP, Q);
    T c[PPL][3], g[PPL][3], zero[PPL][3];
    load_points<T, PPL>(A, b, Q, c);
    set_level<T, GROUP, PPL>(K, Q, c);
#pragma unroll
    for (int q = 0; q < PPL; ++q) zero[q][0] = zero[q][1] = zero[q][2] = T(0);
    double sums[7];
    const double f = eval_cost_grad<T, GROUP, PPL, FAST>(K, Q, c, zero, g, sums);
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
        if (Q.interior[q] && A.out_grad) {
            double* dst = A.out_grad + ((size_t)b * (A.N - 6) + (Q.p0 + q - 3)) * 3;
            dst[0] = (double)g[q][0]; dst[1] = (double)g[q][1]; dst[2] = (double)g[q][2];
        }
    }
    if (lane % GROUP == 0) {
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
// LDS: hist[slot][ROW] of HPair<T> with ROW = TPB*(N-6) columns (each lane reads and writes only
//      the columns of its own points: LDS is a per-lane register extension here, no cross-lane
//      traffic and no barriers), then ys[slot][TPB] and alpha[age][TPB] doubles.
// Control flow: an outer trip per L-BFGS iteration (trip 0 = the initial evaluation) with ONE
// evaluation site inside the line-search loop, so the two groups of a wave re-converge at every
// iteration boundary and run the (dominant) two-loop recursion together.
#ifndef VIGO_TWOLOOP_WIN
#define VIGO_TWOLOOP_WIN 2
#endif
#ifndef VIGO_DOUBLE_ADD
#define VIGO_DOUBLE_ADD 1
#endif
#ifndef VIGO_TWOLOOP_STEADY
#define VIGO_TWOLOOP_STEADY 1
#endif
#ifndef VIGO_TWOLOOP_MARKSTEIN
#define VIGO_TWOLOOP_MARKSTEIN 1
#endif
// dev switch, measured and left off (profiles/README.md, round 3): 1 = the byte offsets of the 14 ring slots by AGE worked
// out once per two-loop (14 x {sub, wrap}, pinned in SGPRs) and each of the 28 fetches taking its offset from that table
// by a static index, instead of a running offset stepped and wrapped before every fetch (4 SALU instructions per fetch).
// The step loses its 4 SALU instructions and gains 4 s_nop: they had been sitting in the wait states the DPP moves of
// the butterfly need after the add that feeds them.  +2 % at B = 1024, +3 % on the full-chip batches.
#ifndef VIGO_RING_TABLE
#define VIGO_RING_TABLE 0
#endif
// dev builds only (-DVIGO_PROFILE_SECTIONS=1, tools/exp_sections.py): shader-clock totals of the sections of an
// iteration, written over out_x[b][0..9] — never defined in the shipped library
#ifndef VIGO_PROFILE_SECTIONS
#define VIGO_PROFILE_SECTIONS 0
#endif
#if VIGO_PROFILE_SECTIONS
#define VIGO_TICK(acc)                                                   \
    do {                                                                 \
        __builtin_amdgcn_sched_barrier(0);                               \
        const long long now_ = (long long)__builtin_readcyclecounter();  \
        __builtin_amdgcn_sched_barrier(0);                               \
        acc += (double)(now_ - tick_);                                   \
        tick_ = now_;                                                    \
    } while (0)
#else
#define VIGO_TICK(acc) do { } while (0)
#endif

// WPS = waves per SIMD the register budget is cut for.  1 (512 registers per lane, no scratch) unless the batch
// has more waves than the chip has SIMDs AND the history leaves room for eight waves in a CU's LDS (fp32 state,
// or fp64 trajectories of up to ~21 control points): capping the registers at 256 (a few hundred bytes of
// scratch per lane) then lets two waves share a SIMD's issue slots — +17 % for fp32 at 65 536 x 32, +36 % for
// fp64 at 16 384 x 16 — but costs 6 % when every wave has a SIMD to itself anyway.  Same arithmetic, same bits.
// OBS == false: the instantiation the launcher picks for calls without an obstacle list (A.obs == nullptr): no staging
// code, no obstacle loop, six sums per evaluation instead of seven — the same bits, fewer live registers
// RH = history pairs besides the newest that stay in registers (ages 1 .. RH; 0 for more than one point per lane: all in
// LDS): 1 normally; 4 or 5 in the level instantiations launched on batches that fill the chip (VIGO_LEVEL_RH below):
// fewer ring slots in LDS are more resident waves per CU
template <typename T, int GROUP, int PPL, bool FAST, int WPS = 1, bool OBS = true, int D = 3, int RH = (PPL == 1 ? 1 : 0)>
__global__ void __launch_bounds__(kWave, WPS) k_optimize(SolveArgs A, const DevConst* __restrict__ Kp) {
    static_assert(PPL == 1 ? (RH >= 1 && RH <= 6) : RH == 0, "register-held history only with one point per lane");
    const DevConst& K = *Kp;  // uniform address: scalar loads at the use sites, not 100+ live SGPRs
    constexpr int TPB = kWave / GROUP;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int N = A.N, NI = N - 6;
    const int ROW = TPB * NI;
    // one more column per slot that holds zeros for good: the history of every control point that is NOT free (the
    // three fixed points at either end, lanes beyond N).  Their s and y are identically zero, so the two-loop needs
    // no per-step select to keep their d at zero (two v_cndmask on the dependent chain of each of its 32 steps).
    const int ROWP = ROW + 1;
    using YS = YSv<FAST>;
    // A history slot = its ROWP records followed by the slot's {ys, 1/ys} per trajectory, so the steady-state ring
    // walks ONE byte offset for both (slot stride in bytes, a multiple of 16 for the ds_read_b128s)
    constexpr int kYsSlotBytes = (TPB * (int)sizeof(YS) + 15) & ~15;
    const int slotB = ROWP * (int)sizeof(HPair<T, D>) + kYsSlotBytes;
    const int m = K.mem_size;
    // REG1 (one control point per lane): the two newest history pairs (ages 0 and 1) stay in
    // registers, LDS holds the older m - 2 — at N = 64, m = 16 that is 38.3 KB instead of 43.8 KB
    // per wave, i.e. four resident waves per CU (one per SIMD) instead of three.
    constexpr bool REG1 = (PPL == 1);
    const int ms = REG1 ? (m > RH + 1 ? m - (RH + 1) : 0) : m;   // history slots in LDS
    HPair<T, D>* hist = reinterpret_cast<HPair<T, D>*>(lds_raw);
    double* ys_tab = reinterpret_cast<double*>(lds_raw + (size_t)ms * slotB);   // alphas, obstacle table

    const int lane = threadIdx.x;
    const int grp = lane / GROUP;
    int b = blockIdx.x * TPB + grp;
    if (A.active_idx) {                       // vigo_rebound_rounds: the compacted active set
        if (b >= *A.active_count) return;
        b = A.active_idx[b];
    } else if (b >= A.B) {
        return;
    }

    LaneProblem<T, PPL> Q;
    load_problem<T, GROUP, PPL>(A, K, b, lane % GROUP, Q);
    // x holds this lane's control points: free variables where interior, fixed boundary points
    // elsewhere (their g, d, s, y are identically zero so they never move).
    T x[PPL][3];
    load_points<T, PPL>(A, b, Q, x);
    set_level<T, GROUP, PPL>(K, Q, x);
    {
        // A solve is launched as ONE general kernel (D = 3), or — calls without obstacles, one point per lane — as the
        // D = 2 instantiation, which carries only x and y through the recursion (two thirds of the history in LDS, of
        // the dot products and of the stencils), followed by the general kernel: a wave whose trajectories are ALL
        // level is solved by the first launch and skipped by the second, every other wave the other way round.  A
        // level trajectory that shares a wave with one that is not is solved here with its z terms masked: the same
        // bits either way, so a result never depends on which trajectory it was paired with.
        const bool wave_level = __all(Q.level);
        if (D == 2 && !wave_level) return;
        if (D == 3 && A.level_waves_elsewhere && wave_level) return;
    }
#if VIGO_PROFILE_SECTIONS
    long long tick_ = (long long)__builtin_readcyclecounter();
    double t_eval = 0, t_ls = 0, t_upd = 0, t_two = 0, t_tail = 0, t_pre = 0, t_trial = 0, t_cal = 0;
#endif
    // history column of each owned point; points that are not free read the zero column and never write
    HPair<T, D>* hl[PPL];
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
        const int p = Q.p0 + q;
        hl[q] = Q.interior[q] ? hist + (grp * NI + (p - 3)) : hist + ROW;
    }
    for (int slot = lane; slot < ms; slot += kWave) {
        HPair<T, D> zero;
#pragma unroll
        for (int a = 0; a < D; ++a) zero.s[a] = zero.y[a] = T(0);
        *reinterpret_cast<HPair<T, D>*>(lds_raw + (size_t)slot * slotB + (size_t)ROW * sizeof(HPair<T, D>)) = zero;
    }
    __syncthreads();   // one wave per workgroup: orders the zero column before the first history read
    YS* ys_l = reinterpret_cast<YS*>(lds_raw + (size_t)ROWP * sizeof(HPair<T, D>)) + grp;   // slot 0; slot k at + k * slotB bytes
    double* al_l = ys_tab + grp;
    auto hist_at = [&](int q, int slot) -> HPair<T, D>& {
        return *reinterpret_cast<HPair<T, D>*>(reinterpret_cast<char*>(hl[q]) + (size_t)slot * slotB);
    };
    auto ys_at = [&](int slot) -> YS& { return *reinterpret_cast<YS*>(reinterpret_cast<char*>(ys_l) + (size_t)slot * slotB); };
    if (!OBS) { Q.obs = nullptr; Q.o_begin = Q.o_end = 0; }
    if (OBS && A.obs) {
        // stage this trajectory's obstacles once: the predicted positions and thresholds of
        // BT.cpp:1011-1020 by the expressions of obstacle_term(), sizes in T arithmetic
        constexpr int kEnt = kObsTabEntries<GROUP>;
        double* tab = ys_tab + (size_t)m * TPB + (size_t)grp * kObsTabDoubles<GROUP>;
        const int cnt = Q.o_end - Q.o_begin;
        const int steps = K.pred_num / 2 + 1;
        int fit = kEnt / steps;
        if (fit > kObsTabObs) fit = kObsTabObs;
        Q.o_tab = cnt < fit ? cnt : fit;
        Q.o_steps = steps;
        Q.obs_tab = tab;
        Q.obs_size = tab + 3 * kEnt;
        for (int e = lane % GROUP; e < Q.o_tab * steps; e += GROUP) {
            const int j = e / steps, n = 2 * (e - j * steps);
            const double* o = A.obs + 9 * (size_t)(Q.o_begin + j);
            const T tn = (T)((double)n * K.ts);
            tab[3 * e + 0] = (double)((T)o[0] + tn * (T)o[3]);
            tab[3 * e + 1] = (double)((T)o[1] + tn * (T)o[4]);
            tab[3 * e + 2] = (double)((T(1) - (T)(n / K.pred_num) * T(0.2)) * (T)K.thr_dyn);
        }
        for (int j = lane % GROUP; j < Q.o_tab; j += GROUP) {
            const double* o = A.obs + 9 * (size_t)(Q.o_begin + j);
            const T hx = (T)o[6] / 2, hy = (T)o[7] / 2;
            tab[3 * kEnt + j] = (double)(T)sqrt(hx * hx + hy * hy);
        }
        __syncthreads();  // one wave per workgroup: orders the staging writes before the lanes' reads
    }

    T g[PPL][3], xp[PPL][3], gp[PPL][3], d[PPL][3];
#pragma unroll
    for (int q = 0; q < PPL; ++q)
#pragma unroll
        for (int a = 0; a < 3; ++a) g[q][a] = xp[q][a] = gp[q][a] = d[q][a] = T(0);
    constexpr int kRH = RH > 0 ? RH : 1;
    T sR[kRH][PPL][3], yR[kRH][PPL][3];   // REG1: the pairs of ages 1 .. RH of the next two-loop (sR[j] = age j + 1)
    YS ysR[kRH]{};
#pragma unroll
    for (int q = 0; q < PPL; ++q)
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int j = 0; j < kRH; ++j) sR[j][q][a] = yR[j][q][a] = T(0);
    double sums[7];
    int evals = 0;
    int ret = LBERR_UNKNOWN;
    int k = 0, end = 0, last = 0;
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
            for (int q = 0; q < PPL; ++q)
#pragma unroll
                for (int a = 0; a < D; ++a) { xp[q][a] = x[q][a]; gp[q][a] = g[q][a]; }  // LB:1172-1173
            dginit = sums[4];  // g.d for the d just built (reduced at the end of the two-loop below)
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
                for (int q = 0; q < PPL; ++q)
#pragma unroll
                    for (int a = 0; a < D; ++a)
                        x[q][a] = FAST ? fmaT((T)step, d[q][a], xp[q][a]) : xp[q][a] + (T)step * d[q][a];
            }

            VIGO_TICK(t_pre);
            fx = eval_cost_grad<T, GROUP, PPL, FAST, OBS, D>(K, Q, x, d, g, sums);  // the only evaluation site (LB:828, :1132)
            VIGO_TICK(t_eval);
            ++evals;
            if (first) break;

            const double dg = sums[4];
            const double ftest1 = finit + step * dgtest;
            ++count;

            // LB:832-866: six exits tested in order, the first that holds wins.  Evaluated as one select chain
            // in reverse order and ONE branch (six exec-mask branches cost more issue slots than the compares)
            {
                int code = 0;
                if (fx <= ftest1 && fabs(dg) <= K.gtol * (-dginit)) code = count;                      // LB:862-866 (count >= 1)
                if (K.max_linesearch <= count) code = LBERR_MAXIMUMLINESEARCH;                         // LB:857-860
                if (brackt && (stmax - stmin) <= K.xtol * stmax) code = LBERR_WIDTHTOOSMALL;           // LB:852-855
                if (step == stpmin && (ftest1 < fx || dgtest <= dg)) code = LBERR_MINIMUMSTEP;         // LB:847-850
                if (step == stpmax && fx <= ftest1 && dg <= dgtest) code = LBERR_MAXIMUMSTEP;          // LB:842-845
                if (brackt && ((step <= stmin || stmax <= step) || uinfo != 0)) code = LBERR_ROUNDING_ERROR;   // LB:837-840
                if (code != 0) { ls = code; break; }
            }

            const double cmin = K.ftol <= K.gtol ? K.ftol : K.gtol;
            if (stage1 && fx <= ftest1 && cmin * dginit <= dg) stage1 = 0;

            // LB:883-920: the interval update runs on the modified function while stage1 holds and
            // the decrease is insufficient; one call site, operands selected here.
            const bool mod = stage1 && ftest1 < fx && fx <= xf;
            double axf = mod ? xf - xt * dgtest : xf, axd = mod ? xd - dgtest : xd;
            double ayf = mod ? yf - yt * dgtest : yf, ayd = mod ? yd - dgtest : yd;
            const double atf = mod ? fx - step * dgtest : fx, atd = mod ? dg - dgtest : dg;
            VIGO_TICK(t_ls);
            uinfo = trial_interval(xt, axf, axd, yt, ayf, ayd, step, atf, atd, stmin, stmax, brackt);
            VIGO_TICK(t_trial);
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

        VIGO_TICK(t_ls);
        double xnorm = sqrt(sums[5]), gnorm = sqrt(sums[6]);
        if (first) {
            first = false;
#pragma unroll
            for (int q = 0; q < PPL; ++q)
#pragma unroll
                for (int a = 0; a < D; ++a) d[q][a] = -g[q][a];  // LB:1144
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
            store_points<T, PPL>(A, b, Q, x);
#pragma unroll
            for (int q = 0; q < PPL; ++q)
#pragma unroll
                for (int a = 0; a < D; ++a) { x[q][a] = xp[q][a]; g[q][a] = gp[q][a]; }
            ret = ls;
            break;
        }

        // convergence test, LB:1200-1225 (norms came with the last evaluation)
        if (xnorm < 1.0) xnorm = 1.0;
        if (gnorm / xnorm <= K.g_epsilon) { ret = LB_CONVERGENCE; break; }
        if (K.max_iterations != 0 && K.max_iterations < k + 1) { ret = LBERR_MAXIMUMITERATION; break; }

        // s, y, ys, yy — LB:1264-1276
        T sv[PPL][3], yv[PPL][3];
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
#pragma unroll
            for (int a = 0; a < D; ++a) { sv[q][a] = x[q][a] - xp[q][a]; yv[q][a] = g[q][a] - gp[q][a]; }
            if (!REG1 && Q.interior[q]) {
                HPair<T, D> hp;
#pragma unroll
                for (int a = 0; a < D; ++a) { hp.s[a] = sv[q][a]; hp.y[a] = yv[q][a]; }
                hist_at(q, end) = hp;
            }
        }
        double ysyy[2] = {dot_lane<FAST, T, PPL, D>(yv, sv), dot_lane<FAST, T, PPL, D>(yv, yv)};
        // (level trajectory: the z product the general kernel adds here is (+0) * (+0); it turns a lane partial of -0 into
        // +0, and the sign of a zero ys decides the sign of the infinities the reference then divides into being)
        if (D == 2) ysyy[0] += 0.0;
        group_sum<GROUP, 2>(ysyy);
        const double ys = ysyy[0], yy = ysyy[1];
        // the two-loop divides by ys of each pair (LB:1300, :1312); FAST keeps its reciprocal instead
        const YS ys_div = make_ys(ys, static_cast<YS*>(nullptr));
        if (!REG1) ys_at(end) = ys_div;
        const bool haveR = REG1 && k >= RH + 1;   // sR[RH - 1] holds a pair (age RH now): it moves to the LDS ring below

        // two-loop recursion, LB:1286-1316, fully unrolled over the pair's age with a register
        // window of kWin pairs (static index age % kWin): the pair needed kWin steps ahead is
        // fetched from LDS into the window slot the current step has just consumed, so the
        // dependent chain never waits for LDS and does no address arithmetic or copies.
        const int bound = (m <= k) ? m : k;
        // !REG1: slot of the pair just stored (age 0); REG1: slot of the pair of age RH + 1 (the last one written)
        const int newest = REG1 ? last : end;
        if (!REG1) end = (end + 1 == m) ? 0 : end + 1;
        ++k;
#pragma unroll
        for (int q = 0; q < PPL; ++q)
#pragma unroll
            for (int a = 0; a < D; ++a) d[q][a] = -g[q][a];

        constexpr int kWin = VIGO_TWOLOOP_WIN;
        T Ps[kWin][PPL][3], Py[kWin][PPL][3];
        YS Pys[kWin];
        auto fetch = [&](int age, T (&s_)[PPL][3], T (&y_)[PPL][3], YS& ys_) {
            // (age is a literal after unrolling: the register cases fold away)
            if (REG1 && age == 0) {
#pragma unroll
                for (int q = 0; q < PPL; ++q)
#pragma unroll
                    for (int a = 0; a < D; ++a) { s_[q][a] = sv[q][a]; y_[q][a] = yv[q][a]; }
                ys_ = ys_div;
                return;
            }
            if (REG1 && age >= 1 && age <= RH) {
                const int j = (age - 1 < kRH && age >= 1) ? age - 1 : 0;
#pragma unroll
                for (int q = 0; q < PPL; ++q)
#pragma unroll
                    for (int a = 0; a < D; ++a) { s_[q][a] = sR[j][q][a]; y_[q][a] = yR[j][q][a]; }
                ys_ = ysR[j];
                return;
            }
            int slot = REG1 ? newest - (age - (RH + 1)) : newest - age;
            if (slot < 0) slot += ms;
#pragma unroll
            for (int q = 0; q < PPL; ++q) {
                const HPair<T, D> h = hist_at(q, slot);
#pragma unroll
                for (int a = 0; a < D; ++a) { s_[q][a] = h.s[a]; y_[q][a] = h.y[a]; }
            }
            ys_ = ys_at(slot);
        };
        // STEADY: the history is full (bound == kMaxMem, every iteration after the 16th): `age <
        // bound` is true at compile time, so the 32 steps are straight-line code — no exec-mask
        // blocks, no merge copies of the window registers, LDS fetches hoisted freely — and the
        // alphas stay in registers.  Same operations in the same order as the general path.
        auto two_loop = [&](auto steady_tag) -> bool {
            constexpr bool STEADY = decltype(steady_tag)::value;
            constexpr bool MARK = STEADY && !FAST && VIGO_TWOLOOP_MARKSTEIN;
            double amin = 1.0, amax = 1.0;
            // (general path: every live lane of the wave is in the same iteration, so the number of pairs is taken
            // through an SGPR — `age < bnd` becomes a scalar branch instead of a compare + exec-mask block per step)
            const int bnd = STEADY ? kMaxMem : __builtin_amdgcn_readfirstlane(bound);
            double al_reg[STEADY ? kMaxMem : 1];
            // STEADY: the LDS ring (kMaxMem - RH - 1 slots) is walked with running byte offsets — one
            // add and a wrap per fetch instead of slot arithmetic and two quarter-rate multiplies
            constexpr int kRing = kMaxMem - (RH + 1);
            constexpr int kFirstRing = RH + 1;    // the youngest age that lives in the ring
            const int stepB = slotB;
            // (the slot index is the same in every live lane — all trajectories of a wave are in the
            // same iteration — and is taken through an SGPR so the ring walk is scalar work; the caller
            // checks the uniformity)
            const int lastU = STEADY ? __builtin_amdgcn_readfirstlane(last) : 0;
            int curB = lastU * stepB;   // slot of the pair of age RH + 1
            auto ring_fetch = [&](T (&s_)[PPL][3], T (&y_)[PPL][3], YS& ys_) {
#pragma unroll
                for (int q = 0; q < PPL; ++q) {
                    const HPair<T, D> h = *reinterpret_cast<const HPair<T, D>*>(reinterpret_cast<const char*>(hl[q]) + curB);
#pragma unroll
                    for (int a = 0; a < D; ++a) { s_[q][a] = h.s[a]; y_[q][a] = h.y[a]; }
                }
                ys_ = *reinterpret_cast<const YS*>(reinterpret_cast<const char*>(ys_l) + curB);
            };
            auto ring_older = [&]() {   // towards higher ages: one slot down, wrapping
                curB -= stepB;
                if (curB < 0) curB += kRing * stepB;
            };
            auto ring_newer = [&]() {
                curB += stepB;
                if (curB >= kRing * stepB) curB -= kRing * stepB;
            };
            // (VIGO_RING_TABLE) ringT[i] = byte offset of the pair of age RH + 1 + i
            int ringT[kRing];
            if (STEADY && VIGO_RING_TABLE) {
                int c = curB;
#pragma unroll
                for (int i = 0; i < kRing; ++i) {
                    ringT[i] = c;
                    // (an opaque SGPR value: otherwise the chain above is re-materialised at every use)
                    asm volatile("" : "+s"(ringT[i]));
                    c -= stepB;
                    if (c < 0) c += kRing * stepB;
                }
            }
            // the pair of age `age` (a literal after unrolling) into a window slot; the running offset (table off) relies
            // on the two loops asking for the ages in ring order
            auto ring_fetch_age = [&](int age, bool newer_next, T (&s_)[PPL][3], T (&y_)[PPL][3], YS& ys_) {
                if (VIGO_RING_TABLE) {
                    curB = ringT[(age - kFirstRing >= 0 && age - kFirstRing < kRing) ? age - kFirstRing : 0];
                    ring_fetch(s_, y_, ys_);
                } else {
                    ring_fetch(s_, y_, ys_);
                    if (newer_next) ring_newer(); else ring_older();
                }
            };
#pragma unroll
            for (int q = 0; q < PPL; ++q)
#pragma unroll
                for (int a = 0; a < D; ++a) { Ps[0][q][a] = sv[q][a]; Py[0][q][a] = yv[q][a]; }
            Pys[0] = ys_div;
#pragma unroll
            for (int age = 1; age < kWin; ++age)
                if (age < bnd) {
                    // (a window of more than two pairs starts with ring slots in it: the ring pointer moves with them)
                    if (STEADY && age >= kFirstRing) ring_fetch_age(age, false, Ps[age], Py[age], Pys[age]);
                    else fetch(age, Ps[age], Py[age], Pys[age]);
                }
#pragma unroll
            for (int age = 0; age < kMaxMem; ++age) {      // newest -> oldest, LB:1294-1303
                if (age < bnd) {
                    const int w = age % kWin;
                    double al = group_sum1<GROUP>(dot_lane<FAST, T, PPL, D>(Ps[w], d));
                    al = over_ys<MARK>(al, Pys[w], amin, amax);
                    if (STEADY) al_reg[STEADY ? age : 0] = al;
                    else al_l[age * TPB] = al;         // alpha_j parks in LDS at a static offset
                    {
                        const T na = (T)(-al);     // (points that are not free hold y = 0: no select, see ROWP)
#pragma unroll
                        for (int q = 0; q < PPL; ++q)
#pragma unroll
                            for (int a = 0; a < D; ++a) d[q][a] = FAST ? fmaT(na, Py[w][q][a], d[q][a]) : d[q][a] + na * Py[w][q][a];
                    }
                    if (age + kWin < kMaxMem && age + kWin < bnd) {
                        if (STEADY && age + kWin >= kFirstRing) ring_fetch_age(age + kWin, false, Ps[w], Py[w], Pys[w]);
                        else fetch(age + kWin, Ps[w], Py[w], Pys[w]);
                    }
                }
            }
            if (STEADY) {
                // the ring pointer has gone once around (age kMaxMem == the slot of age RH + 1); the second
                // loop starts fetching at age kMaxMem - 1 - kWin.  The compiler must not keep the
                // first loop's 14 pairs alive in AGPRs for it (24 register moves per pair cost more
                // VALU slots than three ds_read_b128): LDS is declared clobbered here.
                if (!VIGO_RING_TABLE) {
#pragma unroll
                    for (int i = 0; i < kWin + 1; ++i) ring_newer();
                }
                asm volatile("" ::: "memory");
            }
            {
                const T sc = (T)(ys / yy);  // LB:1305
#pragma unroll
                for (int q = 0; q < PPL; ++q)
#pragma unroll
                    for (int a = 0; a < D; ++a) d[q][a] *= sc;
            }
            // the window now holds the ages [max(0, bound - kWin), bound)
#pragma unroll
            for (int age = kMaxMem - 1; age >= 0; --age) {  // oldest -> newest, LB:1307-1316
                if (age < bnd) {
                    const int w = age % kWin;
                    double beta = group_sum1<GROUP>(dot_lane<FAST, T, PPL, D>(Py[w], d));
                    beta = over_ys<MARK>(beta, Pys[w], amin, amax);
                    const double cod = (STEADY ? al_reg[STEADY ? age : 0] : al_l[age * TPB]) - beta;
                    {
                        const T co = (T)cod;
#pragma unroll
                        for (int q = 0; q < PPL; ++q)
#pragma unroll
                            for (int a = 0; a < D; ++a) d[q][a] = FAST ? fmaT(co, Ps[w][q][a], d[q][a]) : d[q][a] + co * Ps[w][q][a];
                    }
                    if (age - kWin >= 0) {
                        if (STEADY && age - kWin >= kFirstRing) ring_fetch_age(age - kWin, true, Ps[w], Py[w], Pys[w]);
                        else fetch(age - kWin, Ps[w], Py[w], Pys[w]);
                    }
                }
            }
            if (!MARK) return false;
            bool bad = !(amin >= 0x1p-500) || !(amax <= 0x1p500);
#pragma unroll
            for (int q = 0; q < PPL; ++q)
#pragma unroll
                for (int a = 0; a < D; ++a) bad |= !(d[q][a] == d[q][a]);
            return bad;
        };
        VIGO_TICK(t_upd);
        if (VIGO_TWOLOOP_STEADY && PPL == 1 && bound == kMaxMem && !__any(last != __builtin_amdgcn_readfirstlane(last))) {
            if (__any(two_loop(std::true_type{}))) {
                // a dividend outside the range Markstein's sequence is proven for (or a NaN): the same
                // recursion again from d = -g on the general path, which divides for real
                asm volatile("" ::: "memory");
#pragma unroll
                for (int q = 0; q < PPL; ++q)
#pragma unroll
                    for (int a = 0; a < D; ++a) d[q][a] = -g[q][a];
                two_loop(std::false_type{});
            }
        } else {
            two_loop(std::false_type{});
        }
        // A coefficient that is not finite (ys = 0: the reference has no ys > 0 guard, LB:1300) turns the zero d of the
        // points that are not free into NaN (0 * inf) — and through their lanes' partials every later dot product of
        // that two-loop, which oracle/vigo_oracle.c's emulation mirrors.  Those points never move: their d is reset
        // here, once per iteration instead of in every step.
#pragma unroll
        for (int q = 0; q < PPL; ++q)
#pragma unroll
            for (int a = 0; a < D; ++a) d[q][a] = Q.interior[q] ? d[q][a] : T(0);
        VIGO_TICK(t_two);
        if (REG1) {
            // the pair of age RH turns age RH + 1 for the next two-loop: it leaves the registers for the LDS
            // ring (overwriting the pair that would be age m); the younger ones move up by one, the new pair is age 1
            if (haveR && ms > 0) {
#pragma unroll
                for (int q = 0; q < PPL; ++q) {
                    if (Q.interior[q]) {
                        HPair<T, D> hp;
#pragma unroll
                        for (int a = 0; a < D; ++a) { hp.s[a] = sR[kRH - 1][q][a]; hp.y[a] = yR[kRH - 1][q][a]; }
                        hist_at(q, end) = hp;
                    }
                }
                ys_at(end) = ysR[kRH - 1];
                last = end;
                end = (end + 1 == ms) ? 0 : end + 1;
            }
#pragma unroll
            for (int j = kRH - 1; j > 0; --j) {
#pragma unroll
                for (int q = 0; q < PPL; ++q)
#pragma unroll
                    for (int a = 0; a < D; ++a) { sR[j][q][a] = sR[j - 1][q][a]; yR[j][q][a] = yR[j - 1][q][a]; }
                ysR[j] = ysR[j - 1];
            }
#pragma unroll
            for (int q = 0; q < PPL; ++q)
#pragma unroll
                for (int a = 0; a < D; ++a) { sR[0][q][a] = sv[q][a]; yR[0][q][a] = yv[q][a]; }
            ysR[0] = ys_div;
        }
        sums[4] = group_sum1<GROUP>(dot_lane<FAST, T, PPL, D>(g, d));  // dginit of the next line search (LB:746)
        step = 1.0;  // LB:1321
        VIGO_TICK(t_tail);
        VIGO_TICK(t_cal);    // back-to-back: the cost of one probe
    }

    // results.  On success / convergence / iteration cap the last evaluated point is x itself.
    if (ret >= 0 || ret == LBERR_MAXIMUMITERATION) store_points<T, PPL>(A, b, Q, x);
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
        if (Q.interior[q] && A.out_x) {
            double* dst = A.out_x + ((size_t)b * NI + (Q.p0 + q - 3)) * 3;
            dst[0] = (double)x[q][0]; dst[1] = (double)x[q][1]; dst[2] = (double)x[q][2];
        }
    }
#if VIGO_PROFILE_SECTIONS
    if (lane % GROUP == 0 && A.out_x) {
        double* o = A.out_x + (size_t)b * NI * 3;
        o[0] = t_eval; o[1] = t_ls; o[2] = t_upd; o[3] = t_two; o[4] = t_tail; o[5] = (double)k; o[6] = (double)evals;
        o[7] = t_pre; o[8] = t_trial; o[9] = t_cal;
    }
#endif
    if (lane % GROUP == 0) {
        if (A.out_status) A.out_status[(size_t)b * (A.status_stride ? A.status_stride : 1)] = ret;
        if (A.out_fx) A.out_fx[b] = fx;
        if (A.out_iters) A.out_iters[b] = k;
        if (A.out_evals) A.out_evals[b] = evals;
    }
}

template <typename T, int GROUP, bool FAST, int D = 3>
size_t optimize_lds_bytes(int N, int m, int ppl, bool with_obstacles, int rh = 1) {
    const int TPB = kWave / GROUP;
    const int ms = ppl == 1 ? (m > rh + 1 ? m - (rh + 1) : 0) : m;   // REG1: ages 0 .. rh live in registers
    // per slot: one record per free control point + the zero column, then {ys, 1/ys} per trajectory (see k_optimize)
    const size_t slot = ((size_t)TPB * (N - 6) + 1) * sizeof(HPair<T, D>) + (((size_t)TPB * sizeof(YSv<FAST>) + 15) & ~(size_t)15);
    size_t h = (size_t)ms * slot;
    h += (size_t)m * TPB * sizeof(double);        // the alphas of the general two-loop
    if (with_obstacles) h += (size_t)TPB * kObsTabDoubles<GROUP> * sizeof(double);
    return h;
}

constexpr size_t kLdsPerWorkgroup = 160 * 1024;

}  // namespace

#ifndef VIGO_SOLVER_PART
#define VIGO_SOLVER_PART 0
#endif

#if VIGO_SOLVER_PART == 0
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
    K.strict_z = P.strict_z != 0;
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

#endif  // VIGO_SOLVER_PART == 0

// (GROUP, PPL) for N control points: 32 x 1 up to 32, then 64 x {1, 2, 4}
static inline int shape_for(int N) { return N <= 32 ? 0 : (N <= 64 ? 1 : (N <= 128 ? 2 : 3)); }

#if VIGO_SOLVER_PART == 0
template <typename T, bool FAST>
static int launch_cost_grad_t(hipStream_t s, const SolveArgs& a, const DevConst* kd) {
    const int shape = shape_for(a.N);
    const int tpb = shape == 0 ? 2 : 1;
    dim3 grid((a.B + tpb - 1) / tpb), block(kWave);
    switch (shape) {
        case 0: hipLaunchKernelGGL((k_cost_grad<T, 32, 1, FAST>), grid, block, 0, s, a, kd); break;
        case 1: hipLaunchKernelGGL((k_cost_grad<T, 64, 1, FAST>), grid, block, 0, s, a, kd); break;
        case 2: hipLaunchKernelGGL((k_cost_grad<T, 64, 2, FAST>), grid, block, 0, s, a, kd); break;
        default: hipLaunchKernelGGL((k_cost_grad<T, 64, 4, FAST>), grid, block, 0, s, a, kd); break;
    }
    return (int)hipGetLastError();
}

int launch_cost_grad(hipStream_t s, const SolveArgs& a, const DevConst& k, const DevConst* kd, int precision) {
    (void)k;
    if (a.B <= 0) return hipSuccess;
    if (precision == VIGO_PREC_F32) return launch_cost_grad_t<float, false>(s, a, kd);
    if (precision == VIGO_PREC_F64_FAST) return launch_cost_grad_t<double, true>(s, a, kd);
    return launch_cost_grad_t<double, false>(s, a, kd);
}
#endif  // VIGO_SOLVER_PART == 0

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: the "already raised" flags live in the handle
// (LaunchState, one per handle = per device), never in function statics, so a second device of the same process
// gets its own.  SLOT numbers one k_optimize instantiation.
template <typename KernelT>
static int raise_dynamic_lds(LaunchState& L, int slot, KernelT kernel) {
    if (L.lds_attr_set & (1ull << slot)) return (int)hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)kLdsPerWorkgroup);
    if (e != hipSuccess) return (int)e;
    L.lds_attr_set |= 1ull << slot;
    return (int)hipSuccess;
}

// dev switch: 0 = never launch the level (D = 2) instantiation
#ifndef VIGO_LEVEL_KERNEL
#define VIGO_LEVEL_KERNEL 1
#endif
// History pairs besides the newest that the level kernel keeps in registers on batches with more waves than SIMDs
// (1 = as everywhere), where LDS decides how many waves a CU holds.  N <= 32 (two trajectories per wave): 4 pairs,
// 24.4 -> 19.3 KB, six -> EIGHT waves per CU (256 VGPRs, no spills): 1.74 -> 1.47 ms at 16 384 x 32 (2 pairs: 1.60,
// 3: 1.61, 5: 1.54).  32 < N <= 64 (one per wave): 5 pairs, 26.8 -> 19.2 KB, six -> eight: 1.85 -> 1.54 ms at 8192 x 64
// (4 pairs, seven waves: 1.71; 6 pairs: 1.57).  Same arithmetic, same bits.
#ifndef VIGO_LEVEL_RH
#define VIGO_LEVEL_RH 4
#endif
#ifndef VIGO_LEVEL_RH64
#define VIGO_LEVEL_RH64 5
#endif
template <typename T, int GROUP, int PPL, bool FAST, bool OBS>
static int launch_optimize_t(hipStream_t s, const SolveArgs& a_in, const DevConst& k, const DevConst* kd, LaunchState& L) {
    const int tpb = kWave / GROUP;
    SolveArgs a = a_in;
    dim3 grid((a.B + tpb - 1) / tpb), block(kWave);
    // slot = arithmetic (fp32 / fp64 / fp64 fast) x shape (GROUP, PPL) x waves per SIMD x with / without obstacles;
    // the level instantiations (one point per lane, no obstacles) follow from 48 on
    const int arith = std::is_same<T, float>::value ? 0 : (FAST ? 2 : 1);
    const int shape = GROUP == 32 ? 0 : (PPL == 1 ? 1 : (PPL == 2 ? 2 : 3));
    const int slot = (arith * 4 + shape) * 4;
    // a solver wavefront per SIMD (4 per CU) is full occupancy for these kernels; unknown SIMD count: never switch
    const int simds = L.simd_count > 0 ? L.simd_count : (1 << 30);
    auto go = [&](auto kernel, int sl, size_t lds) -> int {
        int e = raise_dynamic_lds(L, sl, kernel);
        if (e != (int)hipSuccess) return e;
        hipLaunchKernelGGL(kernel, grid, block, lds, s, a, kd);
        return (int)hipGetLastError();
    };
    // Calls that can hold level trajectories (no z planning) and have an instantiation for them are two launches: FIRST the
    // level kernel, whose waves with a trajectory that is not level exit at once, THEN the general kernel, whose waves
    // of level trajectories do.  The order matters: each launch decides from the control points it finds; a level
    // trajectory's z is untouched by the first launch, so the second still sees it level and skips it — the other way
    // round, a trajectory just outside the band that the general solve smooths into it would be solved a second time.
    constexpr bool kHasLevel = VIGO_LEVEL_KERNEL && PPL == 1 && !OBS;
    const bool two = kHasLevel && !k.plan_in_z && !k.strict_z;
    a.level_waves_elsewhere = two ? 1 : 0;
    const size_t lds = optimize_lds_bytes<T, GROUP, FAST, 3>(a.N, k.mem_size, PPL, a.obs != nullptr);
    if (lds > kLdsPerWorkgroup) return (int)hipErrorInvalidValue;  // refused earlier by vigo_optimize
    int e = (int)hipSuccess;
    if constexpr (kHasLevel) {
        if (two) {
            const size_t lds2 = optimize_lds_bytes<T, GROUP, FAST, 2>(a.N, k.mem_size, PPL, false);
            const int slot2 = 48 + arith * 4 + shape * 2;
            // fp64, two trajectories per wave, more waves than SIMDs: keep VIGO_LEVEL_RH pairs besides the newest in
            // registers when that buys a further resident wave per CU (N = 32, m = 16: 24.4 -> 22.7 KB, six -> seven)
            constexpr int kRH = GROUP == 32 ? VIGO_LEVEL_RH : VIGO_LEVEL_RH64;
            constexpr bool kHasRH = std::is_same<T, double>::value && kRH > 1;
            bool done = false;
            if constexpr (kHasRH) {
                const size_t lds3 = optimize_lds_bytes<T, GROUP, FAST, 2>(a.N, k.mem_size, PPL, false, kRH);
                if ((int)grid.x > simds && lds2 > kLdsPerWorkgroup / 8 && kLdsPerWorkgroup / lds3 > kLdsPerWorkgroup / lds2) {
                    // (more than four waves per CU put two on a SIMD: the register-capped build, 256 VGPRs)
                    constexpr int kWps = kRH >= 4 ? 2 : 1;
                    e = go(&k_optimize<T, GROUP, PPL, FAST, kWps, OBS, 2, kRH>, 60 + (FAST ? 1 : 0) + (GROUP == 64 ? 2 : 0), lds3);
                    done = true;
                }
            }
            if (!done) {
                if ((int)grid.x > simds && lds2 <= kLdsPerWorkgroup / 8) e = go(&k_optimize<T, GROUP, PPL, FAST, 2, OBS, 2>, slot2 + 1, lds2);
                else e = go(&k_optimize<T, GROUP, PPL, FAST, 1, OBS, 2>, slot2, lds2);
            }
            if (e != (int)hipSuccess) return e;
        }
    }
    if (PPL == 1 && (int)grid.x > simds && lds <= kLdsPerWorkgroup / 8) {   // more waves than SIMDs and 8 fit a CU: two per SIMD
        if constexpr (PPL == 1) e = go(&k_optimize<T, GROUP, PPL, FAST, 2, OBS, 3>, slot + (OBS ? 1 : 3), lds);
        else e = (int)hipErrorInvalidValue;
    } else {
        e = go(&k_optimize<T, GROUP, PPL, FAST, 1, OBS, 3>, slot + (OBS ? 0 : 2), lds);
    }
    return e;
}

template <typename T, bool FAST, bool OBS>
static int launch_optimize_p(hipStream_t s, const SolveArgs& a, const DevConst& k, const DevConst* kd, LaunchState& L) {
    // (one trajectory per wave for N <= 32 — 64 x 1 with half the lanes idle, no divergence between
    // the two line searches — measured slower: 0.462 vs 0.445 ms at B = 1024, 6.40 vs 3.60 ms at B = 16384)
    switch (shape_for(a.N)) {
        // (a 16-lane x 2-point shape saves one butterfly level but measured 22 % slower: 1.76 M vs 2.26 M/s)
        case 0: return launch_optimize_t<T, 32, 1, FAST, OBS>(s, a, k, kd, L);
        case 1: return launch_optimize_t<T, 64, 1, FAST, OBS>(s, a, k, kd, L);
        case 2: return launch_optimize_t<T, 64, 2, FAST, OBS>(s, a, k, kd, L);
        default: return launch_optimize_t<T, 64, 4, FAST, OBS>(s, a, k, kd, L);
    }
}

template <bool OBS>
static int launch_optimize_o(hipStream_t s, const SolveArgs& a, const DevConst& k, const DevConst* kd, int precision, LaunchState& L) {
    if (precision == VIGO_PREC_F32) return launch_optimize_p<float, false, OBS>(s, a, k, kd, L);
    if (precision == VIGO_PREC_F64_FAST) return launch_optimize_p<double, true, OBS>(s, a, k, kd, L);
    return launch_optimize_p<double, false, OBS>(s, a, k, kd, L);
}

// This file is compiled TWICE (csrc/Makefile).  VIGO_SOLVER_PART == 1: only the k_optimize instantiations for calls WITH
// an obstacle list, built with machine LICM (their inner obstacle loops want their invariants hoisted: 1.85 vs 1.92 ms on
// config 5a); VIGO_SOLVER_PART == 0: everything else, built without it (-3 % on configs 2 and 4).
#if VIGO_SOLVER_PART == 1
int launch_optimize_with_obstacles(hipStream_t s, const SolveArgs& a, const DevConst& k, const DevConst* kd, int precision, LaunchState& L) {
    return launch_optimize_o<true>(s, a, k, kd, precision, L);
}
#else
int launch_optimize_with_obstacles(hipStream_t s, const SolveArgs& a, const DevConst& k, const DevConst* kd, int precision, LaunchState& L);
int launch_optimize(hipStream_t s, const SolveArgs& a, const DevConst& k, const DevConst* kd, int precision, LaunchState& L) {
    if (a.B <= 0) return hipSuccess;
    if (a.obs != nullptr) return launch_optimize_with_obstacles(s, a, k, kd, precision, L);
    // Measured (tools/exp_solver.py, VIGO_EXP_MATRIX=1: N = 16 ... 200 x the three arithmetic modes): the instantiation
    // without obstacle code is 4 - 18 % faster everywhere except f64_fast at 32 < N <= 64 on batches with more waves
    // than SIMDs (8 % slower there): those keep the general kernel, which treats a missing list as no obstacles.
    // (Only where the level instantiation cannot apply — z planning on: with it, level waves go to the D = 2 kernel, 20 %
    // faster than either, and the corner is not worth keeping them from it.)
    if (precision == VIGO_PREC_F64_FAST && a.N > 32 && a.N <= 64 && L.simd_count > 0 && a.B > L.simd_count && (k.plan_in_z || k.strict_z || !VIGO_LEVEL_KERNEL))
        return launch_optimize_with_obstacles(s, a, k, kd, precision, L);
    return launch_optimize_o<false>(s, a, k, kd, precision, L);
}

// bytes of LDS one trajectory-solve workgroup needs; the C ABI refuses N it cannot hold
size_t optimize_lds_requirement(int N, int mem_size, int precision) {
    const bool g32 = N <= 32;
    const int ppl = N <= 64 ? 1 : (N <= 128 ? 2 : 4);
    if (precision == VIGO_PREC_F32) return g32 ? optimize_lds_bytes<float, 32, false>(N, mem_size, ppl, true) : optimize_lds_bytes<float, 64, false>(N, mem_size, ppl, true);
    if (precision == VIGO_PREC_F64_FAST) return g32 ? optimize_lds_bytes<double, 32, true>(N, mem_size, ppl, true) : optimize_lds_bytes<double, 64, true>(N, mem_size, ppl, true);
    return g32 ? optimize_lds_bytes<double, 32, false>(N, mem_size, ppl, true) : optimize_lds_bytes<double, 64, false>(N, mem_size, ppl, true);
}
#endif  // VIGO_SOLVER_PART

}  // namespace vigo
