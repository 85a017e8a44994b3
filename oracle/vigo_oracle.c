/*
 * vigo_oracle.c — TEST INFRASTRUCTURE ONLY (see vigo_oracle.h for the pinning status).
 *
 * Plain-C, single-thread, fp64 restatement of the reference hot path.  Arithmetic is
 * written in the reference's own evaluation order (including its pow() calls and its
 * quirks) so that, compiled like the reference (g++/gcc -O3, no -ffast-math, x86-64 SSE2,
 * no FMA contraction), it reproduces the reference's roundings.
 *
 * Citations: BT = include/trajectory_planner/bsplineTraj.{h,cpp}, LB = .../solver/lbfgs.hpp,
 * BS = .../bspline.cpp, PO = .../polyTrajOctomap.cpp, PS = .../polyTrajSolver.cpp.
 */
#include "vigo_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------ */
/* Evaluation mode.                                                                      */
/*   0        reference order: sequential sums, glibc pow() — what the reference computes. */
/*   32 / 64  "device emulation": the SAME formulas, but (a) every per-trajectory sum is    */
/*            formed like the HIP kernels do — a per-control-point partial (lane p = point  */
/*            p) followed by the butterfly tree v[i] += v[i^m], m = 1,2,..,GROUP/2 — and    */
/*            (b) x*x, x*x*x, sqrt replace pow(x,2), pow(x,3), pow(x,0.5).  In this mode     */
/*            the oracle is expected to match the GPU bit for bit, which pins the kernels'  */
/*            control flow; mode 0 vs mode 32/64 on the CPU quantifies what those two       */
/*            (rounding-level) differences do to a 50-iteration solve.                      */
/* ------------------------------------------------------------------------------------ */
static int g_emu_group = 0;
static int g_emu_ppl = 1;   /* consecutive control points owned by one lane */
static int g_emu_fast = 0;  /* mirror VIGO_PREC_F64_FAST: the kernels' explicit fma()s and the
                               reciprocal-multiply of the two-loop recursion */
void vgo_set_emulation_fast(int on) { g_emu_fast = on; }
#define FAST (g_emu_group && g_emu_fast)
void vgo_set_emulation(int group) { g_emu_group = group; g_emu_ppl = 1; }
void vgo_set_emulation2(int group, int ppl) { g_emu_group = group; g_emu_ppl = ppl > 0 ? ppl : 1; }
int vgo_get_emulation(void) { return g_emu_group; }

static inline double P2(double x) { return g_emu_group ? x * x : pow(x, 2); }
static inline double P3(double x) { return g_emu_group ? (x * x) * x : pow(x, 3); }
static inline double PHALF(double x) { return g_emu_group ? sqrt(x) : pow(x, 0.5); }

/* per-point partials pts[group*ppl] -> lane partial (the lane's points in index order, starting
 * from 0.0) -> butterfly all-reduce; returns lane 0's value */
static double lane_tree_sum(const double* pts, int group) {
    double a[64], b[64];
    for (int i = 0; i < group; ++i) {
        double s = pts[i * g_emu_ppl];
        for (int q = 1; q < g_emu_ppl; ++q) s += pts[i * g_emu_ppl + q];
        a[i] = s;
    }
    for (int m = 1; m < group; m <<= 1) {
        for (int i = 0; i < group; ++i) b[i] = a[i] + a[i ^ m];
        for (int i = 0; i < group; ++i) a[i] = b[i];
    }
    return a[0];
}

/* ------------------------------------------------------------------------------------ */
/* defaults: cfg/bspline_interactive/bspline_planner_param.yaml:4-19, BT.h:46-47,        */
/* BT.cpp:695-699, LB:942-954                                                            */
/* ------------------------------------------------------------------------------------ */
void vgo_default_params(vigo_params_t* p) {
    memset(p, 0, sizeof(*p));
    p->dthresh = 0.5;
    p->dist_thresh_dynamic = 0.5;
    p->ts_ctrl = 0.2;
    p->ts = 0.1;
    p->pred_horizon = 2.0;
    p->uncertain_factor = 1.0;
    p->w_distance = 1.0;
    p->w_smoothness = 1.0;
    p->w_feasibility = 1.0;
    p->w_dynamic = 1.0;
    p->min_height = 0.7;
    p->max_height = 1.3;
    p->plan_in_z = 0;
    p->mem_size = 16;
    p->max_iterations = 200;
    p->max_linesearch = 40;
    p->past = 0;
    p->g_epsilon = 0.01;
    p->delta = 1e-5;
    p->min_step = 1e-20;
    p->max_step = 1e20;
    p->f_dec_coeff = 1e-4;
    p->s_curv_coeff = 0.9;
    p->xtol = 1.0e-16;
}

/* ------------------------------------------------------------------------------------ */
/* voxel map (own contract, include/vigo.h "voxel map")                                  */
/* ------------------------------------------------------------------------------------ */
void vgo_grid_init(vgo_grid_t* g, int nx, int ny, int nz, const double origin[3], double res,
                   const uint8_t* vox) {
    g->nx = nx; g->ny = ny; g->nz = nz;
    g->res = res;
    g->vox = vox;
    for (int a = 0; a < 3; ++a) g->origin[a] = origin[a];
    g->bmin[0] = origin[0]; g->bmin[1] = origin[1]; g->bmin[2] = origin[2];
    g->bmax[0] = origin[0] + nx * res;
    g->bmax[1] = origin[1] + ny * res;
    g->bmax[2] = origin[2] + nz * res;
}

/* returns the voxel byte, or 0xFF (all bits set) when p is outside the box */
static unsigned grid_byte(const vgo_grid_t* g, const double p[3]) {
    int ix = (int)floor((p[0] - g->origin[0]) / g->res);
    int iy = (int)floor((p[1] - g->origin[1]) / g->res);
    int iz = (int)floor((p[2] - g->origin[2]) / g->res);
    if (ix < 0 || iy < 0 || iz < 0 || ix >= g->nx || iy >= g->ny || iz >= g->nz) return 0xFFu;
    return g->vox[((size_t)ix * g->ny + iy) * g->nz + iz];
}

int vgo_is_inflated_occupied(const vgo_grid_t* g, const double p[3]) {
    return (int)(grid_byte(g, p) & 1u);
}

int vgo_is_unknown(const vgo_grid_t* g, const double p[3]) {
    return (int)((grid_byte(g, p) >> 1) & 1u);
}

/* endpoints, then int(dist/res)-1 interior probes spaced res along the line */
int vgo_is_inflated_occupied_line(const vgo_grid_t* g, const double p1[3], const double p2[3]) {
    if (vgo_is_inflated_occupied(g, p1) || vgo_is_inflated_occupied(g, p2)) return 1;
    double d[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
    double dist = sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    double inc[3] = {d[0] / dist * g->res, d[1] / dist * g->res, d[2] / dist * g->res};
    int steps = (int)(dist / g->res);
    for (int i = 1; i < steps; ++i) {
        double q[3] = {p1[0] + i * inc[0], p1[1] + i * inc[1], p1[2] + i * inc[2]};
        if (vgo_is_inflated_occupied(g, q)) return 1;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* cost terms.  Gradients are accumulated into 3*N arrays laid out like Eigen's 3xN       */
/* column-major matrix (BT.cpp:807-810).                                                 */
/* ------------------------------------------------------------------------------------ */

/* Eigen fixed-size 3-vector reductions on x86-64/SSE2 reduce a 2-packet first:
 * (x0 + x1) + x2.  (Un-vectorised Eigen would give x0 + (x1 + x2): parity unpinned.) */
static inline double sum3(double a, double b, double c) { return (a + b) + c; }

/* a cost contribution of control point i: added to the running total (reference order) or to
 * lane i's partial (device emulation) */
#define LANE_ADD(i, v) do { double v_ = (v); if (lanes) lanes[(i)] += v_; else cost += v_; } while (0)
#define TERM_RESULT() (lanes ? lane_tree_sum(lanes, g_emu_group) : cost)

/* BT.cpp:823-932 */
static double distance_term(const vigo_params_t* P, int N, const double* c, const int32_t* goff,
                            const double* gpv, const uint8_t* gunk, double* G, double* lanes) {
    const double dth = P->dthresh;
    double cost = 0.0;
    double a = 3.0 * dth, b = -3.0 * pow(dth, 2), cc = pow(dth, 3);
    const double hth = 0.2; /* heightDistThresh BT.cpp:836 */
    double ah = 3.0 * hth, bh = -3 * pow(hth, 2), ch = pow(hth, 3);
    for (int i = 3; i <= N - 3 - 1; ++i) {
        const double* ci = c + 3 * i;
        double* Gi = G + 3 * i;
        for (int32_t j = goff[i]; j < goff[i + 1]; ++j) {
            const double* p = gpv + 6 * (size_t)j;
            const double* v = p + 3;
            double dist = FAST ? fma(ci[2] - p[2], v[2], fma(ci[1] - p[1], v[1], (ci[0] - p[0]) * v[0]))
                               : sum3((ci[0] - p[0]) * v[0], (ci[1] - p[1]) * v[1], (ci[2] - p[2]) * v[2]);
            int unknown = gunk ? (gunk[j] != 0) : 0;
            double e = dth - dist;
            double ct, k, gt[3];
            if (e <= -1.0 * dth) {            /* too far: NOT scaled by the unknown factor */
                ct = P3(-e);
                k = 3.0 * P2(-e);
                gt[0] = k * v[0]; gt[1] = k * v[1]; gt[2] = k * v[2];
            } else if (e > 0 && e <= dth) {   /* e == dth lands here, not in the quadratic */
                ct = P3(e);
                k = -3.0 * P2(e);
                gt[0] = k * v[0]; gt[1] = k * v[1]; gt[2] = k * v[2];
                if (unknown) {
                    ct *= P->uncertain_factor;
                    gt[0] *= P->uncertain_factor; gt[1] *= P->uncertain_factor; gt[2] *= P->uncertain_factor;
                }
            } else if (e >= dth) {
                ct = a * P2(e) + b * e + cc;
                k = -(2 * a * e + b);
                gt[0] = k * v[0]; gt[1] = k * v[1]; gt[2] = k * v[2];
                if (unknown) {
                    ct *= P->uncertain_factor;
                    gt[0] *= P->uncertain_factor; gt[1] *= P->uncertain_factor; gt[2] *= P->uncertain_factor;
                }
            } else {
                continue;                     /* -dth < e <= 0: no penalty */
            }
            if (!P->plan_in_z) gt[2] = 0.0;
            LANE_ADD(i, ct);
            Gi[0] += gt[0]; Gi[1] += gt[1]; Gi[2] += gt[2];
        }
        if (P->plan_in_z) {
            /* BT.cpp:897-930, including the x-row gradient and the heightDistMax band test */
            double hmin = ci[2] - P->min_height;
            double hmax = ci[2] - P->max_height;
            if (hmin < 0) {
                double e = hth - hmin;
                LANE_ADD(i, ah * P2(e) + bh * e + ch);
                Gi[0] += -(2 * ah * e + bh) * -1.0;
            } else if (hmin >= 0 && hmax < hth) {
                double e = hth - hmin;
                LANE_ADD(i, P3(e));
                Gi[0] += -3.0 * P2(e) * -1.0;
            }
            if (hmax > 0) {
                double e = hth + hmax;
                LANE_ADD(i, ah * P2(e) + bh * e + ch);
                Gi[0] += -(2 * ah * e + bh) * 1.0;
            } else if (hmax <= 0 && hmax >= -hth) {
                double e = hth + hmax;
                LANE_ADD(i, P3(e));
                Gi[0] += -3.0 * P2(e) * 1.0;
            }
        }
    }
    return TERM_RESULT();
}

/* (device emulation only) THE LEVEL RULE of the kernels (include/vigo.h, vigo_solver.hip set_level): with plan_in_z off and
 * all N control points at one height to 2^-40 relative, the smoothness and feasibility terms of the z axis — whose
 * values then differ by rounding noise only — are taken as exactly zero, cost and gradient, so z never moves.  The
 * reference-order mode knows no such rule: it restates the reference, which lets z drift by that noise. */
/* the kernels decide ONCE per solve, from the control points the solve starts with (a trajectory just outside the band
 * that the smoothing pulls into it later stays a general one to the end): vgo_optimize pins the answer here for its
 * evaluations; -1 = decide from the points given (the standalone cost / gradient) */
static int g_level_pinned = -1;
static int traj_level(const vigo_params_t* P, int N, const double* c) {
    if (!g_emu_group || P->plan_in_z || P->strict_z) return 0;
    if (g_level_pinned >= 0) return g_level_pinned;
    double mn = INFINITY, mx = -INFINITY;
    for (int i = 0; i < N; ++i) { mn = fmin(mn, c[3 * i + 2]); mx = fmax(mx, c[3 * i + 2]); }
    return (mx - mn) <= 0x1p-40 * fmax(1.0, fmax(fabs(mn), fabs(mx)));
}

/* BT.cpp:934-950 (axes = 3; 2 under the level rule) */
static double smoothness_term(int N, const double* c, double* G, double* lanes, int axes) {
    double cost = 0.0;
    for (int i = 0; i < N - 3; ++i) {
        double jk[3], gt[3];
        for (int a = 0; a < 3; ++a) {
            jk[a] = FAST ? fma(3.0, c[3 * (i + 1) + a], fma(-3.0, c[3 * (i + 2) + a], c[3 * (i + 3) + a])) - c[3 * i + a]
                         : ((c[3 * (i + 3) + a] - 3 * c[3 * (i + 2) + a]) + 3 * c[3 * (i + 1) + a]) - c[3 * i + a];
            if (a >= axes) jk[a] = 0.0;
        }
        LANE_ADD(i, sum3(jk[0] * jk[0], jk[1] * jk[1], jk[2] * jk[2]));
        for (int a = 0; a < 3; ++a) gt[a] = 2.0 * jk[a];
        for (int a = 0; a < axes; ++a) {
            G[3 * i + a] += -gt[a];
            if (FAST) {
                G[3 * (i + 1) + a] = fma(3.0, gt[a], G[3 * (i + 1) + a]);
                G[3 * (i + 2) + a] = fma(-3.0, gt[a], G[3 * (i + 2) + a]);
            } else {
                G[3 * (i + 1) + a] += 3.0 * gt[a];
                G[3 * (i + 2) + a] += -3.0 * gt[a];
            }
            G[3 * (i + 3) + a] += gt[a];
        }
    }
    return TERM_RESULT();
}

/* BT.cpp:952-999; limits hard-coded to 1.0 (BT.cpp:955-956) */
static double feasibility_term(const vigo_params_t* P, int N, const double* c, double* G, double* lanes, int axes) {
    double cost = 0.0;
    const double maxVel = 1.0, maxAcc = 1.0;
    const double ts = P->ts_ctrl;
    double tsInvSqr = 1 / pow(ts, 2);
    for (int i = 0; i < N - 1; ++i) {
        for (int j = 0; j < axes; ++j) {
            double vi = (c[3 * (i + 1) + j] - c[3 * i + j]) / ts;
            if (vi > maxVel) {
                LANE_ADD(i, P2(vi - maxVel) * tsInvSqr);
                G[3 * i + j] += -2 * (vi - maxVel) / ts * tsInvSqr;
                G[3 * (i + 1) + j] += 2 * (vi - maxVel) / ts * tsInvSqr;
            } else if (vi < -maxVel) {
                LANE_ADD(i, P2(vi + maxVel) * tsInvSqr);
                G[3 * i + j] += -2 * (vi + maxVel) / ts * tsInvSqr;
                G[3 * (i + 1) + j] += 2 * (vi + maxVel) / ts * tsInvSqr;
            }
        }
    }
    for (int i = 0; i < N - 2; ++i) {
        for (int j = 0; j < axes; ++j) {
            double ai = (FAST ? fma(-2.0, c[3 * (i + 1) + j], c[3 * (i + 2) + j]) + c[3 * i + j]
                              : (c[3 * (i + 2) + j] - 2 * c[3 * (i + 1) + j]) + c[3 * i + j]) * tsInvSqr;
            if (ai > maxAcc) {
                LANE_ADD(i, P2(ai - maxAcc));
                G[3 * i + j] += 2 * (ai - maxAcc) * tsInvSqr;
                G[3 * (i + 1) + j] += -4 * (ai - maxAcc) * tsInvSqr;
                G[3 * (i + 2) + j] += 2 * (ai - maxAcc) * tsInvSqr;
            } else if (ai < -maxAcc) {
                LANE_ADD(i, P2(ai + maxAcc));
                G[3 * i + j] += 2 * (ai + maxAcc) * tsInvSqr;
                G[3 * (i + 1) + j] += -4 * (ai + maxAcc) * tsInvSqr;
                G[3 * (i + 2) + j] += 2 * (ai + maxAcc) * tsInvSqr;
            }
        }
    }
    return TERM_RESULT();
}

/* BT.cpp:1001-1064 */
static double dynamic_term(const vigo_params_t* P, int N, const double* c, int n_obs,
                           const double* obs, double* G, double* lanes) {
    double cost = 0;
    if (n_obs == 0) return cost;  /* BT.cpp:1003 */
    const int skipFactor = 2;
    int predictionNum = (int)(P->pred_horizon / P->ts);
    const double thr0 = P->dist_thresh_dynamic;
    double a = 3.0 * thr0, b = -3 * pow(thr0, 2), cc = pow(thr0, 3);
    for (int i = 3; i <= N - 3 - 1; ++i) {
        const double* ci = c + 3 * i;
        double* Gi = G + 3 * i;
        for (int j = 0; j < n_obs; ++j) {
            const double* op = obs + 9 * (size_t)j;
            const double* ov = op + 3;
            const double* os = op + 6;
            double size = PHALF(P2(os[0] / 2) + P2(os[1] / 2));
            for (int n = 0; n <= predictionNum; n += skipFactor) {
                double tn = (double)(n * P->ts);
                double px = op[0] + tn * ov[0];
                double py = op[1] + tn * ov[1];
                /* integer division n/predictionNum (BT.cpp:1020) */
                double thr = (1 - (double)(n / predictionNum) * 0.2) * thr0;
                double dx = ci[0] - px, dy = ci[1] - py, dz = 0.0;
                double nrm = sqrt(sum3(dx * dx, dy * dy, dz * dz));
                double dist = nrm - size;
                double e = thr - dist;
                double gx = dx / nrm, gy = dy / nrm, gz = dz / nrm;
                if (e <= 0) {
                    /* no punishment */
                } else if (e > 0 && e <= thr) {
                    LANE_ADD(i, P3(e));
                    double k = -3.0 * P2(e);
                    Gi[0] += k * gx; Gi[1] += k * gy; Gi[2] += k * gz;
                } else if (e >= thr) {
                    LANE_ADD(i, (a * P2(e) + b * e + cc));
                    double k = -(2 * a * e + b);
                    Gi[0] += k * gx; Gi[1] += k * gy; Gi[2] += k * gz;
                }
            }
        }
    }
    return TERM_RESULT();
}

#define VGO_MAX_N VIGO_MAX_CTRL_POINTS

/* BT.cpp:802-821 */
double vgo_cost_grad(const vigo_params_t* P, int N, const double* ctrl, const int32_t* goff,
                     const double* gpv, const uint8_t* gunk, int n_obs, const double* obs,
                     const double w[4], double* grad_free, double* grad_full, double* terms) {
    double Gd[3 * VGO_MAX_N], Gs[3 * VGO_MAX_N], Gf[3 * VGO_MAX_N], Go[3 * VGO_MAX_N];
    memset(Gd, 0, sizeof(double) * 3 * N);
    memset(Gs, 0, sizeof(double) * 3 * N);
    memset(Gf, 0, sizeof(double) * 3 * N);
    memset(Go, 0, sizeof(double) * 3 * N);
    double L[4][VGO_MAX_N];   /* per-point cost partials (device emulation) */
    memset(L, 0, sizeof(L));
    const int emu = g_emu_group != 0;
    const int level = traj_level(P, N, ctrl);
    double cd = distance_term(P, N, ctrl, goff, gpv, gunk, Gd, emu ? L[0] : NULL);
    double cs = smoothness_term(N, ctrl, Gs, emu ? L[1] : NULL, level ? 2 : 3);
    double cf = feasibility_term(P, N, ctrl, Gf, emu ? L[2] : NULL, level ? 2 : 3);
    double co = dynamic_term(P, N, ctrl, n_obs, obs, Go, emu ? L[3] : NULL);
    double total = w[0] * cd + w[1] * cs + w[2] * cf + w[3] * co;
    for (int e = 0; e < 3 * N; ++e) {
        double t = FAST ? fma(w[3], Go[e], fma(w[2], Gf[e], fma(w[1], Gs[e], w[0] * Gd[e])))
                        : w[0] * Gd[e] + w[1] * Gs[e] + w[2] * Gf[e] + w[3] * Go[e];
        if (level && e % 3 == 2) t = 0.0;
        if (grad_full) grad_full[e] = t;
        if (grad_free && e >= 9 && e < 3 * (N - 3)) grad_free[e - 9] = t;
    }
    if (terms) { terms[0] = cd; terms[1] = cs; terms[2] = cf; terms[3] = co; }
    return total;
}

/* ------------------------------------------------------------------------------------ */
/* L-BFGS with More-Thuente line search, restated from LB:295-391, :506-714, :716-937,    */
/* :1024-1349.  Status codes are the reference's (LB:20-80).                             */
/* ------------------------------------------------------------------------------------ */
enum {
    ST_CONVERGENCE = 0, ST_STOP = 1, ST_ALREADY_MINIMIZED = 2,
    ST_ERR_UNKNOWN = -1024, ST_ERR_LOGIC, ST_ERR_CANCELED, ST_ERR_INVALID_N, ST_ERR_INVALID_MEMSIZE,
    ST_ERR_INVALID_GEPSILON, ST_ERR_INVALID_TESTPERIOD, ST_ERR_INVALID_DELTA, ST_ERR_INVALID_MINSTEP,
    ST_ERR_INVALID_MAXSTEP, ST_ERR_INVALID_FDECCOEFF, ST_ERR_INVALID_SCURVCOEFF, ST_ERR_INVALID_XTOL,
    ST_ERR_INVALID_MAXLINESEARCH, ST_ERR_OUTOFINTERVAL, ST_ERR_INCORRECT_TMINMAX, ST_ERR_ROUNDING,
    ST_ERR_MINIMUMSTEP, ST_ERR_MAXIMUMSTEP, ST_ERR_MAXIMUMLINESEARCH, ST_ERR_MAXIMUMITERATION,
    ST_ERR_WIDTHTOOSMALL, ST_ERR_INVALIDPARAMETERS, ST_ERR_INCREASEGRADIENT
};

/* (device emulation) the kernels keep the d of the control points that are not free — the fixed end points and the
 * lanes beyond N — at zero by giving them zero history; a two-loop coefficient that is not finite (ys = 0: no ys > 0
 * guard, LB:1300) turns that zero into NaN (0 * inf), and through those lanes' partials every later dot product of
 * the same two-loop.  g_emu_poisoned is that state, set and cleared by vgo_lbfgs around its two-loop. */
static int g_emu_poisoned = 0;
static double dotn(const double* a, const double* b, int n) {
    if (g_emu_group && g_emu_poisoned) return NAN;
    if (g_emu_group) {
        /* device emulation: control point p owns elements 3(p-3)..3(p-3)+2 */
        double pts[VIGO_MAX_CTRL_POINTS];
        memset(pts, 0, sizeof(pts));
        for (int i = 0; i < n / 3; ++i)
            pts[i + 3] = FAST ? fma(a[3 * i + 2], b[3 * i + 2], fma(a[3 * i + 1], b[3 * i + 1], a[3 * i] * b[3 * i]))
                              : sum3(a[3 * i] * b[3 * i], a[3 * i + 1] * b[3 * i + 1], a[3 * i + 2] * b[3 * i + 2]);
        return lane_tree_sum(pts, g_emu_group);
    }
    double s = 0.;
    for (int i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

/* one end of the line-search interval: step, value, slope */
typedef struct { double t, f, d; } ls_point;

/* minimiser of the cubic through (u,fu,du),(v,fv,dv)   LB:308-324 */
static double cubic_min(double u, double fu, double du, double v, double fv, double dv) {
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

/* safeguarded variant   LB:338-366 */
static double cubic_min_bounded(double u, double fu, double du, double v, double fv, double dv,
                                double xmin, double xmax) {
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

/* LB:377-379 */
static double quad_min(double u, double fu, double du, double v, double fv) {
    double a = v - u;
    return u + du / ((fu - fv) / a + du) / 2 * a;
}

/* LB:389-391 */
static double quad_min_secant(double u, double du, double v, double dv) {
    double a = u - v;
    return v + dv / (dv - du) * a;
}

/* LB:506-714.  X = best point, Y = other end, T = trial (T->t receives the new trial). */
static int trial_interval(ls_point* X, ls_point* Y, ls_point* T, double tmin, double tmax,
                          int* brackt) {
    int bound;
    int dsign = T->d * (X->d / fabs(X->d)) < 0.;
    double mc, mq, newt;

    if (*brackt) {
        double lo = X->t <= Y->t ? X->t : Y->t;
        double hi = X->t >= Y->t ? X->t : Y->t;
        if (T->t <= lo || hi <= T->t) return ST_ERR_OUTOFINTERVAL;
        if (0. <= X->d * (T->t - X->t)) return ST_ERR_INCREASEGRADIENT;
        if (tmax < tmin) return ST_ERR_INCORRECT_TMINMAX;
    }

    if (X->f < T->f) {                       /* case 1: higher value, minimum bracketed */
        *brackt = 1;
        bound = 1;
        mc = cubic_min(X->t, X->f, X->d, T->t, T->f, T->d);
        mq = quad_min(X->t, X->f, X->d, T->t, T->f);
        newt = (fabs(mc - X->t) < fabs(mq - X->t)) ? mc : mc + 0.5 * (mq - mc);
    } else if (dsign) {                      /* case 2: lower value, slopes of opposite sign */
        *brackt = 1;
        bound = 0;
        mc = cubic_min(X->t, X->f, X->d, T->t, T->f, T->d);
        mq = quad_min_secant(X->t, X->d, T->t, T->d);
        newt = (fabs(mc - T->t) > fabs(mq - T->t)) ? mc : mq;
    } else if (fabs(T->d) < fabs(X->d)) {    /* case 3: lower value, same sign, |slope| shrinks */
        bound = 1;
        mc = cubic_min_bounded(X->t, X->f, X->d, T->t, T->f, T->d, tmin, tmax);
        mq = quad_min_secant(X->t, X->d, T->t, T->d);
        if (*brackt) newt = (fabs(T->t - mc) < fabs(T->t - mq)) ? mc : mq;
        else         newt = (fabs(T->t - mc) > fabs(T->t - mq)) ? mc : mq;
    } else {                                 /* case 4: lower value, same sign, no shrink */
        bound = 0;
        if (*brackt)          newt = cubic_min(T->t, T->f, T->d, Y->t, Y->f, Y->d);
        else if (X->t < T->t) newt = tmax;
        else                  newt = tmin;
    }

    /* interval update LB:664-684 */
    if (X->f < T->f) {
        *Y = *T;
    } else {
        if (dsign) *Y = *X;
        *X = *T;
    }

    if (tmax < newt) newt = tmax;
    if (newt < tmin) newt = tmin;

    if (*brackt && bound) {
        mq = X->t + 0.66 * (Y->t - X->t);
        if (X->t < Y->t) { if (mq < newt) newt = mq; }
        else             { if (newt < mq) newt = mq; }
    }
    T->t = newt;
    return 0;
}

typedef struct {
    int n;
    vgo_eval_fn eval; void* ctx;
    vgo_trace_fn trace; void* tctx;
    const vigo_params_t* P;
    int evals;
} ls_env;

/* LB:716-937 */
static int more_thuente(ls_env* E, double* x, double* f, double* g, const double* s, double* stp,
                        const double* xp, double stpmin, double stpmax) {
    const vigo_params_t* P = E->P;
    const int n = E->n;
    int count = 0, brackt = 0, stage1 = 1, uinfo = 0;
    if (*stp <= 0.) return ST_ERR_INVALIDPARAMETERS;
    double dginit = dotn(g, s, n);
    if (0 < dginit) return ST_ERR_INCREASEGRADIENT;

    double finit = *f;
    double dgtest = P->f_dec_coeff * dginit;
    double width = stpmax - stpmin;
    double prev_width = 2.0 * width;
    ls_point X = {0., finit, dginit}, Y = {0., finit, dginit};
    double stmin, stmax;

    for (;;) {
        if (E->trace) E->trace(E->tctx, x, g, X.f, *stp, n);

        if (brackt) {
            stmin = X.t <= Y.t ? X.t : Y.t;
            stmax = X.t >= Y.t ? X.t : Y.t;
        } else {
            stmin = X.t;
            stmax = *stp + 4.0 * (*stp - X.t);
        }
        if (*stp < stpmin) *stp = stpmin;
        if (stpmax < *stp) *stp = stpmax;

        if ((brackt && ((*stp <= stmin || stmax <= *stp) || P->max_linesearch <= count + 1 || uinfo != 0)) ||
            (brackt && (stmax - stmin <= P->xtol * stmax))) {
            *stp = X.t;
        }

        if (FAST) {
            for (int i = 0; i < n; ++i) x[i] = fma(*stp, s[i], xp[i]);
        } else {
            for (int i = 0; i < n; ++i) x[i] = xp[i];
            for (int i = 0; i < n; ++i) x[i] += *stp * s[i];
        }

        *f = E->eval(E->ctx, x, g, n);
        ++E->evals;
        double dg = dotn(g, s, n);
        double ftest1 = finit + *stp * dgtest;
        ++count;

        if (brackt && ((*stp <= stmin || stmax <= *stp) || uinfo != 0)) return ST_ERR_ROUNDING;
        if (*stp == stpmax && *f <= ftest1 && dg <= dgtest) return ST_ERR_MAXIMUMSTEP;
        if (*stp == stpmin && (ftest1 < *f || dgtest <= dg)) return ST_ERR_MINIMUMSTEP;
        if (brackt && (stmax - stmin) <= P->xtol * stmax) return ST_ERR_WIDTHTOOSMALL;
        if (P->max_linesearch <= count) return ST_ERR_MAXIMUMLINESEARCH;
        if (*f <= ftest1 && fabs(dg) <= P->s_curv_coeff * (-dginit)) return count;

        double cmin = P->f_dec_coeff <= P->s_curv_coeff ? P->f_dec_coeff : P->s_curv_coeff;
        if (stage1 && *f <= ftest1 && cmin * dginit <= dg) stage1 = 0;

        ls_point T;
        if (stage1 && ftest1 < *f && *f <= X.f) {
            /* modified function LB:883-908 */
            ls_point Xm = {X.t, X.f - X.t * dgtest, X.d - dgtest};
            ls_point Ym = {Y.t, Y.f - Y.t * dgtest, Y.d - dgtest};
            T.t = *stp; T.f = *f - *stp * dgtest; T.d = dg - dgtest;
            uinfo = trial_interval(&Xm, &Ym, &T, stmin, stmax, &brackt);
            X.t = Xm.t; Y.t = Ym.t;
            X.f = Xm.f + Xm.t * dgtest;
            Y.f = Ym.f + Ym.t * dgtest;
            X.d = Xm.d + dgtest;
            Y.d = Ym.d + dgtest;
            *stp = T.t;
        } else {
            T.t = *stp; T.f = *f; T.d = dg;
            uinfo = trial_interval(&X, &Y, &T, stmin, stmax, &brackt);
            *stp = T.t;
            /* the reference passes f and dg by pointer (LB:918): when the trial becomes the
             * new best point they are unchanged, so nothing to write back */
        }

        if (brackt) {
            if (0.66 * prev_width <= fabs(Y.t - X.t)) *stp = X.t + 0.5 * (Y.t - X.t);
            prev_width = width;
            width = fabs(Y.t - X.t);
        }
    }
}

/* LB:1024-1349 */
int vgo_lbfgs(int n, double* x, double* fx_out, vgo_eval_fn eval, void* ctx,
              const vigo_params_t* P, int* out_iters, int* out_evals, vgo_trace_fn trace,
              void* tctx) {
    const int m = P->mem_size;
    if (out_iters) *out_iters = 0;
    if (out_evals) *out_evals = 0;
    if (n <= 0) return ST_ERR_INVALID_N;
    if (m <= 0) return ST_ERR_INVALID_MEMSIZE;
    if (P->g_epsilon < 0.) return ST_ERR_INVALID_GEPSILON;
    if (P->past < 0) return ST_ERR_INVALID_TESTPERIOD;
    if (P->delta < 0.) return ST_ERR_INVALID_DELTA;
    if (P->min_step < 0.) return ST_ERR_INVALID_MINSTEP;
    if (P->max_step < P->min_step) return ST_ERR_INVALID_MAXSTEP;
    if (P->f_dec_coeff < 0.) return ST_ERR_INVALID_FDECCOEFF;
    if (P->s_curv_coeff <= P->f_dec_coeff || 1. <= P->s_curv_coeff) return ST_ERR_INVALID_SCURVCOEFF;
    if (P->xtol < 0.) return ST_ERR_INVALID_XTOL;
    if (P->max_linesearch <= 0) return ST_ERR_INVALID_MAXLINESEARCH;

    double* work = (double*)calloc((size_t)n * (4 + 2 * (size_t)m) + 2 * (size_t)m + (size_t)(P->past > 0 ? P->past : 0), sizeof(double));
    double* xp = work;
    double* g = xp + n;
    double* gp = g + n;
    double* d = gp + n;
    double* S = d + n;                    /* m rows of n */
    double* Y = S + (size_t)m * n;        /* m rows of n */
    double* ysv = Y + (size_t)m * n;      /* m */
    double* alpha = ysv + m;              /* m */
    double* pf = P->past > 0 ? alpha + m : NULL;

    ls_env E = {n, eval, ctx, trace, tctx, P, 0};
    int ret;
    int k = 0;
    double fx = eval(ctx, x, g, n);
    ++E.evals;
    if (pf) pf[0] = fx;
    for (int i = 0; i < n; ++i) d[i] = -g[i];

    double xnorm = sqrt(dotn(x, x, n));
    double gnorm = sqrt(dotn(g, g, n));
    if (xnorm < 1.0) xnorm = 1.0;
    if (gnorm / xnorm <= P->g_epsilon) {
        ret = ST_ALREADY_MINIMIZED;
    } else {
        double step = 1.0 / sqrt(dotn(d, d, n));
        int end = 0;
        k = 1;
        for (;;) {
            memcpy(xp, x, sizeof(double) * n);
            memcpy(gp, g, sizeof(double) * n);

            int ls = more_thuente(&E, x, &fx, g, d, &step, xp, P->min_step, P->max_step);
            if (ls < 0) {
                memcpy(x, xp, sizeof(double) * n);
                memcpy(g, gp, sizeof(double) * n);
                ret = ls;
                break;
            }
            xnorm = sqrt(dotn(x, x, n));
            gnorm = sqrt(dotn(g, g, n));
            if (xnorm < 1.0) xnorm = 1.0;
            if (gnorm / xnorm <= P->g_epsilon) { ret = ST_CONVERGENCE; break; }

            if (pf) {
                if (P->past <= k) {
                    double rate = (pf[k % P->past] - fx) / fx;
                    if (fabs(rate) < P->delta) { ret = ST_STOP; break; }
                }
                pf[k % P->past] = fx;
            }
            if (P->max_iterations != 0 && P->max_iterations < k + 1) { ret = ST_ERR_MAXIMUMITERATION; break; }

            double* sv = S + (size_t)end * n;
            double* yv = Y + (size_t)end * n;
            for (int i = 0; i < n; ++i) sv[i] = x[i] - xp[i];
            for (int i = 0; i < n; ++i) yv[i] = g[i] - gp[i];
            double ys = dotn(yv, sv, n);
            double yy = dotn(yv, yv, n);
            ysv[end] = FAST ? 1.0 / ys : ys;   /* FAST keeps the reciprocal of ys */

            int bound = (m <= k) ? m : k;
            ++k;
            end = (end + 1) % m;

            for (int i = 0; i < n; ++i) d[i] = -g[i];
            int j = end;
            g_emu_poisoned = 0;
#define POISON_IF_NOT_FINITE(c) do { if (g_emu_group && !((c) - (c) == 0.0)) g_emu_poisoned = 1; } while (0)
            for (int i = 0; i < bound; ++i) {
                j = (j + m - 1) % m;
                const double* sj = S + (size_t)j * n;
                const double* yj = Y + (size_t)j * n;
                alpha[j] = dotn(sj, d, n);
                if (FAST) alpha[j] *= ysv[j]; else alpha[j] /= ysv[j];
                double na = -alpha[j];
                if (FAST) { for (int e = 0; e < n; ++e) d[e] = fma(na, yj[e], d[e]); }
                else      { for (int e = 0; e < n; ++e) d[e] += na * yj[e]; }
                POISON_IF_NOT_FINITE(na);
            }
            double sc = ys / yy;
            for (int e = 0; e < n; ++e) d[e] *= sc;
            POISON_IF_NOT_FINITE(sc);
            for (int i = 0; i < bound; ++i) {
                const double* sj = S + (size_t)j * n;
                const double* yj = Y + (size_t)j * n;
                double beta = dotn(yj, d, n);
                if (FAST) beta *= ysv[j]; else beta /= ysv[j];
                double co = alpha[j] - beta;
                if (FAST) { for (int e = 0; e < n; ++e) d[e] = fma(co, sj[e], d[e]); }
                else      { for (int e = 0; e < n; ++e) d[e] += co * sj[e]; }
                POISON_IF_NOT_FINITE(co);
                j = (j + 1) % m;
            }
#undef POISON_IF_NOT_FINITE
            g_emu_poisoned = 0;   /* the kernels reset the d of those points after the two-loop */
            step = 1.0;
        }
    }
    if (fx_out) *fx_out = fx;
    if (out_iters) *out_iters = k;
    if (out_evals) *out_evals = E.evals;
    free(work);
    return ret;
}

/* ------------------------------------------------------------------------------------ */
/* BT.cpp:687-718 + :796-821                                                             */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    const vigo_params_t* P;
    int N;
    double* ctrl;
    const int32_t* goff; const double* gpv; const uint8_t* gunk;
    int n_obs; const double* obs;
    const double* w;
} solve_ctx;

static double solve_eval(void* vctx, const double* x, double* g, int n) {
    solve_ctx* S = (solve_ctx*)vctx;
    memcpy(S->ctrl + 9, x, sizeof(double) * n); /* BT.cpp:803: controlPoints keeps the last trial */
    return vgo_cost_grad(S->P, S->N, S->ctrl, S->goff, S->gpv, S->gunk, S->n_obs, S->obs, S->w, g, NULL, NULL);
}

int vgo_optimize(const vigo_params_t* P, int N, double* ctrl, const int32_t* goff,
                 const double* gpv, const uint8_t* gunk, int n_obs, const double* obs,
                 const double w[4], double* x_out, double* fx_out, int* iters, int* evals) {
    int n = 3 * (N - 6);
    double x[3 * VGO_MAX_N];
    memcpy(x, ctrl + 9, sizeof(double) * n);
    solve_ctx S = {P, N, ctrl, goff, gpv, gunk, n_obs, obs, w};
    g_level_pinned = traj_level(P, N, ctrl);      /* (device emulation) the level rule's decision for this solve */
    int ret = vgo_lbfgs(n, x, fx_out, solve_eval, &S, P, iters, evals, NULL, NULL);
    g_level_pinned = -1;
    if (x_out) memcpy(x_out, x, sizeof(double) * n);
    return ret;
}

static void batch_slice(int b, int N, const int32_t* guide_off, const int32_t* obs_off,
                        const double* obs, int n_obs_shared, const double* weights,
                        const vigo_params_t* P, const int32_t** goff, int* n_obs,
                        const double** obs_b, double w[4]) {
    *goff = guide_off + (size_t)b * N;
    if (obs_off) {
        *n_obs = obs_off[b + 1] - obs_off[b];
        *obs_b = obs + 9 * (size_t)obs_off[b];
    } else {
        *n_obs = n_obs_shared;
        *obs_b = obs;
    }
    if (weights) {
        for (int q = 0; q < 4; ++q) w[q] = weights[4 * (size_t)b + q];
    } else {
        w[0] = P->w_distance; w[1] = P->w_smoothness; w[2] = P->w_feasibility; w[3] = P->w_dynamic;
    }
}

void vgo_cost_grad_batch(const vigo_params_t* P, int B, int N, const double* ctrl,
                         const int32_t* guide_off, const double* gpv, const uint8_t* gunk,
                         const int32_t* obs_off, const double* obs, int n_obs_shared,
                         const double* weights, double* out_cost, double* out_grad,
                         double* out_terms) {
    int n = 3 * (N - 6);
    for (int b = 0; b < B; ++b) {
        const int32_t* goff; int n_obs; const double* ob; double w[4];
        batch_slice(b, N, guide_off, obs_off, obs, n_obs_shared, weights, P, &goff, &n_obs, &ob, w);
        out_cost[b] = vgo_cost_grad(P, N, ctrl + 3 * (size_t)N * b, goff, gpv, gunk, n_obs, ob, w,
                                    out_grad ? out_grad + (size_t)n * b : NULL, NULL,
                                    out_terms ? out_terms + 4 * (size_t)b : NULL);
    }
}

void vgo_optimize_batch(const vigo_params_t* P, int B, int N, double* ctrl,
                        const int32_t* guide_off, const double* gpv, const uint8_t* gunk,
                        const int32_t* obs_off, const double* obs, int n_obs_shared,
                        const double* weights, double* out_x, int32_t* out_status,
                        double* out_fx, int32_t* out_iters, int32_t* out_evals) {
    int n = 3 * (N - 6);
    for (int b = 0; b < B; ++b) {
        const int32_t* goff; int n_obs; const double* ob; double w[4];
        batch_slice(b, N, guide_off, obs_off, obs, n_obs_shared, weights, P, &goff, &n_obs, &ob, w);
        double fx; int it, ev;
        int st = vgo_optimize(P, N, ctrl + 3 * (size_t)N * b, goff, gpv, gunk, n_obs, ob, w,
                              out_x ? out_x + (size_t)n * b : NULL, &fx, &it, &ev);
        if (out_status) out_status[b] = st;
        if (out_fx) out_fx[b] = fx;
        if (out_iters) out_iters[b] = it;
        if (out_evals) out_evals[b] = ev;
    }
}

/* ------------------------------------------------------------------------------------ */
/* uniform B-spline  BS.cpp:19-72                                                        */
/* ------------------------------------------------------------------------------------ */
static inline double knot(int i, int degree, double ts) { return (i - degree) * ts; }

void vgo_bspline_at(int degree, int ncp, const double* cp, double ts, double t, double out[3]) {
    int knotsNum = ncp - 1 + degree + 1 + 1;
    double duration = knot(knotsNum - degree - 1, degree, ts);
    double tb = fmin(fmax(0.0, t), duration);
    int k = degree;
    while (1) {
        if (knot(k + 1, degree, ts) >= tb) break;
        ++k;
    }
    double d[8][3];
    for (int i = 0; i <= degree; ++i)
        for (int a = 0; a < 3; ++a) d[i][a] = cp[3 * (k - degree + i) + a];
    for (int r = 1; r <= degree; ++r) {
        for (int i = degree; i >= r; --i) {
            double alpha = (tb - knot(i + k - degree, degree, ts)) /
                           (knot(i + 1 + k - r, degree, ts) - knot(i + k - degree, degree, ts));
            for (int a = 0; a < 3; ++a) d[i][a] = (1 - alpha) * d[i - 1][a] + alpha * d[i][a];
        }
    }
    for (int a = 0; a < 3; ++a) out[a] = d[degree][a];
}

/* bspline::parameterizeToBspline, BS.cpp:74-138: the (K+4)x(K+2) system and its least-squares
 * solution by column-pivoted Householder QR (the decomposition the reference asks Eigen for,
 * BS.cpp:129-131; Eigen itself is absent here, so the pivot order is this file's own: largest
 * remaining column norm, recomputed every step).  points [K][3], cond [4][3] (vel0, vel1, acc0,
 * acc1), ctrl_out [K+2][3].  Returns 0, or -1 for K < 4 / ts <= 0 (the reference exit(0)s). */
int vgo_bspline_fit(int K, double ts, const double* points, const double* cond, double* ctrl_out) {
    if (ts <= 0 || K <= 3) return -1;
    const int R = K + 4, Cn = K + 2;
    double* A = (double*)calloc((size_t)R * Cn, sizeof(double));
    double* b = (double*)calloc((size_t)R * 3, sizeof(double));
    int* perm = (int*)malloc(sizeof(int) * Cn);
    double* v = (double*)malloc(sizeof(double) * R);
#define AT(r, c) A[(size_t)(r) * Cn + (c)]
    for (int i = 0; i < K; ++i) {                       /* BS.cpp:102-104 */
        AT(i, i) = (1 / 6.0) * 1; AT(i, i + 1) = (1 / 6.0) * 4; AT(i, i + 2) = (1 / 6.0) * 1;
        for (int a = 0; a < 3; ++a) b[3 * i + a] = points[3 * i + a];
    }
    AT(K, 0) = (1 / 2.0 / ts) * -1; AT(K, 2) = (1 / 2.0 / ts) * 1;                    /* :106 */
    AT(K + 1, K - 1) = (1 / 2.0 / ts) * -1; AT(K + 1, K + 1) = (1 / 2.0 / ts) * 1;    /* :107 */
    AT(K + 2, 0) = (1 / ts / ts) * 1; AT(K + 2, 1) = (1 / ts / ts) * -2; AT(K + 2, 2) = (1 / ts / ts) * 1;              /* :108 */
    AT(K + 3, K - 1) = (1 / ts / ts) * 1; AT(K + 3, K) = (1 / ts / ts) * -2; AT(K + 3, K + 1) = (1 / ts / ts) * 1;      /* :109 */
    for (int i = 0; i < 4; ++i)
        for (int a = 0; a < 3; ++a) b[3 * (K + i) + a] = cond ? cond[3 * i + a] : 0.0;    /* :119-123 */
    for (int c = 0; c < Cn; ++c) perm[c] = c;
    for (int c = 0; c < Cn; ++c) {
        int best = c;
        double bestn = -1;
        for (int cc = c; cc < Cn; ++cc) {
            double n2 = 0;
            for (int r = c; r < R; ++r) n2 += AT(r, cc) * AT(r, cc);
            if (n2 > bestn) { bestn = n2; best = cc; }
        }
        if (best != c) {
            for (int r = 0; r < R; ++r) { double t = AT(r, c); AT(r, c) = AT(r, best); AT(r, best) = t; }
            int t = perm[c]; perm[c] = perm[best]; perm[best] = t;
        }
        double nrm = sqrt(bestn);
        if (nrm == 0) break;
        double alpha = AT(c, c) > 0 ? -nrm : nrm;
        double vn = 0;
        for (int r = c; r < R; ++r) v[r] = AT(r, c);
        v[c] -= alpha;
        for (int r = c; r < R; ++r) vn += v[r] * v[r];
        if (vn == 0) continue;
        for (int cc = c; cc < Cn; ++cc) {
            double sdot = 0;
            for (int r = c; r < R; ++r) sdot += v[r] * AT(r, cc);
            sdot = 2 * sdot / vn;
            for (int r = c; r < R; ++r) AT(r, cc) -= sdot * v[r];
        }
        for (int a = 0; a < 3; ++a) {
            double sdot = 0;
            for (int r = c; r < R; ++r) sdot += v[r] * b[3 * r + a];
            sdot = 2 * sdot / vn;
            for (int r = c; r < R; ++r) b[3 * r + a] -= sdot * v[r];
        }
    }
    for (int a = 0; a < 3; ++a) {
        for (int r = Cn - 1; r >= 0; --r) {
            double sacc = b[3 * r + a];
            for (int cc = r + 1; cc < Cn; ++cc) sacc -= AT(r, cc) * v[cc];
            v[r] = sacc / AT(r, r);
        }
        for (int r = 0; r < Cn; ++r) ctrl_out[3 * perm[r] + a] = v[r];
    }
#undef AT
    free(A); free(b); free(perm); free(v);
    return 0;
}

void vgo_bspline_fit_batch(int B, int K, double ts, const double* points, const double* conds, double* ctrl_out) {
    for (int i = 0; i < B; ++i)
        vgo_bspline_fit(K, ts, points + (size_t)i * K * 3, conds ? conds + (size_t)i * 12 : NULL,
                        ctrl_out + (size_t)i * (K + 2) * 3);
}

void vgo_bspline_derivative(int degree, int ncp, const double* cp, double ts, double* out) {
    for (int i = 0; i < ncp - 1; ++i) {
        double den = knot(i + degree + 1, degree, ts) - knot(i + 1, degree, ts);
        for (int a = 0; a < 3; ++a)
            out[3 * i + a] = degree * (cp[3 * (i + 1) + a] - cp[3 * i + a]) / den;
    }
}

void vgo_traj_eval(int N, const double* ctrl, double ts_ctrl, int deriv, double t, double out[3]) {
    if (deriv == 0) { vgo_bspline_at(3, N, ctrl, ts_ctrl, t, out); return; }
    double v[3 * VGO_MAX_N];
    vgo_bspline_derivative(3, N, ctrl, ts_ctrl, v);
    if (deriv == 1) { vgo_bspline_at(2, N - 1, v, ts_ctrl, t, out); return; }
    double a[3 * VGO_MAX_N];
    vgo_bspline_derivative(2, N - 1, v, ts_ctrl, a);
    vgo_bspline_at(1, N - 2, a, ts_ctrl, t, out);
}

int vgo_sample_times(double tmax, double dt, double* times, int cap) {
    int k = 0;
    for (double t = 0; t <= tmax; t += dt) {
        if (times && k < cap) times[k] = t;
        ++k;
    }
    return k;
}

/* BT.h:307-325 */
static int traj_collision_ncr(const vgo_grid_t* g, int N, const double* ctrl, double ts_ctrl, double dt,
                              double not_check_ratio, int* first_idx);
int vgo_traj_collision(const vgo_grid_t* g, int N, const double* ctrl, double ts_ctrl, double dt,
                       int* first_idx) {
    return traj_collision_ncr(g, N, ctrl, ts_ctrl, dt, 0.0, first_idx);   /* notCheckRatio_ = 0.0, BT.h:58 */
}
/* BT.h:307-325 with notCheckRatio_ as a parameter (BT.h:313) */
static int traj_collision_ncr(const vgo_grid_t* g, int N, const double* ctrl, double ts_ctrl, double dt,
                              double not_check_ratio, int* first_idx) {
    double duration = (N - 3) * ts_ctrl; /* knots(N) BS.cpp:27 */
    int k = 0;
    if (first_idx) *first_idx = -1;
    for (double t = 0; t <= (1.0 - not_check_ratio) * duration; t += dt, ++k) {
        double p[3];
        vgo_bspline_at(3, N, ctrl, ts_ctrl, t, p);
        if (vgo_is_inflated_occupied(g, p)) {
            if (first_idx) *first_idx = k;
            return 1;
        }
    }
    return 0;
}

/* BT.h:344-368 */
int vgo_traj_dynamic_collision(int N, const double* ctrl, double ts_ctrl, double dt, int n_obs,
                               const double* obs) {
    double duration = (N - 3) * ts_ctrl;
    for (double t = 0; t <= duration; t += dt) {
        double p[3];
        vgo_bspline_at(3, N, ctrl, ts_ctrl, t, p);
        for (int i = 0; i < n_obs; ++i) {
            const double* op = obs + 9 * (size_t)i;
            double size = fmin(op[6] / 2, op[7] / 2);
            double dx = p[0] - op[0], dy = p[1] - op[1];
            double dist = sqrt(sum3(dx * dx, dy * dy, 0.0)) - size;
            if (dist < 0) return 1;
        }
    }
    return 0;
}

void vgo_ctrl_occupancy(const vgo_grid_t* g, int N, const double* ctrl, uint8_t* pt, uint8_t* line) {
    for (int i = 0; i < N; ++i) {
        pt[i] = (uint8_t)vgo_is_inflated_occupied(g, ctrl + 3 * i);
        line[i] = (i == 0) ? 0 : (uint8_t)vgo_is_inflated_occupied_line(g, ctrl + 3 * (i - 1), ctrl + 3 * i);
    }
}

/* ------------------------------------------------------------------------------------ */
/* the rebound loop's bookkeeping between two A* calls (BT.cpp:403-445, :573-685)         */
/* ------------------------------------------------------------------------------------ */
/* findCollisionSeg, BT.cpp:403-445: pairs (first, second) into seg[2 * cap]; returns how many it found (may exceed cap) */
int vgo_find_collision_seg(const vgo_grid_t* g, int N, const double* ctrl, double not_check_ratio, int32_t* seg, int cap) {
    int n = 0;
    int previousHasCollision = 0;
    int endIdx = (int)((N - 3 - 1) - not_check_ratio * (N - 2 * 3));
    int pairStartIdx = 3, pairEndIdx = 3;
    for (int i = 3; i <= endIdx; ++i) {
        const double* p = ctrl + 3 * i;
        int hasCollision = vgo_is_inflated_occupied(g, p);
        if (hasCollision != previousHasCollision) {
            if (hasCollision) {
                pairStartIdx = i - 1;
            } else {
                pairEndIdx = i;
                if (n < cap) { seg[2 * n] = pairStartIdx; seg[2 * n + 1] = pairEndIdx; }
                ++n;
            }
        }
        if (hasCollision && i == endIdx - 1) { /* corner case, BT.cpp:426-430 */
            pairEndIdx = N - 1;
            if (n < cap) { seg[2 * n] = pairStartIdx; seg[2 * n + 1] = pairEndIdx; }
            ++n;
        }
        if (i != 3 && !previousHasCollision && !hasCollision) {
            if (vgo_is_inflated_occupied_line(g, ctrl + 3 * (i - 1), p)) {
                if (n < cap) { seg[2 * n] = i - 1; seg[2 * n + 1] = i; }
                ++n;
            }
        }
        previousHasCollision = hasCollision;
    }
    return n;
}

static int index_in_seg(const int32_t* seg, int n, int idx) { /* BT.h:370-377 */
    for (int k = 0; k < n; ++k)
        if (idx >= seg[2 * k] && idx <= seg[2 * k + 1]) return 1;
    return 0;
}

/* isControlPointRequireNewGuide, BT.h:417-429 (guide pairs of control point i in CSR form) */
static int require_new_guide(const vigo_params_t* P, int N, const double* ctrl, const int32_t* goff, const double* gpv, int i) {
    (void)N;
    const double* c = ctrl + 3 * i;
    if (goff && gpv) {
        for (int j = goff[i]; j < goff[i + 1]; ++j) {
            const double* pv = gpv + 6 * (size_t)j;
            double dist = sum3((c[0] - pv[0]) * pv[3], (c[1] - pv[1]) * pv[4], (c[2] - pv[2]) * pv[5]);
            double distErr = P->dthresh - dist;
            if (distErr > 0) return 0;
        }
    }
    return 1;
}

/* isReguideRequired, BT.cpp:573-608 with compareCollisionSeg (BT.h:379-403): 1 when a segment asks for A*.  new_seg
 * receives findCollisionSeg of the current control points (what collisionSeg_ becomes), *n_new its size. */
int vgo_is_reguide_required(const vigo_params_t* P, const vgo_grid_t* g, int N, const double* ctrl, const int32_t* goff,
                            const double* gpv, const int32_t* prev_seg, int n_prev, double not_check_ratio, int32_t* new_seg,
                            int cap, int* n_new) {
    int n = vgo_find_collision_seg(g, N, ctrl, not_check_ratio, new_seg, cap);
    *n_new = n;
    if (n > cap) return 1; /* (more than the state holds: the device hands such a trajectory to the host) */
    int need = 0;
    for (int k = 0; k < n; ++k) {
        int first = new_seg[2 * k], second = new_seg[2 * k + 1];
        for (int i = first + 1; i <= second - 1; ++i) {
            if (index_in_seg(prev_seg, n_prev, i)) need |= require_new_guide(P, N, ctrl, goff, gpv, i); /* overlapped point */
            else need = 1;                                                                              /* new collision point */
        }
        if (second - first - 1 == 0) { /* line collision */
            for (int i = first; i <= second; ++i) {
                if (index_in_seg(prev_seg, n_prev, i)) need |= require_new_guide(P, N, ctrl, goff, gpv, i);
                else need = 1;
            }
        }
    }
    return need;
}

/* One pass of the loop body of optimizeTrajectory (BT.cpp:619-679) for a trajectory, as far as it needs no A*:
 * gates, success exit, the failCount >= 4 hand-over, isReguideRequired, weight doubling.  st follows
 * vigo_rebound_state_t (include/vigo.h); weights = (distance, smoothness, feasibility, dynamic).  Returns the status. */
int vgo_rebound_decide(const vigo_params_t* P, const vgo_grid_t* g, int N, const double* ctrl, const int32_t* goff,
                       const double* gpv, int n_obs, const double* obs, double gate_dt, double not_check_ratio,
                       double* weights, vigo_rebound_state_t* st) {
    int hasCollision = traj_collision_ncr(g, N, ctrl, P->ts_ctrl, gate_dt, not_check_ratio, NULL);
    int hasDynamicCollision = n_obs > 0 ? vgo_traj_dynamic_collision(N, ctrl, P->ts_ctrl, gate_dt, n_obs, obs) : 0;
    st->gate_static = hasCollision;
    st->gate_dynamic = hasDynamicCollision;
    st->rounds += 1;
    if (!hasCollision && !hasDynamicCollision) { st->status = VIGO_RB_DONE; return st->status; }   /* BT.cpp:628-631 */
    if (st->fail_count >= 4) { st->status = VIGO_RB_NEEDS_HOST; return st->status; }               /* BT.cpp:640-648: A* */
    if (hasCollision) {
        int32_t new_seg[2 * VIGO_MAX_COLLISION_SEGS];
        int n_new = 0;
        int n_prev = st->n_seg < 0 ? 0 : (st->n_seg > VIGO_MAX_COLLISION_SEGS ? VIGO_MAX_COLLISION_SEGS : st->n_seg);
        if (vgo_is_reguide_required(P, g, N, ctrl, goff, gpv, st->seg, n_prev, not_check_ratio, new_seg,
                                    VIGO_MAX_COLLISION_SEGS, &n_new)) {
            st->status = VIGO_RB_NEEDS_HOST;                                                         /* BT.cpp:659-665: A* */
            return st->status;
        }
        st->n_seg = n_new;                                                                           /* BT.cpp:575 */
        for (int k = 0; k < 2 * n_new; ++k) st->seg[k] = new_seg[k];
        weights[0] *= 2.0;                                                                           /* BT.cpp:672 */
        st->fail_count += 1;
    }
    if (hasDynamicCollision) weights[3] *= 2.0;                                                      /* BT.cpp:677-679 */
    return st->status;
}

/* ------------------------------------------------------------------------------------ */
/* corridor checker  PO.cpp:547-589, octomap OcTree::search semantics on the dense grid   */
/* ------------------------------------------------------------------------------------ */
static int oct_point_collides(const vgo_grid_t* g, float x, float y, float z) {
    /* PO.cpp:572-577 metric bounds */
    if (x < g->bmin[0] || x > g->bmax[0] || y < g->bmin[1] || y > g->bmax[1] ||
        z < g->bmin[2] || z > g->bmax[2])
        return 1;
    /* octomap coordToKey: (int)floor(resolution_factor * coordinate) (+32768, cancelled by the
     * grid's own key offset).  The grid origin must be a multiple of res. */
    double rf = 1.0 / g->res;
    int kx = (int)floor(rf * (double)x) - (int)floor(g->origin[0] / g->res + 0.5);
    int ky = (int)floor(rf * (double)y) - (int)floor(g->origin[1] / g->res + 0.5);
    int kz = (int)floor(rf * (double)z) - (int)floor(g->origin[2] / g->res + 0.5);
    if (kx < 0 || ky < 0 || kz < 0 || kx >= g->nx || ky >= g->ny || kz >= g->nz) return 1; /* NULL node */
    unsigned v = g->vox[((size_t)kx * g->ny + ky) * g->nz + kz];
    if (v & 2u) return 1;          /* unknown: search() == NULL and ignoreUnknown == false */
    return (v & 4u) ? 1 : 0;       /* isNodeOccupied */
}

int vgo_box_collision(const vgo_grid_t* g, float px, float py, float pz, const double box[3],
                      double map_res) {
    double xmin = px - box[0] / 2, xmax = px + box[0] / 2;
    double ymin = py - box[1] / 2, ymax = py + box[1] / 2;
    double zmin = pz - box[2] / 2, zmax = pz + box[2] / 2;
    int xNum = (int)((xmax - xmin) / map_res);
    int yNum = (int)((ymax - ymin) / map_res);
    int zNum = (int)((zmax - zmin) / map_res);
    for (int xi = 0; xi <= xNum; ++xi)
        for (int yi = 0; yi <= yNum; ++yi)
            for (int zi = 0; zi <= zNum; ++zi) {
                float qx = (float)(xmin + xi * map_res);
                float qy = (float)(ymin + yi * map_res);
                float qz = (float)(zmin + zi * map_res);
                if (oct_point_collides(g, qx, qy, qz)) return 1;
            }
    return 0;
}

/* pow(t, d) of PS.cpp:1035-1039.  Mode 0: libm's pow(), i.e. the reference as it runs on THIS host (glibc's pow
 * is within 0.51 ulp: the correctly rounded power in ~99.9 % of the cases, its neighbour otherwise, and which one
 * depends on the libm build).  Mode 1: the correctly rounded power by exact integer arithmetic — the
 * platform-independent definition the HIP sampler implements (csrc/vigo_exact_pow.hpp).  Written independently of
 * that header: a base-2^32 big integer with 64-bit accumulation. */
static int g_pow_mode = 0;
void vgo_set_pow_mode(int exact) { g_pow_mode = exact ? 1 : 0; }
int vgo_get_pow_mode(void) { return g_pow_mode; }

double vgo_pow_exact(double t, int d) {
    if (d <= 0) return 1.0;
    if (d == 1) return t;
    if (t != t || t == 0.0 || t == INFINITY || t == -INFINITY) return pow(t, (double)d);   /* exact specials */
    int e2;
    double fr = frexp(fabs(t), &e2);                       /* |t| = fr * 2^e2, fr in [0.5, 1) */
    uint64_t m = (uint64_t)ldexp(fr, 53);                  /* 53-bit integer significand (exact, subnormals too) */
    long long ex = (long long)e2 - 53;                     /* |t| = m * 2^ex */
    uint32_t limb[64];
    int n = 1;
    memset(limb, 0, sizeof limb);
    limb[0] = 1;
    for (int i = 0; i < d; ++i) {                          /* limb <- limb * m, schoolbook in base 2^32 */
        const uint32_t digit[2] = {(uint32_t)m, (uint32_t)(m >> 32)};
        uint32_t out[64];
        memset(out, 0, sizeof out);
        for (int k = 0; k < 2; ++k) {
            uint64_t carry = 0;
            for (int j = 0; j < n; ++j) {
                const uint64_t cur = (uint64_t)out[j + k] + (uint64_t)limb[j] * digit[k] + carry;
                out[j + k] = (uint32_t)cur;
                carry = cur >> 32;
            }
            for (int idx = n + k; carry; ++idx) {
                const uint64_t cur = (uint64_t)out[idx] + carry;
                out[idx] = (uint32_t)cur;
                carry = cur >> 32;
            }
        }
        n += 2;
        while (n > 1 && out[n - 1] == 0) --n;
        memcpy(limb, out, sizeof out);
    }
    long long E = ex * d;                                  /* t^d = limb * 2^E */
    int topw = n - 1, topb = 31;
    while (!((limb[topw] >> topb) & 1u)) --topb;
    long long top = 32LL * topw + topb;                    /* index of the leading bit */
    long long exp2 = top + E;
    int neg = (t < 0) && (d & 1);
    if (exp2 > 1023) return neg ? -INFINITY : INFINITY;
    long long keep = exp2 >= -1022 ? 53 : 53 - (-1022 - exp2);
    if (keep < 0) return neg ? -0.0 : 0.0;
    long long drop = top + 1 - keep;                       /* low bits to round away */
    uint64_t q = 0;
    if (drop <= 0) {
        for (long long i = top; i >= 0; --i) q = (q << 1) | ((limb[i >> 5] >> (i & 31)) & 1u);
        drop = 0;
    } else {
        for (long long i = top; i >= drop; --i) q = (q << 1) | ((limb[i >> 5] >> (i & 31)) & 1u);
        int half = (limb[(drop - 1) >> 5] >> ((drop - 1) & 31)) & 1u, rest = 0;
        for (long long i = drop - 2; i >= 0 && !rest; --i) rest = (limb[i >> 5] >> (i & 31)) & 1u;
        if (half && (rest || (q & 1u))) ++q;               /* round to nearest, ties to even */
    }
    double r = ldexp((double)q, (int)(E + drop));          /* representable by construction */
    return neg ? -r : r;
}

static inline double POWD(double t, int d) { return g_pow_mode ? vgo_pow_exact(t, d) : pow(t, d); }

void vgo_poly_pos(int deg, const double* cx, const double* cy, const double* cz, double t,
                  double out[3]) {
    double x = 0, y = 0, z = 0;
    for (int d = 0; d < deg + 1; ++d) {
        const double pw = POWD(t, d);
        x += cx[d] * pw;
        y += cy[d] * pw;
        z += cz[d] * pw;
    }
    out[0] = x; out[1] = y; out[2] = z;
}

/* PS.cpp:1125-1137 for S segments: sample k of segment s at out[(s * stride + k) * 3], accumulated clock */
void vgo_poly_sample(int S, int deg, const double* coeffs, const int32_t* n_samp, const double* delT, int stride,
                     double* out_pos, float* out_f32) {
    for (int s = 0; s < S; ++s) {
        const double* c = coeffs + (size_t)s * 3 * (deg + 1);
        double t = 0;
        for (int k = 0; k < n_samp[s] && k < stride; ++k) {
            double p[3];
            vgo_poly_pos(deg, c, c + (deg + 1), c + 2 * (deg + 1), t, p);
            for (int a = 0; a < 3; ++a) {
                if (out_pos) out_pos[((size_t)s * stride + k) * 3 + a] = p[a];
                if (out_f32) out_f32[((size_t)s * stride + k) * 3 + a] = (float)p[a];
            }
            t += delT[s];
        }
    }
}

int vgo_corridor_check_segment(const vgo_grid_t* g, int deg, const double* coeffs, int n_samp,
                               double delT, const double box[3], double map_res, int* first_idx,
                               int* count) {
    const double* cx = coeffs;
    const double* cy = coeffs + (deg + 1);
    const double* cz = coeffs + 2 * (deg + 1);
    int first = -1, cnt = 0;
    double t = 0;
    for (int k = 0; k < n_samp; ++k) {
        double p[3];
        vgo_poly_pos(deg, cx, cy, cz, t, p);
        if (vgo_box_collision(g, (float)p[0], (float)p[1], (float)p[2], box, map_res)) {
            if (first < 0) first = k;
            ++cnt;
        }
        t += delT;
    }
    if (first_idx) *first_idx = first;
    if (count) *count = cnt;
    return cnt > 0;
}

void vgo_corridor_check_batch(const vgo_grid_t* g, int S, int deg, const double* coeffs, const int32_t* n_samp,
                              const double* delT, const double box[3], double map_res, uint8_t* flag,
                              int32_t* first, int32_t* count) {
    for (int s = 0; s < S; ++s) {
        int fi = -1, cn = 0;
        flag[s] = (uint8_t)vgo_corridor_check_segment(g, deg, coeffs + (size_t)s * 3 * (deg + 1), n_samp[s], delT[s], box,
                                                      map_res, &fi, &cn);
        if (first) first[s] = fi;
        if (count) count[s] = cn;
    }
}

/* ------------------------------------------------------------------------------------ */
/* trilinear ESDF (own definition: samples at voxel centres, clamped to the lattice)      */
/* ------------------------------------------------------------------------------------ */
void vgo_esdf_query(int nx, int ny, int nz, const double origin[3], double res, const float* dist,
                    const double p[3], double* out_d, double out_g[3]) {
    int n[3] = {nx, ny, nz};
    int i0[3];
    double f[3];
    for (int a = 0; a < 3; ++a) {
        double u = (p[a] - origin[a]) / res - 0.5;
        double fl = floor(u);
        int i = (int)fl;
        double fr = u - fl;
        if (i < 0) { i = 0; fr = 0.0; }
        if (i > n[a] - 2) { i = n[a] - 2; fr = 1.0; }
        i0[a] = i; f[a] = fr;
    }
    double v[2][2][2];
    for (int dx = 0; dx < 2; ++dx)
        for (int dy = 0; dy < 2; ++dy)
            for (int dz = 0; dz < 2; ++dz)
                v[dx][dy][dz] = (double)dist[((size_t)(i0[0] + dx) * ny + (i0[1] + dy)) * nz + (i0[2] + dz)];
    double c00 = v[0][0][0] * (1 - f[0]) + v[1][0][0] * f[0];
    double c01 = v[0][0][1] * (1 - f[0]) + v[1][0][1] * f[0];
    double c10 = v[0][1][0] * (1 - f[0]) + v[1][1][0] * f[0];
    double c11 = v[0][1][1] * (1 - f[0]) + v[1][1][1] * f[0];
    double c0 = c00 * (1 - f[1]) + c10 * f[1];
    double c1 = c01 * (1 - f[1]) + c11 * f[1];
    *out_d = c0 * (1 - f[2]) + c1 * f[2];
    /* d/dx */
    double gx00 = v[1][0][0] - v[0][0][0], gx01 = v[1][0][1] - v[0][0][1];
    double gx10 = v[1][1][0] - v[0][1][0], gx11 = v[1][1][1] - v[0][1][1];
    double gx0 = gx00 * (1 - f[1]) + gx10 * f[1];
    double gx1 = gx01 * (1 - f[1]) + gx11 * f[1];
    out_g[0] = (gx0 * (1 - f[2]) + gx1 * f[2]) / res;
    /* d/dy */
    double gy0 = c10 - c00, gy1 = c11 - c01;
    out_g[1] = (gy0 * (1 - f[2]) + gy1 * f[2]) / res;
    /* d/dz */
    out_g[2] = (c1 - c0) / res;
}

/* The fp32 twin of the query (include/vigo.h vigo_esdf_query_f32): the same definition with every operation in float,
 * each rounded once (-ffp-contract=off; FLT_EVAL_METHOD == 0 on x86-64), the lattice scale as ONE float reciprocal.
 * out4 = {d, gx, gy, gz}. */
void vgo_esdf_query_f32(int nx, int ny, int nz, const double origin[3], double res, const float* dist,
                        const float p[3], float out4[4]) {
    int n[3] = {nx, ny, nz};
    int i0[3];
    float f[3];
    const float inv_res = 1.0f / (float)res;
    for (int a = 0; a < 3; ++a) {
        float u = (p[a] - (float)origin[a]) * inv_res - 0.5f;
        float fl = floorf(u);
        int i;
        float fr = u - fl;
        if (!(fl >= 0.0f)) { i = 0; fr = 0.0f; }
        else if (fl > (float)(n[a] - 2)) { i = n[a] - 2; fr = 1.0f; }
        else i = (int)fl;
        i0[a] = i; f[a] = fr;
    }
    float v[2][2][2];
    for (int dx = 0; dx < 2; ++dx)
        for (int dy = 0; dy < 2; ++dy)
            for (int dz = 0; dz < 2; ++dz)
                v[dx][dy][dz] = dist[((size_t)(i0[0] + dx) * ny + (i0[1] + dy)) * nz + (i0[2] + dz)];
    float c00 = v[0][0][0] * (1 - f[0]) + v[1][0][0] * f[0];
    float c01 = v[0][0][1] * (1 - f[0]) + v[1][0][1] * f[0];
    float c10 = v[0][1][0] * (1 - f[0]) + v[1][1][0] * f[0];
    float c11 = v[0][1][1] * (1 - f[0]) + v[1][1][1] * f[0];
    float c0 = c00 * (1 - f[1]) + c10 * f[1];
    float c1 = c01 * (1 - f[1]) + c11 * f[1];
    out4[0] = c0 * (1 - f[2]) + c1 * f[2];
    float gx00 = v[1][0][0] - v[0][0][0], gx01 = v[1][0][1] - v[0][0][1];
    float gx10 = v[1][1][0] - v[0][1][0], gx11 = v[1][1][1] - v[0][1][1];
    float gx0 = gx00 * (1 - f[1]) + gx10 * f[1];
    float gx1 = gx01 * (1 - f[1]) + gx11 * f[1];
    out4[1] = (gx0 * (1 - f[2]) + gx1 * f[2]) * inv_res;
    float gy0 = c10 - c00, gy1 = c11 - c01;
    out4[2] = (gy0 * (1 - f[2]) + gy1 * f[2]) * inv_res;
    out4[3] = (c1 - c0) * inv_res;
}

void vgo_esdf_query_f32_batch(int nx, int ny, int nz, const double origin[3], double res, const float* dist, int64_t Q,
                              const float* pts, float* out4) {
    for (int64_t q = 0; q < Q; ++q) vgo_esdf_query_f32(nx, ny, nz, origin, res, dist, pts + 3 * q, out4 + 4 * q);
}

void vgo_esdf_query_batch(int nx, int ny, int nz, const double origin[3], double res, const float* dist, int64_t Q,
                          const double* pts, double* out_d, double* out_g) {
    for (int64_t q = 0; q < Q; ++q) vgo_esdf_query(nx, ny, nz, origin, res, dist, pts + 3 * q, out_d + q, out_g + 3 * q);
}

/* ------------------------------------------------------------------------------------ */
/* the evaluate callback of BT.cpp:796-800 as a public symbol, so tests can hand the SAME  */
/* objective to vgo_lbfgs and to the verbatim reference lbfgs_optimize (oracle/_ref)       */
/* ------------------------------------------------------------------------------------ */
void* vgo_solve_ctx_new(const vigo_params_t* P, int N, double* ctrl, const int32_t* goff,
                        const double* gpv, const uint8_t* gunk, int n_obs, const double* obs,
                        const double* w) {
    solve_ctx* S = (solve_ctx*)malloc(sizeof(solve_ctx));
    S->P = P; S->N = N; S->ctrl = ctrl; S->goff = goff; S->gpv = gpv; S->gunk = gunk;
    S->n_obs = n_obs; S->obs = obs; S->w = w;
    return S;
}
void vgo_solve_ctx_free(void* ctx) { free(ctx); }
double vgo_solve_eval(void* ctx, const double* x, double* g, int n) { return solve_eval(ctx, x, g, n); }
