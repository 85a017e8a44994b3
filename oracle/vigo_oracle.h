/*
 * vigo_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * Single-thread fp64 CPU restatement of the reference's hot path, used as the parity
 * checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing in
 * the product path (trajectory_planner_amd/, include/) may link or call this.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - L-BFGS + More-Thuente (vgo_lbfgs): PINNED bit-for-bit against the verbatim reference
 *     include/trajectory_planner/solver/lbfgs.hpp compiled from /root/reference into
 *     oracle/_ref/libref_lbfgs.so (oracle/ref_lbfgs_harness.cpp), and by golden traces
 *     under tests/golden/ generated from it.
 *   - cost terms, B-spline evaluation, collision gates, corridor checker: restated from
 *     bsplineTraj.cpp / bspline.cpp / polyTrajOctomap.cpp, which cannot be built here
 *     (need Eigen, ROS, map_manager, octomap).  The reference ships no fixtures or
 *     assertions for them => PARITY UNPINNED by reference data; pinned only by closed-form
 *     known answers and finite-difference checks in tests/.
 *   - voxel map semantics (mapManager::occMap, octomap::OcTree): the dependency is not in
 *     /root/reference; the contract below is this build's own (include/vigo.h).
 */
#ifndef VIGO_ORACLE_H
#define VIGO_ORACLE_H

#include <stdint.h>
#include "../include/vigo.h"

#ifdef __cplusplus
extern "C" {
#endif

/* same values as vigo_default_params (cfg/bspline_interactive/bspline_planner_param.yaml,
 * BT.cpp:695-699, LB:942-954); separate symbol so the oracle never links the product */
void vgo_default_params(vigo_params_t* p);

/* 0 = reference order (default); 32 / 64 = emulate the HIP kernels' lane-tree sums and
 * pow-free powers so results can be compared with the GPU bit for bit (vigo_oracle.c top). */
void vgo_set_emulation(int group);
void vgo_set_emulation2(int group, int points_per_lane);
/* with emulation on: also mirror VIGO_PREC_F64_FAST (explicit fma, reciprocal-multiply) */
void vgo_set_emulation_fast(int on);
int vgo_get_emulation(void);

/* dense voxel map, same contract as vigo_set_grid (include/vigo.h) */
typedef struct {
    int nx, ny, nz;
    double origin[3];
    double res;
    const uint8_t* vox;      /* [nx][ny][nz], bit0 inflated-occ, bit1 unknown, bit2 occupied */
    double bmin[3], bmax[3]; /* metric bounds for the corridor checker */
} vgo_grid_t;

void vgo_grid_init(vgo_grid_t* g, int nx, int ny, int nz, const double origin[3], double res,
                   const uint8_t* vox);
int vgo_is_inflated_occupied(const vgo_grid_t* g, const double p[3]);
int vgo_is_unknown(const vgo_grid_t* g, const double p[3]);
int vgo_is_inflated_occupied_line(const vgo_grid_t* g, const double p1[3], const double p2[3]);

/* BT.cpp:802-1064.  One trajectory.  goff[N+1] is the trajectory's own CSR slice
 * (absolute indices into gpv/gunk).  w[4] = weights.  grad_free: n = 3*(N-6) values
 * (BT.cpp:819); grad_full (may be NULL): 3*N values incl. the fixed points; terms[4]
 * (may be NULL): un-weighted distance, smoothness, feasibility, dynamic costs. */
double vgo_cost_grad(const vigo_params_t* P, int N, const double* ctrl,
                     const int32_t* goff, const double* gpv, const uint8_t* gunk,
                     int n_obs, const double* obs, const double w[4],
                     double* grad_free, double* grad_full, double* terms);

/* LB:1024-1349 restated.  eval(ctx, x, g, n) -> f.  Returns the reference's status code.
 * trace (may be NULL) is called where the reference calls proc_progress (LB:778-788). */
typedef double (*vgo_eval_fn)(void* ctx, const double* x, double* g, int n);
typedef void (*vgo_trace_fn)(void* tctx, const double* x, const double* g, double fx,
                             double step, int n);
int vgo_lbfgs(int n, double* x, double* fx_out, vgo_eval_fn eval, void* ctx,
              const vigo_params_t* P, int* out_iters, int* out_evals,
              vgo_trace_fn trace, void* tctx);

/* BT.cpp:687-718 for one trajectory: ctrl in/out (last evaluated point), x_out (n) the
 * vector lbfgs returns.  Returns the L-BFGS status. */
int vgo_optimize(const vigo_params_t* P, int N, double* ctrl,
                 const int32_t* goff, const double* gpv, const uint8_t* gunk,
                 int n_obs, const double* obs, const double w[4],
                 double* x_out, double* fx_out, int* iters, int* evals);

/* The evaluate callback of BT.cpp:796-800 as a public symbol (ctx from vgo_solve_ctx_new): lets
 * tests drive vgo_lbfgs and the verbatim reference lbfgs_optimize with the same objective. */
void* vgo_solve_ctx_new(const vigo_params_t* P, int N, double* ctrl, const int32_t* goff,
                        const double* gpv, const uint8_t* gunk, int n_obs, const double* obs,
                        const double* w);
void vgo_solve_ctx_free(void* ctx);
double vgo_solve_eval(void* ctx, const double* x, double* g, int n);

/* batch drivers over the include/vigo.h layouts (single thread; used for timing too) */
void vgo_cost_grad_batch(const vigo_params_t* P, int B, int N, const double* ctrl,
                         const int32_t* guide_off, const double* gpv, const uint8_t* gunk,
                         const int32_t* obs_off, const double* obs, int n_obs_shared,
                         const double* weights, double* out_cost, double* out_grad,
                         double* out_terms);
void vgo_optimize_batch(const vigo_params_t* P, int B, int N, double* ctrl,
                        const int32_t* guide_off, const double* gpv, const uint8_t* gunk,
                        const int32_t* obs_off, const double* obs, int n_obs_shared,
                        const double* weights, double* out_x, int32_t* out_status,
                        double* out_fx, int32_t* out_iters, int32_t* out_evals);

/* BS.cpp:19-72: uniform B-spline of `degree` over ncp control points (rows of 3). */
/* BS.cpp:74-138 (least-squares fit of waypoints + boundary conditions to control points) */
int vgo_bspline_fit(int K, double ts, const double* points, const double* cond, double* ctrl_out);
void vgo_bspline_fit_batch(int B, int K, double ts, const double* points, const double* conds, double* ctrl_out);
void vgo_bspline_at(int degree, int ncp, const double* cp, double ts, double t, double out[3]);
/* derivative control points (BS.cpp:64-72): out has ncp-1 rows */
void vgo_bspline_derivative(int degree, int ncp, const double* cp, double ts, double* out);
/* cubic spline (degree 3) value / 1st / 2nd derivative at t, as bsplineTraj uses them */
void vgo_traj_eval(int N, const double* ctrl, double ts_ctrl, int deriv, double t,
                   double out[3]);
/* number of samples of `for (t=0; t<=tmax; t+=dt)` and the accumulated times */
int vgo_sample_times(double tmax, double dt, double* times, int cap);

/* BT.h:307-325 / :344-368 / BT.cpp:403-445 map queries */
int vgo_traj_collision(const vgo_grid_t* g, int N, const double* ctrl, double ts_ctrl,
                       double dt, int* first_idx);
int vgo_traj_dynamic_collision(int N, const double* ctrl, double ts_ctrl, double dt,
                               int n_obs, const double* obs);
void vgo_ctrl_occupancy(const vgo_grid_t* g, int N, const double* ctrl, uint8_t* pt,
                        uint8_t* line);

/* the rebound loop's bookkeeping between two A* calls: findCollisionSeg (BT.cpp:403-445), isReguideRequired
 * (BT.cpp:573-608, BT.h:379-429) and one pass of the loop body of optimizeTrajectory (BT.cpp:619-679) for a trajectory
 * that needs no A* — the CPU statement of vigo_rebound_rounds' decision kernel */
int vgo_find_collision_seg(const vgo_grid_t* g, int N, const double* ctrl, double not_check_ratio, int32_t* seg, int cap);
int vgo_is_reguide_required(const vigo_params_t* P, const vgo_grid_t* g, int N, const double* ctrl, const int32_t* goff,
                            const double* gpv, const int32_t* prev_seg, int n_prev, double not_check_ratio, int32_t* new_seg,
                            int cap, int* n_new);
int vgo_rebound_decide(const vigo_params_t* P, const vgo_grid_t* g, int N, const double* ctrl, const int32_t* goff,
                       const double* gpv, int n_obs, const double* obs, double gate_dt, double not_check_ratio,
                       double* weights, vigo_rebound_state_t* st);

/* PO.cpp:547-589 box sweep at one (float) sample position */
int vgo_box_collision(const vgo_grid_t* g, float px, float py, float pz,
                      const double box[3], double map_res);
/* PS.cpp:1026-1056 position of one segment polynomial at local time t */
void vgo_poly_pos(int deg, const double* cx, const double* cy, const double* cz, double t,
                  double out[3]);
/* pow(t, d) inside vgo_poly_pos: 0 = libm pow() (the reference on this host), 1 = the correctly rounded power
 * by exact integer arithmetic (what the HIP sampler implements) */
void vgo_set_pow_mode(int exact);
int vgo_get_pow_mode(void);
double vgo_pow_exact(double t, int d);
/* PS.cpp:1125-1137 positions of S segments (fp64 and/or after pose2Octomap's float cast) */
void vgo_poly_sample(int S, int deg, const double* coeffs, const int32_t* n_samp, const double* delT, int stride,
                     double* out_pos, float* out_f32);
void vgo_corridor_check_batch(const vgo_grid_t* g, int S, int deg, const double* coeffs, const int32_t* n_samp,
                              const double* delT, const double box[3], double map_res, uint8_t* flag,
                              int32_t* first, int32_t* count);
/* PO.cpp:634-656 restricted to one segment sampled with accumulated t += delT */
int vgo_corridor_check_segment(const vgo_grid_t* g, int deg, const double* coeffs,
                               int n_samp, double delT, const double box[3], double map_res,
                               int* first_idx, int* count);

/* config 5: trilinear ESDF value + gradient (no reference counterpart) */
void vgo_esdf_query(int nx, int ny, int nz, const double origin[3], double res,
                    const float* dist, const double p[3], double* out_d, double out_g[3]);

void vgo_esdf_query_f32(int nx, int ny, int nz, const double origin[3], double res, const float* dist,
                        const float p[3], float out4[4]);
void vgo_esdf_query_f32_batch(int nx, int ny, int nz, const double origin[3], double res, const float* dist, int64_t Q,
                              const float* pts, float* out4);
void vgo_esdf_query_batch(int nx, int ny, int nz, const double origin[3], double res, const float* dist, int64_t Q,
                          const double* pts, double* out_d, double* out_g);

#ifdef __cplusplus
}
#endif
#endif
