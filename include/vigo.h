/*
 * vigo.h — C ABI of libvigo_hip.so, the MI355X (gfx950) batched back-end for the
 * ViGO B-spline optimizer hot path and the min-snap corridor collision checker of
 * hanyujin02/trajectory_planner.
 *
 * Every entry point names the reference interface it replaces (paths relative to the
 * reference repo, BT = include/trajectory_planner/bsplineTraj.{h,cpp},
 * LB = include/trajectory_planner/solver/lbfgs.hpp, BS = .../bspline.cpp,
 * PO = .../polyTrajOctomap.cpp, PS = .../polyTrajSolver.cpp).
 *
 * Conventions
 *   - plain C types only; no C++/torch types cross this boundary.
 *   - every array argument is a DEVICE pointer unless the function name ends in
 *     `_host` (those stage through an internal device workspace).
 *   - return value: 0 = VIGO_OK, negative = vigo_status_t error.  Per-trajectory solver
 *     results use the reference's own L-BFGS codes (LB:20-80) in `out_status`.
 *   - a handle owns its device buffers and HIP stream binding; calls on one handle are
 *     serialized by the caller; there is no global state.  A handle belongs to the device it was
 *     created on: that device must be the calling thread's current HIP device during every call
 *     (one process per GPU, or hipSetDevice first) — launches go to the bound stream / the current
 *     device's default stream.
 *   - all floating point arrays are fp64 (the reference's arithmetic, BT.h:22) unless the
 *     name says f32.
 *
 * Batch layouts (B trajectories, N control points each, n = 3*(N-6) free scalars):
 *   ctrl        double[B][N][3]      == B copies of Eigen::MatrixXd(3,N) column-major
 *                                       (optData::controlPoints, BT.h:22)
 *   guide_off   int32 [B*N + 1]      CSR offsets: control point (b,i) owns guide pairs
 *                                       [guide_off[b*N+i], guide_off[b*N+i+1])
 *                                       (optData::guidePoints[i][j], BT.h:23-24)
 *   guide_pv    double[G][6]         (p.x p.y p.z v.x v.y v.z) per pair
 *   guide_unk   uint8 [G]            map_->isUnknown(p) per pair (BT.cpp:841); may be NULL
 *                                       (=> all known).  vigo_guides_unknown() fills it.
 *   obs_off     int32 [B + 1]        CSR offsets of dynamic obstacles per trajectory, or NULL
 *                                       with n_obs_shared obstacles shared by the whole batch
 *   obs         double[O][9]         (pos.xyz vel.xyz size.xyz)  (BT.h:26-28)
 *                                       guide_pv == NULL / obs == NULL mean "no guides" / "no
 *                                       obstacles" whatever the offsets say (they are not read)
 *   weights     double[B][4]         (distance, smoothness, feasibility, dynamic) per
 *                                       trajectory; NULL => the handle's params.  The rebound
 *                                       loop doubles them per trajectory (BT.cpp:667,672,678).
 */
#ifndef VIGO_H
#define VIGO_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vigo_context* vigo_handle_t;

typedef enum {
    VIGO_OK = 0,
    VIGO_ERR_INVALID_ARG = -1,
    VIGO_ERR_NO_DEVICE = -2,     /* no HIP device / HIP runtime error at create      */
    VIGO_ERR_HIP = -3,           /* a HIP call failed; see vigo_last_error()          */
    VIGO_ERR_UNSUPPORTED_N = -4, /* N < 7 or N > VIGO_MAX_CTRL_POINTS                 */
    VIGO_ERR_NO_GRID = -5,       /* a map query was issued before vigo_set_grid*()    */
    VIGO_ERR_UNSUPPORTED = -6    /* parameter combination not implemented             */
} vigo_status_t;

/* vigo_optimize additionally needs mem_size * (N-6) * 48 B (fp64) of LDS <= 160 KiB:
 * N <= 219 at mem_size 16; larger N returns VIGO_ERR_UNSUPPORTED_N. */
enum { VIGO_MAX_CTRL_POINTS = 256, VIGO_MAX_MEM_SIZE = 16 };

/* arithmetic mode of the solver / cost kernels */
typedef enum {
    VIGO_PREC_F64 = 0,       /* fp64 state + reductions (parity-gated default)        */
    VIGO_PREC_F32 = 1,       /* fp32 state, fp64 reductions (throughput mode)         */
    VIGO_PREC_F64_FAST = 2   /* fp64 with fused multiply-adds in dot products / axpys /
                                stencils (what GCC's default contraction does to the
                                reference on ARM) and one reciprocal per history pair
                                instead of a division per two-loop step; same 1e-4 gate  */
} vigo_precision_t;

/*
 * Parameters of the hot path.  Field comments give the reference member they mirror.
 * vigo_default_params() loads cfg/bspline_interactive/bspline_planner_param.yaml values and
 * the solver settings of BT.cpp:695-699 / LB:942-954.
 */
typedef struct vigo_params_s {
    /* cost terms */
    double dthresh;              /* dthresh_               BT.cpp:35   */
    double dist_thresh_dynamic;  /* distThreshDynamic_     BT.cpp:143  */
    double ts_ctrl;              /* controlPointsTs_ = 0.2 BT.h:47     */
    double ts;                   /* ts_ (bspline_traj/timestep) BT.cpp:26 */
    double pred_horizon;         /* predHorizon_           BT.cpp:134  */
    double uncertain_factor;     /* uncertainAwareFactor_  BT.cpp:125  */
    double w_distance;           /* weightDistance_        BT.cpp:62   */
    double w_smoothness;         /* weightSmoothness_      BT.cpp:71   */
    double w_feasibility;        /* weightFeasibility_     BT.cpp:80   */
    double w_dynamic;            /* weightDynamicObstacle_ BT.cpp:89   */
    double min_height;           /* minHeight_             BT.cpp:107  */
    double max_height;           /* maxHeight_             BT.cpp:116  */
    int32_t plan_in_z;           /* planInZAxis_           BT.cpp:98   */
    /* L-BFGS (lbfgs_parameter_t, LB:87-191) */
    int32_t mem_size;            /* BT.cpp:697 (16)  */
    int32_t max_iterations;      /* BT.cpp:698 (200); 0 (= unbounded in lbfgs.hpp) is refused */
    int32_t max_linesearch;      /* LB:948 (40)      */
    int32_t past;                /* LB:945 (0); only 0 is supported */
    int32_t strict_z;            /* 1: do NOT apply the level rule (below, at vigo_optimize): the reference's arithmetic on the z
                                    axis whatever the input.  Default 0.  (Was a reserved field: same layout.) */
    double g_epsilon;            /* BT.cpp:699 (0.01) */
    double delta;                /* LB:946 */
    double min_step;             /* LB:949 */
    double max_step;             /* LB:950 */
    double f_dec_coeff;          /* LB:951 ftol */
    double s_curv_coeff;         /* LB:952 gtol */
    double xtol;                 /* LB:953 */
} vigo_params_t;

/* ---- lifecycle ------------------------------------------------------------------- */

/* Replaces: bsplineTraj::bsplineTraj()/init() device-side state (BT.cpp:9-22). */
int vigo_create(vigo_handle_t* out, int device_ordinal);
int vigo_destroy(vigo_handle_t h);
/* Binds all later launches of this handle to a hipStream_t (NULL = default stream). */
int vigo_set_stream(vigo_handle_t h, void* hip_stream);
/* Replaces: bsplineTraj::initParam() (BT.cpp:24-172) for the hot-path subset. */
void vigo_default_params(vigo_params_t* p);
int vigo_set_params(vigo_handle_t h, const vigo_params_t* p);
int vigo_get_params(vigo_handle_t h, vigo_params_t* p);
int vigo_set_precision(vigo_handle_t h, int vigo_precision);
/* Text of the last HIP/runtime failure on this handle (never NULL). */
const char* vigo_last_error(vigo_handle_t h);
/* Library/ABI version, and whether the code object was built for gfx950. */
int vigo_abi_version(void);
const char* vigo_build_arch(void);
/* "solver:<12 hex> all:<12 hex>": content hashes of the library's sources at build time (the first over the files that
 * decide the solve kernels' code).  Profiles record it, so a counter file can be told from a stale one. */
const char* vigo_build_id(void);

/* ---- voxel map ------------------------------------------------------------------- */

/*
 * Dense voxel-map contract standing in for mapManager::occMap (external package
 * map_manager, un-vendored; call sites BT.h:197,199,312,319,332, BT.cpp:292,412,435,841).
 *   voxels: uint8[nx][ny][nz] (z fastest), bit0 = inflated-occupied, bit1 = unknown,
 *           bit2 = occupied (un-inflated; used by the corridor checker's octree semantics).
 *   index  = floor((p - origin) / res) per axis; outside the box => occupied AND unknown.
 * The map is SNAPSHOTTED (packed to one bit per voxel per plane in HBM).
 * Replaces: bsplineTraj::setMap (BT.cpp:187-195), polyTrajOctomap::updateMap (PO.cpp:133-145).
 */
int vigo_set_grid(vigo_handle_t h, int nx, int ny, int nz, const double origin[3],
                  double res, const uint8_t* voxels_dev);
int vigo_set_grid_host(vigo_handle_t h, int nx, int ny, int nz, const double origin[3],
                       double res, const uint8_t* voxels_host);
/* Inflation of a byte grid on the device, before vigo_set_grid / vigo_pack_grid: bit0 (inflated-occupied)
 * := OR of bit2 (occupied) over the box |dx| <= rx, |dy| <= ry, |dz| <= rz voxels — what map_manager's
 * occMap does with the robot size (cfg/bspline_interactive/occupancy_map.yaml:9) before the planner's
 * isInflatedOccupied queries (BT.cpp:292,412,435).  In place; other bits are kept. */
int vigo_inflate_grid(vigo_handle_t h, int nx, int ny, int nz, uint8_t* voxels_dev, int rx, int ry, int rz);
/* Size in bytes of the packed snapshot for a grid of these dims (3 bit planes). */
size_t vigo_grid_packed_bytes(int nx, int ny, int nz);
/* Pack a byte grid into the snapshot format into a caller-owned device buffer (so it can
 * be broadcast with RCCL), and adopt an already packed snapshot (copy into the handle). */
int vigo_pack_grid(vigo_handle_t h, int nx, int ny, int nz, const uint8_t* voxels_dev,
                   uint32_t* packed_dev);
int vigo_set_grid_packed(vigo_handle_t h, int nx, int ny, int nz, const double origin[3],
                         double res, const uint32_t* packed_dev);
/* Metric bounds used by the corridor checker's out-of-bounds test
 * (octomap getMetricMin/Max, PO.cpp:572-577).  Default: the grid box. */
int vigo_set_metric_bounds(vigo_handle_t h, const double bmin[3], const double bmax[3]);

/* Point queries, Q points double[Q][3] -> uint8[Q].
 * which: 0 = isInflatedOccupied, 1 = isUnknown  (BT.cpp:412,841). */
int vigo_query_points(vigo_handle_t h, int which, int64_t Q, const double* pts,
                      uint8_t* out);
/* map_->isUnknown(guidePoint) for every guide pair (loop-invariant per solve, BT.cpp:841). */
int vigo_guides_unknown(vigo_handle_t h, int64_t G, const double* guide_pv,
                        uint8_t* out_unk);

/* ---- ViGO cost / gradient --------------------------------------------------------- */

/*
 * Replaces: bsplineTraj::costFunction (BT.cpp:802-821) = getDistanceCost (:823-932) +
 * getSmoothnessCost (:934-950) + getFeasibilityCost (:952-999) + getDynamicObstacleCost
 * (:1001-1064), evaluated for B trajectories in one launch.
 *   out_cost  double[B]            total weighted cost
 *   out_grad  double[B][N-6][3]    gradient w.r.t. the free control points (BT.cpp:819)
 *   out_terms double[B][4] or NULL un-weighted (distance, smoothness, feasibility, dynamic)
 */
int vigo_cost_grad(vigo_handle_t h, int B, int N, const double* ctrl,
                   const int32_t* guide_off, const double* guide_pv, const uint8_t* guide_unk,
                   const int32_t* obs_off, const double* obs, int n_obs_shared,
                   const double* weights,
                   double* out_cost, double* out_grad, double* out_terms);

/*
 * Replaces: bsplineTraj::optimize (BT.cpp:687-718) -> lbfgs::lbfgs_optimize (LB:1024-1349)
 * with line_search_morethuente (LB:716-937), for B trajectories in one launch.
 *   ctrl       in: initial control points; out: optData_.controlPoints as the reference
 *              leaves them, i.e. the LAST EVALUATED point (BT.cpp:803), not L-BFGS' x.
 *   out_x      double[B][N-6][3] or NULL: the x vector lbfgs_optimize returns
 *              (reverted to xp on line-search failure, LB:1192)
 *   out_status int32[B]  lbfgs_optimize return code (LB:20-80)
 *   out_fx     double[B] final objective (LB:1328)
 *   out_iters  int32[B]  iteration counter k at exit;  out_evals int32[B] cost evaluations
 *   (any out_* except ctrl may be NULL)
 *
 * THE LEVEL RULE (vigo_cost_grad, vigo_optimize, vigo_rebound_rounds; not in the reference).  With plan_in_z = 0 the
 * z coordinate of a control point feels the smoothness and feasibility terms only (BT.cpp:856-858 zeroes the guide
 * term's z gradient, BT.cpp:1022 the obstacle term's).  When all N control points of a trajectory lie at one height
 * — zmax - zmin <= 2^-40 * max(1, |zmin|, |zmax|): a level path as the least-squares fit leaves it — those two z
 * terms are differences of values that differ by rounding noise; the kernels take them as exactly zero (cost and
 * gradient), so such a trajectory's z never moves, where the reference lets it drift by that noise (measured:
 * <= 6e-14 m over 50 iterations).  The deviation is ~12 orders of magnitude inside the 1e-4 parity bar, and exact for
 * a path that is level to the bit (decided once per call, from the control points the call starts with).  What it
 * buys: waves whose trajectories are all level are solved (calls without obstacles, N <= 64) by an instantiation
 * that carries x and y only — two thirds of the L-BFGS history, of every dot product and of the stencils, and on
 * batches that fill the chip most of that history in registers, so that a CU holds eight waves instead of four
 * (-8 % at 1024 x 32, -35 % on such batches).  The rule is applied
 * per trajectory, also where a level trajectory shares a wave with one that is not: results never depend on the
 * batch a trajectory travels in.  The oracle's device-emulation mode applies the same rule; its reference-order mode
 * restates the reference and does not.  vigo_params_t.strict_z = 1 switches the rule off for a handle.
 */
int vigo_optimize(vigo_handle_t h, int B, int N, double* ctrl,
                  const int32_t* guide_off, const double* guide_pv, const uint8_t* guide_unk,
                  const int32_t* obs_off, const double* obs, int n_obs_shared,
                  const double* weights,
                  double* out_x, int32_t* out_status, double* out_fx,
                  int32_t* out_iters, int32_t* out_evals);

/*
 * Validation of the CSR lists a solve will index (they live in device memory, so vigo_optimize / vigo_cost_grad /
 * vigo_traj_dynamic_collision cannot check them per call): offsets start at 0, never decrease and end within
 * G guide pairs / O obstacles.  Synchronous (one small kernel and a 4-byte read back) — an integration-time
 * check, not part of the hot path.  guide_off / obs_off may be NULL (skipped).
 * Returns the number of violations found (0 = the lists are safe to pass), or a negative vigo_status_t.
 * No reference counterpart: the reference's vector<vector<>> cannot be inconsistent.
 */
int vigo_check_lists(vigo_handle_t h, int B, int N, const int32_t* guide_off, int64_t G,
                     const int32_t* obs_off, int64_t O);

/* ---- B-spline fit, evaluation and the rebound-loop gates --------------------------- */

/*
 * Replaces: bspline::parameterizeToBspline (BS.cpp:74-138) as bsplineTraj::updatePath calls it
 * (BT.cpp:314), for B paths of K waypoints each in one launch: least-squares solution of the
 * (K+4)x(K+2) system [1 4 1]/6 | velocity rows | acceleration rows (three colPivHouseholderQr
 * solves per path in the reference).  The factorisation depends on (K, ts) only; it is computed on
 * the device at the first call with a new (K, ts) and cached in the handle.
 *   points   double[B][K][3]   waypoints
 *   conds    double[B][4][3]   start vel, end vel, start acc, end acc (the startEndConditions order of BT.cpp:290, BS.cpp:117-121);
 *                              NULL => all zero
 *   ctrl_out double[B][K+2][3] control points (the `ctrl` layout of vigo_optimize, N = K + 2)
 * 4 <= K <= VIGO_MAX_CTRL_POINTS - 2 (the reference exit(0)s below 4 points, BS.cpp:83-87).
 */
int vigo_bspline_fit(vigo_handle_t h, int B, int K, double ts, const double* points,
                     const double* conds, double* ctrl_out);

/*
 * Replaces: bspline::at (BS.cpp:32-58) on bspline(3, ctrl, ts_ctrl) and its
 * getDerivative() chains (BS.cpp:64-72).  deriv = 0,1,2.
 *   times double[T] shared by the batch; out double[B][T][3].
 */
int vigo_bspline_eval(vigo_handle_t h, int B, int N, const double* ctrl, int deriv,
                      int T, const double* times, double* out);

/*
 * Replaces: bsplineTraj::hasCollisionTrajectory (BT.h:307-325): sample the spline every
 * dt = res/max_vel/2 and test isInflatedOccupied.
 *   out_flag uint8[B]; out_first int32[B] index of the first colliding sample or -1.
 */
int vigo_traj_collision(vigo_handle_t h, int B, int N, const double* ctrl, double dt,
                        uint8_t* out_flag, int32_t* out_first);
/*
 * Replaces: bsplineTraj::hasDynamicCollisionTrajectory (BT.h:344-368).
 */
int vigo_traj_dynamic_collision(vigo_handle_t h, int B, int N, const double* ctrl, double dt,
                                const int32_t* obs_off, const double* obs, int n_obs_shared,
                                uint8_t* out_flag);
/*
 * Replaces the map queries of bsplineTraj::findCollisionSeg (BT.cpp:403-445):
 *   out_pt   uint8[B][N]  isInflatedOccupied(ctrl[i])
 *   out_line uint8[B][N]  isInflatedOccupiedLine(ctrl[i-1], ctrl[i]) (entry 0 = 0)
 * The segment bookkeeping itself stays on the host (it is a serial scan of these flags).
 */
int vigo_ctrl_occupancy(vigo_handle_t h, int B, int N, const double* ctrl,
                        uint8_t* out_pt, uint8_t* out_line);

/* ---- the rebound loop between two A* calls, device-resident ------------------------------ */

/*
 * Replaces: the while loop of bsplineTraj::optimizeTrajectory (BT.cpp:611-685) for B trajectories, for as long as
 * a trajectory needs nothing from the host.  One round per trajectory =
 *     hasCollisionTrajectory / hasDynamicCollisionTrajectory     (BT.cpp:620-626, BT.h:307-368; the dynamic gate
 *                                                                  only when the trajectory has obstacles)
 *     neither          -> VIGO_RB_DONE                            (BT.cpp:628-631)
 *     failCount >= 4   -> VIGO_RB_NEEDS_HOST                      (forced A* re-guide, BT.cpp:640-654)
 *     static collision -> isReguideRequired (BT.cpp:573-608: findCollisionSeg :403-445 on the new control points,
 *                         comparison with the previous segments, isControlPointRequireNewGuide BT.h:417-429):
 *                         required -> VIGO_RB_NEEDS_HOST (A*, BT.cpp:656-665; nothing of the state is touched, the
 *                         host repeats the step itself); else collisionSeg_ := the new segments,
 *                         weightDistance *= 2, ++failCount      (BT.cpp:666-674)
 *     dynamic collision -> weightDynamicObstacle *= 2             (BT.cpp:677-679)
 *     optimize()                                                  (BT.cpp:680, = vigo_optimize on the still-active set,
 *                                                                  compacted on the device)
 * max_rounds rounds are queued without a host round trip; trajectories that are done or wait for the host are
 * skipped, and once a round hands a trajectory to the host the rest of the call is a no-op for the whole batch: the
 * optimize() the still-active trajectories owe is left to the next call (their solve_first is set), where it shares
 * one launch with the re-guided ones — the waiting trajectories are on the batch's critical path.  A trajectory whose state has solve_first != 0 is optimized once before its first gate (the
 * optimize() of BT.cpp:612, or the one that follows a host-side re-guide).
 *   ctrl, guide_*, obs_*       as vigo_optimize (guide_unk from vigo_guides_unknown); ctrl in/out
 *   weights   double[B][4]     in/out, REQUIRED (the loop doubles them per trajectory)
 *   gate_dt                    sample step of the gates, map_->getRes() / maxVel_ / 2 (BT.h:312)
 *   not_check_ratio            notCheckRatio_ of findCollisionSeg (BT.cpp:408; 0 in the reference)
 *   state     vigo_rebound_state_t[B] in/out (device memory)
 * Needs vigo_set_grid.  The 30 ms wall-clock budget of BT.cpp:633 stays with the caller (between calls).
 */
enum { VIGO_MAX_COLLISION_SEGS = 48 };
typedef enum { VIGO_RB_ACTIVE = 0, VIGO_RB_DONE = 1, VIGO_RB_NEEDS_HOST = 2 } vigo_rebound_status_t;
typedef struct {
    int32_t status;        /* vigo_rebound_status_t; only VIGO_RB_ACTIVE entries are worked on      */
    int32_t solve_first;   /* in: optimize before the first gate; out: an optimize() is still owed    */
    int32_t fail_count;    /* failCount, BT.cpp:613                                                 */
    int32_t gate_static;   /* out: last hasCollisionTrajectory result                               */
    int32_t gate_dynamic;  /* out: last hasDynamicCollisionTrajectory result                        */
    int32_t rounds;        /* out: += gate passes made by the call                                  */
    int32_t lbfgs_status;  /* out: lbfgs_optimize return code of the last optimize() (LB:20-80)     */
    int32_t n_seg;         /* collisionSeg_ (BT.h:73) as isReguideRequired left it: n_seg pairs     */
    int32_t seg[2 * VIGO_MAX_COLLISION_SEGS];   /* (first, second); more segments than fit -> NEEDS_HOST */
} vigo_rebound_state_t;

int vigo_rebound_rounds(vigo_handle_t h, int B, int N, double* ctrl,
                        const int32_t* guide_off, const double* guide_pv, const uint8_t* guide_unk,
                        const int32_t* obs_off, const double* obs, int n_obs_shared,
                        double* weights, double gate_dt, double not_check_ratio, int max_rounds,
                        vigo_rebound_state_t* state);

/* ---- min-snap QP and corridor collision checker ------------------------------------ */

/*
 * Replaces: polyTrajSolver::solve (PS.cpp:849-904) with constructP/constructA/constructBound
 * (:241-846), avgTimeAllocation (:125-138), updateCorridorParam (:985-1012) and the rescale to
 * un-normalised local time (:874-878) — the three per-axis QPs the reference hands to OSQP —
 * for T waypoint paths of W waypoints (W-1 degree-7 segments) each, one wavefront per path,
 * solved exactly (null-space elimination of the equality rows + dual active set on the corridor
 * boxes) instead of ADMM to eps 1e-3.
 *   waypoints  double[T][W][3]
 *   corridor   double[T][W-1]  corridor half-size per segment (0 = no boxes there,
 *                              PS.cpp:992); NULL = no corridor constraint (makePlanAddingWaypoint)
 *   conds      double[T][4][3] init vel, end vel, init acc, end acc (PS.h updateInitVel/...); NULL = 0
 *   out_coeffs double[T][W-1][3][deg+1]   == the `coeffs` layout of vigo_corridor_check (S = T*(W-1))
 *   out_knots  double[T][W]    desiredTime_ (PS.cpp:125-138)
 *   out_status int32[T]        0 solved, -1 numerical failure (coincident waypoints, > 1024 boxes),
 *                              -2 infeasible corridor (the reference keeps a stale solution silently)
 * deg must be 7; 2 <= W <= 11; diff/cont as polynomial/continuity degrees of cfg/planner*.yaml.
 */
int vigo_minsnap(vigo_handle_t h, int T, int W, int deg, int diff, int cont, double desired_vel,
                 double corridor_res, const double* waypoints, const double* corridor,
                 const double* conds, double* out_coeffs, double* out_knots, int32_t* out_status);


/*
 * Replaces: polyTrajOctomap::checkCollisionTraj (PO.cpp:634-656) -> checkCollision
 * (:547-568) -> checkCollisionPoint (:571-589), fed by polyTrajSolver::getTrajectory /
 * getPose (PS.cpp:1125-1137, :1026-1056), for S independent polynomial segments.
 *   coeffs   double[S][3][deg+1]  x,y,z coefficients in un-normalised local time
 *   dur      double[S]            segment duration
 *   n_samp   int32[S]             samples per segment: t_k = k * delT[s], k < n_samp[s]
 *   delT     double[S]
 *   box[3], map_res               collision_box / map_resolution (cfg/planner_interactive.yaml)
 *   out_flag uint8[S]; out_first int32[S] first colliding sample index or -1;
 *   out_count int32[S] number of colliding samples (may be NULL)
 * Every sample's verdict is the reference walk's; how they are reached is not (csrc/vigo_corridor.hip): segments of
 * more than 512 samples are cut into spans of 32 or 16 samples, and a span is decided by ONE evaluation when the kernel
 * can prove that all its samples see the same voxel keys (an interval that holds every sample's float position, taken
 * through the reference's own monotone expressions at both ends); spans it cannot decide are cut in four, and what
 * remains goes through the per-sample sweep.  The accumulated clock t += delT is reproduced exactly
 * (vigo_accumulated_time / vigo_clock_table_time below).
 * A pose at NaN or infinity (non-finite coefficients, overflow to float): the reference's lattice count
 * (int)((xmax - xmin) / map_res) is then the conversion of a NaN — undefined in C++, INT_MIN on x86, where the sweep
 * makes no pass and the pose does NOT collide.  Both entry points below follow x86 (the oracle on this host does);
 * a caller that wants such poses refused tests them itself (the host facade does: host/src/polyTrajOctomap.cpp).
 */
int vigo_corridor_check(vigo_handle_t h, int S, int deg, const double* coeffs,
                        const int32_t* n_samp, const double* delT,
                        const double box[3], double map_res,
                        uint8_t* out_flag, int32_t* out_first, int32_t* out_count);

/*
 * Replaces: polyTrajOctomap::checkCollision(point3d) (PO.cpp:547-568) for M already-sampled
 * poses, as the reference's checkCollisionTraj(trajectory, ...) overloads use it (PO.cpp:619-656).
 *   pts double[M][3] (cast to float like pose2Octomap); out uint8[M].
 */
int vigo_box_collision_points(vigo_handle_t h, int64_t M, const double* pts, const double box[3],
                              double map_res, uint8_t* out);

/*
 * Replaces: polyTrajSolver::getTrajectory (PS.cpp:1125-1137) -> getPose (PS.cpp:1026-1056), positions only,
 * for S independent polynomial segments: the sampler vigo_corridor_check runs internally, on its own.
 *   coeffs, n_samp, delT as in vigo_corridor_check; sample k of segment s is written at index s * stride + k
 *   (samples k >= stride are not produced);
 *   out_pos     double[S][stride][3] or NULL   x, y, z as getPose returns them
 *   out_pos_f32 float [S][stride][3] or NULL   the same after pose2Octomap's cast to octomap::point3d (PO.cpp:634-656)
 * pow(t, d) of PS.cpp:1035-1039 is evaluated as the correctly rounded power (see vigo_exact_pow below).
 */
int vigo_poly_sample(vigo_handle_t h, int S, int deg, const double* coeffs, const int32_t* n_samp,
                     const double* delT, int stride, double* out_pos, float* out_pos_f32);

/* The reference's sample clock: t_k of `for (t = 0; ...; t += delT)` (PS.cpp:1129), i.e. the
 * k-fold floating-point accumulation, evaluated in closed form (host utility, no GPU). */
double vigo_accumulated_time(double delT, int64_t k);

/* The same clock through the table vigo_corridor_check builds once per segment (csrc/vigo_exact_time.hpp: one piece
 * per run of equal increments, a binary search per lookup): t_k for 0 <= k <= k_last from the table made for k_last,
 * or NaN when the kernel would build none (delT outside [2^-1000, 1e300), more than 128 pieces) and fall back to
 * vigo_accumulated_time.  Host utility for the tests (no GPU): must equal vigo_accumulated_time(delT, k) bit for bit. */
double vigo_clock_table_time(double delT, int64_t k_last, int64_t k);

/* pow(t, d) of polyTrajSolver::getPose (PS.cpp:1035-1039) for an integer 0 <= d <= 15 as the sampler kernels
 * evaluate it: the CORRECTLY ROUNDED power (libm's pow returns it or its neighbour, depending on the libm
 * build).  Host utilities, no GPU:
 *   vigo_exact_pow          the kernels' two-tier evaluation (NaN for d outside [0, 15]);
 *   vigo_exact_pow_dd       its first tier alone — the running double-double product — and whether that tier
 *                           could certify the rounding (*ambiguous = 0) or defers to the second;
 *   vigo_exact_pow_integer  its second tier alone: exact integer arithmetic, rounded once. */
double vigo_exact_pow(double t, int d);
double vigo_exact_pow_dd(double t, int d, int* ambiguous);
double vigo_exact_pow_integer(double t, int d);

/* ---- ESDF trilinear query (config 5; no reference counterpart) -------
 * vigo_set_esdf copies the row-major float lattice dist_dev[nx][ny][nz] (device memory) into the handle's own layout
 * (one 128-B line per group of 1x3x3 trilinear cells): 3.56x the lattice's bytes of device memory, stream-ordered, the caller's
 * buffer is not referenced afterwards.  nx, ny, nz >= 2. */

int vigo_set_esdf(vigo_handle_t h, int nx, int ny, int nz, const double origin[3],
                  double res, const float* dist_dev);
int vigo_esdf_query(vigo_handle_t h, int64_t Q, const double* pts,
                    double* out_dist, double* out_grad);
/* The same query at the I/O width SURVEY.md §8(d) config 5 states for the fp32 lattice: pts float[Q][3] in (12 B),
 * out float[Q][4] = {distance, d/dx, d/dy, d/dz} out (16 B, 16-byte aligned), all arithmetic in fp32, every operation
 * rounded once: u = (p - (float)origin) * inv_res - 0.5f with inv_res = 1.0f / (float)res, floorf, clamps, the blend
 * x then y then z, gradient differences * inv_res.  Own definition (no reference counterpart); oracle twin
 * vgo_esdf_query_f32.  Differs from the fp64 entry by fp32 rounding only (~1e-6 relative). */
int vigo_esdf_query_f32(vigo_handle_t h, int64_t Q, const float* pts, float* out_dist_grad);

#ifdef __cplusplus
}
#endif
#endif /* VIGO_H */
