// vigo_internal.hpp — host-side types shared by the C-ABI layer and the kernel launchers.
// Not part of the public boundary (that is include/vigo.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/vigo.h"

namespace vigo {

// Scalars derived from vigo_params_t once on the host, with the reference's own expressions
// (BT.cpp:835, :959, :1007-1009) so the device sees the same bits as the CPU path.
struct DevConst {
    // distance term (BT.cpp:835)
    double dth, da, db, dc;
    double unc_factor;
    // height band (BT.cpp:836-837, plan_in_z only)
    double hth, ha, hb, hc, min_h, max_h;
    // feasibility (BT.cpp:955-959)
    double ts_ctrl, ts_inv_sqr;
    // dynamic obstacles (BT.cpp:1007-1009)
    double ts, thr_dyn, oa, ob, oc;
    int pred_num;
    int plan_in_z;
    int strict_z;      // 1: no level rule
    // default weights
    double w[4];
    // L-BFGS (LB:87-191)
    int mem_size, max_iterations, max_linesearch;
    double g_epsilon, min_step, max_step, ftol, gtol, xtol;
};

DevConst make_dev_const(const vigo_params_t& P);

struct SolveArgs {
    int B, N;
    double* ctrl;  // in/out for optimize, read-only for cost_grad
    const int32_t* guide_off;
    const double* guide_pv;
    const uint8_t* guide_unk;
    const int32_t* obs_off;
    const double* obs;
    int n_obs_shared;
    const double* weights;
    // (vigo_rebound_rounds) the launch works on trajectory active_idx[slot] for slot < *active_count instead of
    // b = slot < B; lbfgs_status_stride != 0: out_status is the lbfgs_status field of a vigo_rebound_state_t array
    const int32_t* active_idx;
    const int32_t* active_count;
    int status_stride;     // in int32 units between consecutive trajectories' out_status (0 = 1)
    // optimize outputs
    double* out_x;
    int32_t* out_status;
    double* out_fx;
    int32_t* out_iters;
    int32_t* out_evals;
    // (set by the solve launcher) 1: the waves whose trajectories are all level are solved by a second launch (D = 2)
    int level_waves_elsewhere;
    // cost_grad outputs
    double* out_cost;
    double* out_grad;
    double* out_terms;
};

// Packed voxel snapshot in HBM: three bit planes (inflated-occupied, unknown, occupied), each
// nx*ny rows of nzw 32-bit words, z fastest (bit k of word w = voxel z = 32*w + k).
struct GridView {
    const uint32_t* planes;  // 3 * plane_words
    size_t plane_words;
    int nx, ny, nz, nzw;
    double origin[3];
    double res;
    double bmin[3], bmax[3];
    int key0[3];  // octomap key of voxel index 0 minus 32768 (corridor checker)
};

// ESDF samples in HBM as one 128-B cache line per group of trilinear cells: line (x, by, bz) holds the 2 x 4 x 4 values
// [x, x+1] x [3 by, 3 by + 3] x [3 bz, 3 bz + 3] (z fastest), i.e. the 1 x 3 x 3 whole cells starting at (x, 3 by, 3 bz):
// all eight corners of ANY cell lie in ONE line (one base address + constant offsets), at 3.56x the lattice's bytes.
// Uniformly random queries are bound by the cache lines they touch (a 256^3 lattice lives in the Infinity Cache, not
// in L2): the row-major lattice costs 4.1 lines per cell, disjoint 4x4x4 bricks 2.3, overlapping 4x4x4 bricks 1.33.
struct EsdfView {
    const float* dist;        // (nx - 1) * nby * nbz lines of 32 floats
    int nx, ny, nz;
    int nby, nbz;             // lines along y and z
    double origin[3];
    double res;
};
inline int esdf_bricks_along(int n) { return (n + 1) / 3; }   // ceil((n - 1) / 3) groups cover the n - 1 cells, n >= 2
inline size_t esdf_bricked_floats(int nx, int ny, int nz) {
    return (size_t)(nx - 1) * esdf_bricks_along(ny) * esdf_bricks_along(nz) * 32;
}

// Per-handle (= per-device) launch state: which kernel instantiations already had their dynamic-LDS limit raised on
// the handle's device, and that device's SIMD count.  Nothing of this kind is kept in function statics.
struct LaunchState {
    uint64_t lds_attr_set = 0;   // bit per k_optimize instantiation (vigo_solver.hip)
    bool minsnap_attr_set = false;
    int simd_count = 0;          // 4 per CU; 0 = unknown
};

// launchers (each returns hipError_t as int)
// k: host copy (launch geometry), kd: the same constants in device memory (read by the kernels)
int launch_cost_grad(hipStream_t s, const SolveArgs& a, const DevConst& k, const DevConst* kd, int precision);
int launch_optimize(hipStream_t s, const SolveArgs& a, const DevConst& k, const DevConst* kd, int precision, LaunchState& L);
// LDS bytes one solve workgroup needs for N control points (must stay <= 160 KiB)
size_t optimize_lds_requirement(int N, int mem_size, int precision);

int launch_pack_grid(hipStream_t s, int nx, int ny, int nz, const uint8_t* vox, uint32_t* packed);
int launch_inflate(hipStream_t s, int nx, int ny, int nz, uint8_t* vox, uint32_t* planeA, uint32_t* planeB, int rx, int ry, int rz);
int launch_query_points(hipStream_t s, const GridView& g, int which, int64_t Q, const double* pts,
                        int pt_stride, uint8_t* out);
int launch_bspline_eval(hipStream_t s, int B, int N, const double* ctrl, double ts_ctrl, int deriv,
                        int T, const double* times, double* out);
int launch_traj_collision(hipStream_t s, const GridView& g, int B, int N, const double* ctrl,
                          double ts_ctrl, int T, const double* times, uint8_t* out_flag,
                          int32_t* out_first);
int launch_traj_dynamic_collision(hipStream_t s, int B, int N, const double* ctrl, double ts_ctrl,
                                  int T, const double* times, const int32_t* obs_off,
                                  const double* obs, int n_obs_shared, uint8_t* out_flag);
int launch_fill_sample_times(hipStream_t s, double dt, int T, double* times);
int launch_ctrl_occupancy(hipStream_t s, const GridView& g, int B, int N, const double* ctrl,
                          uint8_t* out_pt, uint8_t* out_line);
int launch_corridor_check(hipStream_t s, const GridView& g, int S, int deg, const double* coeffs,
                          const int32_t* n_samp, const double* delT, const double box[3],
                          double map_res, uint8_t* out_flag, int32_t* out_first, int32_t* out_count);
int launch_box_points(hipStream_t s, const GridView& g, int64_t M, const double* pts, const double box[3],
                      double map_res, uint8_t* out);
// polyTrajSolver::getTrajectory for S segments: sample k of segment s at out[(s * stride + k) * 3] (fp64 and/or float)
int launch_poly_sample(hipStream_t s, int S, int deg, const double* coeffs, const int32_t* n_samp, const double* delT,
                       int stride, double* out_pos, float* out_f32);
// the gate + decision pass of vigo_rebound_rounds (one wave per trajectory) and the compaction of the active set
struct ReboundArgs {
    int B, N;
    const double* ctrl;
    const int32_t* guide_off;
    const double* guide_pv;
    const int32_t* obs_off;
    const double* obs;
    int n_obs_shared;
    double* weights;
    vigo_rebound_state_t* state;
    double ts_ctrl;
    int T;                   // samples of the whole trajectory (dynamic gate, BT.h:345)
    int T_static;            // samples up to (1 - not_check_ratio) * duration (static gate, BT.h:313); <= T
    const double* times;
    double dthresh, not_check_ratio;
    int32_t* flags;          // see k_rebound_compact
};
int launch_rebound_decide(hipStream_t s, const GridView& g, const ReboundArgs& a);
// mode 0: status == ACTIVE && solve_first (clears solve_first); mode 1: status == ACTIVE.  Ascending order.
int launch_rebound_compact(hipStream_t s, int B, vigo_rebound_state_t* state, int mode, int32_t* idx, int32_t* flags);
// counts CSR violations of guide_off[B*N+1] / obs_off[B+1] into *bad (device int, zeroed by the launcher)
int launch_check_lists(hipStream_t s, int B, int N, const int32_t* guide_off, int64_t G, const int32_t* obs_off, int64_t O,
                       int* bad);
int launch_esdf_query(hipStream_t s, const EsdfView& e, int64_t Q, const double* pts,
                      double* out_dist, double* out_grad);
// fp32 I/O and arithmetic: pts float[Q][3], out float[Q][4] = {d, gx, gy, gz} (16-byte aligned)
int launch_esdf_query_f32(hipStream_t s, const EsdfView& e, int64_t Q, const float* pts, float* out4);
// row-major [nx][ny][nz] -> the bricked layout of EsdfView
int launch_esdf_brick(hipStream_t s, int nx, int ny, int nz, const float* src, float* dst);
// batched B-spline fit (vigo_fit.hip): one-off device factorisation per (K, ts), then the fit
size_t fit_work_doubles(int K);
size_t fit_pinv_doubles(int K);
int launch_fit_setup(hipStream_t s, int K, double ts, double* work, double* pinvT);
int launch_bspline_fit(hipStream_t s, int B, int K, const double* pinvT, const double* points,
                       const double* conds, double* out);

// batched min-snap QP (vigo_minsnap.hip)
size_t minsnap_lds_bytes(int W, int cont);
int minsnap_max_waypoints();
int launch_minsnap(hipStream_t s, int T, int W, int deg, int diff, int cont, double vel, double corridor_res,
                   const double* wp, const double* corridor, const double* conds, double* out_coeffs,
                   double* out_knots, int32_t* out_status, LaunchState& L);

}  // namespace vigo

struct vigo_context {
    int device = 0;
    hipStream_t stream = nullptr;
    vigo_params_t params;
    vigo::DevConst dc;
    vigo::DevConst* dc_dev = nullptr;  // device copy, refreshed by vigo_set_params IN STREAM ORDER (see there)
    // pinned staging ring for those refreshes: slot i may be rewritten once dc_event[i] (its last copy) is done
    static constexpr int kDcSlots = 4;
    vigo::DevConst* dc_stage = nullptr;   // hipHostMalloc'ed [kDcSlots]
    hipEvent_t dc_event[kDcSlots] = {};
    int dc_next = 0;
    int precision = VIGO_PREC_F64;
    vigo::LaunchState launch;
    std::string last_error;
    // voxel snapshot
    uint32_t* grid_planes = nullptr;
    size_t grid_capacity_bytes = 0;
    vigo::GridView grid{};
    bool has_grid = false;
    // esdf
    float* esdf = nullptr;
    size_t esdf_capacity = 0;
    vigo::EsdfView esdf_view{};
    bool has_esdf = false;
    // least-squares operator of the B-spline fit for (fit_K, fit_ts), transposed (vigo_fit.hip)
    double* fit_pinvT = nullptr;
    size_t fit_capacity = 0;   // doubles
    int fit_K = 0;
    double fit_ts = 0.0;
    // sample clock of the gates, cached per (dt, tmax): filled on the device, no host round trip per call
    double* times_dev = nullptr;
    size_t times_cap = 0;      // doubles
    double times_dt = -1.0, times_tmax = -1.0;
    int times_T = -1;
    hipStream_t times_stream = nullptr;
    // vigo_rebound_rounds: count (16 ints, first used) + compacted indices
    int32_t* rebound_idx = nullptr;
    size_t rebound_cap = 0;    // trajectories
    // scratch (sample-time tables, corridor checkpoints, staging of *_host calls)
    void* scratch = nullptr;
    size_t scratch_bytes = 0;
};
