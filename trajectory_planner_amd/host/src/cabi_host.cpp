// cabi_host.cpp — small extern "C" entry points of libtrajectory_planner_vigo.so so the Python
// tests can exercise the host-side pieces that have no GPU part (the .bt reader and the min-snap
// QP) with ctypes.  Not part of include/vigo.h (that is the device ABI).
#ifdef VIGO_WITH_ROS
#error "tools of the in-tree dense map (standin/dense_occmap.h): not part of a build against map_manager"
#endif
#include <trajectory_planner/bsplineTraj.h>
#include <trajectory_planner/octomapBt.h>
#include <trajectory_planner/path_search/astarOcc.h>
#include <trajectory_planner/piecewiseLinearTraj.h>
#include <trajectory_planner/polyTrajOctomap.h>
#include <trajectory_planner/polyTrajSolver.h>

#include <chrono>
#include <cstring>
#include <mutex>

#include "workerPool.h"

extern "C" {

// returns 0 on success; info: nodes_header, nodes_parsed, bytes, occupied, free, nx, ny, nz; origin[3]; res
int vigo_host_bt_info(const char* path, long long* info, double* origin, double* res) {
    trajPlanner::BtInfo bi;
    const double inflate[3] = {0, 0, 0};
    auto m = trajPlanner::loadOctomapBt(path, inflate, 0, &bi);
    if (!m) return -1;
    info[0] = bi.nodes_header; info[1] = bi.nodes_parsed; info[2] = bi.bytes_consumed; info[3] = bi.occupied; info[4] = bi.free_;
    info[5] = m->nx(); info[6] = m->ny(); info[7] = m->nz();
    for (int a = 0; a < 3; ++a) origin[a] = m->origin()(a);
    *res = m->getRes();
    return 0;
}

// dense voxels of a .bt (caller allocates nx*ny*nz bytes as reported by vigo_host_bt_info with the same arguments)
int vigo_host_bt_load(const char* path, const double* inflate, int margin, unsigned char* out, long long cap, int* dims, double* origin) {
    auto m = trajPlanner::loadOctomapBt(path, inflate, margin, nullptr);
    if (!m) return -1;
    dims[0] = m->nx(); dims[1] = m->ny(); dims[2] = m->nz();
    for (int a = 0; a < 3; ++a) origin[a] = m->origin()(a);
    if ((long long)m->voxels().size() > cap) return -2;
    std::memcpy(out, m->voxels().data(), m->voxels().size());
    return 0;
}

// dense voxels of an ASCII .pcd at resolution res: first call with out == NULL to get dims/origin/points
int vigo_host_pcd_load(const char* path, double res, const double* inflate, int margin, unsigned char* out, long long cap, int* dims,
                       double* origin, long long* points) {
    auto m = trajPlanner::loadPcdAscii(path, res, inflate, margin, points);
    if (!m) return -1;
    dims[0] = m->nx(); dims[1] = m->ny(); dims[2] = m->nz();
    for (int a = 0; a < 3; ++a) origin[a] = m->origin()(a);
    if (!out) return 0;
    if ((long long)m->voxels().size() > cap) return -2;
    std::memcpy(out, m->voxels().data(), m->voxels().size());
    return 0;
}

// bsplineTraj's host prologue of makePlan() (BT.cpp:333-350: findCollisionSeg -> pathSearch -> assignGuidePointsSemiCircle)
// on a dense byte grid, after updatePath() over n_path poses (zero start / end conditions).  cfg: distance_threshold,
// min_height, max_height, max_obstacle_size[3].  Outputs (caller-sized, cap doubles / ints each):
//   ctrl_out  3 * N control points (column by column), returns N through *n_ctrl
//   seg_out   pairs of the collision segments AFTER pathSearch, *n_seg of them; -1 in *n_seg when A* failed
//   guide_off N + 1 offsets into guide_out, which holds (point, direction) 6-tuples per control point, in push order
//   path_off / path_out: the A* paths (with the segment ends put in, BT.cpp:455-457) as xyz triples
// Returns 0, -1 when updatePath refuses the path, -2 when a buffer is too small.
int vigo_host_bspline_prologue(const unsigned char* vox, const int* dims, const double* origin, double res, int n_path, const double* path_xyz,
                               const double* cfg, double* ctrl_out, int* n_ctrl, int* seg_out, int* n_seg, int* guide_off, double* guide_out,
                               int* path_off, double* path_out, int cap) {
    auto m = std::make_shared<mapManager::occMap>(dims[0], dims[1], dims[2], Eigen::Vector3d(origin[0], origin[1], origin[2]), res);
    std::memcpy(m->voxels().data(), vox, m->voxels().size());
    ros::NodeHandle nh;
    nh.setParam("bspline_traj/distance_threshold", cfg[0]);
    nh.setParam("bspline_traj/min_height", cfg[1]);
    nh.setParam("bspline_traj/max_height", cfg[2]);
    nh.setParam("bspline_traj/max_obstacle_size", std::vector<double>{cfg[3], cfg[4], cfg[5]});
    nh.setParam("bspline_traj/max_path_length", 1000.0);
    trajPlanner::bsplineTraj bt(nh);
    bt.setMap(m);
    nav_msgs::Path path;
    for (int i = 0; i < n_path; ++i) {
        geometry_msgs::PoseStamped ps;
        ps.pose.position.x = path_xyz[3 * i]; ps.pose.position.y = path_xyz[3 * i + 1]; ps.pose.position.z = path_xyz[3 * i + 2];
        path.poses.push_back(ps);
    }
    if (!bt.updatePath(path, std::vector<Eigen::Vector3d>(4, Eigen::Vector3d(0, 0, 0)))) return -1;
    const Eigen::MatrixXd c = bt.getControlPoints();
    const int N = (int)c.cols();
    if (3 * N > cap) return -2;
    *n_ctrl = N;
    for (int i = 0; i < N; ++i) for (int k = 0; k < 3; ++k) ctrl_out[3 * i + k] = c(k, i);
    std::vector<std::pair<int, int>> seg;
    std::vector<std::vector<Eigen::Vector3d>> paths;
    bt.findCollisionSeg(c, seg);
    if (!bt.pathSearch(seg, paths)) { *n_seg = -1; return 0; }
    bt.assignGuidePointsSemiCircle(paths, seg);
    if (2 * (int)seg.size() > cap) return -2;
    *n_seg = (int)seg.size();
    for (size_t i = 0; i < seg.size(); ++i) { seg_out[2 * i] = seg[i].first; seg_out[2 * i + 1] = seg[i].second; }
    const trajPlanner::optData& od = bt.getOptData();
    int g = 0;
    for (int i = 0; i < N; ++i) {
        guide_off[i] = g;
        for (size_t j = 0; j < od.guidePoints[i].size(); ++j) {
            if (6 * (g + 1) > cap) return -2;
            for (int k = 0; k < 3; ++k) { guide_out[6 * g + k] = od.guidePoints[i][j](k); guide_out[6 * g + 3 + k] = od.guideDirections[i][j](k); }
            ++g;
        }
    }
    guide_off[N] = g;
    int q = 0;
    for (size_t i = 0; i < paths.size(); ++i) {
        path_off[i] = q;
        for (const auto& v : paths[i]) {
            if (3 * (q + 1) > cap) return -2;
            for (int k = 0; k < 3; ++k) path_out[3 * q + k] = v(k);
            ++q;
        }
    }
    path_off[paths.size()] = q;
    return 0;
}

// one AStar::AstarSearch on a dense byte grid (bit 0 = inflated-occupied): returns the number of path points written
// (xyz triples, start side first), -1 when no path is found, -2 when path_out is too small
int vigo_host_astar(const unsigned char* vox, const int* dims, const double* origin, double res, const int* pool, double min_height,
                    double max_height, double step, const double* start, const double* end, double* path_out, int cap) {
    auto m = std::make_shared<mapManager::occMap>(dims[0], dims[1], dims[2], Eigen::Vector3d(origin[0], origin[1], origin[2]), res);
    std::memcpy(m->voxels().data(), vox, m->voxels().size());
    AStar a;
    a.initGridMap(m, Eigen::Vector3i(pool[0], pool[1], pool[2]), min_height, max_height);
    if (!a.AstarSearch(step, Eigen::Vector3d(start[0], start[1], start[2]), Eigen::Vector3d(end[0], end[1], end[2]))) return -1;
    const std::vector<Eigen::Vector3d> path = a.getPath();
    if ((int)path.size() > cap) return -2;
    for (size_t i = 0; i < path.size(); ++i)
        for (int k = 0; k < 3; ++k) path_out[3 * i + k] = path[i](k);
    return (int)path.size();
}

// min-snap through n_wp waypoints (xyz triples); corridor == NULL: equality-constrained only.
// coeffs_out: 3 * (n_wp-1) * (deg+1) doubles (x block, y block, z block), knots_out: n_wp doubles.
int vigo_host_minsnap(int n_wp, const double* wp, int deg, int diff, int cont, double vel, const double* corridor,
                      double corridor_res, double* coeffs_out, double* knots_out) {
    std::vector<trajPlanner::pose> path;
    for (int i = 0; i < n_wp; ++i) path.push_back(trajPlanner::pose(wp[3 * i], wp[3 * i + 1], wp[3 * i + 2]));
    trajPlanner::polyTrajSolver s(deg, diff, cont, vel);
    s.updatePath(path);
    if (corridor) s.setCorridorConstraint(std::vector<double>(corridor, corridor + n_wp - 1), corridor_res);
    if (!s.solve()) return -1;
    const int n = (n_wp - 1) * (deg + 1);
    for (int a = 0; a < 3; ++a) std::memcpy(coeffs_out + (size_t)a * n, s.getSolution(a).data(), sizeof(double) * n);
    std::memcpy(knots_out, s.getTimeKnot().data(), sizeof(double) * n_wp);
    return 0;
}

// trajPlanner::pwlTraj over n_wp poses (x, y, z, yaw): updatePath(path[, desired_vel], use_yaw), makePlan(traj, delT).
// desired_vel <= 0: the class default.  traj_out: up to cap poses (x, y, z, yaw); knots_out: up to 2 n_wp doubles.
// Returns the number of trajectory poses, *n_knots the number of time knots; -2 when traj_out is too small.
int vigo_host_pwl(int n_wp, const double* wp, int use_yaw, double desired_vel, double delT, double* traj_out, int cap, double* knots_out,
                  int* n_knots) {
    std::vector<trajPlanner::pose> path;
    for (int i = 0; i < n_wp; ++i) path.push_back(trajPlanner::pose(wp[4 * i], wp[4 * i + 1], wp[4 * i + 2], wp[4 * i + 3]));
    ros::NodeHandle nh;
    trajPlanner::pwlTraj pw(nh);
    if (desired_vel > 0) pw.updatePath(path, desired_vel, use_yaw != 0);
    else pw.updatePath(path, use_yaw != 0);
    std::vector<trajPlanner::pose> traj;
    pw.makePlan(traj, delT);
    const std::vector<double> k = pw.getTimeKnot();
    *n_knots = (int)k.size();
    for (size_t i = 0; i < k.size(); ++i) knots_out[i] = k[i];
    if ((int)traj.size() > cap) return -2;
    for (size_t i = 0; i < traj.size(); ++i) { traj_out[4 * i] = traj[i].x; traj_out[4 * i + 1] = traj[i].y; traj_out[4 * i + 2] = traj[i].z; traj_out[4 * i + 3] = traj[i].yaw; }
    return (int)traj.size();
}

// min-snap with SOFT interior waypoints (polyTrajSolver::setSoftConstraint, PS.cpp:943-958): soft[3] = half sizes per axis
int vigo_host_minsnap_soft(int n_wp, const double* wp, int deg, int diff, int cont, double vel, const double* soft,
                           double* coeffs_out, double* knots_out) {
    std::vector<trajPlanner::pose> path;
    for (int i = 0; i < n_wp; ++i) path.push_back(trajPlanner::pose(wp[3 * i], wp[3 * i + 1], wp[3 * i + 2]));
    trajPlanner::polyTrajSolver s(deg, diff, cont, vel);
    s.updatePath(path);
    s.setSoftConstraint(soft[0], soft[1], soft[2]);
    if (!s.solve()) return -1;
    const int n = (n_wp - 1) * (deg + 1);
    for (int a = 0; a < 3; ++a) std::memcpy(coeffs_out + (size_t)a * n, s.getSolution(a).data(), sizeof(double) * n);
    std::memcpy(knots_out, s.getTimeKnot().data(), sizeof(double) * n_wp);
    return 0;
}

// the same solve, then polyTrajSolver::getPose / getVel / getAcc at n_t times: out[n_t][9] = position, velocity, acceleration
int vigo_host_minsnap_eval(int n_wp, const double* wp, int deg, int diff, int cont, double vel, int n_t, const double* t,
                           double* out) {
    std::vector<trajPlanner::pose> path;
    for (int i = 0; i < n_wp; ++i) path.push_back(trajPlanner::pose(wp[3 * i], wp[3 * i + 1], wp[3 * i + 2]));
    trajPlanner::polyTrajSolver s(deg, diff, cont, vel);
    s.updatePath(path);
    if (!s.solve()) return -1;
    for (int k = 0; k < n_t; ++k) {
        const trajPlanner::pose p = s.getPose(t[k]);
        const Eigen::Vector3d v = s.getVel(t[k]), a = s.getAcc(t[k]);
        double* o = out + 9 * (size_t)k;
        o[0] = p.x; o[1] = p.y; o[2] = p.z;
        for (int q = 0; q < 3; ++q) { o[3 + q] = v(q); o[6 + q] = a(q); }
    }
    return 0;
}

// BASELINE configs[0]: one polyTrajOctomap::makePlan() (cfg/planner_interactive.yaml values passed in
// `cfg`: box[3], map_resolution, sample_delta_time, desired_velocity, initial_radius, shrinking_factor,
// corridor_res, maximum_iteration_num, traj_timeout, mode) on a dense byte grid (vigo.h voxel
// contract).  traj_out: up to traj_cap xyz triples.  info_out: valid, iterations, samples, duration,
// seconds of makePlan.  Needs the GPU (the box sweep of every sample runs there); -1 on failure.
int vigo_host_poly_plan(int nx, int ny, int nz, const double* origin, double res, const unsigned char* voxels, int n_wp,
                        const double* wp, const double* cfg, double* traj_out, int traj_cap, double* info_out) {
    auto map = std::make_shared<mapManager::occMap>(nx, ny, nz, Eigen::Vector3d(origin[0], origin[1], origin[2]), res);
    std::memcpy(map->voxels().data(), voxels, (size_t)nx * ny * nz);
    ros::NodeHandle nh;
    nh.setParam("collision_box", std::vector<double>{cfg[0], cfg[1], cfg[2]});
    nh.setParam("map_resolution", cfg[3]);
    nh.setParam("sample_delta_time", cfg[4]);
    nh.setParam("desired_velocity", cfg[5]);
    nh.setParam("initial_radius", cfg[6]);
    nh.setParam("shrinking_factor", cfg[7]);
    nh.setParam("corridor_res", cfg[8]);
    nh.setParam("maximum_iteration_num", cfg[9]);
    nh.setParam("traj_timeout", cfg[10]);
    nh.setParam("mode", cfg[11]);
    nh.setParam("polynomial_degree", 7.0);
    nh.setParam("differential_degree", 4.0);
    nh.setParam("continuity_degree", 4.0);
    trajPlanner::polyTrajOctomap planner(nh);
    planner.setMap(map);
    std::vector<trajPlanner::pose> path, traj;
    for (int i = 0; i < n_wp; ++i) path.push_back(trajPlanner::pose(wp[3 * i], wp[3 * i + 1], wp[3 * i + 2]));
    if (planner.checkCollision(path.front())) { /* first device call: creates the handle, snapshots the map */ }
    planner.updatePath(path);
    const auto t0 = std::chrono::steady_clock::now();
    planner.makePlan(traj, cfg[4]);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const int n = (int)traj.size() < traj_cap ? (int)traj.size() : traj_cap;
    for (int i = 0; i < n; ++i) { traj_out[3 * i] = traj[i].x; traj_out[3 * i + 1] = traj[i].y; traj_out[3 * i + 2] = traj[i].z; }
    info_out[0] = planner.isValid() ? 1.0 : 0.0;
    info_out[1] = planner.getIterations();
    info_out[2] = (double)traj.size();
    info_out[3] = planner.getDuration();
    info_out[4] = secs;
    return 0;
}

}  // extern "C"

// ---- the same prologue for MANY paths on ONE map (the workload generator of bench.py / tests: product code, no oracle) ----
namespace {
// planners parked between jobs: each owns an A* node pool sized by max_obstacle_size, shares the map
struct PlannerPool {
    std::shared_ptr<mapManager::occMap> map;
    ros::NodeHandle nh;
    std::mutex m;
    std::vector<std::unique_ptr<trajPlanner::bsplineTraj>> idle;
    std::unique_ptr<trajPlanner::bsplineTraj> take() {
        {
            std::lock_guard<std::mutex> lk(m);
            if (!idle.empty()) { auto p = std::move(idle.back()); idle.pop_back(); return p; }
        }
        std::unique_ptr<trajPlanner::bsplineTraj> p(new trajPlanner::bsplineTraj(nh));
        p->setMap(map);
        return p;
    }
    void give(std::unique_ptr<trajPlanner::bsplineTraj> p) {
        std::lock_guard<std::mutex> lk(m);
        idle.push_back(std::move(p));
    }
};

void initPool(PlannerPool& pool, const unsigned char* vox, const int* dims, const double* origin, double res, const double* cfg) {
    pool.map = std::make_shared<mapManager::occMap>(dims[0], dims[1], dims[2], Eigen::Vector3d(origin[0], origin[1], origin[2]), res);
    std::memcpy(pool.map->voxels().data(), vox, pool.map->voxels().size());
    pool.nh.setParam("bspline_traj/distance_threshold", cfg[0]);
    pool.nh.setParam("bspline_traj/min_height", cfg[1]);
    pool.nh.setParam("bspline_traj/max_height", cfg[2]);
    pool.nh.setParam("bspline_traj/max_obstacle_size", std::vector<double>{cfg[3], cfg[4], cfg[5]});
    pool.nh.setParam("bspline_traj/max_path_length", 1000.0);
}

// guide pairs of one planner, control point by control point in push order, into one trajectory's slot of the
// per-trajectory staging (cnt[N], pairs appended to pv)
void collectGuides(const trajPlanner::optData& od, int N, std::vector<int>& cnt, std::vector<double>& pv) {
    cnt.assign(N, 0);
    pv.clear();
    for (int i = 0; i < N; ++i) {
        cnt[i] = (int)od.guidePoints[i].size();
        for (size_t j = 0; j < od.guidePoints[i].size(); ++j) {
            for (int k = 0; k < 3; ++k) pv.push_back(od.guidePoints[i][j](k));
            for (int k = 0; k < 3; ++k) pv.push_back(od.guideDirections[i][j](k));
        }
    }
}
}  // namespace

extern "C" {

// n paths of n_pts poses each (xyz) on one dense byte grid -> per path: status (0 planned, -1 updatePath refused the path or
// the control-point count is not N, -2 A* failed: no guides), N control points, the number of collision segments after
// pathSearch, and the guide pairs of makePlan()'s prologue (BT.cpp:333-350) as CSR over n * N control points:
// guide_off[n * N + 1], guide_pv[6 per pair] (at most cap_pairs; -2 is returned when they do not fit).
// ctrl_in != NULL: SKIP updatePath and replay the host steps the rebound loop takes at failCount >= 4 (BT.cpp:640-648:
// findCollisionSeg -> pathSearch -> assignGuidePointsSemiCircle) on these CURRENT control points [n][N][3], starting from
// empty lists: the pairs returned are the ones that step would APPEND.  cfg as in vigo_host_bspline_prologue.
int vigo_host_bspline_guides_batch(const unsigned char* vox, const int* dims, const double* origin, double res, int n, int n_pts,
                                   const double* path_xyz, const double* ctrl_in, int N, const double* cfg, double* ctrl_out, int* status,
                                   int* n_seg, int* guide_off, double* guide_pv, long long cap_pairs) {
    if (n < 0 || N < 7 || (!path_xyz && !ctrl_in)) return -1;
    PlannerPool pool;
    initPool(pool, vox, dims, origin, res, cfg);
    std::vector<std::vector<int>> cnt(n);
    std::vector<std::vector<double>> pv(n);
    vigo_host::parallelFor((size_t)n, [&](size_t t) {
        auto bt = pool.take();
        status[t] = 0;
        n_seg[t] = 0;
        cnt[t].assign(N, 0);
        bool have = true;
        if (ctrl_in) {
            Eigen::MatrixXd c(3, N);
            for (int i = 0; i < N; ++i) for (int k = 0; k < 3; ++k) c(k, i) = ctrl_in[((size_t)t * N + i) * 3 + k];
            bt->setControlPoints(c);
        } else {
            nav_msgs::Path path;
            for (int i = 0; i < n_pts; ++i) {
                geometry_msgs::PoseStamped ps;
                const double* q = path_xyz + ((size_t)t * n_pts + i) * 3;
                ps.pose.position.x = q[0]; ps.pose.position.y = q[1]; ps.pose.position.z = q[2];
                path.poses.push_back(ps);
            }
            have = bt->updatePath(path, std::vector<Eigen::Vector3d>(4, Eigen::Vector3d(0, 0, 0))) && bt->getControlPoints().cols() == N;
        }
        if (!have) {
            status[t] = -1;
        } else {
            const Eigen::MatrixXd c = bt->getControlPoints();
            if (ctrl_out) for (int i = 0; i < N; ++i) for (int k = 0; k < 3; ++k) ctrl_out[((size_t)t * N + i) * 3 + k] = c(k, i);
            std::vector<std::pair<int, int>> seg;
            std::vector<std::vector<Eigen::Vector3d>> paths;
            bt->findCollisionSeg(c, seg);
            if (!bt->pathSearch(seg, paths)) {
                status[t] = -2;
            } else {
                bt->assignGuidePointsSemiCircle(paths, seg);
                n_seg[t] = (int)seg.size();
                collectGuides(bt->getOptData(), N, cnt[t], pv[t]);
            }
        }
        pool.give(std::move(bt));
    });
    long long g = 0;
    for (int t = 0; t < n; ++t)
        for (int i = 0; i < N; ++i) { guide_off[(size_t)t * N + i] = (int)g; g += cnt[t][i]; }
    guide_off[(size_t)n * N] = (int)g;
    if (g > cap_pairs) return -2;
    long long w = 0;
    for (int t = 0; t < n; ++t) { std::memcpy(guide_pv + 6 * w, pv[t].data(), pv[t].size() * sizeof(double)); w += (long long)pv[t].size() / 6; }
    return 0;
}

}  // extern "C"

// mapAdapter::rasterise (the generic route: four public map methods only) over the whole box of a dense map must give
// back that map's inflated-occupied and unknown bits.  Returns the number of differing voxels (0 = agreement), -1 on
// failure; dims_out receives the rasterised grid's extents.
extern "C" long long vigo_host_rasterise_check(int nx, int ny, int nz, const double* origin, double res, const unsigned char* voxels,
                                               const double* box_min, const double* box_max, int* dims_out) {
    auto map = std::make_shared<mapManager::occMap>(nx, ny, nz, Eigen::Vector3d(origin[0], origin[1], origin[2]), res);
    std::memcpy(map->voxels().data(), voxels, (size_t)nx * ny * nz);
    trajPlanner::mapRegion region;
    region.set = true;
    region.boxMin = Eigen::Vector3d(box_min[0], box_min[1], box_min[2]);
    region.boxMax = Eigen::Vector3d(box_max[0], box_max[1], box_max[2]);
    std::vector<uint8_t> vox;
    int dims[3];
    double o[3];
    if (!trajPlanner::mapAdapter::rasterise(*map, region, vox, dims, o)) return -1;
    for (int a = 0; a < 3; ++a) dims_out[a] = dims[a];
    long long bad = 0;
    for (int ix = 0; ix < dims[0]; ++ix)
        for (int iy = 0; iy < dims[1]; ++iy)
            for (int iz = 0; iz < dims[2]; ++iz) {
                const Eigen::Vector3d c(o[0] + (ix + 0.5) * res, o[1] + (iy + 0.5) * res, o[2] + (iz + 0.5) * res);
                const unsigned want = map->byteAt(c);          // 0xFF outside the dense map: occupied and unknown
                const unsigned got = vox[((size_t)ix * dims[1] + iy) * dims[2] + iz];
                if ((got & 1u) != (want & 1u) || ((got >> 1) & 1u) != ((want >> 1) & 1u) || ((got >> 2) & 1u) != (got & 1u)) ++bad;
            }
    return bad;
}
