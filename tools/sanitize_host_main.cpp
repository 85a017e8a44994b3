#include <trajectory_planner/octomapBt.h>
#include <trajectory_planner/polyTrajSolver.h>
#include <trajectory_planner/bspline.h>
#include <trajectory_planner/path_search/astarOcc.h>
#include <trajectory_planner/piecewiseLinearTraj.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <string>
ros::Time ros::Time::now() { return ros::Time(); }   // (the in-tree stand-in's clock lives in bsplineTraj.cpp, which needs HIP)

int main(int argc, char** argv) {
    using namespace trajPlanner;
    int fails = 0;
    const double inflate[3] = {0.1, 0.1, 0.0};
    for (int i = 1; i < argc; ++i) {
        const std::string arg = argv[i];
        if (arg.size() > 4 && arg.substr(arg.size() - 4) == ".pcd") {
            long long n = 0;
            auto m = loadPcdAscii(arg, 0.1, inflate, 1, &n);
            std::printf("%s: %s points %lld dims %d %d %d\n", argv[i], m ? "ok" : "FAILED", n, m ? m->nx() : 0, m ? m->ny() : 0, m ? m->nz() : 0);
            if (!m) ++fails;
            continue;
        }
        BtInfo bi;
        auto m = loadOctomapBt(argv[i], inflate, 2, &bi);
        std::printf("%s: %s nodes %lld/%lld dims %d %d %d\n", argv[i], m ? "ok" : "FAILED", bi.nodes_parsed, bi.nodes_header, m ? m->nx() : 0, m ? m->ny() : 0, m ? m->nz() : 0);
        if (!m || bi.nodes_parsed != bi.nodes_header) ++fails;
    }
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(-1, 1);
    for (int trial = 0; trial < 300; ++trial) {
        const int W = 2 + trial % 10;
        std::vector<pose> path;
        double x = 0, y = 0;
        for (int i = 0; i < W; ++i) { path.push_back(pose(x, y, 1.0 + 0.1 * U(rng))); x += 1.0 + 2.0 * std::fabs(U(rng)); y += 2.0 * U(rng); }
        polyTrajSolver s(7, 4, 4, 1.0);
        s.updatePath(path);
        if (trial % 3) s.setCorridorConstraint(std::vector<double>(W - 1, 0.05 + 0.5 * std::fabs(U(rng))), 8.0);
        const bool ok = s.solve();
        if (ok) { std::vector<pose> tr; s.getTrajectory(tr, 0.1); if (tr.empty()) ++fails; }
    }
    // B-spline fit + evaluation
    for (int K = 4; K < 70; K += 7) {
        std::vector<Eigen::Vector3d> pts, cond(4, Eigen::Vector3d(0.1, -0.2, 0.0));
        for (int i = 0; i < K; ++i) pts.push_back(Eigen::Vector3d(0.25 * i, 0.1 * U(rng), 1.0));
        Eigen::MatrixXd C;
        if (!bspline::parameterizeToBspline(0.2, pts, cond, C) || C.cols() != K + 2) ++fails;
        bspline b(3, C, 0.2);
        for (double t = 0; t <= b.getDuration(); t += 0.05) (void)b.at(t);
        (void)b.getDerivative().getDerivative().at(0.3);
    }
    // A* on a small map with a wall and a gap (guide search of the B-spline planner)
    {
        auto m = std::make_shared<mapManager::occMap>(64, 64, 20, Eigen::Vector3d(-3.2, -3.2, 0.0), 0.1);
        for (int iy = 0; iy < 64; ++iy)
            for (int iz = 0; iz < 20; ++iz)
                if (iy < 20 || iy > 26) m->at(32, iy, iz) |= 5;
        AStar a;
        a.initGridMap(m, Eigen::Vector3i(100, 100, 100), 0.0, 2.0);
        for (int k = 0; k < 20; ++k) {
            const bool found = a.AstarSearch(0.1, Eigen::Vector3d(-2.0, 2.0 * U(rng), 1.0), Eigen::Vector3d(2.0, 2.0 * U(rng), 1.0));
            if (found && a.getPath().size() < 2) ++fails;
        }
    }
    // A* on random box worlds: a found path is connected on the 26-neighbourhood lattice, stays out of inflated
    // voxels and ends next to the (possibly shifted, AS.cpp:58-102) end points; hostile queries (identical,
    // outside the pool, NaN) come back false
    {
        long found = 0, asked = 0, broken = 0;
        for (int world = 0; world < 12; ++world) {
            auto m = std::make_shared<mapManager::occMap>(80, 80, 24, Eigen::Vector3d(-4.0, -4.0, 0.0), 0.1);
            const int nb = 2 + world;
            for (int b = 0; b < nb; ++b) {
                const int cx = 10 + (int)(30 * (U(rng) + 1)), cy = 10 + (int)(30 * (U(rng) + 1)), hx = 1 + (int)(4 * std::fabs(U(rng))), hy = 1 + (int)(6 * std::fabs(U(rng)));
                for (int x = std::max(0, cx - hx); x < std::min(80, cx + hx); ++x)
                    for (int y = std::max(0, cy - hy); y < std::min(80, cy + hy); ++y)
                        for (int z = 0; z < 24; ++z) m->at(x, y, z) |= 5;
            }
            AStar a;
            a.initGridMap(m, Eigen::Vector3i(100, 100, 100), 0.0, 2.0);
            for (int k = 0; k < 40; ++k) {
                const Eigen::Vector3d s(3.5 * U(rng), 3.5 * U(rng), 0.5 + 1.0 * std::fabs(U(rng))), e(3.5 * U(rng), 3.5 * U(rng), 0.5 + 1.0 * std::fabs(U(rng)));
                ++asked;
                if (!a.AstarSearch(0.1, s, e)) continue;
                ++found;
                const std::vector<Eigen::Vector3d> path = a.getPath();
                bool ok = path.size() >= 1;
                for (size_t i = 0; i < path.size() && ok; ++i) {
                    if (i > 0 && (path[i] - path[i - 1]).norm() > 0.1 * std::sqrt(3.0) + 1e-9) ok = false;
                    if (i > 0 && i + 1 < path.size() && m->isInflatedOccupied(path[i])) ok = false;
                }
                if (!ok) { ++broken; ++fails; }
            }
            const double nan = std::nan("");
            if (a.AstarSearch(0.1, Eigen::Vector3d(60, 0, 1), Eigen::Vector3d(0, 0, 1))) ++fails;             // outside the pool
            if (a.AstarSearch(0.1, Eigen::Vector3d(nan, 0, 1), Eigen::Vector3d(0, 0, 1))) ++fails;
            (void)a.AstarSearch(0.1, Eigen::Vector3d(0.33, 0.2, 1), Eigen::Vector3d(0.33, 0.2, 1));            // identical ends
            (void)a.AstarSearch(0.0, Eigen::Vector3d(0, 0, 1), Eigen::Vector3d(1, 0, 1));                       // zero step
        }
        std::printf("A*: %ld of %ld random queries found a path, %ld broken\n", found, asked, broken);
    }
    // pwlTraj (the rotate-then-move fallback) on random paths incl. coincident waypoints, both yaw modes; soft waypoint boxes
    {
        ros::NodeHandle nh;
        for (int k = 0; k < 200; ++k) {
            const int n = 1 + (int)(5 * std::fabs(U(rng)));
            std::vector<pose> path;
            for (int i = 0; i < n; ++i) path.push_back(pose(3 * U(rng), 3 * U(rng), 1 + 0.2 * U(rng), 3.0 * U(rng)));
            if (k % 7 == 0 && n >= 2) path[1] = path[0];
            pwlTraj pw(nh);
            if (k & 1) pw.updatePath(path, k % 3 == 0); else pw.updatePath(path, 0.5 + std::fabs(U(rng)), k % 3 == 0);
            std::vector<pose> traj;
            pw.makePlan(traj, 0.1);
            if (n >= 2 && traj.empty()) ++fails;
            (void)pw.getPose(-1.0); (void)pw.getPose(0.5 * pw.getDuration()); (void)pw.getPose(1e9); (void)pw.getFirstPose();
        }
        for (int k = 0; k < 60; ++k) {
            const int n = 3 + (int)(4 * std::fabs(U(rng)));
            std::vector<pose> path;
            for (int i = 0; i < n; ++i) path.push_back(pose(1.5 * i + 0.4 * U(rng), 1.0 * U(rng), 1.0 + 0.1 * U(rng)));
            polyTrajSolver sv(7, 4, 4, 1.0);
            sv.updatePath(path);
            sv.setSoftConstraint(0.3 * std::fabs(U(rng)), 0.3 * std::fabs(U(rng)), k % 2 ? 0.0 : 0.1);
            if (!sv.solve()) ++fails;
            (void)sv.getVel(0.3); (void)sv.getAcc(0.3); (void)sv.getPos(0.3);
        }
    }
    std::printf("%s\n", fails ? "FAILED" : "sanitizer run complete, no failures");
    return fails;
}
