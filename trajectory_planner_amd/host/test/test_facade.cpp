// test_facade.cpp — drives the C++ facades the way the reference's nodes drive the originals
// (src/bspline_node.cpp:227-231, :344-378; src/poly_RRT_node.cpp:148-150).  Runs on the GPU box
// (tests/test_gpu_facade.py); prints one "KEY value" line per check and exits non-zero on failure.
#include <trajectory_planner/bsplineTraj.h>
#include <trajectory_planner/polyTrajOctomap.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <iostream>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <cstring>
#include <set>
#include <thread>

using trajPlanner::bsplineTraj;

static int fails = 0;
#define CHECK(cond, msg) do { if (!(cond)) { std::printf("FAIL %s\n", msg); ++fails; } else std::printf("ok   %s\n", msg); } while (0)

static std::shared_ptr<mapManager::occMap> makeMap() {
    // 12.8 m cube at 0.1 m, origin on the key lattice; a wall with a gap and a pillar
    auto m = std::make_shared<mapManager::occMap>(128, 128, 40, Eigen::Vector3d(-6.4, -6.4, -0.5), 0.1);
    auto box = [&](double x0, double x1, double y0, double y1, double z0, double z1) {
        for (int ix = 0; ix < 128; ++ix) for (int iy = 0; iy < 128; ++iy) for (int iz = 0; iz < 40; ++iz) {
            const double x = -6.4 + (ix + 0.5) * 0.1, y = -6.4 + (iy + 0.5) * 0.1, z = -0.5 + (iz + 0.5) * 0.1;
            if (x >= x0 && x <= x1 && y >= y0 && y <= y1 && z >= z0 && z <= z1) m->at(ix, iy, iz) |= 4;
            if (x >= x0 - 0.4 && x <= x1 + 0.4 && y >= y0 - 0.4 && y <= y1 + 0.4 && z >= z0 - 0.15 && z <= z1 + 0.15) m->at(ix, iy, iz) |= 1;
        }
    };
    box(-0.3, 0.3, -0.8, 0.8, -0.5, 3.5);   // pillar across the straight path
    return m;
}

static ros::NodeHandle makeParams() {
    ros::NodeHandle nh;   // cfg/bspline_interactive/bspline_planner_param.yaml
    nh.setParam("bspline_traj/timestep", 0.1);
    nh.setParam("bspline_traj/distance_threshold", 0.5);
    nh.setParam("bspline_traj/distance_threshold_dynamic", 0.5);
    nh.setParam("bspline_traj/weight_distance", 1.0);
    nh.setParam("bspline_traj/weight_smoothness", 1.0);
    nh.setParam("bspline_traj/weight_feasibility", 1.0);
    nh.setParam("bspline_traj/weight_dynamic_obstacle", 1.0);
    nh.setParam("bspline_traj/plan_in_z_axis", 0.0);
    nh.setParam("bspline_traj/min_height", 0.7);
    nh.setParam("bspline_traj/max_height", 1.3);
    nh.setParam("bspline_traj/uncertain_aware_factor", 1.0);
    nh.setParam("bspline_traj/max_path_length", 20.0);
    nh.setParam("bspline_traj/max_obstacle_size", std::vector<double>{5, 5, 3});
    nh.setParam("bspline_traj/prediction_horizon", 2.0);
    return nh;
}

static nav_msgs::Path straight(double x0, double y0, double x1, double y1, double z, double spacing) {
    nav_msgs::Path p;
    const double len = std::hypot(x1 - x0, y1 - y0);
    const int n = (int)(len / spacing);
    for (int i = 0; i <= n; ++i) {
        geometry_msgs::PoseStamped ps;
        ps.pose.position.x = x0 + (x1 - x0) * i / n;
        ps.pose.position.y = y0 + (y1 - y0) * i / n;
        ps.pose.position.z = z;
        p.poses.push_back(ps);
    }
    return p;
}

int main() {
    auto map = makeMap();
    const std::vector<Eigen::Vector3d> cond(4, Eigen::Vector3d(0, 0, 0));

    // ---- single planner, the bspline_node.cpp sequence ----
    bsplineTraj bst(makeParams());
    bst.setMap(map);
    bst.updateMaxVel(2.0);
    bst.updateMaxAcc(3.0);
    nav_msgs::Path path = straight(-3.0, 0.05, 3.0, 0.0, 1.0, bst.getControlPointDist());
    CHECK(bst.updatePath(path, cond), "updatePath accepts a free goal");
    Eigen::MatrixXd c0 = bst.getControlPoints();
    CHECK(c0.cols() == (int)path.poses.size() + 2, "fit returns K+2 control points");
    CHECK(bst.hasCollisionTrajectory(c0), "the straight path collides before planning (device gate)");
    const bool ok = bst.makePlan();
    CHECK(ok, "makePlan succeeds around the pillar");
    Eigen::MatrixXd c1 = bst.getControlPoints();
    CHECK(!bst.hasCollisionTrajectory(c1), "the planned trajectory is collision free (device gate)");
    Eigen::Vector3d firstHit;
    CHECK(bst.isCurrTrajValid(firstHit), "isCurrTrajValid (host evalTraj + map) agrees");
    bool fixedKept = true;
    for (int i = 0; i < 3; ++i) for (int a = 0; a < 3; ++a)
        fixedKept = fixedKept && c0(a, i) == c1(a, i) && c0(a, c0.cols() - 1 - i) == c1(a, c1.cols() - 1 - i);
    CHECK(fixedKept, "the first/last three control points are untouched (BT.cpp:690-691)");
    CHECK(std::fabs(bst.getDuration() - (c1.cols() - 3) * 0.2) < 1e-12, "duration = (N-3) * controlPointsTs");
    geometry_msgs::PoseStamped p0 = bst.getPose(0.0), pe = bst.getPose(bst.getDuration());
    CHECK(std::fabs(p0.pose.position.x + 3.0) < 0.2 && std::fabs(pe.pose.position.x - 3.0) < 0.2, "getPose spans start to goal (least-squares fit, not interpolation)");
    CHECK(std::fabs(p0.pose.orientation.w * p0.pose.orientation.w + p0.pose.orientation.z * p0.pose.orientation.z - 1.0) < 1e-12, "yaw-only unit quaternion");
    CHECK(bst.getLinearFactor() > 0 && std::isfinite(bst.getLinearFactor()), "linear feasibility re-parameterisation factor");
    std::printf("INFO solver status %d, linear factor %.4f\n", bst.getLastSolverStatus(), bst.getLinearFactor());

    // the lbfgs_evaluate_t seam on the device: finite-difference check of costFunction
    {
        const int N = c1.cols(), n = 3 * (N - 6);
        std::vector<double> x(c1.data() + 9, c1.data() + 9 + n), g(n), gp(n);
        const double f0 = bst.costFunction(x.data(), g.data(), n);
        double worst = 0;
        for (int k = 0; k < n; k += 7) {
            if (k % 3 == 2) continue;   // plan_in_z_axis = false zeroes the z gradient of the guide term by design
            std::vector<double> xp = x, xm = x;
            xp[k] += 1e-6; xm[k] -= 1e-6;
            const double fd = (bst.costFunction(xp.data(), gp.data(), n) - bst.costFunction(xm.data(), gp.data(), n)) / 2e-6;
            worst = std::fmax(worst, std::fabs(fd - g[k]) / std::fmax(1.0, std::fabs(g[k])));
        }
        CHECK(std::isfinite(f0) && worst < 1e-4, "costFunction gradient matches central differences");
        // the four terms on their own add up to the total (unit weights) and so do their gradients
        double cd, cs, cf, co;
        Eigen::MatrixXd gd, gs, gf, go;
        bst.getDistanceCost(c1, cd, gd);
        bst.getSmoothnessCost(c1, cs, gs);
        bst.getFeasibilityCost(c1, cf, gf);
        bst.getDynamicObstacleCost(c1, co, go);
        double gerr = 0;
        for (int k = 0; k < n; ++k) gerr = std::fmax(gerr, std::fabs(gd.data()[9 + k] + gs.data()[9 + k] + gf.data()[9 + k] + go.data()[9 + k] - g[k]));
        CHECK(std::fabs(cd + cs + cf + co - f0) <= 1e-12 * std::fmax(1.0, f0) && gerr <= 1e-12 * std::fmax(1.0, f0) && cs > 0, "getDistance/Smoothness/Feasibility/DynamicObstacleCost sum to costFunction");
        CHECK(bsplineTraj::solverCostFunction(&bst, x.data(), gp.data(), n) == f0, "solverCostFunction is the lbfgs_evaluate_t-shaped costFunction");
    }

    // a goal inside an obstacle is refused (BT.cpp:291-295)
    {
        bsplineTraj b2(makeParams());
        b2.setMap(map);
        nav_msgs::Path bad = straight(-3.0, 0.0, 0.0, 0.0, 1.0, 0.25);
        CHECK(!b2.updatePath(bad, cond), "updatePath refuses an occupied goal");
    }

    // ---- batch: 48 planners through ONE rebound loop ----
    {
        std::vector<std::unique_ptr<bsplineTraj>> owners;
        std::vector<bsplineTraj*> ps;
        for (int i = 0; i < 48; ++i) {
            owners.emplace_back(new bsplineTraj(makeParams()));
            bsplineTraj* b = owners.back().get();
            b->setMap(map);
            b->updateMaxVel(2.0);
            b->updateMaxAcc(3.0);
            ps.push_back(b);
        }
        // updatePathBatch: one device least-squares fit for all planners; planner 0 also runs the
        // single-path host fit for comparison
        std::vector<nav_msgs::Path> paths;
        for (int i = 0; i < 48; ++i) {
            const double y = -2.4 + 0.1 * i;   // some hit the pillar, some pass beside it
            paths.push_back(straight(-3.0, y, 3.0, y + 0.03, 1.0, 0.25));
        }
        std::vector<bool> up = bsplineTraj::updatePathBatch(ps, paths, std::vector<std::vector<Eigen::Vector3d>>(48, cond));
        int accepted = 0;
        for (bool u : up) accepted += u;
        CHECK(accepted == 48, "updatePathBatch accepts all 48 paths");
        {
            bsplineTraj single(makeParams());
            single.setMap(map);
            single.updatePath(paths[5], cond);
            Eigen::MatrixXd a = single.getControlPoints(), d = ps[5]->getControlPoints();
            double worst = 0;
            for (int c = 0; c < a.cols() && c < d.cols(); ++c) for (int r = 0; r < 3; ++r) worst = std::fmax(worst, std::fabs(a(r, c) - d(r, c)));
            CHECK(a.cols() == d.cols() && worst < 1e-10, "device batch fit == host single-path fit (1e-10)");
        }
        std::vector<bool> res = bsplineTraj::makePlanBatch(ps);
        int good = 0, clean = 0;
        for (size_t i = 0; i < ps.size(); ++i) {
            good += res[i];
            if (res[i] && ps[i]->isCurrTrajValid()) ++clean;
        }
        std::printf("INFO batch: %d of %zu planned, %d verified collision free\n", good, ps.size(), clean);
        CHECK(good >= 40 && clean == good, "makePlanBatch plans the batch and every success is collision free");
    }

    // ---- application-level rate: 1024 planners through updatePathBatch + makePlanBatch (host A*, guide
    //      assignment and bookkeeping included), reported, not gated — once with the rebound loop resident on the
    //      device between A* calls (vigo_rebound_rounds, the default) and once driven round by round from the host
    //      (the round-1 path): control points, success flags and solver status must be identical ----
    {
        const int NP = 1024;
        struct Run { std::vector<std::unique_ptr<bsplineTraj>> owners; std::vector<bsplineTraj*> ps; std::vector<bool> res; double msU = 0, msP = 0; };
        std::vector<nav_msgs::Path> paths;
        for (int i = 0; i < NP; ++i) {
            const double y = -2.6 + 5.2 * (i % 97) / 96.0, tilt = 0.4 * ((i * 37) % 11 - 5) / 5.0;
            paths.push_back(straight(-3.0, y, 3.0, y + tilt, 1.0, 0.25));
        }
        auto construct = [&](Run& R) {
            for (int i = 0; i < NP; ++i) {
                R.owners.emplace_back(new bsplineTraj(makeParams()));
                R.owners.back()->setMap(map);
                R.owners.back()->updateMaxVel(2.0);
                R.owners.back()->updateMaxAcc(3.0);
                R.ps.push_back(R.owners.back().get());
            }
        };
        auto plan = [&](bool deviceLoop, Run& R) {
            bsplineTraj::setDeviceResidentRebound(deviceLoop);
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<bool> up = bsplineTraj::updatePathBatch(R.ps, paths, std::vector<std::vector<Eigen::Vector3d>>(NP, cond));
            const auto t1 = std::chrono::steady_clock::now();
            R.res = bsplineTraj::makePlanBatch(R.ps);
            const auto t2 = std::chrono::steady_clock::now();
            bsplineTraj::setDeviceResidentRebound(true);
            R.msU = std::chrono::duration<double, std::milli>(t1 - t0).count();
            R.msP = std::chrono::duration<double, std::milli>(t2 - t1).count();
        };
        auto runBatch = [&](bool deviceLoop, Run& R) { construct(R); plan(deviceLoop, R); };
        Run warm, dev, host;
        runBatch(true, warm);                      // first use of the kernels and buffers of this size
        runBatch(true, dev);
        runBatch(false, host);
        int good = 0, clean = 0, same = 0;
        for (int i = 0; i < NP; ++i) {
            good += dev.res[i];
            if (dev.res[i] && dev.ps[i]->isCurrTrajValid()) ++clean;
            const Eigen::MatrixXd a = dev.ps[i]->getControlPoints(), b = host.ps[i]->getControlPoints();
            bool eq = dev.res[i] == host.res[i] && a.cols() == b.cols() && dev.ps[i]->getLastSolverStatus() == host.ps[i]->getLastSolverStatus();
            if (eq) eq = std::memcmp(a.data(), b.data(), sizeof(double) * 3 * a.cols()) == 0;
            same += eq;
            if (!eq && same + 5 > i) {
                double worst = 0;
                for (int c = 0; c < a.cols() && c < b.cols(); ++c) for (int r = 0; r < 3; ++r) worst = std::fmax(worst, std::fabs(a(r, c) - b(r, c)));
                std::printf("INFO mismatch planner %d: success %d/%d, solver status %d/%d, max |diff| %.3e\n", i, (int)dev.res[i], (int)host.res[i],
                            dev.ps[i]->getLastSolverStatus(), host.ps[i]->getLastSolverStatus(), worst);
            }
        }
        std::printf("INFO 1024 planners, rebound loop on the device: updatePathBatch %.2f ms, makePlanBatch %.2f ms (%.0f plans/s end to end), %d planned, %d verified collision free\n",
                    dev.msU, dev.msP, NP / ((dev.msU + dev.msP) * 1e-3), good, clean);
        std::printf("INFO 1024 planners, rebound loop driven by the host: updatePathBatch %.2f ms, makePlanBatch %.2f ms (%.0f plans/s end to end)\n",
                    host.msU, host.msP, NP / ((host.msU + host.msP) * 1e-3));
        CHECK(good >= NP * 8 / 10 && clean == good, "1024-planner batch: every success is collision free");
        CHECK(same == NP, "device-resident rebound loop == host-driven loop (control points, success, solver status, bit for bit)");
        warm.owners.clear(); warm.ps.clear();                            // (release the node pools of the runs no longer needed)
        host.owners.clear(); host.ps.clear();
        // several batches in flight: one host thread per batch, each planning its own 1024 planners (own handles, HIP
        // streams and staging buffers).  A planner service keeps its threads: each worker first plans a warm-up batch,
        // so its thread-local stream and buffers exist, then all start their timed batch together.
        for (int inflight : {2, 4}) {
            std::vector<Run> timed(inflight);
            for (int w = 0; w < inflight; ++w) construct(timed[w]);      // (each planner owns a 19 MB A* node pool: ~20 GB per batch)
            std::atomic<int> ready{0};
            std::vector<std::chrono::steady_clock::time_point> tStart(inflight), tEnd(inflight);
            auto worker = [&](int id) {
                plan(true, timed[id]);                                   // warm-up: the same planners plan the same paths twice
                ready.fetch_add(1);
                while (ready.load() < inflight) std::this_thread::yield();
                tStart[id] = std::chrono::steady_clock::now();
                plan(true, timed[id]);
                tEnd[id] = std::chrono::steady_clock::now();
            };
            std::vector<std::thread> pool;
            for (int w = 0; w < inflight; ++w) pool.emplace_back(worker, w);
            for (auto& t : pool) t.join();
            const double ms = std::chrono::duration<double, std::milli>(*std::max_element(tEnd.begin(), tEnd.end()) -
                                                                        *std::min_element(tStart.begin(), tStart.end())).count();
            int okAll = 0, eqAll = 0;
            for (int i = 0; i < NP; ++i) {
                bool ok = true, eq = true;
                const Eigen::MatrixXd b = dev.ps[i]->getControlPoints();
                for (int w = 0; w < inflight; ++w) {
                    ok = ok && timed[w].res[i];
                    const Eigen::MatrixXd a = timed[w].ps[i]->getControlPoints();
                    eq = eq && a.cols() == b.cols() && std::memcmp(a.data(), b.data(), sizeof(double) * 3 * a.cols()) == 0;
                }
                okAll += ok;
                eqAll += eq;
            }
            std::printf("INFO %d batches of 1024 in flight (one host thread each, updatePathBatch + makePlanBatch): %.2f ms, %.0f plans/s, %d planned in all\n",
                        inflight, ms, inflight * NP / (ms * 1e-3), okAll);
            CHECK(eqAll == NP, "concurrent batches give the single-batch control points");
        }
        // ONE call with 2048 / 4096 planners: makePlanBatch splits it into two / four pipelined parts itself (companion host
        // threads, own handles and streams) — the rate several caller threads got above, from a single call; the same plans
        // as the unsplit call
        for (int mult : {2, 4}) {
            const int NP2 = mult * NP;
            std::vector<nav_msgs::Path> paths2;
            for (int k = 0; k < mult; ++k) paths2.insert(paths2.end(), paths.begin(), paths.end());
            auto construct2 = [&](Run& R) {
                for (int i = 0; i < NP2; ++i) {
                    R.owners.emplace_back(new bsplineTraj(makeParams()));
                    R.owners.back()->setMap(map);
                    R.owners.back()->updateMaxVel(2.0);
                    R.owners.back()->updateMaxAcc(3.0);
                    R.ps.push_back(R.owners.back().get());
                }
            };
            auto plan2 = [&](size_t threshold, Run& R) {
                bsplineTraj::setBatchPipelineThreshold(threshold);
                const auto t0 = std::chrono::steady_clock::now();
                bsplineTraj::updatePathBatch(R.ps, paths2, std::vector<std::vector<Eigen::Vector3d>>(NP2, cond));
                const auto t1 = std::chrono::steady_clock::now();
                R.res = bsplineTraj::makePlanBatch(R.ps);
                const auto t2 = std::chrono::steady_clock::now();
                bsplineTraj::setBatchPipelineThreshold(2048);
                R.msU = std::chrono::duration<double, std::milli>(t1 - t0).count();
                R.msP = std::chrono::duration<double, std::milli>(t2 - t1).count();
            };
            // (one run alive at a time: a planner owns a 19 MB A* node pool, 4096 of them are 78 GB)
            struct Kept { std::vector<Eigen::MatrixXd> ctrl; std::vector<int> status; std::vector<bool> res; double msU = 0, msP = 0; };
            auto runKept = [&](size_t threshold) {
                Run R;
                construct2(R);
                plan2(threshold, R); plan2(threshold, R);      // (twice: the first call of this size grows staging buffers / creates the companions' streams)
                Kept k;
                for (int i = 0; i < NP2; ++i) { k.ctrl.push_back(R.ps[i]->getControlPoints()); k.status.push_back(R.ps[i]->getLastSolverStatus()); }
                k.res = R.res; k.msU = R.msU; k.msP = R.msP;
                return k;
            };
            const Kept whole = runKept(0), split = runKept(2048);
            int same2 = 0, good2 = 0;
            for (int i = 0; i < NP2; ++i) {
                const Eigen::MatrixXd &a = whole.ctrl[i], &b = split.ctrl[i];
                same2 += whole.res[i] == split.res[i] && a.cols() == b.cols() && std::memcmp(a.data(), b.data(), sizeof(double) * 3 * a.cols()) == 0 &&
                         whole.status[i] == split.status[i];
                good2 += split.res[i];
            }
            std::printf("INFO %d planners in ONE makePlanBatch call: unsplit %.2f ms (%.0f plans/s), %d pipelined parts %.2f ms (%.0f plans/s); updatePathBatch %.2f ms; %d planned, %d of %d identical\n",
                        NP2, whole.msP, NP2 / (whole.msP * 1e-3), mult, split.msP, NP2 / (split.msP * 1e-3), split.msU, good2, same2, NP2);
            CHECK(same2 == NP2 && good2 >= NP2 * 8 / 10, "makePlanBatch of 2048 / 4096 planners as pipelined parts == the unsplit call (control points, success, solver status)");
        }
    }

    // ---- a batch of planners with DIFFERENT yaml values and maps: each is solved with its own parameters (the batch is
    //      split into groups the device state fits), i.e. exactly as if it had been planned alone ----
    {
        auto otherMap = makeMap();
        auto mk = [&](int variant, const std::shared_ptr<mapManager::occMap>& m) {
            ros::NodeHandle nh = makeParams();
            if (variant == 1) nh.setParam("bspline_traj/distance_threshold", 0.7);
            if (variant == 2) { nh.setParam("bspline_traj/weight_smoothness", 2.5); nh.setParam("bspline_traj/timestep", 0.05); }
            std::unique_ptr<bsplineTraj> q(new bsplineTraj(nh));
            q->setMap(m);
            q->updateMaxVel(variant == 3 ? 1.5 : 2.0);
            q->updateMaxAcc(3.0);
            return q;
        };
        std::vector<std::unique_ptr<bsplineTraj>> mixed, alone;
        std::vector<bsplineTraj*> ps;
        std::vector<nav_msgs::Path> paths;
        for (int i = 0; i < 20; ++i) {
            const int variant = i % 5;                                   // 4 = the default parameters on a second map object
            mixed.push_back(mk(variant, variant == 4 ? otherMap : map));
            alone.push_back(mk(variant, variant == 4 ? otherMap : map));
            ps.push_back(mixed.back().get());
            paths.push_back(straight(-3.0, -0.5 + 0.06 * i, 3.0, 0.3 - 0.04 * i, 1.0, 0.25));
        }
        // (both sides take the host's single-path fit: the batched fit differs from it in the 10th digit, which 50 L-BFGS
        // iterations amplify to 1e-3 — with the same control points in, the plans must come out identical)
        std::vector<bool> up(ps.size());
        for (size_t i = 0; i < ps.size(); ++i) up[i] = ps[i]->updatePath(paths[i], cond);
        std::vector<bool> res = bsplineTraj::makePlanBatch(ps);
        int same = 0;
        for (size_t i = 0; i < ps.size(); ++i) {
            const bool u = alone[i]->updatePath(paths[i], cond);
            const bool r = u && alone[i]->makePlan();
            const Eigen::MatrixXd a = ps[i]->getControlPoints(), b = alone[i]->getControlPoints();
            double worst = 0;
            for (int c = 0; c < a.cols() && c < b.cols(); ++c) for (int k = 0; k < 3; ++k) worst = std::fmax(worst, std::fabs(a(k, c) - b(k, c)));
            const bool ok = (u == (bool)up[i]) && (r == (bool)res[i]) && a.cols() == b.cols() && worst == 0.0;
            if (!ok) std::printf("INFO mixed batch planner %zu (variant %zu): update %d/%d plan %d/%d cols %d/%d worst %.3e\n", i, i % 5, (int)up[i], (int)u,
                                 (int)res[i], (int)r, (int)a.cols(), (int)b.cols(), worst);
            same += ok;
        }
        CHECK(same == (int)ps.size(), "a batch of planners with different parameters and maps == each planned alone");
    }

    // ---- updatePathBatch runs the planners' prologues on the host workers; the previous path length that
    //      adjustPathLengthDirect hands from one call to the next (the reference's function-static, BT.cpp:755) must
    //      still go down the line in order: paths longer than max_path_length (7 m) make it matter ----
    {
        auto mk = [&]() {
            ros::NodeHandle nh = makeParams();
            nh.setParam("bspline_traj/max_path_length", 7.0);      // the reference's default, BT.cpp:152
            std::unique_ptr<bsplineTraj> q(new bsplineTraj(nh));
            q->setMap(map);
            q->updateMaxVel(2.0);
            q->updateMaxAcc(3.0);
            return q;
        };
        std::vector<nav_msgs::Path> paths;
        for (int i = 0; i < 40; ++i) {
            const double y = 2.0 + 0.08 * i;                       // clear of the pillar
            const double x1 = (i % 5 == 2) ? 5.5 : ((i % 7 == 3) ? 4.9 : 2.0 + 0.05 * i);   // 11.5 m, 10.9 m, or 5 .. 7 m long
            paths.push_back(straight(-6.0, y, x1, y + 0.05, 1.0, 0.25));
        }
        const nav_msgs::Path reset = straight(-3.0, 3.0, 0.0, 3.0, 1.0, 0.25);   // leaves the shared value at 3 m
        std::vector<std::unique_ptr<bsplineTraj>> seqO, batO;
        std::vector<bsplineTraj*> bat;
        for (int i = 0; i < 40; ++i) { seqO.push_back(mk()); batO.push_back(mk()); bat.push_back(batO.back().get()); }
        auto dummy = mk();
        dummy->updatePath(reset, cond);
        std::vector<bool> us(40);
        for (int i = 0; i < 40; ++i) us[i] = seqO[i]->updatePath(paths[i], cond);
        dummy->updatePath(reset, cond);
        std::vector<bool> ub = bsplineTraj::updatePathBatch(bat, paths, std::vector<std::vector<Eigen::Vector3d>>(40, cond));
        int same = 0, shortened = 0;
        for (int i = 0; i < 40; ++i) {
            const Eigen::MatrixXd a = seqO[i]->getControlPoints(), b = batO[i]->getControlPoints();
            double worst = 0;
            for (int c = 0; c < a.cols() && c < b.cols(); ++c) for (int k = 0; k < 3; ++k) worst = std::fmax(worst, std::fabs(a(k, c) - b(k, c)));
            same += (bool)us[i] == (bool)ub[i] && a.cols() == b.cols() && worst < 1e-9;
            shortened += a.cols() < (int)paths[i].poses.size() + 2;
        }
        // and the value left behind is the same: one more planner after each run
        auto tailS = mk(), tailB = mk();
        dummy->updatePath(reset, cond);
        for (int i = 0; i < 40; ++i) seqO[i]->updatePath(paths[i], cond);
        tailS->updatePath(paths[2], cond);
        dummy->updatePath(reset, cond);
        bsplineTraj::updatePathBatch(bat, paths, std::vector<std::vector<Eigen::Vector3d>>(40, cond));
        tailB->updatePath(paths[2], cond);
        std::printf("INFO updatePathBatch vs one planner after another: %d of 40 equal, %d paths shortened by the length rule\n", same, shortened);
        CHECK(same == 40 && shortened > 0 && tailS->getControlPoints().cols() == tailB->getControlPoints().cols(),
              "updatePathBatch (prologues on the host workers) == updatePath one planner after another, previous path length handed down in order");
    }

    // ---- polyTrajOctomap checker ----
    {
        ros::NodeHandle nh;
        nh.setParam("collision_box", std::vector<double>{0.4, 0.4, 0.2});
        nh.setParam("map_resolution", 0.2);
        nh.setParam("sample_delta_time", 0.1);
        trajPlanner::polyTrajOctomap poly(nh);
        std::cout << poly << std::endl;   // src/poly_RRT_node.cpp:68
        poly.setMap(map);
        std::vector<trajPlanner::pose> wp{{-3, 0, 1}, {0, 2.0, 1}, {3, 0, 1}};
        poly.updatePath(wp);
        // two degree-7 segments, linear in local time: through the pillar's side region then away
        std::vector<double> xs(16, 0), ys(16, 0), zs(16, 0);
        xs[0] = -3; xs[1] = 1.0; ys[0] = 0; ys[1] = 0.0; zs[0] = 1;          // segment 0: straight at y = 0 (hits the pillar)
        xs[8] = 0; xs[9] = 1.0; ys[8] = 2.0; ys[9] = 0.0; zs[8] = 1;         // segment 1: y = 2 (free)
        poly.setSolution(7, xs, ys, zs, {0.0, 3.0, 6.0});
        std::vector<trajPlanner::pose> traj;
        poly.makePlan(traj, 0.1);
        std::set<int> seg;
        const bool hit = poly.checkCollisionTraj(traj, 0.1, seg);
        CHECK(hit && seg.count(0) == 1 && seg.count(1) == 0 && !poly.isValid(), "checkCollisionTraj blames segment 0 only");
        CHECK(poly.checkCollision(trajPlanner::pose(0, 0, 1)) && !poly.checkCollision(trajPlanner::pose(-3, 0, 1)), "checkCollision box sweep");
        CHECK(poly.checkCollisionPoint(trajPlanner::pose(0, 0, 1)) && !poly.checkCollisionPoint(trajPlanner::pose(0, 2, 1)), "checkCollisionPoint");
        CHECK(std::fabs(poly.getDuration() - 6.0) < 1e-12 && std::fabs(poly.getPose(1.5).pose.position.x + 1.5) < 1e-12, "getDuration / getPose");
    }

    // ---- polyTrajOctomap::makePlanBatch: 24 waypoint paths beside / through the pillar, one QP launch + one sweep launch per round ----
    {
        std::vector<std::unique_ptr<trajPlanner::polyTrajOctomap>> owners;
        std::vector<trajPlanner::polyTrajOctomap*> ps;
        for (int i = 0; i < 24; ++i) {
            ros::NodeHandle nh;
            nh.setParam("collision_box", std::vector<double>{0.4, 0.4, 0.2});
            nh.setParam("map_resolution", 0.2);
            nh.setParam("sample_delta_time", 0.1);
            nh.setParam("mode", 0.0);
            nh.setParam("initial_radius", 0.5);
            nh.setParam("shrinking_factor", 0.8);
            nh.setParam("corridor_res", 8.0);
            nh.setParam("maximum_iteration_num", 30.0);
            nh.setParam("traj_timeout", 0.5);
            owners.emplace_back(new trajPlanner::polyTrajOctomap(nh));
            owners.back()->setMap(map);
            const double y = 1.6 + 0.12 * i;   // legs pass the pillar (|y| <= 0.8 + margin) at increasing distance
            owners.back()->updatePath(std::vector<trajPlanner::pose>{{-3, y, 1}, {-1, y + 0.3, 1}, {1, y + 0.3, 1}, {3, y, 1}});
            ps.push_back(owners.back().get());
        }
        std::vector<std::vector<trajPlanner::pose>> trajs;
        std::vector<bool> res = trajPlanner::polyTrajOctomap::makePlanBatch(ps, trajs);
        int valid = 0, agree = 0;
        for (size_t i = 0; i < ps.size(); ++i) {
            valid += res[i];
            std::vector<int> idx;
            const bool hit = ps[i]->checkCollisionTraj(trajs[i], idx);
            agree += (res[i] == !hit) || !res[i];
        }
        // the same planner alone gives the same answer and the same trajectory
        trajPlanner::polyTrajOctomap* solo = ps[3];
        std::vector<trajPlanner::pose> t2;
        solo->makePlan(t2, 0.1);
        double worst = 0;
        for (size_t k = 0; k < t2.size() && k < trajs[3].size(); ++k) worst = std::fmax(worst, std::fabs(t2[k].x - trajs[3][k].x) + std::fabs(t2[k].y - trajs[3][k].y));
        std::printf("INFO poly batch: %d of %zu valid; solo-vs-batch max diff %.3e\n", valid, ps.size(), worst);
        CHECK(valid >= 20 && agree == (int)ps.size(), "polyTrajOctomap::makePlanBatch: valid plans are collision free");
        CHECK(t2.size() == trajs[3].size() && worst < 1e-6 && solo->isValid() == res[3], "batch plan == single makePlan (device QP == host QP)");
    }

    // ---- polyTrajOctomap::makePlanBatch in the ADDING-WAYPOINT mode (PO.cpp:259-386), mixed with corridor-mode planners:
    //      the paths grow as waypoints are inserted, so each round groups the planners by waypoint count; every
    //      planner is compared with a twin planned on its own (host QP) ----
    {
        auto mk = [&](int i, bool addingWaypoints) {
            ros::NodeHandle nh;
            nh.setParam("collision_box", std::vector<double>{0.4, 0.4, 0.2});
            nh.setParam("map_resolution", 0.2);
            nh.setParam("sample_delta_time", 0.1);
            nh.setParam("mode", addingWaypoints ? 1.0 : 0.0);
            nh.setParam("initial_radius", 0.5);
            nh.setParam("shrinking_factor", 0.8);
            nh.setParam("corridor_res", 8.0);
            nh.setParam("maximum_iteration_num", 12.0);
            nh.setParam("traj_timeout", 0.5);
            std::unique_ptr<trajPlanner::polyTrajOctomap> q(new trajPlanner::polyTrajOctomap(nh));
            q->setMap(map);
            const double c = 0.75 + 0.05 * i;   // the leg over the pillar (|y| <= 0.8): from grazing it to well clear
            if (i % 3 == 0) q->updatePath(std::vector<trajPlanner::pose>{{-3, 0.02 * i, 1}, {-1, 0.1, 1}, {1, 0.1, 1}, {3, 0, 1}});   // through it: waypoints are inserted until the limit
            else if (i % 5 == 4) q->updatePath(std::vector<trajPlanner::pose>{{-3, 0, 1}, {0, c + 0.2, 1}, {3, 0, 1}});
            else q->updatePath(std::vector<trajPlanner::pose>{{-3, 0, 1}, {-1.1, c, 1}, {1.1, c, 1}, {3, 0, 1}});
            return q;
        };
        std::vector<std::unique_ptr<trajPlanner::polyTrajOctomap>> owners, twins;
        std::vector<trajPlanner::polyTrajOctomap*> ps;
        for (int i = 0; i < 32; ++i) {
            const bool adding = i % 4 != 3;               // 24 adding-waypoint planners, 8 corridor planners
            owners.push_back(mk(i, adding));
            twins.push_back(mk(i, adding));
            ps.push_back(owners.back().get());
        }
        std::vector<std::vector<trajPlanner::pose>> trajs;
        std::vector<bool> res = trajPlanner::polyTrajOctomap::makePlanBatch(ps, trajs);
        int same = 0, validAdding = 0, grown = 0;
        double worst = 0;
        for (size_t i = 0; i < ps.size(); ++i) {
            std::vector<trajPlanner::pose> t2;
            twins[i]->makePlan(t2, 0.1);
            bool eq = twins[i]->isValid() == res[i] && t2.size() == trajs[i].size() && twins[i]->getPath().size() == ps[i]->getPath().size();
            double w = 0;
            for (size_t k = 0; eq && k < t2.size(); ++k)
                w = std::fmax(w, std::fabs(t2[k].x - trajs[i][k].x) + std::fabs(t2[k].y - trajs[i][k].y) + std::fabs(t2[k].z - trajs[i][k].z));
            eq = eq && w < 1e-9;
            worst = std::fmax(worst, w);
            same += eq;
            if (i % 4 != 3) { validAdding += res[i]; grown += ps[i]->getPath().size() > 4; }
        }
        std::printf("INFO poly batch, adding-waypoint mode: %d of 24 valid, %d paths grew; batch vs solo: %d of 32 identical plans, max diff %.3e\n",
                    validAdding, grown, same, worst);
        CHECK(same == 32 && grown > 0 && validAdding > 0, "polyTrajOctomap::makePlanBatch, adding-waypoint mode: batch plan == single makePlan");
    }

    // ---- a corridor-mode path of MORE waypoints than the device QP takes (13 > 11) inside a batch: solved by the host QP
    //      inside the same rounds, WITH the round's corridor boxes (PO.cpp:421-424) — the same plan as the planner alone ----
    {
        auto mk = [&](double y0, int W) {
            ros::NodeHandle nh;
            nh.setParam("collision_box", std::vector<double>{0.4, 0.4, 0.2});
            nh.setParam("map_resolution", 0.2);
            nh.setParam("sample_delta_time", 0.1);
            nh.setParam("mode", 0.0);
            nh.setParam("initial_radius", 0.5);
            nh.setParam("shrinking_factor", 0.8);
            nh.setParam("corridor_res", 8.0);
            nh.setParam("maximum_iteration_num", 30.0);
            nh.setParam("traj_timeout", 5.0);
            std::unique_ptr<trajPlanner::polyTrajOctomap> q(new trajPlanner::polyTrajOctomap(nh));
            q->setMap(map);
            std::vector<trajPlanner::pose> wp;
            for (int i = 0; i < W; ++i) {
                // a straight line at y0 with ONE waypoint pulled 1.3 m aside just before the pillar: the free min-snap
                // polynomial rings and undershoots y0 by ~7.5 cm over the pillar (into its voxels for y0 <= 1.07), and
                // only a corridor shrunk below ~0.17 m holds it back (host QP, tools: vigo_host_minsnap)
                const double x = -3.0 + 6.0 * i / (W - 1);
                wp.push_back(trajPlanner::pose(x, y0 + ((W == 13 && i == 5) ? 1.3 : 0.0), 1));
            }
            q->updatePath(wp);
            return q;
        };
        std::vector<std::unique_ptr<trajPlanner::polyTrajOctomap>> owners, twins;
        std::vector<trajPlanner::polyTrajOctomap*> ps;
        const double ys[4] = {1.04, 1.06, 1.07, 2.4};
        for (int i = 0; i < 4; ++i) { owners.push_back(mk(ys[i], 13)); twins.push_back(mk(ys[i], 13)); ps.push_back(owners.back().get()); }
        for (int i = 0; i < 4; ++i) { owners.push_back(mk(ys[i], 5)); twins.push_back(mk(ys[i], 5)); ps.push_back(owners.back().get()); }
        std::vector<std::vector<trajPlanner::pose>> trajs;
        std::vector<bool> res = trajPlanner::polyTrajOctomap::makePlanBatch(ps, trajs);
        int same = 0, shrunk = 0, validLong = 0;
        double worst = 0;
        for (size_t i = 0; i < ps.size(); ++i) {
            std::vector<trajPlanner::pose> t2;
            twins[i]->makePlan(t2, 0.1);
            bool eq = twins[i]->isValid() == res[i] && t2.size() == trajs[i].size() && twins[i]->getIterations() == ps[i]->getIterations();
            double w = 0;
            for (size_t k = 0; eq && k < t2.size(); ++k)
                w = std::fmax(w, std::fabs(t2[k].x - trajs[i][k].x) + std::fabs(t2[k].y - trajs[i][k].y) + std::fabs(t2[k].z - trajs[i][k].z));
            eq = eq && w < 1e-6;
            worst = std::fmax(worst, w);
            same += eq;
            if (i < 4) { shrunk += ps[i]->getIterations() > 1 && res[i]; validLong += res[i]; }
        }
        std::printf("INFO poly batch with 13-waypoint corridor planners: %d of 8 identical to the planner alone (max diff %.3e); of the 4 long paths %d valid, %d of them only after the corridor had shrunk\n",
                    same, worst, validLong, shrunk);
        CHECK(same == 8 && shrunk > 0, "a 13-waypoint corridor-mode planner in a batch is planned with its corridor boxes: batch plan == single makePlan");
    }

    // ---- the device ordinal of a planner (setDevice): ordinal 0 again after a plan re-creates the handle and re-uploads the
    //      map with the same result; an ordinal the machine does not have is refused without a crash ----
    {
        bsplineTraj a(makeParams()), b(makeParams());
        a.setMap(map);
        b.setMap(map);
        nav_msgs::Path path = straight(-3.0, 0.1, 3.0, 0.1, 1.0, 0.25);
        std::vector<Eigen::Vector3d> cond(4, Eigen::Vector3d(0, 0, 0));
        b.setDevice(0);
        const bool okA = a.updatePath(path, cond) && a.makePlan();
        bool okB = b.updatePath(path, cond) && b.makePlan();
        b.setDevice(1 << 20);                              // no such card
        const bool refused = !(b.updatePath(path, cond) && b.makePlan());
        b.setDevice(0);                                    // back: a new handle, the map uploaded again
        okB = okB && b.updatePath(path, cond) && b.makePlan();
        const Eigen::MatrixXd ca = a.getControlPoints(), cb = b.getControlPoints();
        bool eq = ca.cols() == cb.cols();
        for (int i = 0; eq && i < ca.cols(); ++i) for (int k = 0; k < 3; ++k) eq = eq && ca(k, i) == cb(k, i);
        CHECK(okA && okB && refused && eq, "setDevice: ordinal 0 plans as the default does, a missing card is refused, moving back re-creates the handle");
        ros::NodeHandle nh;
        nh.setParam("collision_box", std::vector<double>{0.4, 0.4, 0.2});
        nh.setParam("map_resolution", 0.2);
        trajPlanner::polyTrajOctomap poly(nh);
        poly.setMap(map);
        poly.setDevice(0);
        const bool hit0 = poly.checkCollision(trajPlanner::pose(0, 0, 1));
        poly.setDevice(1 << 20);
        const bool hitNo = poly.checkCollision(trajPlanner::pose(-3, 0, 1));   // no device: "colliding" (flags default to 1), no crash
        poly.setDevice(0);
        CHECK(hit0 && hitNo && !poly.checkCollision(trajPlanner::pose(-3, 0, 1)), "polyTrajOctomap::setDevice: same, and back");
    }

    // ---- degenerate inputs: the classes answer false / "not found", never crash ----
    {
        const double nan = std::nan("");
        bsplineTraj fresh(makeParams());
        CHECK(!fresh.makePlan(), "makePlan without a map and a path is false");
        fresh.setMap(map);
        CHECK(!fresh.makePlan(), "makePlan without a path is false");
        nav_msgs::Path none, three = straight(-3.0, 2.5, -2.5, 2.5, 1.0, 0.25), same = straight(-3.0, 2.5, -3.0, 2.5, 1.0, 0.25);
        three.poses.resize(3);
        CHECK(!fresh.updatePath(none, cond), "updatePath refuses an empty path");
        {   // 3 poses: the prologue may interpolate them to 4+ fit points (BT.cpp:279-331); either way no exit(0)
            const bool up3 = fresh.updatePath(three, cond);
            CHECK(!up3 || fresh.getControlPoints().cols() >= 6, "updatePath of 3 poses: refused, or a spline of >= 6 control points");
        }
        nav_msgs::Path poisoned = straight(-3.0, 2.5, 3.0, 2.5, 1.0, 0.25);
        poisoned.poses[4].pose.position.y = nan;
        const bool upNan = fresh.updatePath(poisoned, cond);
        const bool planNan = upNan && fresh.makePlan();
        CHECK(!planNan, "a NaN waypoint never yields a plan");
        (void)fresh.updatePath(same, cond);
        (void)fresh.makePlan();
        (void)fresh.getPose(0.3);
        std::vector<bsplineTraj*> withNull{&fresh, nullptr};
        // soft waypoint constraints (yaml soft_constraint / constraint_radius, PO.cpp:98-107): the interior waypoint is a box
        {
            ros::NodeHandle nhs;
            nhs.setParam("collision_box", std::vector<double>{0.4, 0.4, 0.2});
            nhs.setParam("map_resolution", 0.2);
            nhs.setParam("sample_delta_time", 0.1);
            nhs.setParam("traj_timeout", 0.5);
            nhs.setParam("mode", false);
            trajPlanner::polyTrajOctomap hardP(nhs);
            nhs.setParam("soft_constraint", true);
            nhs.setParam("constraint_radius", 0.3);
            trajPlanner::polyTrajOctomap softP(nhs);
            const std::vector<trajPlanner::pose> wps{{-3, 2.2, 1}, {0, 3.4, 1}, {3, 2.2, 1}};     // clear of the pillar
            std::vector<trajPlanner::pose> th, ts;
            hardP.setMap(map); hardP.updatePath(wps); hardP.makePlan(th, 0.1);
            softP.setMap(map); softP.updatePath(wps); softP.makePlan(ts, 0.1);
            auto nearest = [&](const std::vector<trajPlanner::pose>& t) {
                double best = 1e9;
                for (const auto& q : t) best = std::fmin(best, std::hypot(q.x - wps[1].x, q.y - wps[1].y));
                return best;
            };
            std::vector<trajPlanner::polyTrajOctomap*> both{&hardP, &softP};
            std::vector<std::vector<trajPlanner::pose>> tb;
            std::vector<bool> rb2 = trajPlanner::polyTrajOctomap::makePlanBatch(both, tb);
            bool sameSoft = tb.size() == 2 && tb[1].size() == ts.size();
            for (size_t i = 0; sameSoft && i < ts.size(); ++i) sameSoft = std::fabs(tb[1][i].x - ts[i].x) < 1e-9 && std::fabs(tb[1][i].y - ts[i].y) < 1e-9;
            std::printf("INFO soft waypoint constraint: hard plan passes the waypoint at %.4f m, soft plan (radius 0.3) at %.4f m\n", nearest(th), nearest(ts));
            CHECK(hardP.isValid() && softP.isValid() && nearest(th) < 0.06 && nearest(ts) > 0.1 && nearest(ts) < 0.3 * std::sqrt(2.0) + 0.06,
                  "soft_constraint: the interior waypoint is cut within its box, not interpolated");
            CHECK(rb2.size() == 2 && rb2[0] && rb2[1] && sameSoft, "makePlanBatch plans a soft-constraint planner on its own, same plan");
        }

        // batch entry points with unprepared planners
        bsplineTraj idle(makeParams());
        std::vector<bsplineTraj*> two{&fresh, &idle};
        std::vector<bool> r2 = bsplineTraj::makePlanBatch(two);
        CHECK(r2.size() == 2 && !r2[1], "makePlanBatch reports an unprepared planner as failed");
        CHECK(bsplineTraj::makePlanBatch(std::vector<bsplineTraj*>{}).empty(), "makePlanBatch of nothing");

        ros::NodeHandle nh;
        nh.setParam("collision_box", std::vector<double>{0.4, 0.4, 0.2});
        nh.setParam("map_resolution", 0.2);
        nh.setParam("sample_delta_time", 0.1);
        nh.setParam("traj_timeout", 0.2);
        trajPlanner::polyTrajOctomap poly(nh);
        poly.setMap(map);
        std::vector<trajPlanner::pose> traj;
        poly.makePlan(traj, 0.1);                                    // no path at all
        CHECK(traj.empty() && !poly.isValid(), "polyTrajOctomap without a path plans nothing");
        poly.updatePath(std::vector<trajPlanner::pose>{{1, 2, 1}});
        poly.makePlan(traj, 0.1);
        CHECK(traj.size() == 1, "a single waypoint is its own trajectory (PO.cpp:229-233)");
        poly.updatePath(std::vector<trajPlanner::pose>{{1, 2, 1}, {1, 2, 1}, {3, 2, 1}});      // coincident waypoints
        poly.makePlan(traj, 0.1);
        bool fin = true;
        for (const auto& q : traj) fin = fin && std::isfinite(q.x) && std::isfinite(q.y) && std::isfinite(q.z);
        CHECK(fin, "coincident waypoints give a finite trajectory (min-snap or the piecewise-linear fallback)");
        poly.updatePath(std::vector<trajPlanner::pose>{{1, 2, 1}, {nan, 2, 1}, {3, 2, 1}});
        poly.makePlan(traj, 0.1);
        CHECK(!poly.isValid() || traj.empty(), "a NaN waypoint never yields a valid min-snap plan");
        std::vector<trajPlanner::polyTrajOctomap*> pnone;
        std::vector<std::vector<trajPlanner::pose>> tn;
        CHECK(trajPlanner::polyTrajOctomap::makePlanBatch(pnone, tn).empty(), "polyTrajOctomap::makePlanBatch of nothing");
    }

    // ---- randomised stress (VIGO_FACADE_FUZZ=<rounds>, default 3): random box worlds, random straight paths with
    //      free end points, dynamic obstacles on a third of the planners.  No success rate is demanded (a random
    //      world may wall a path in); what is demanded: no crash, and every plan reported successful is collision
    //      free by the host-side gate (isCurrTrajValid: host spline + host copy of the map) ----
    {
        const char* fz = std::getenv("VIGO_FACADE_FUZZ");
        const int rounds = fz ? std::atoi(fz) : 3;
        unsigned long long st = 0x9E3779B97F4A7C15ull;
        auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0; };
        int planned = 0, total = 0, dirty = 0, clipped = 0, polyValid = 0, polyTotal = 0, polyDirty = 0, loopMismatch = 0, obstaclePlans = 0;
        for (int round = 0; round < rounds; ++round) {
            auto m = std::make_shared<mapManager::occMap>(128, 128, 40, Eigen::Vector3d(-6.4, -6.4, -0.5), 0.1);
            const int nb = 3 + (int)(rnd() * 10);
            for (int b = 0; b < nb; ++b) {
                const double cx = -4.5 + 9.0 * rnd(), cy = -4.5 + 9.0 * rnd(), hx = 0.15 + 0.6 * rnd(), hy = 0.15 + 0.6 * rnd(), top = 0.5 + 3.0 * rnd();
                for (int ix = 0; ix < 128; ++ix) for (int iy = 0; iy < 128; ++iy) for (int iz = 0; iz < 40; ++iz) {
                    const double x = -6.4 + (ix + 0.5) * 0.1, y = -6.4 + (iy + 0.5) * 0.1, z = -0.5 + (iz + 0.5) * 0.1;
                    if (std::fabs(x - cx) <= hx && std::fabs(y - cy) <= hy && z <= top) m->at(ix, iy, iz) |= 4;
                    if (std::fabs(x - cx) <= hx + 0.4 && std::fabs(y - cy) <= hy + 0.4 && z <= top + 0.15) m->at(ix, iy, iz) |= 1;
                }
            }
            auto freePoint = [&](double& x, double& y) {
                for (int tries = 0; tries < 200; ++tries) {
                    x = -5.0 + 10.0 * rnd(); y = -5.0 + 10.0 * rnd();
                    if (!m->isInflatedOccupied(Eigen::Vector3d(x, y, 1.0))) return true;
                }
                return false;
            };
            const int NPR = 40 + (int)(rnd() * 60);
            std::vector<std::unique_ptr<bsplineTraj>> owners, twins;
            std::vector<bsplineTraj*> ps, ps2;
            std::vector<nav_msgs::Path> paths;
            std::vector<double> vels, readyVel;
            std::vector<std::vector<std::vector<Eigen::Vector3d>>> obsOf;   // per planner: {pos, vel, size} or empty
            for (int i = 0; i < NPR; ++i) {
                double x0, y0, x1, y1;
                if (!freePoint(x0, y0) || !freePoint(x1, y1) || std::hypot(x1 - x0, y1 - y0) < 2.0) continue;
                // every planner has a twin with the same inputs: the twins run the rebound loop from the host
                owners.emplace_back(new bsplineTraj(makeParams()));
                twins.emplace_back(new bsplineTraj(makeParams()));
                vels.push_back(1.0 + 2.0 * rnd());
                const double acc = 2.0 + 2.0 * rnd();
                for (bsplineTraj* q : {owners.back().get(), twins.back().get()}) {
                    q->setMap(m);
                    q->updateMaxVel(vels.back());
                    q->updateMaxAcc(acc);
                }
                if (i % 3 == 0) {
                    const double f = rnd();
                    std::vector<Eigen::Vector3d> op{Eigen::Vector3d(x0 + (x1 - x0) * f + 0.8 * (rnd() - 0.5), y0 + (y1 - y0) * f + 0.8 * (rnd() - 0.5), 1.0)};
                    std::vector<Eigen::Vector3d> ov{Eigen::Vector3d(rnd() - 0.5, rnd() - 0.5, 0.0)};
                    std::vector<Eigen::Vector3d> os{Eigen::Vector3d(0.4 + 0.4 * rnd(), 0.4 + 0.4 * rnd(), 1.5)};
                    obsOf.push_back({op, ov, os});
                } else {
                    obsOf.push_back({});
                }
                ps.push_back(owners.back().get());
                ps2.push_back(twins.back().get());
                paths.push_back(straight(x0, y0, x1, y1, 1.0, 0.25));
            }
            // (updatePath clear()s the obstacles, BT.cpp:393-401: they are set after it, like src/bspline_node.cpp does)
            std::vector<bool> up = bsplineTraj::updatePathBatch(ps, paths, std::vector<std::vector<Eigen::Vector3d>>(ps.size(), cond));
            std::vector<bool> up2 = bsplineTraj::updatePathBatch(ps2, paths, std::vector<std::vector<Eigen::Vector3d>>(ps2.size(), cond));
            std::vector<bsplineTraj*> ready, ready2;
            for (size_t i = 0; i < ps.size(); ++i) {
                if (!obsOf[i].empty()) {
                    ps[i]->updateDynamicObstacles(obsOf[i][0], obsOf[i][1], obsOf[i][2]);
                    ps2[i]->updateDynamicObstacles(obsOf[i][0], obsOf[i][1], obsOf[i][2]);
                }
                if (up[i] && up2[i]) { ready.push_back(ps[i]); ready2.push_back(ps2[i]); readyVel.push_back(vels[i]); }
            }
            std::vector<bool> res = bsplineTraj::makePlanBatch(ready);
            bsplineTraj::setDeviceResidentRebound(false);
            std::vector<bool> resHost = bsplineTraj::makePlanBatch(ready2);
            bsplineTraj::setDeviceResidentRebound(true);
            for (size_t i = 0; i < ready.size(); ++i) {
                const Eigen::MatrixXd a = ready[i]->getControlPoints(), b = ready2[i]->getControlPoints();
                const bool eq = res[i] == resHost[i] && a.cols() == b.cols() && std::memcmp(a.data(), b.data(), sizeof(double) * 3 * a.cols()) == 0;
                loopMismatch += eq ? 0 : 1;
                obstaclePlans += ready[i]->hasDynamicObstacles() ? 1 : 0;
            }
            for (size_t i = 0; i < ready.size(); ++i) {
                ++total;
                if (!res[i]) continue;
                ++planned;
                bool hit = false, finite = true;
                const double dur = ready[i]->getDuration();
                for (double t = 0.0; t <= dur; t += 0.01) {
                    const geometry_msgs::PoseStamped ps1 = ready[i]->getPose(t);
                    const Eigen::Vector3d q(ps1.pose.position.x, ps1.pose.position.y, ps1.pose.position.z);
                    finite = finite && std::isfinite(q(0)) && std::isfinite(q(1)) && std::isfinite(q(2));
                    hit = hit || m->isInflatedOccupied(q);
                }
                // the reference's own criterion (host evalTraj at the gate's sample step + the host map) must hold;
                // the 10 ms sampling is finer than that gate (res / maxVel / 2 in spline time) and may see a clipped
                // voxel corner between two gate samples — the reference's gate has the same blind spot: reported only
                bool gateHit = false;                      // BT.h:307-325 on the host: host spline, host map, same sample step
                {
                    trajPlanner::bspline sp(3, ready[i]->getControlPoints(), ready[i]->getControlPointTs());
                    const double dtg = m->getRes() / readyVel[i] / 2.0;
                    for (double t = 0.0; t <= sp.getDuration(); t += dtg) gateHit = gateHit || m->isInflatedOccupied(sp.at(t));
                }
                if (gateHit || !ready[i]->isCurrTrajValid() || !finite || !(dur > 0)) ++dirty;
                if (hit) ++clipped;
            }
            // min-snap planners through the same world: 3-6 free waypoints each, corridor mode
            std::vector<std::unique_ptr<trajPlanner::polyTrajOctomap>> pown;
            std::vector<trajPlanner::polyTrajOctomap*> pp;
            std::vector<std::vector<trajPlanner::pose>> pwps;
            for (int i = 0; i < 24; ++i) {
                ros::NodeHandle nh;
                nh.setParam("collision_box", std::vector<double>{0.4, 0.4, 0.2});
                nh.setParam("map_resolution", 0.2);
                nh.setParam("sample_delta_time", 0.1);
                nh.setParam("mode", (double)(i % 2));
                nh.setParam("initial_radius", 0.5);
                nh.setParam("shrinking_factor", 0.8);
                nh.setParam("corridor_res", 8.0);
                nh.setParam("maximum_iteration_num", 20.0);
                nh.setParam("traj_timeout", 0.5);
                std::vector<trajPlanner::pose> wp;
                const int nw = 3 + (int)(rnd() * 4);
                double x, y;
                bool okp = true;
                for (int k = 0; k < nw && okp; ++k) { okp = freePoint(x, y); wp.push_back(trajPlanner::pose(x, y, 1.0)); }
                if (!okp) continue;
                pown.emplace_back(new trajPlanner::polyTrajOctomap(nh));
                pown.back()->setMap(m);
                pown.back()->updatePath(wp);
                pp.push_back(pown.back().get());
                pwps.push_back(wp);
            }
            std::vector<std::vector<trajPlanner::pose>> trajs;
            std::vector<bool> pres = trajPlanner::polyTrajOctomap::makePlanBatch(pp, trajs);
            for (size_t i = 0; i < pp.size(); ++i) {
                ++polyTotal;
                if (!pres[i]) continue;
                ++polyValid;
                // a valid plan starts at the first waypoint and ends at the last (never a default-constructed pose)
                bool hit = trajs[i].size() < 2 || std::fabs(trajs[i].front().x - pwps[i].front().x) + std::fabs(trajs[i].front().y - pwps[i].front().y) > 1e-6 ||
                           std::fabs(trajs[i].back().x - pwps[i].back().x) + std::fabs(trajs[i].back().y - pwps[i].back().y) > 1e-6;
                for (const auto& q : trajs[i]) {
                    if (!(std::isfinite(q.x) && std::isfinite(q.y) && std::isfinite(q.z))) hit = true;
                    else if (m->byteAt(Eigen::Vector3d(q.x, q.y, q.z)) & 4u) hit = true;      // the pose itself inside an occupied voxel
                }
                if (hit) ++polyDirty;
            }
        }
        std::printf("INFO stress: %d rounds; bsplineTraj %d of %d planned, %d failed the host gate, %d clip a voxel between gate samples; polyTrajOctomap %d of %d valid, %d dirty\n",
                    rounds, planned, total, dirty, clipped, polyValid, polyTotal, polyDirty);
        CHECK(dirty == 0 && planned > 0, "stress: every bsplineTraj success is collision free on the host copy of the map");
        std::printf("INFO stress: %d plans carried dynamic obstacles; device-resident vs host-driven rebound loop: %d of %d differ\n", obstaclePlans, loopMismatch, total);
        CHECK(loopMismatch == 0, "stress: device-resident rebound loop == host-driven loop on every planner (bit for bit)");
        CHECK(polyDirty == 0, "stress: every valid polyTrajOctomap plan keeps its poses out of occupied voxels");
    }

    std::printf("%s (%d failures)\n", fails ? "FAILED" : "PASSED", fails);
    return fails ? 1 : 0;
}
