// polyTrajOctomap.cpp — min-snap + corridor planner facade (see the header).  Behaviour follows
// polyTrajOctomap.cpp:14-110 (parameters), :178-192, :226-545 (planning loops), :547-689 (checker,
// getters); the box sweep of every trajectory sample runs on the device.
#include <trajectory_planner/polyTrajOctomap.h>

#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cmath>
#include <iostream>

#include "../../../include/vigo.h"
#include "devbuf.h"

using std::cout;
using std::endl;

namespace trajPlanner {

static double nowSec() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

polyTrajOctomap::polyTrajOctomap() : polyTrajOctomap(ros::NodeHandle()) {}

polyTrajOctomap::polyTrajOctomap(const ros::NodeHandle& nh) : nh_(nh) {
    // PO.cpp:14-108: same (un-namespaced) keys and fall-back values
    if (!nh_.getParam("collision_box", collisionBox_) || collisionBox_.size() < 3) collisionBox_ = {0.5, 0.5, 0.5};
    if (!nh_.getParam("polynomial_degree", polyDegree_)) polyDegree_ = 7;
    if (!nh_.getParam("differential_degree", diffDegree_)) diffDegree_ = 4;
    if (!nh_.getParam("continuity_degree", continuityDegree_)) continuityDegree_ = 4;
    if (!nh_.getParam("desired_velocity", desiredVel_)) desiredVel_ = 1.0;
    if (!nh_.getParam("map_resolution", mapRes_)) mapRes_ = 0.2;
    if (!nh_.getParam("maximum_iteration_num", maxIter_)) maxIter_ = 20;
    if (!nh_.getParam("traj_timeout", timeout_)) timeout_ = 0.1;
    if (!nh_.getParam("mode", mode_)) mode_ = true;
    if (!nh_.getParam("sample_delta_time", delT_)) delT_ = 0.1;
    if (!nh_.getParam("initial_radius", initR_)) initR_ = 0.5;
    if (!nh_.getParam("shrinking_factor", fs_)) fs_ = 0.8;
    if (!nh_.getParam("corridor_res", corridorRes_)) corridorRes_ = 5.0;
    if (!nh_.getParam("soft_constraint", softConstraint_)) softConstraint_ = false;                       // PO.cpp:98-107
    if (softConstraint_ && !nh_.getParam("constraint_radius", softConstraintRadius_)) softConstraintRadius_ = 0.5;
}

polyTrajOctomap::~polyTrajOctomap() {
    if (dev_) vigo_destroy(dev_);
}

void polyTrajOctomap::setMap(const std::shared_ptr<mapManager::occMap>& map) {
    map_ = map;
    mapStamp_ = 0;
}

bool polyTrajOctomap::syncDevice() {
    if (!map_) return false;
    if (hipSetDevice(deviceOrdinal_) != hipSuccess) {    // this planner's GPU current on the calling thread (stream, staging)
        cout << "[Trajectory Planner INFO]: HIP device " << deviceOrdinal_ << " is not available (no CPU fallback)." << endl;
        return false;
    }
    if (!dev_ && vigo_create(&dev_, deviceOrdinal_) != VIGO_OK) {
        cout << "[Trajectory Planner INFO]: no HIP device for the corridor checker (no CPU fallback)." << endl;
        dev_ = nullptr;
        return false;
    }
    // launches and staging copies of this call go to the calling thread's stream
    if (vigo_set_stream(dev_, vigo_host::threadStream()) != VIGO_OK) return false;
    return mapAdapter::uploadSnapshot(dev_, map_, mapRegion_, mapStamp_);
}

// see bsplineTraj::setDevice
void polyTrajOctomap::setDevice(int ordinal) {
    if (ordinal == deviceOrdinal_) return;
    if (dev_) { vigo_destroy(dev_); dev_ = nullptr; }
    mapStamp_ = 0;
    deviceOrdinal_ = ordinal;
}

void polyTrajOctomap::updatePath(const nav_msgs::Path& path) {
    std::vector<pose> trajPath;
    for (const auto& p : path.poses) trajPath.push_back(pose(p.pose.position.x, p.pose.position.y, p.pose.position.z));
    this->updatePath(trajPath);
}

void polyTrajOctomap::updatePath(const std::vector<pose>& path) { this->path_ = path; extKnots_.clear(); }
void polyTrajOctomap::updateInitVel(double vx, double vy, double vz) { initVel_[0] = vx; initVel_[1] = vy; initVel_[2] = vz; }
void polyTrajOctomap::updateInitAcc(double ax, double ay, double az) { initAcc_[0] = ax; initAcc_[1] = ay; initAcc_[2] = az; }
void polyTrajOctomap::setDefaultInit() { updateInitVel(0, 0, 0); updateInitAcc(0, 0, 0); }

void polyTrajOctomap::setSolution(int polyDegree, const std::vector<double>& xSol, const std::vector<double>& ySol,
                                  const std::vector<double>& zSol, const std::vector<double>& timeKnot) {
    extDegree_ = polyDegree;
    xSol_ = xSol; ySol_ = ySol; zSol_ = zSol;
    extKnots_ = timeKnot;
    trajSolver_.reset();
}

// PO.cpp:178-186
void polyTrajOctomap::insertWaypoint(const std::set<int>& seg) {
    for (auto rit = seg.rbegin(); rit != seg.rend(); ++rit) {
        const int idx = *rit;
        if (idx < 0 || idx + 1 >= (int)path_.size()) continue;
        const pose p1 = path_[idx], p2 = path_[idx + 1];
        path_.insert(path_.begin() + idx + 1, pose((p1.x + p2.x) / 2, (p1.y + p2.y) / 2, (p1.z + p2.z) / 2));
    }
}

const std::vector<double>& polyTrajOctomap::timeKnots() {
    if (trajSolver_) return trajSolver_->getTimeKnot();
    if (!extKnots_.empty()) return extKnots_;
    return pwlKnots_;
}

bool polyTrajOctomap::sweepPoints(const std::vector<pose>& pts, std::vector<uint8_t>& flags) {
    flags.assign(pts.size(), 1);
    if (pts.empty()) return true;
    if (!syncDevice()) return false;
    std::vector<double> xyz(pts.size() * 3);
    for (size_t i = 0; i < pts.size(); ++i) { xyz[3 * i] = pts[i].x; xyz[3 * i + 1] = pts[i].y; xyz[3 * i + 2] = pts[i].z; }
    static thread_local vigo_host::StagingBuf dP, dF;      // reused by every sweep of this thread
    bool ok = dP.upload(xyz.data(), xyz.size() * 8) && dF.alloc(pts.size());
    const double box[3] = {collisionBox_[0], collisionBox_[1], collisionBox_[2]};
    ok = ok && vigo_box_collision_points(dev_, (int64_t)pts.size(), (const double*)dP.p, box, mapRes_, (uint8_t*)dF.p) == VIGO_OK;
    ok = ok && dF.download(flags.data(), pts.size());
    // A pose at NaN or infinity: the reference's sweep makes no pass there (the lattice count is the conversion of a
    // NaN, INT_MIN on x86 — the device entry point follows that, include/vigo.h) and would publish the trajectory.  The
    // facade refuses such a pose instead: a degenerate min-snap solution (coincident waypoints) then falls back to the
    // piecewise-linear plan like any colliding one.
    for (size_t i = 0; i < pts.size(); ++i)
        if (!(std::isfinite(pts[i].x) && std::isfinite(pts[i].y) && std::isfinite(pts[i].z))) flags[i] = 1;
    if (!ok && dev_) cout << "[Trajectory Planner INFO]: device box sweep failed: " << vigo_last_error(dev_) << endl;
    return ok;
}

bool polyTrajOctomap::checkCollision(const pose& p) {
    std::vector<uint8_t> f;
    sweepPoints({p}, f);
    return f[0] != 0;
}

// PO.cpp:571-589 on the dense map (host: a single lookup)
bool polyTrajOctomap::checkCollisionPoint(const pose& p, bool ignoreUnknown) {
    if (!map_) return true;
    const unsigned v = mapAdapter::nodeBits(map_, mapRegion_, (float)p.x, (float)p.y, (float)p.z);   // pose2Octomap + search()
    if (v & mapAdapter::kOutside) return true;                  // beyond getMetricMin/Max
    if (v & mapAdapter::kUnknown) return !ignoreUnknown;        // no node there
    return (v & mapAdapter::kOccupied) != 0;                    // isNodeOccupied
}

// PO.cpp:619-632
bool polyTrajOctomap::checkCollisionTraj(const std::vector<pose>& trajectory, std::vector<int>& collisionIdx) {
    std::vector<uint8_t> f;
    sweepPoints(trajectory, f);
    bool has = false;
    for (size_t i = 0; i < f.size(); ++i)
        if (f[i]) { has = true; collisionIdx.push_back((int)i); }
    return has;
}

// PO.cpp:634-656: t accumulates delT per sample; first time-knot interval containing t (inclusive)
bool polyTrajOctomap::checkCollisionTraj(const std::vector<pose>& trajectory, double delT, std::set<int>& collisionSeg) {
    collisionSeg.clear();
    std::vector<uint8_t> f;
    sweepPoints(trajectory, f);
    const std::vector<double>& knots = timeKnots();
    double t = 0;
    bool has = false;
    for (size_t k = 0; k < trajectory.size(); ++k) {
        if (f[k]) {
            has = true;
            for (size_t i = 0; i + 1 < knots.size(); ++i)
                if (t >= knots[i] && t <= knots[i + 1]) { collisionSeg.insert((int)i); break; }
        }
        t += delT;
    }
    return has;
}

// ---- the fallback of PO.cpp:308-318, :373-383, :528-541: a fresh pwlTraj over the waypoints (its own 1.0 m/s and
// 0.5 rad/s, piecewiseLinearTraj.h:20-21 — not the planner's desired velocity), sampled at delT ----
void polyTrajOctomap::pwlPlan(std::vector<pose>& trajectory, double delT) {
    pwlTrajSolver_.reset(new pwlTraj(nh_));
    trajectory.clear();
    pwlKnots_.clear();
    if (path_.empty()) return;
    pwlTrajSolver_->updatePath(path_);
    pwlTrajSolver_->makePlan(trajectory, delT);
    pwlKnots_ = pwlTrajSolver_->getTimeKnot();
}

// polyTrajSolver::getPose on an externally supplied polynomial (PS.cpp:1026-1056)
pose polyTrajOctomap::extPose(double t) {
    pose p;
    for (size_t i = 0; i + 1 < extKnots_.size(); ++i) {
        if (t >= extKnots_[i] && t <= extKnots_[i + 1]) {
            t = (double)(t - extKnots_[i]);
            const int c0 = (extDegree_ + 1) * (int)i;
            double x = 0, y = 0, z = 0;
            for (int d = 0; d < extDegree_ + 1; ++d) {
                x += xSol_[c0 + d] * std::pow(t, d);
                y += ySol_[c0 + d] * std::pow(t, d);
                z += zSol_[c0 + d] * std::pow(t, d);
            }
            if (t == 0) t = 0.01;
            double dx = 0, dy = 0;
            for (int d = 0; d < extDegree_ + 1; ++d) {
                dx += d * xSol_[c0 + d] * std::pow(t, d - 1);
                dy += d * ySol_[c0 + d] * std::pow(t, d - 1);
            }
            p.x = x; p.y = y; p.z = z; p.yaw = std::atan2(dy, dx);
            break;
        }
    }
    return p;
}

// PO.cpp:472-545
void polyTrajOctomap::makePlanCorridorConstraint(std::vector<pose>& trajectory, double delT) {
    this->setDefaultInit();
    trajSolver_.reset(new polyTrajSolver(polyDegree_, diffDegree_, continuityDegree_, desiredVel_));
    trajSolver_->updatePath(path_);
    trajSolver_->updateInitVel(initVel_[0], initVel_[1], initVel_[2]);
    trajSolver_->updateInitAcc(initAcc_[0], initAcc_[1], initAcc_[2]);
    std::vector<double> corridorSizeVec(path_.size() - 1, initR_);
    int countIter = 0;
    bool valid = false;
    const double t0 = nowSec();
    while (!valid) {
        if (nowSec() - t0 >= timeout_) { cout << "[Trajectory Planner INFO]: Timeout." << endl; break; }
        trajSolver_->setCorridorConstraint(corridorSizeVec, corridorRes_);
        if (softConstraint_) trajSolver_->setSoftConstraint(softConstraintRadius_, softConstraintRadius_, 0);   // PO.cpp:290-292, :354-356, :426-428
        trajSolver_->solve();
        // an infeasible corridor keeps the previous polynomial, like the reference; with none to keep (the very
        // first corridor was infeasible, and shrinking it cannot help) there is nothing to sample: not found
        if (!trajSolver_->hasSolution()) break;
        trajSolver_->getTrajectory(trajectory, delT);
        std::set<int> collisionSeg;
        valid = !this->checkCollisionTraj(trajectory, delT, collisionSeg);
        if (!valid)
            this->adjustCorridorSize(collisionSeg, corridorSizeVec);
        ++countIter;
        if (countIter > maxIter_) break;
    }
    lastIterations_ = countIter;
    findValidTraj_ = valid;
}

// PO.cpp:259-386
void polyTrajOctomap::makePlanAddingWaypoint(std::vector<pose>& trajectory, double delT) {
    this->setDefaultInit();
    trajSolver_.reset(new polyTrajSolver(polyDegree_, diffDegree_, continuityDegree_, desiredVel_));
    trajSolver_->updateInitVel(initVel_[0], initVel_[1], initVel_[2]);
    trajSolver_->updateInitAcc(initAcc_[0], initAcc_[1], initAcc_[2]);
    trajSolver_->updatePath(path_);
    int countIter = 0;
    bool valid = false;
    const double t0 = nowSec();
    while (!valid) {
        if (nowSec() - t0 >= timeout_) { cout << "[Trajectory Planner INFO]: Timeout." << endl; break; }
        if (softConstraint_) trajSolver_->setSoftConstraint(softConstraintRadius_, softConstraintRadius_, 0);   // PO.cpp:290-292, :354-356, :426-428
        trajSolver_->solve();
        if (!trajSolver_->hasSolution()) break;   // degenerate path (e.g. coincident waypoints): nothing to sample
        trajSolver_->getTrajectory(trajectory, delT);
        std::set<int> collisionSeg;
        valid = !this->checkCollisionTraj(trajectory, delT, collisionSeg);
        if (!valid) {
            this->insertWaypoint(collisionSeg);
            trajSolver_->updatePath(path_);   // (the reference never refreshes the solver's path here)
        }
        ++countIter;
        if (countIter > maxIter_) break;
    }
    lastIterations_ = countIter;
    findValidTraj_ = valid;
}

void polyTrajOctomap::makePlan(std::vector<pose>& trajectory, double delT) {
    this->findValidTraj_ = false;
    if (this->path_.empty()) return;
    if (this->path_.size() == 1) { trajectory = this->path_; this->findValidTraj_ = true; return; }
    if (!extKnots_.empty() && !trajSolver_) {
        // an installed polynomial: one pass of the loop body (sample -> device sweep)
        trajectory.clear();
        for (double t = 0; t < extKnots_.back(); t += delT) trajectory.push_back(extPose(t));
        trajectory.push_back(path_.back());
        std::set<int> collisionSeg;
        findValidTraj_ = !this->checkCollisionTraj(trajectory, delT, collisionSeg);
        return;
    }
    if (mode_) makePlanAddingWaypoint(trajectory, delT);
    else makePlanCorridorConstraint(trajectory, delT);
    if (findValidTraj_) {
        cout << "[Trajectory Planner INFO]: Found valid trajectory!" << endl;
    } else {
        cout << "[Trajectory Planner INFO]: Not found. Return the best. Please consider piecewise linear trajectory!!" << endl;
        trajSolver_.reset();
        pwlPlan(trajectory, delT);
    }
}

// makePlan() of many planners in lock-step, BOTH planning loops (adding waypoints PO.cpp:259-386, corridor constraint
// PO.cpp:388-545).  Per round: the active planners are grouped by (waypoint count, mode) and every group's QPs are
// ONE vigo_minsnap launch (corridor boxes for the corridor mode, none for the adding-waypoint mode, whose paths grow
// as waypoints are inserted: the groups are re-formed every round); then every sample of every candidate trajectory
// goes through ONE vigo_box_collision_points launch; the per-planner bookkeeping (shrink the corridor of the
// colliding segments / insert waypoints there, iteration and time limits, the PWL fallback) stays on the host.
// A path that outgrows the device QP (more than 11 waypoints) is solved by the host QP of the same algorithm
// inside the same round.
std::vector<bool> polyTrajOctomap::makePlanBatch(const std::vector<polyTrajOctomap*>& ps, std::vector<std::vector<pose>>& trajectories) {
    const size_t P = ps.size();
    std::vector<bool> result(P, false);
    trajectories.assign(P, {});
    if (P == 0) return result;
    // planners the batch cannot take (installed polynomial, a single waypoint, another polynomial degree, another
    // map or sweep geometry than the first planner's, soft waypoint constraints — the device QP takes the waypoints
    // as equalities) plan on their own
    std::vector<size_t> grp;
    bool toldSoft = false;
    for (size_t i = 0; i < P; ++i) {
        polyTrajOctomap* p = ps[i];
        p->findValidTraj_ = false;
        const bool batchable = p->extKnots_.empty() && p->path_.size() >= 2 && p->polyDegree_ == 7 && p->diffDegree_ == ps[0]->diffDegree_ &&
                               p->continuityDegree_ == ps[0]->continuityDegree_ && p->desiredVel_ == ps[0]->desiredVel_ &&
                               p->corridorRes_ == ps[0]->corridorRes_ && p->deviceOrdinal_ == ps[0]->deviceOrdinal_ && p->map_ == ps[0]->map_ && sameRegion(p->mapRegion_, ps[0]->mapRegion_) && p->collisionBox_ == ps[0]->collisionBox_ &&
                               p->mapRes_ == ps[0]->mapRes_ && !p->softConstraint_;
        if (batchable) grp.push_back(i);
        else {
            if (p->softConstraint_ && !toldSoft) {
                // (once per call: such a planner leaves the batched device path — the device QP takes waypoints as equalities)
                cout << "[Trajectory Planner INFO]: soft waypoint constraints: planned on the host path, outside the device batch." << endl;
                toldSoft = true;
            }
            p->makePlan(trajectories[i], p->delT_);
            result[i] = p->findValidTraj_;
        }
    }
    if (grp.empty()) return result;
    polyTrajOctomap* lead = ps[grp[0]];
    if (!lead->syncDevice()) return result;
    const int D = 8, kMaxDevWaypoints = 11;
    struct State { std::vector<double> corridor; int iters = 0; bool active = true; double t0 = 0; };
    const size_t G = grp.size();
    std::vector<State> st(G);
    for (size_t g = 0; g < G; ++g) {
        polyTrajOctomap* p = ps[grp[g]];
        p->setDefaultInit();
        p->trajSolver_.reset(new polyTrajSolver(p->polyDegree_, p->diffDegree_, p->continuityDegree_, p->desiredVel_));
        p->trajSolver_->updatePath(p->path_);
        if (!p->mode_) st[g].corridor.assign(p->path_.size() - 1, p->initR_);
        st[g].t0 = nowSec();
    }
    static thread_local vigo_host::StagingBuf bWp, bCor, bCo, bKn, bSt, bPts, bFl;   // reused by every batch of this thread
    bool ok = true;
    while (ok) {
        std::vector<size_t> act;
        for (size_t g = 0; g < G; ++g) if (st[g].active) act.push_back(g);
        if (act.empty()) break;
        // ---- the QPs: one launch per (waypoint count, mode) among the active planners ----
        std::vector<bool> solved(G, false);
        for (size_t a0 = 0; a0 < act.size() && ok; ++a0) {
            const size_t g0 = act[a0];
            if (solved[g0]) continue;
            polyTrajOctomap* p0 = ps[grp[g0]];
            const int W = (int)p0->path_.size(), K = W - 1;
            if (W > kMaxDevWaypoints) {                       // beyond the device QP: the host QP, same algorithm
                // (corridor mode: the boxes of this round before every solve, like makePlanCorridorConstraint — PO.cpp:421-424;
                // an infeasible corridor keeps the previous polynomial there as here)
                if (!p0->mode_) p0->trajSolver_->setCorridorConstraint(st[g0].corridor, p0->corridorRes_);
                p0->trajSolver_->solve();
                solved[g0] = true;
                continue;
            }
            std::vector<size_t> members;
            for (size_t a = a0; a < act.size(); ++a) {
                polyTrajOctomap* q = ps[grp[act[a]]];
                if (!solved[act[a]] && (int)q->path_.size() == W && q->mode_ == p0->mode_) { members.push_back(act[a]); solved[act[a]] = true; }
            }
            const int T = (int)members.size();
            std::vector<double> hWp, hCor, hCo((size_t)T * K * 3 * D);
            std::vector<int32_t> hSt(T);
            for (size_t g : members) {
                for (const pose& q : ps[grp[g]]->path_) { hWp.push_back(q.x); hWp.push_back(q.y); hWp.push_back(q.z); }
                if (!p0->mode_) hCor.insert(hCor.end(), st[g].corridor.begin(), st[g].corridor.end());
            }
            ok = bWp.upload(hWp.data(), hWp.size() * 8) && (p0->mode_ || bCor.upload(hCor.data(), hCor.size() * 8)) &&
                 bCo.alloc(hCo.size() * 8) && bKn.alloc((size_t)T * W * 8) && bSt.alloc((size_t)T * 4) &&
                 vigo_minsnap(lead->dev_, T, W, 7, lead->diffDegree_, lead->continuityDegree_, lead->desiredVel_, lead->corridorRes_,
                              (const double*)bWp.p, p0->mode_ ? nullptr : (const double*)bCor.p, nullptr, (double*)bCo.p, (double*)bKn.p,
                              (int32_t*)bSt.p) == VIGO_OK &&
                 bCo.download(hCo.data(), hCo.size() * 8) && bSt.download(hSt.data(), (size_t)T * 4);
            if (!ok) break;
            // install the solutions (an infeasible corridor keeps the previous one, like the reference)
            for (int a = 0; a < T; ++a) {
                if (hSt[a] != 0) continue;
                std::vector<double> xs(K * D), ys(K * D), zs(K * D);
                for (int sgm = 0; sgm < K; ++sgm)
                    for (int d = 0; d < D; ++d) {
                        xs[sgm * D + d] = hCo[(((size_t)a * K + sgm) * 3 + 0) * D + d];
                        ys[sgm * D + d] = hCo[(((size_t)a * K + sgm) * 3 + 1) * D + d];
                        zs[sgm * D + d] = hCo[(((size_t)a * K + sgm) * 3 + 2) * D + d];
                    }
                ps[grp[members[a]]]->trajSolver_->installSolution(xs, ys, zs);
            }
        }
        if (!ok) break;
        // ---- sample every candidate, sweep all samples at once ----
        const int T = (int)act.size();
        std::vector<double> pts;
        std::vector<size_t> first(T + 1, 0);
        for (int a = 0; a < T; ++a) {
            polyTrajOctomap* p = ps[grp[act[a]]];
            if (!p->trajSolver_->hasSolution()) {      // nothing to sample (first corridor infeasible, degenerate path): not found
                trajectories[grp[act[a]]].clear();
                st[act[a]].active = false;
                first[a + 1] = pts.size() / 3;
                continue;
            }
            p->trajSolver_->getTrajectory(trajectories[grp[act[a]]], p->delT_);
            for (const pose& q : trajectories[grp[act[a]]]) { pts.push_back(q.x); pts.push_back(q.y); pts.push_back(q.z); }
            first[a + 1] = pts.size() / 3;
        }
        const size_t M = pts.size() / 3;
        std::vector<uint8_t> flags(M, 1);
        const double box[3] = {lead->collisionBox_[0], lead->collisionBox_[1], lead->collisionBox_[2]};
        if (M) {
            ok = bPts.upload(pts.data(), M * 24) && bFl.alloc(M) &&
                 vigo_box_collision_points(lead->dev_, (int64_t)M, (const double*)bPts.p, box, lead->mapRes_, (uint8_t*)bFl.p) == VIGO_OK &&
                 bFl.download(flags.data(), M);
            if (!ok) break;
            for (size_t k = 0; k < M; ++k)             // a pose at NaN or infinity is refused (see sweepPoints)
                if (!(std::isfinite(pts[3 * k]) && std::isfinite(pts[3 * k + 1]) && std::isfinite(pts[3 * k + 2]))) flags[k] = 1;
        }
        for (int a = 0; a < T; ++a) {
            const size_t g = act[a];
            polyTrajOctomap* p = ps[grp[g]];
            if (!st[g].active) continue;               // retired above without a polynomial
            const std::vector<double>& knots = p->trajSolver_->getTimeKnot();
            std::set<int> collisionSeg;   // PO.cpp:634-656
            double t = 0;
            bool has = false;
            for (size_t k = first[a]; k < first[a + 1]; ++k) {
                if (flags[k]) {
                    has = true;
                    for (size_t i = 0; i + 1 < knots.size(); ++i)
                        if (t >= knots[i] && t <= knots[i + 1]) { collisionSeg.insert((int)i); break; }
                }
                t += p->delT_;
            }
            ++st[g].iters;
            if (!has) { p->findValidTraj_ = true; st[g].active = false; }
            else {
                if (p->mode_) {
                    p->insertWaypoint(collisionSeg);               // PO.cpp:178-186
                    p->trajSolver_->updatePath(p->path_);
                } else {
                    for (int sgm : collisionSeg) st[g].corridor[sgm] *= p->fs_;   // adjustCorridorSize, PO.cpp:188-192
                }
                if (st[g].iters > p->maxIter_ || nowSec() - st[g].t0 >= p->timeout_ * (double)G) st[g].active = false;
            }
            p->lastIterations_ = st[g].iters;
        }
    }
    for (size_t g = 0; g < G; ++g) {
        polyTrajOctomap* p = ps[grp[g]];
        if (!p->findValidTraj_) { p->trajSolver_.reset(); p->pwlPlan(trajectories[grp[g]], p->delT_); }   // PO.cpp:459-467
        result[grp[g]] = p->findValidTraj_;
    }
    return result;
}

void polyTrajOctomap::makePlan() {
    std::vector<pose> trajectory;
    this->makePlan(trajectory, this->delT_);
}

void polyTrajOctomap::makePlan(nav_msgs::Path& trajectory, double delT) {
    std::vector<pose> tmp;
    this->makePlan(tmp, delT);
    this->trajMsgConverter(tmp, trajectory);
}

void polyTrajOctomap::trajMsgConverter(const std::vector<pose>& trajectoryTemp, nav_msgs::Path& trajectory) {
    trajectory.poses.clear();
    for (const pose& p : trajectoryTemp) {
        geometry_msgs::PoseStamped ps;
        ps.header.frame_id = "map";
        ps.pose.position.x = p.x; ps.pose.position.y = p.y; ps.pose.position.z = p.z;
        ps.pose.orientation = quaternion_from_rpy(0, 0, p.yaw);
        trajectory.poses.push_back(ps);
    }
    trajectory.header.frame_id = "map";
}

// PO.cpp:658-677
geometry_msgs::PoseStamped polyTrajOctomap::getPose(double t) {
    if (t > this->getDuration()) t = this->getDuration();
    if (!trajSolver_ && extKnots_.empty()) {                     // PO.cpp:672-674: the fallback answers in its own words
        if (pwlTrajSolver_) return pwlTrajSolver_->getPose(t);
        geometry_msgs::PoseStamped none;
        none.header.frame_id = "map";
        return none;
    }
    pose p = trajSolver_ ? trajSolver_->getPose(t) : extPose(t);
    geometry_msgs::PoseStamped ps;
    ps.pose.position.x = p.x; ps.pose.position.y = p.y; ps.pose.position.z = p.z;
    ps.pose.orientation = quaternion_from_rpy(0, 0, p.yaw);
    ps.header.frame_id = "map";
    return ps;
}

// PO.cpp:679-689
double polyTrajOctomap::getDuration() {
    if (this->path_.size() == 1) return 0.0;
    const std::vector<double>& k = timeKnots();
    return k.empty() ? 0.0 : k.back();
}

}  // namespace trajPlanner
