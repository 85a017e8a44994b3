// bsplineTraj.cpp — trajPlanner::bsplineTraj over the MI355X back-end.
//
// Behaviour follows the reference's bsplineTraj.{h,cpp} (cited per function as BT.cpp / BT.h);
// the numerics of optimize(), the collision gates and isUnknown(guide) run in libvigo_hip.so
// through the C ABI of include/vigo.h.  Own implementation: host bookkeeping only.
#include <trajectory_planner/bsplineTraj.h>

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <cstdlib>
#include <cmath>
#include <fstream>
#include <cstring>
#include <iostream>
#include <set>

#include "../../../include/vigo.h"
#include "devbuf.h"
#include "workerPool.h"

using std::cout;
using std::endl;

namespace {

using vigo_host::DevBuf;
using vigo_host::StagingBuf;

using vigo_host::parallelFor;

double wallSeconds() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

#ifndef VIGO_WITH_ROS
ros::Time ros::Time::now() {
    ros::Time t;
    t.sec = wallSeconds();
    return t;
}
#endif

namespace trajPlanner {

// The reference keeps the length of the previous call's path in a function-static shared by all instances (BT.cpp:755):
// kept here as one process-wide value, read and written through adjustPathLengthWith so that updatePathBatch can run
// the planners' prologues on several host threads and still hand every planner the value its predecessor left.
// (atomic only so that two host threads planning two batches do not race on it formally; WHICH value a planner of one
// batch sees when another batch runs at the same time is as unspecified as it is for the reference's static)
namespace {
std::atomic<double> g_prevPathLength{0.0};
}


bsplineTraj::bsplineTraj() {}

bsplineTraj::bsplineTraj(const ros::NodeHandle& nh) : nh_(nh) { this->initParam(); }

bsplineTraj::~bsplineTraj() {
    if (dev_) vigo_destroy(dev_);
}

void bsplineTraj::init(const ros::NodeHandle& nh) {
    this->nh_ = nh;
    this->initParam();
}

// BT.cpp:24-172: same keys, same fall-back values
void bsplineTraj::initParam() {
    auto get = [this](const char* key, double& dst, double fallback) {
        if (!this->nh_.getParam(key, dst)) dst = fallback;
    };
    get("bspline_traj/timestep", ts_, 0.1);
    get("bspline_traj/distance_threshold", dthresh_, 0.5);
    get("bspline_traj/max_vel", maxVel_, 1.0);
    get("bspline_traj/max_acc", maxAcc_, 0.5);
    get("bspline_traj/weight_distance", weightDistance_, 0.5);
    get("bspline_traj/weight_smoothness", weightSmoothness_, 1.0);
    get("bspline_traj/weight_feasibility", weightFeasibility_, 1.0);
    get("bspline_traj/weight_dynamic_obstacle", weightDynamicObstacle_, 1.0);
    if (!nh_.getParam("bspline_traj/plan_in_z_axis", planInZAxis_)) planInZAxis_ = true;
    get("bspline_traj/min_height", minHeight_, 0.5);
    get("bspline_traj/max_height", maxHeight_, 2.0);
    get("bspline_traj/uncertain_aware_factor", uncertainAwareFactor_, 2.0);
    get("bspline_traj/prediction_horizon", predHorizon_, 2.0);
    get("bspline_traj/distance_threshold_dynamic", distThreshDynamic_, 1.0);
    get("bspline_traj/max_path_length", maxPathLength_, 7.0);
    std::vector<double> mos;
    if (!nh_.getParam("bspline_traj/max_obstacle_size", mos) || mos.size() < 3) maxObstacleSize_ = Eigen::Vector3d(10.0, 10.0, 10.0);
    else maxObstacleSize_ = Eigen::Vector3d(mos[0], mos[1], mos[2]);
}

// BT.cpp:187-195
void bsplineTraj::setMap(const std::shared_ptr<mapManager::occMap>& map) {
    this->map_ = map;
    this->pathSearch_.reset(new AStar);
    int maxGridX = 2 * int(this->maxObstacleSize_(0) / this->map_->getRes());
    int maxGridY = 2 * int(this->maxObstacleSize_(1) / this->map_->getRes());
    int maxGridZ = 2 * int(this->maxObstacleSize_(2) / this->map_->getRes());
    this->pathSearch_->initGridMap(map, Eigen::Vector3i(maxGridX, maxGridY, maxGridZ), this->minHeight_, this->maxHeight_);
    this->mapStamp_ = 0;  // force a new device snapshot
}

void bsplineTraj::setMapRegion(const Eigen::Vector3d& boxMin, const Eigen::Vector3d& boxMax) {
    this->mapRegion_.set = true;
    this->mapRegion_.boxMin = boxMin;
    this->mapRegion_.boxMax = boxMax;
    this->mapStamp_ = 0;
}

void bsplineTraj::refreshMap() {
    mapAdapter::bumpGeneration(this->map_.get());   // every handle holding a snapshot of this map is stale, not only this planner's
    this->mapStamp_ = 0;
}

// one process per GPU is the deployment the back-end is built for (HIP_VISIBLE_DEVICES picks the card); a process that
// drives several cards gives each planner its ordinal before the planner's first device call.  A planner that already
// holds a handle on another card lets go of it: the next call creates a new one there and uploads the map again.
void bsplineTraj::setDevice(int ordinal) {
    if (ordinal == deviceOrdinal_) return;
    if (dev_) { vigo_destroy(dev_); dev_ = nullptr; }
    mapStamp_ = 0;
    deviceOrdinal_ = ordinal;
}

void bsplineTraj::updateMaxVel(double maxVel) { this->maxVel_ = maxVel; }
void bsplineTraj::updateMaxAcc(double maxAcc) { this->maxAcc_ = maxAcc; }

// every hot-path parameter the device sees for this planner
void bsplineTraj::fillParams(vigo_params_s* Pp) const {
    vigo_params_t& P = *Pp;
    vigo_default_params(&P);
    P.dthresh = dthresh_;
    P.dist_thresh_dynamic = distThreshDynamic_;
    P.ts_ctrl = controlPointsTs_;
    P.ts = ts_;
    P.pred_horizon = predHorizon_;
    P.uncertain_factor = uncertainAwareFactor_;
    P.w_distance = weightDistance_;
    P.w_smoothness = weightSmoothness_;
    P.w_feasibility = weightFeasibility_;
    P.w_dynamic = weightDynamicObstacle_;
    P.min_height = minHeight_;
    P.max_height = maxHeight_;
    P.plan_in_z = planInZAxis_ ? 1 : 0;
    P.mem_size = 16;          // BT.cpp:697
    P.max_iterations = 200;   // BT.cpp:698
    P.g_epsilon = 0.01;       // BT.cpp:699
}

// Two planners may share a device batch when the lead's handle state fits both: the same map object AND the same
// box of it to snapshot (setMapRegion), the same control-point count, gate step (maxVel_) and every parameter of fillParams() except the four weights, which
// travel per trajectory.
bool bsplineTraj::sameBatchKey(const bsplineTraj& o) const {
    if (deviceOrdinal_ != o.deviceOrdinal_ || map_ != o.map_ || !sameRegion(mapRegion_, o.mapRegion_) || maxVel_ != o.maxVel_ || notCheckRatio_ != o.notCheckRatio_ ||
        optData_.controlPoints.cols() != o.optData_.controlPoints.cols())
        return false;
    vigo_params_t a, b;
    this->fillParams(&a);
    o.fillParams(&b);
    a.w_distance = b.w_distance; a.w_smoothness = b.w_smoothness; a.w_feasibility = b.w_feasibility; a.w_dynamic = b.w_dynamic;
    return std::memcmp(&a, &b, sizeof(a)) == 0;
}

// handle creation, parameter push and (re)snapshot of the map when it changed (mapAdapter)
bool bsplineTraj::syncDevice() {
    // the planner's GPU is made current on the calling thread: its stream and staging buffers are per (thread, device)
    if (hipSetDevice(deviceOrdinal_) != hipSuccess) {
        cout << "[BsplineTraj]: HIP device " << deviceOrdinal_ << " is not available (there is no CPU fallback)." << endl;
        return false;
    }
    if (!dev_) {
        if (vigo_create(&dev_, deviceOrdinal_) != VIGO_OK) {
            cout << "[BsplineTraj]: no HIP device for the ViGO back-end (there is no CPU fallback)." << endl;
            dev_ = nullptr;
            return false;
        }
    }
    // launches and staging copies of this call go to the calling thread's stream (two host threads planning two
    // batches then overlap on the device)
    if (vigo_set_stream(dev_, vigo_host::threadStream()) != VIGO_OK) return false;
    vigo_params_t P;
    this->fillParams(&P);
    if (vigo_set_params(dev_, &P) != VIGO_OK) return false;
    if (map_ && !mapAdapter::uploadSnapshot(dev_, map_, mapRegion_, mapStamp_)) return false;
    return true;
}

// BT.cpp:207-245
bool bsplineTraj::inputPathCheck(const nav_msgs::Path& path, nav_msgs::Path& adjustedPath, double dt, double& finalTime) {
    if (path.poses.size() == 0) return true;
    std::vector<Eigen::Vector3d> curveFitPoints, adjustedCurveFitPoints;
    this->pathMsgToEigenPoints(path, curveFitPoints);
    this->adjustPathLengthDirect(curveFitPoints, adjustedCurveFitPoints);
    for (size_t i = 0; i + 1 < adjustedCurveFitPoints.size(); ++i) {
        double dist = (adjustedCurveFitPoints[i] - adjustedCurveFitPoints[i + 1]).norm();
        if (dist > this->controlPointDistance_ * 1.5) return false;
    }
    Eigen::Vector3d prevPoint;
    std::vector<Eigen::Vector3d> adjustedPoints;
    for (size_t i = 0; i < adjustedCurveFitPoints.size(); ++i) {
        Eigen::Vector3d p = adjustedCurveFitPoints[i];
        if (i == 0) {
            adjustedPoints.push_back(p);
            prevPoint = p;
        } else if ((p - prevPoint).norm() >= this->controlPointDistance_ * 0.8) {
            adjustedPoints.push_back(p);
            prevPoint = p;
        }
    }
    adjustedPoints.push_back(adjustedPoints.back());
    this->eigenPointsToPathMsg(adjustedPoints, adjustedPath);
    finalTime = (adjustedCurveFitPoints.size() - 1) * dt;
    return true;
}

// BT.cpp:247-288
bool bsplineTraj::fillPath(const nav_msgs::Path& path, nav_msgs::Path& adjustedPath) {
    const int n = int(path.poses.size());
    if (n <= 1) return false;
    auto P = [&](int i) { return Eigen::Vector3d(path.poses[i].pose.position.x, path.poses[i].pose.position.y, path.poses[i].pose.position.z); };
    std::vector<Eigen::Vector3d> out;
    if (n == 2) {
        Eigen::Vector3d ps = P(0), pf = P(1);
        out = {ps, (pf - ps) / 3.0 + ps, 2.0 * (pf - ps) / 3.0 + ps, pf};
    } else if (n == 3) {
        Eigen::Vector3d ps = P(0), pm = P(1), pf = P(2);
        out = {ps, (ps + pm) / 2.0, pm, (pm + pf) / 2.0, pf};
    } else {
        adjustedPath = path;
        return true;
    }
    adjustedPath.poses.clear();
    for (const auto& q : out) {
        geometry_msgs::PoseStamped ps;
        ps.pose.position.x = q(0); ps.pose.position.y = q(1); ps.pose.position.z = q(2);
        adjustedPath.poses.push_back(ps);
    }
    return true;
}

// BT.cpp:290-312: everything of updatePath() before the fit — goal check, path-length adjustment,
// filling short paths, clear() — leaving the curve-fit points
bool bsplineTraj::prepareFitPoints(const nav_msgs::Path& adjustedPath, std::vector<Eigen::Vector3d>& adjustedCurveFitPoints) {
    bool wrote = false;
    const double prevIn = g_prevPathLength.load();
    double prevOut = prevIn;
    const bool ok = this->prepareFitPointsWith(adjustedPath, adjustedCurveFitPoints, prevIn, prevOut, wrote);
    if (wrote) g_prevPathLength.store(prevOut);
    return ok;
}

bool bsplineTraj::prepareFitPointsWith(const nav_msgs::Path& adjustedPath, std::vector<Eigen::Vector3d>& adjustedCurveFitPoints, double prevIn,
                                       double& prevOut, bool& wrote) {
    wrote = false;
    prevOut = prevIn;
    adjustedCurveFitPoints.clear();
    if (adjustedPath.poses.empty() || !map_) return false;
    Eigen::Vector3d goal(adjustedPath.poses.back().pose.position.x, adjustedPath.poses.back().pose.position.y,
                         adjustedPath.poses.back().pose.position.z);
    if (this->map_->isInflatedOccupied(goal)) {
        cout << "[bsplineTraj]: Invalid goal position: " << goal(0) << " " << goal(1) << " " << goal(2) << endl;
        return false;
    }
    std::vector<Eigen::Vector3d> adjustedPathVec, inputPathVec;
    this->pathMsgToEigenPoints(adjustedPath, adjustedPathVec);
    this->adjustPathLengthWith(adjustedPathVec, inputPathVec, prevIn, prevOut);
    wrote = true;
    nav_msgs::Path inputPath;
    this->eigenPointsToPathMsg(inputPathVec, inputPath);
    if (inputPath.poses.size() < 4) {
        if (!this->fillPath(adjustedPath, inputPath)) {
            cout << "[bsplineTraj]: Input path point size is less (or equal) than 1." << endl;
            return false;
        }
    }
    this->clear();
    this->pathMsgToEigenPoints(inputPath, adjustedCurveFitPoints);
    return true;
}

// BT.cpp:315-322
void bsplineTraj::installControlPoints(const Eigen::MatrixXd& controlPoints, const std::vector<Eigen::Vector3d>& adjustedCurveFitPoints) {
    this->optData_.controlPoints = controlPoints;
    int controlPointNum = controlPoints.cols();
    this->optData_.guidePoints.assign(controlPointNum, {});
    this->optData_.guideDirections.assign(controlPointNum, {});
    this->optData_.findGuidePoint.assign(controlPointNum, false);
    this->init_ = true;
    this->inputPathVis_ = adjustedCurveFitPoints;
}

// BT.cpp:290-323
bool bsplineTraj::updatePath(const nav_msgs::Path& adjustedPath, const std::vector<Eigen::Vector3d>& startEndConditions) {
    std::vector<Eigen::Vector3d> adjustedCurveFitPoints;
    if (!this->prepareFitPoints(adjustedPath, adjustedCurveFitPoints)) return false;
    Eigen::MatrixXd controlPoints;
    if (!trajPlanner::bspline::parameterizeToBspline(this->controlPointsTs_, adjustedCurveFitPoints, startEndConditions, controlPoints))
        return false;  // the reference exit(0)s here (bspline.cpp:80-91)
    this->installControlPoints(controlPoints, adjustedCurveFitPoints);
    return true;
}

// updatePath() for many planners: the host prologue per planner, then ONE vigo_bspline_fit launch
// per group of equal waypoint count (bspline::parameterizeToBspline, bspline.cpp:74-138, batched).
std::vector<bool> bsplineTraj::updatePathBatch(const std::vector<bsplineTraj*>& planners, const std::vector<nav_msgs::Path>& paths,
                                               const std::vector<std::vector<Eigen::Vector3d>>& startEndConditions) {
    std::vector<bool> ok(planners.size(), false);
    if (paths.size() != planners.size() || startEndConditions.size() != planners.size()) return ok;
    std::vector<std::vector<Eigen::Vector3d>> fitPts(planners.size());
    std::vector<bool> ready(planners.size(), false);
    // The prologues (goal check, path-length adjustment with its line checks against the map, filling) run on the host
    // workers, each as if its predecessor had left a previous path length not above its own max_path_length — then the
    // value does not enter (BT.cpp:762: max(prevPathLength, maxPathLength_)); a serial pass hands the real value down the
    // line and repeats, in order, the rare planner for which it does enter.  Same results as one planner after another.
    std::vector<uint8_t> okv(planners.size(), 0), wrote(planners.size(), 0);
    std::vector<double> prevOut(planners.size(), 0.0);
    parallelFor(planners.size(), [&](size_t i) {
        if (startEndConditions[i].size() != 4) return;
        bool w = false;
        okv[i] = planners[i]->prepareFitPointsWith(paths[i], fitPts[i], 0.0, prevOut[i], w) ? 1 : 0;
        wrote[i] = w ? 1 : 0;
    });
    double prev = g_prevPathLength.load();
    for (size_t i = 0; i < planners.size(); ++i) {
        if (startEndConditions[i].size() == 4 && wrote[i] && prev > planners[i]->maxPathLength_) {
            bool w = false;
            okv[i] = planners[i]->prepareFitPointsWith(paths[i], fitPts[i], prev, prevOut[i], w) ? 1 : 0;
        }
        if (wrote[i]) prev = prevOut[i];
        ready[i] = okv[i] && fitPts[i].size() > 3;
    }
    g_prevPathLength.store(prev);
    std::vector<bool> doneMask(planners.size(), false);
    for (size_t a = 0; a < planners.size(); ++a) {
        if (doneMask[a] || !ready[a]) continue;
        const int K = (int)fitPts[a].size();
        const double ts = planners[a]->controlPointsTs_;
        std::vector<size_t> grp;
        for (size_t b = a; b < planners.size(); ++b)
            if (!doneMask[b] && ready[b] && (int)fitPts[b].size() == K && planners[b]->controlPointsTs_ == ts) { grp.push_back(b); doneMask[b] = true; }
        bsplineTraj* lead = planners[a];
        if (K + 2 > VIGO_MAX_CTRL_POINTS || !lead->syncDevice()) continue;
        const int B = (int)grp.size();
        std::vector<double> pts((size_t)B * K * 3), cond((size_t)B * 12), ctrl((size_t)B * (K + 2) * 3);
        for (int b = 0; b < B; ++b) {
            for (int i = 0; i < K; ++i)
                for (int q = 0; q < 3; ++q) pts[((size_t)b * K + i) * 3 + q] = fitPts[grp[b]][i](q);
            for (int i = 0; i < 4; ++i)
                for (int q = 0; q < 3; ++q) cond[((size_t)b * 4 + i) * 3 + q] = startEndConditions[grp[b]][i](q);
        }
        static thread_local StagingBuf dPts, dCond, dCtrl;
        if (!dPts.upload(pts.data(), pts.size() * 8) || !dCond.upload(cond.data(), cond.size() * 8) || !dCtrl.alloc(ctrl.size() * 8)) continue;
        if (vigo_bspline_fit(lead->dev_, B, K, ts, (const double*)dPts.p, (const double*)dCond.p, (double*)dCtrl.p) != VIGO_OK) {
            cout << "[BsplineTraj]: vigo_bspline_fit failed: " << vigo_last_error(lead->dev_) << endl;
            continue;
        }
        if (!vigo_host::threadSync() || !dCtrl.download(ctrl.data(), ctrl.size() * 8)) continue;
        parallelFor((size_t)B, [&](size_t b) {
            Eigen::MatrixXd controlPoints;
            controlPoints.resize(3, K + 2);
            std::memcpy(controlPoints.data(), ctrl.data() + b * (K + 2) * 3, sizeof(double) * 3 * (K + 2));
            planners[grp[b]]->installControlPoints(controlPoints, fitPts[grp[b]]);
        });
        for (int b = 0; b < B; ++b) ok[grp[b]] = true;
    }
    return ok;
}

void bsplineTraj::updateDynamicObstacles(const std::vector<Eigen::Vector3d>& obstaclesPos, const std::vector<Eigen::Vector3d>& obstaclesVel,
                                         const std::vector<Eigen::Vector3d>& obstaclesSize) {
    this->optData_.dynamicObstaclesPos = obstaclesPos;
    this->optData_.dynamicObstaclesVel = obstaclesVel;
    this->optData_.dynamicObstaclesSize = obstaclesSize;
}

// BT.cpp:333-385
bool bsplineTraj::makePlan() {
    std::vector<bsplineTraj*> one{this};
    return makePlanBatch(one)[0];
}

bool bsplineTraj::makePlan(nav_msgs::Path& trajectory, bool yaw) {
    bool success = this->makePlan();
    trajectory = this->evalTrajToMsg(yaw);
    return success;
}

void bsplineTraj::clear() {
    this->optData_.guidePoints.clear();
    this->optData_.guideDirections.clear();
    this->optData_.dynamicObstaclesPos.clear();
    this->optData_.dynamicObstaclesVel.clear();
    this->optData_.dynamicObstaclesSize.clear();
    this->collisionSeg_.clear();
    this->astarPaths_.clear();
}

// BT.cpp:403-445 (incl. the corner case that can duplicate a segment, :426-430)
void bsplineTraj::findCollisionSeg(const Eigen::MatrixXd& controlPoints, std::vector<std::pair<int, int>>& collisionSeg) {
    collisionSeg.clear();
    bool previousHasCollision = false;
    const int N = controlPoints.cols();
    int endIdx = int((N - bsplineDegree - 1) - this->notCheckRatio_ * (N - 2 * bsplineDegree));
    int pairStartIdx = bsplineDegree, pairEndIdx = bsplineDegree;
    for (int i = bsplineDegree; i <= endIdx; ++i) {
        Eigen::Vector3d p = controlPoints.col(i);
        bool hasCollision = this->map_->isInflatedOccupied(p);
        if (hasCollision != previousHasCollision) {
            if (hasCollision) {
                pairStartIdx = i - 1;
            } else {
                pairEndIdx = i;
                collisionSeg.push_back({pairStartIdx, pairEndIdx});
            }
        }
        if (hasCollision && i == endIdx - 1) {
            pairEndIdx = N - 1;
            collisionSeg.push_back({pairStartIdx, pairEndIdx});
        }
        if (i != bsplineDegree && !previousHasCollision && !hasCollision) {
            if (this->map_->isInflatedOccupiedLine(controlPoints.col(i - 1), p)) collisionSeg.push_back({i - 1, i});
        }
        previousHasCollision = hasCollision;
    }
}

// BT.cpp:447-514 (merge bookkeeping reproduced as written, Appendix B of SURVEY.md)
bool bsplineTraj::pathSearch(std::vector<std::pair<int, int>>& collisionSeg, std::vector<std::vector<Eigen::Vector3d>>& paths) {
    paths.clear();
    std::vector<int> mergeIndices;
    int collisionSegNum = int(collisionSeg.size());
    for (int i = 0; i < collisionSegNum; ++i) {
        std::pair<int, int> seg = collisionSeg[i];
        Eigen::Vector3d pStart = this->optData_.controlPoints.col(seg.first);
        Eigen::Vector3d pEnd = this->optData_.controlPoints.col(seg.second);
        if (this->pathSearch_->AstarSearch(this->map_->getRes(), pStart, pEnd)) {
            std::vector<Eigen::Vector3d> searchedPath = this->pathSearch_->getPath();
            searchedPath[0] = pStart;
            searchedPath.push_back(pEnd);
            paths.push_back(searchedPath);
        } else {
            if (i + 1 < collisionSegNum) {
                std::pair<int, int> nextSeg = collisionSeg[i + 1];
                if (nextSeg.first - seg.second <= 2) {
                    Eigen::Vector3d pEnd2 = this->optData_.controlPoints.col(nextSeg.second);
                    if (this->pathSearch_->AstarSearch(this->map_->getRes(), pStart, pEnd2)) {
                        std::vector<Eigen::Vector3d> searchedPath = this->pathSearch_->getPath();
                        searchedPath[0] = pStart;
                        searchedPath.push_back(pEnd2);
                        paths.push_back(searchedPath);
                        mergeIndices.push_back(i);
                        ++i;
                        continue;
                    }
                }
            }
            cout << "[BsplineTraj]: Path Search Error. Force return." << endl;
            return false;
        }
    }
    if (!mergeIndices.empty()) {
        int midx = 0;
        std::vector<std::pair<int, int>> collisionSegTemp;
        for (int i = 0; i < collisionSegNum; ++i) {
            if (midx < int(mergeIndices.size()) && i == mergeIndices[midx]) {
                collisionSegTemp.push_back({collisionSeg[i].first, collisionSeg[i + 1].second});
                ++i;
                ++midx;
            } else {
                collisionSeg.push_back(collisionSeg[i]);  // BT.cpp:507 pushes into the input...
            }
        }
        collisionSeg = collisionSegTemp;  // ...and :510 overwrites it: unmerged segments are dropped
    }
    return true;
}

// BT.h:196-204
bool bsplineTraj::checkCollisionLine(const Eigen::Vector3d& p1, const Eigen::Vector3d& p2) {
    for (double a = 0.0; a <= 1.0; a += this->map_->getRes()) {
        Eigen::Vector3d pMid = a * p1 + (1 - a) * p2;
        if (this->map_->isInflatedOccupied(pMid)) return true;
    }
    return false;
}

// BT.h:206-240
void bsplineTraj::shortcutPath(const std::vector<Eigen::Vector3d>& path, std::vector<Eigen::Vector3d>& pathSC) {
    pathSC.clear();
    size_t ptr1 = 0, ptr2 = 2;
    pathSC.push_back(path[ptr1]);
    if (path.size() == 1) return;
    if (path.size() == 2) { pathSC.push_back(path[1]); return; }
    while (true) {
        if (ptr2 > path.size() - 1) break;
        if (!this->checkCollisionLine(path[ptr1], path[ptr2])) {
            if (ptr2 >= path.size() - 1) { pathSC.push_back(path[ptr2]); break; }
            ++ptr2;
        } else {
            pathSC.push_back(path[ptr2 - 1]);
            ptr1 = ptr2 - 1;
            ptr2 = ptr1 + 2;
        }
    }
}

// BT.h:251-304
bool bsplineTraj::findGuidePointSemiCircle(int controlPointIdx, const std::pair<int, int>& seg,
                                           const std::vector<Eigen::Vector3d>& path, Eigen::Vector3d& guidePoint) {
    double minAngle = 0.0, maxAngle = PI_const;
    int numControlpoints = seg.second - seg.first - 1;
    double targetAngle;
    Eigen::Vector3d pseudo;
    if (numControlpoints != 0) {
        int order = controlPointIdx - seg.first;
        targetAngle = order * PI_const / (numControlpoints + 2);
        targetAngle = std::min(std::max(minAngle, targetAngle), maxAngle);
        double ratio = double(order) / double(numControlpoints + 1.0);
        pseudo = ratio * (path.back() - path[0]) + path[0];
    } else {
        targetAngle = PI_const / 2.0;
        pseudo = (path[0] + path.back()) / 2.0;
    }
    Eigen::Vector3d direction = path[0] - pseudo;
    for (size_t i = 0; i + 1 < path.size(); ++i) {
        Eigen::Vector3d wpCurr = path[i], wpNext = path[i + 1];
        double angleCurr = angleBetweenVectors(direction, wpCurr - pseudo);
        double angleNext = angleBetweenVectors(direction, wpNext - pseudo);
        if (targetAngle >= angleCurr && targetAngle <= angleNext) {
            double prevAngleDiff = 0.0;
            Eigen::Vector3d prevTempPoint;
            for (double a = 1.0; a >= 0.0; a -= 0.1) {
                Eigen::Vector3d tempPoint = a * wpCurr + (1 - a) * wpNext;
                double angleDiff = angleBetweenVectors(direction, tempPoint - pseudo) - targetAngle;
                if (angleDiff == 0) { guidePoint = tempPoint; return true; }
                if (angleDiff * prevAngleDiff < 0) {
                    double totalDiff = std::abs(angleDiff) + std::abs(prevAngleDiff);
                    guidePoint = std::abs(prevAngleDiff) / totalDiff * (tempPoint - prevTempPoint) + prevTempPoint;
                    return true;
                }
                prevAngleDiff = angleDiff;
                prevTempPoint = tempPoint;
            }
        }
    }
    return false;
}

// BT.cpp:517-571
void bsplineTraj::assignGuidePointsSemiCircle(const std::vector<std::vector<Eigen::Vector3d>>& paths,
                                              const std::vector<std::pair<int, int>>& collisionSeg) {
    std::vector<std::vector<Eigen::Vector3d>> pathsSC;
    this->shortcutPaths(paths, pathsSC);
    const int N = this->optData_.controlPoints.cols();
    Eigen::Vector3d guidePoint, guideDirection;
    for (size_t i = 0; i < collisionSeg.size() && i < pathsSC.size(); ++i) {
        const std::pair<int, int> seg = collisionSeg[i];
        const std::vector<Eigen::Vector3d>& path = pathsSC[i];
        for (int idx = seg.first + 1; idx < seg.second; ++idx) {
            if (idx < 0 || idx >= N) continue;
            this->findGuidePointSemiCircle(idx, seg, path, guidePoint);
            this->optData_.guidePoints[idx].push_back(guidePoint);
            Eigen::Vector3d diff = guidePoint - this->optData_.controlPoints.col(idx);
            this->optData_.guideDirections[idx].push_back(diff / diff.norm());
        }
        if (seg.second - seg.first - 1 == 0) {  // line collision
            this->findGuidePointSemiCircle(seg.first, seg, path, guidePoint);
            Eigen::Vector3d midPoint = (this->optData_.controlPoints.col(seg.first) + this->optData_.controlPoints.col(seg.second)) / 2.0;
            Eigen::Vector3d diff = guidePoint - midPoint;
            guideDirection = diff / diff.norm();
            for (int idx = seg.first - 1; idx <= seg.second + 1; ++idx) {
                if (idx >= bsplineDegree && idx <= N - bsplineDegree - 1) {
                    this->optData_.guidePoints[idx].push_back(guidePoint);
                    this->optData_.guideDirections[idx].push_back(guideDirection);
                }
            }
        }
    }
}

bool bsplineTraj::indexInCollisionSeg(const std::vector<std::pair<int, int>>& collisionSeg, int idx) {
    for (const auto& seg : collisionSeg)
        if (idx >= seg.first && idx <= seg.second) return true;
    return false;
}

int bsplineTraj::findCollisionSegIndex(const std::vector<std::pair<int, int>>& collisionSeg, int idx) {
    int k = 0;
    for (const auto& seg : collisionSeg) {
        if (idx >= seg.first && idx <= seg.second) return k;
        ++k;
    }
    return -1;
}

// BT.h:417-429
bool bsplineTraj::isControlPointRequireNewGuide(int controlPointIdx) {
    Eigen::Vector3d c = this->optData_.controlPoints.col(controlPointIdx);
    for (size_t i = 0; i < this->optData_.guidePoints[controlPointIdx].size(); ++i) {
        double dist = (c - this->optData_.guidePoints[controlPointIdx][i]).dot(this->optData_.guideDirections[controlPointIdx][i]);
        if (this->dthresh_ - dist > 0) return false;
    }
    return true;
}

// BT.h:379-403: the control points of the new collision segments, split into those the previous segments already held
// and the fresh ones (a segment without interior points contributes both of its ends)
void bsplineTraj::compareCollisionSeg(const std::vector<std::pair<int, int>>& prevCollisionSeg, const std::vector<std::pair<int, int>>& newCollisionSeg,
                                      std::vector<int>& newCollisionPoints, std::vector<int>& overlappedCollisionPoints) {
    for (const auto& s : newCollisionSeg) {
        for (int i = s.first + 1; i <= s.second - 1; ++i) (indexInCollisionSeg(prevCollisionSeg, i) ? overlappedCollisionPoints : newCollisionPoints).push_back(i);
        if (s.second - s.first - 1 == 0)
            for (int i = s.first; i <= s.second; ++i) (indexInCollisionSeg(prevCollisionSeg, i) ? overlappedCollisionPoints : newCollisionPoints).push_back(i);
    }
}

// BT.h:250-257
void bsplineTraj::shortcutPaths(const std::vector<std::vector<Eigen::Vector3d>>& paths, std::vector<std::vector<Eigen::Vector3d>>& pathsSC) {
    pathsSC.clear();
    for (const auto& p : paths) {
        std::vector<Eigen::Vector3d> sc;
        this->shortcutPath(p, sc);
        pathsSC.push_back(sc);
    }
}

// BT.cpp:573-608
bool bsplineTraj::isReguideRequired(std::vector<std::pair<int, int>>& reguideCollisionSeg) {
    std::vector<std::pair<int, int>> prev = this->collisionSeg_;
    this->findCollisionSeg(this->optData_.controlPoints, this->collisionSeg_);
    std::vector<int> fresh, overlapped;
    this->compareCollisionSeg(prev, this->collisionSeg_, fresh, overlapped);
    std::set<int> segIdx;
    for (int i : fresh) segIdx.insert(findCollisionSegIndex(this->collisionSeg_, i));
    for (int i : overlapped)
        if (this->isControlPointRequireNewGuide(i)) segIdx.insert(findCollisionSegIndex(this->collisionSeg_, i));
    segIdx.erase(-1);
    if (segIdx.empty()) return false;
    for (int s : segIdx) reguideCollisionSeg.push_back(this->collisionSeg_[s]);
    return true;
}

// ---- device calls ---------------------------------------------------------------------

namespace {
// flattens the planners' optData into the batch layouts of include/vigo.h
struct HostBatch {
    int B = 0, N = 0;
    std::vector<double> ctrl, gpv, obs, weights;
    std::vector<int32_t> goff, ooff;
};
}  // namespace

void bsplineTraj::solveBatch(const std::vector<bsplineTraj*>& ps) {
    // groups of equal N share a launch (vigo_optimize takes one N per call)
    std::vector<bool> doneMask(ps.size(), false);
    for (size_t a = 0; a < ps.size(); ++a) {
        if (doneMask[a]) continue;
        const int N = ps[a]->optData_.controlPoints.cols();
        std::vector<bsplineTraj*> grp;
        for (size_t b = a; b < ps.size(); ++b)
            if (!doneMask[b] && ps[a]->sameBatchKey(*ps[b])) { grp.push_back(ps[b]); doneMask[b] = true; }
        bsplineTraj* lead = grp[0];
        for (auto* p : grp) p->lastStatus_ = VIGO_ERR_HIP;
        if (N < 7 || N > VIGO_MAX_CTRL_POINTS || !lead->syncDevice()) continue;
        HostBatch hb;
        hb.B = (int)grp.size();
        hb.N = N;
        hb.goff.push_back(0);
        hb.ooff.push_back(0);
        for (auto* p : grp) {
            const double* c = p->optData_.controlPoints.data();
            hb.ctrl.insert(hb.ctrl.end(), c, c + 3 * N);
            for (int i = 0; i < N; ++i) {
                const size_t cnt = i < (int)p->optData_.guidePoints.size() ? p->optData_.guidePoints[i].size() : 0;
                for (size_t j = 0; j < cnt; ++j) {
                    const Eigen::Vector3d& g = p->optData_.guidePoints[i][j];
                    const Eigen::Vector3d& v = p->optData_.guideDirections[i][j];
                    for (int q = 0; q < 3; ++q) hb.gpv.push_back(g(q));
                    for (int q = 0; q < 3; ++q) hb.gpv.push_back(v(q));
                }
                hb.goff.push_back((int32_t)(hb.gpv.size() / 6));
            }
            for (size_t j = 0; j < p->optData_.dynamicObstaclesPos.size(); ++j) {
                for (int q = 0; q < 3; ++q) hb.obs.push_back(p->optData_.dynamicObstaclesPos[j](q));
                for (int q = 0; q < 3; ++q) hb.obs.push_back(p->optData_.dynamicObstaclesVel[j](q));
                for (int q = 0; q < 3; ++q) hb.obs.push_back(p->optData_.dynamicObstaclesSize[j](q));
            }
            hb.ooff.push_back((int32_t)(hb.obs.size() / 9));
            hb.weights.push_back(p->weightDistance_);
            hb.weights.push_back(p->weightSmoothness_);
            hb.weights.push_back(p->weightFeasibility_);
            hb.weights.push_back(p->weightDynamicObstacle_);
        }
        const size_t G = hb.gpv.size() / 6;
        // device staging buffers live across calls (one set per host thread): hipMalloc/hipFree per
        // rebound round cost more than the round's kernels
        static thread_local StagingBuf dCtrl, dGoff, dGpv, dGunk, dOoff, dObs, dW, dStatus;
        bool ok = dCtrl.upload(hb.ctrl.data(), hb.ctrl.size() * 8) && dGoff.upload(hb.goff.data(), hb.goff.size() * 4) &&
                  dGpv.upload(hb.gpv.data(), hb.gpv.size() * 8) && dGunk.alloc(G) && dOoff.upload(hb.ooff.data(), hb.ooff.size() * 4) &&
                  dObs.upload(hb.obs.data(), hb.obs.size() * 8) && dW.upload(hb.weights.data(), hb.weights.size() * 8) &&
                  dStatus.alloc(hb.B * 4);
        if (!ok) continue;
        vigo_handle_t h = lead->dev_;
        // map_->isUnknown(guidePoint), BT.cpp:841, hoisted out of the solve
        if (G && vigo_guides_unknown(h, (int64_t)G, (const double*)dGpv.p, (uint8_t*)dGunk.p) != VIGO_OK) continue;
        if (vigo_optimize(h, hb.B, N, (double*)dCtrl.p, (const int32_t*)dGoff.p, G ? (const double*)dGpv.p : nullptr,
                          G ? (const uint8_t*)dGunk.p : nullptr, (const int32_t*)dOoff.p,
                          hb.obs.empty() ? nullptr : (const double*)dObs.p, 0, (const double*)dW.p, nullptr,
                          (int32_t*)dStatus.p, nullptr, nullptr, nullptr) != VIGO_OK) {
            cout << "[BsplineTraj]: vigo_optimize failed: " << vigo_last_error(h) << endl;
            continue;
        }
        std::vector<int32_t> status(hb.B);
        if (!vigo_host::threadSync() || !dCtrl.download(hb.ctrl.data(), hb.ctrl.size() * 8) ||
            !dStatus.download(status.data(), status.size() * 4))
            continue;
        for (int b = 0; b < hb.B; ++b) {
            // optData_.controlPoints = the last evaluated point, as costFunction leaves it (BT.cpp:803)
            std::memcpy(grp[b]->optData_.controlPoints.data(), hb.ctrl.data() + (size_t)b * 3 * N, sizeof(double) * 3 * N);
            grp[b]->lastStatus_ = status[b];
        }
    }
}

void bsplineTraj::gateBatch(const std::vector<bsplineTraj*>& ps, std::vector<uint8_t>& col, std::vector<uint8_t>& dyn) {
    col.assign(ps.size(), 1);
    dyn.assign(ps.size(), 0);
    std::vector<bool> doneMask(ps.size(), false);
    for (size_t a = 0; a < ps.size(); ++a) {
        if (doneMask[a]) continue;
        const int N = ps[a]->optData_.controlPoints.cols();
        std::vector<size_t> idx;
        for (size_t b = a; b < ps.size(); ++b)
            if (!doneMask[b] && ps[a]->sameBatchKey(*ps[b])) {
                idx.push_back(b);
                doneMask[b] = true;
            }
        bsplineTraj* lead = ps[a];
        if (N < 4 || N > VIGO_MAX_CTRL_POINTS || !lead->syncDevice()) continue;
        std::vector<double> ctrl, obs;
        std::vector<int32_t> ooff{0};
        for (size_t b : idx) {
            const double* c = ps[b]->optData_.controlPoints.data();
            ctrl.insert(ctrl.end(), c, c + 3 * N);
            for (size_t j = 0; j < ps[b]->optData_.dynamicObstaclesPos.size(); ++j) {
                for (int q = 0; q < 3; ++q) obs.push_back(ps[b]->optData_.dynamicObstaclesPos[j](q));
                for (int q = 0; q < 3; ++q) obs.push_back(ps[b]->optData_.dynamicObstaclesVel[j](q));
                for (int q = 0; q < 3; ++q) obs.push_back(ps[b]->optData_.dynamicObstaclesSize[j](q));
            }
            ooff.push_back((int32_t)(obs.size() / 9));
        }
        const int B = (int)idx.size();
        static thread_local StagingBuf dCtrl, dFlag, dDyn, dOoff, dObs;
        if (!dCtrl.upload(ctrl.data(), ctrl.size() * 8) || !dFlag.alloc(B) || !dDyn.alloc(B) ||
            !dOoff.upload(ooff.data(), ooff.size() * 4) || !dObs.upload(obs.data(), obs.size() * 8))
            continue;
        const double dt = lead->map_->getRes() / lead->maxVel_ / 2.0;  // BT.h:312
        if (vigo_traj_collision(lead->dev_, B, N, (const double*)dCtrl.p, dt, (uint8_t*)dFlag.p, nullptr) != VIGO_OK) continue;
        std::vector<uint8_t> f(B), d(B, 0);
        if (!obs.empty()) {
            if (vigo_traj_dynamic_collision(lead->dev_, B, N, (const double*)dCtrl.p, dt, (const int32_t*)dOoff.p,
                                            (const double*)dObs.p, 0, (uint8_t*)dDyn.p) != VIGO_OK)
                continue;
        }
        if (!vigo_host::threadSync() || !dFlag.download(f.data(), B)) continue;
        if (!obs.empty() && !dDyn.download(d.data(), B)) continue;
        for (int b = 0; b < B; ++b) {
            col[idx[b]] = f[b];
            // BT.cpp:621-626: the dynamic gate only runs when the planner has obstacles
            dyn[idx[b]] = ps[idx[b]]->optData_.dynamicObstaclesPos.empty() ? 0 : d[b];
        }
    }
}

namespace {
std::atomic<bool> g_deviceResidentRebound{true};
}
void bsplineTraj::setDeviceResidentRebound(bool on) { g_deviceResidentRebound.store(on); }
bool bsplineTraj::deviceResidentRebound() { return g_deviceResidentRebound.load(); }

// BT.cpp:611-685 between two A* calls, on the device, for one group of planners (one batch): upload the planners'
// state, queue the rounds (vigo_rebound_rounds), bring back what the host part of the loop needs.
bool bsplineTraj::deviceRounds(const std::vector<bsplineTraj*>& grp, const std::vector<Rebound*>& rb, int maxRounds) {
    if (grp.empty()) return true;
    bsplineTraj* lead = grp[0];
    const int N = lead->optData_.controlPoints.cols();
    std::vector<int> prevStatus(grp.size());
    for (size_t k = 0; k < grp.size(); ++k) { prevStatus[k] = grp[k]->lastStatus_; grp[k]->lastStatus_ = VIGO_ERR_HIP; }
    if (N < 7 || N > VIGO_MAX_CTRL_POINTS || !lead->syncDevice()) return false;
    HostBatch hb;
    hb.B = (int)grp.size();
    hb.N = N;
    hb.goff.push_back(0);
    hb.ooff.push_back(0);
    std::vector<vigo_rebound_state_t> state(grp.size());
    for (size_t k = 0; k < grp.size(); ++k) {
        bsplineTraj* p = grp[k];
        const double* c = p->optData_.controlPoints.data();
        hb.ctrl.insert(hb.ctrl.end(), c, c + 3 * N);
        for (int i = 0; i < N; ++i) {
            const size_t cnt = i < (int)p->optData_.guidePoints.size() ? p->optData_.guidePoints[i].size() : 0;
            for (size_t j = 0; j < cnt; ++j) {
                const Eigen::Vector3d& g = p->optData_.guidePoints[i][j];
                const Eigen::Vector3d& v = p->optData_.guideDirections[i][j];
                for (int q = 0; q < 3; ++q) hb.gpv.push_back(g(q));
                for (int q = 0; q < 3; ++q) hb.gpv.push_back(v(q));
            }
            hb.goff.push_back((int32_t)(hb.gpv.size() / 6));
        }
        for (size_t j = 0; j < p->optData_.dynamicObstaclesPos.size(); ++j) {
            for (int q = 0; q < 3; ++q) hb.obs.push_back(p->optData_.dynamicObstaclesPos[j](q));
            for (int q = 0; q < 3; ++q) hb.obs.push_back(p->optData_.dynamicObstaclesVel[j](q));
            for (int q = 0; q < 3; ++q) hb.obs.push_back(p->optData_.dynamicObstaclesSize[j](q));
        }
        hb.ooff.push_back((int32_t)(hb.obs.size() / 9));
        hb.weights.push_back(p->weightDistance_);
        hb.weights.push_back(p->weightSmoothness_);
        hb.weights.push_back(p->weightFeasibility_);
        hb.weights.push_back(p->weightDynamicObstacle_);
        vigo_rebound_state_t& st = state[k];
        std::memset(&st, 0, sizeof(st));
        st.status = VIGO_RB_ACTIVE;
        st.lbfgs_status = prevStatus[k];      // kept when this call makes no optimize() for the planner
        st.solve_first = rb[k]->needOptimize ? 1 : 0;
        st.fail_count = rb[k]->failCount;
        st.n_seg = (int32_t)p->collisionSeg_.size();
        if (st.n_seg > VIGO_MAX_COLLISION_SEGS) {
            // more previous segments than the device state holds: this planner's rounds stay with the host (one gate,
            // then reboundStep) — a trajectory of VIGO_MAX_CTRL_POINTS control points can get there, a 7 m path cannot
            st.status = VIGO_RB_NEEDS_HOST;
            st.n_seg = 0;
        } else {
            for (int q = 0; q < st.n_seg; ++q) { st.seg[2 * q] = p->collisionSeg_[q].first; st.seg[2 * q + 1] = p->collisionSeg_[q].second; }
        }
    }
    const size_t G = hb.gpv.size() / 6;
    static thread_local StagingBuf dCtrl, dGoff, dGpv, dGunk, dOoff, dObs, dW, dState;
    bool ok = dCtrl.upload(hb.ctrl.data(), hb.ctrl.size() * 8) && dGoff.upload(hb.goff.data(), hb.goff.size() * 4) &&
              dGpv.upload(hb.gpv.data(), hb.gpv.size() * 8) && dGunk.alloc(G) && dOoff.upload(hb.ooff.data(), hb.ooff.size() * 4) &&
              dObs.upload(hb.obs.data(), hb.obs.size() * 8) && dW.upload(hb.weights.data(), hb.weights.size() * 8) &&
              dState.upload(state.data(), state.size() * sizeof(vigo_rebound_state_t));
    if (!ok) return false;
    vigo_handle_t h = lead->dev_;
    if (G && vigo_guides_unknown(h, (int64_t)G, (const double*)dGpv.p, (uint8_t*)dGunk.p) != VIGO_OK) return false;
    const double dt = lead->map_->getRes() / lead->maxVel_ / 2.0;  // BT.h:312
    if (vigo_rebound_rounds(h, hb.B, N, (double*)dCtrl.p, (const int32_t*)dGoff.p, G ? (const double*)dGpv.p : nullptr,
                            G ? (const uint8_t*)dGunk.p : nullptr, (const int32_t*)dOoff.p, hb.obs.empty() ? nullptr : (const double*)dObs.p, 0,
                            (double*)dW.p, dt, lead->notCheckRatio_, maxRounds, (vigo_rebound_state_t*)dState.p) != VIGO_OK) {
        cout << "[BsplineTraj]: vigo_rebound_rounds failed: " << vigo_last_error(h) << endl;
        return false;
    }
    if (!vigo_host::threadSync() || !dCtrl.download(hb.ctrl.data(), hb.ctrl.size() * 8) ||
        !dW.download(hb.weights.data(), hb.weights.size() * 8) || !dState.download(state.data(), state.size() * sizeof(vigo_rebound_state_t)))
        return false;
    std::vector<bsplineTraj*> overflow;
    std::vector<size_t> overflowAt;
    for (size_t k = 0; k < grp.size(); ++k) {
        bsplineTraj* p = grp[k];
        const vigo_rebound_state_t& st = state[k];
        if (st.rounds == 0 && st.status == VIGO_RB_NEEDS_HOST) { overflow.push_back(p); overflowAt.push_back(k); continue; }
        // optData_.controlPoints = the last evaluated point, as costFunction leaves it (BT.cpp:803)
        std::memcpy(p->optData_.controlPoints.data(), hb.ctrl.data() + k * 3 * N, sizeof(double) * 3 * N);
        p->weightDistance_ = hb.weights[4 * k + 0];
        p->weightDynamicObstacle_ = hb.weights[4 * k + 3];
        p->lastStatus_ = st.lbfgs_status;
        p->collisionSeg_.clear();
        for (int q = 0; q < st.n_seg; ++q) p->collisionSeg_.push_back({st.seg[2 * q], st.seg[2 * q + 1]});
        rb[k]->failCount = st.fail_count;
        rb[k]->needOptimize = st.solve_first != 0;     // an optimize() the device deferred to the next call
        rb[k]->devStatus = st.status;
        rb[k]->gateStatic = st.gate_static != 0;
        rb[k]->gateDynamic = st.gate_dynamic != 0;
    }
    // planners the device state cannot represent: one host-driven round (solve if owed, gate; the caller steps)
    for (size_t q = 0; q < overflow.size(); ++q) {
        std::vector<bsplineTraj*> one{overflow[q]};
        Rebound& r = *rb[overflowAt[q]];
        if (r.needOptimize) solveBatch(one);
        std::vector<uint8_t> col, dyn;
        gateBatch(one, col, dyn);
        r.gateStatic = col[0] != 0;
        r.gateDynamic = dyn[0] != 0;
        r.devStatus = (!r.gateStatic && !r.gateDynamic) ? VIGO_RB_DONE : VIGO_RB_NEEDS_HOST;
    }
    return true;
}

// BT.cpp:687-718
int bsplineTraj::optimize() {
    std::vector<bsplineTraj*> one{this};
    solveBatch(one);
    return lastStatus_;
}

// the lbfgs_evaluate_t seam on the device (BT.cpp:796-821)
double bsplineTraj::costFunction(const double* x, double* grad, const int n) {
    const int N = optData_.controlPoints.cols();
    if (n != 3 * (N - 2 * bsplineDegree) || !syncDevice()) return std::nan("");
    std::memcpy(optData_.controlPoints.data() + 3 * bsplineDegree, x, n * sizeof(double));  // BT.cpp:803
    std::vector<double> gpv, obs;
    std::vector<int32_t> goff{0};
    for (int i = 0; i < N; ++i) {
        for (size_t j = 0; j < optData_.guidePoints[i].size(); ++j) {
            for (int q = 0; q < 3; ++q) gpv.push_back(optData_.guidePoints[i][j](q));
            for (int q = 0; q < 3; ++q) gpv.push_back(optData_.guideDirections[i][j](q));
        }
        goff.push_back((int32_t)(gpv.size() / 6));
    }
    for (size_t j = 0; j < optData_.dynamicObstaclesPos.size(); ++j) {
        for (int q = 0; q < 3; ++q) obs.push_back(optData_.dynamicObstaclesPos[j](q));
        for (int q = 0; q < 3; ++q) obs.push_back(optData_.dynamicObstaclesVel[j](q));
        for (int q = 0; q < 3; ++q) obs.push_back(optData_.dynamicObstaclesSize[j](q));
    }
    const size_t G = gpv.size() / 6;
    DevBuf dCtrl, dGoff, dGpv, dGunk, dObs, dCost, dGrad;
    if (!dCtrl.upload(optData_.controlPoints.data(), 3 * N * 8) || !dGoff.upload(goff.data(), goff.size() * 4) ||
        !dGpv.upload(gpv.data(), gpv.size() * 8) || !dGunk.alloc(G) || !dObs.upload(obs.data(), obs.size() * 8) ||
        !dCost.alloc(8) || !dGrad.alloc((size_t)n * 8))
        return std::nan("");
    if (G && vigo_guides_unknown(dev_, (int64_t)G, (const double*)dGpv.p, (uint8_t*)dGunk.p) != VIGO_OK) return std::nan("");
    if (vigo_cost_grad(dev_, 1, N, (const double*)dCtrl.p, (const int32_t*)dGoff.p, G ? (const double*)dGpv.p : nullptr,
                       G ? (const uint8_t*)dGunk.p : nullptr, nullptr, obs.empty() ? nullptr : (const double*)dObs.p,
                       (int)(obs.size() / 9), nullptr, (double*)dCost.p, (double*)dGrad.p, nullptr) != VIGO_OK)
        return std::nan("");
    double cost = 0;
    if (!vigo_host::threadSync() || !dCost.download(&cost, 8) || !dGrad.download(grad, (size_t)n * 8)) return std::nan("");
    return cost;
}

// BT.cpp:796-800: the lbfgs_evaluate_t-shaped entry (instance pointer first)
double bsplineTraj::solverCostFunction(void* func_data, const double* x, double* grad, const int n) {
    return reinterpret_cast<bsplineTraj*>(func_data)->costFunction(x, grad, n);
}

// One cost term and its gradient for the given control points, evaluated by the device kernel with a
// unit weight on that term (vigo_cost_grad's per-trajectory weights).  Columns of the three fixed control
// points at either end stay zero: the reference computes and then discards them (BT.cpp:819).
bool bsplineTraj::termCost(int term, const Eigen::MatrixXd& controlPoints, double& cost, Eigen::MatrixXd& gradient) {
    cost = 0;
    const int N = controlPoints.cols();
    gradient = Eigen::MatrixXd::Zero(3, N);
    if (N < 7 || N > VIGO_MAX_CTRL_POINTS || (int)optData_.guidePoints.size() < N || !syncDevice()) return false;
    std::vector<double> gpv, obs, w(4, 0.0);
    w[term] = 1.0;
    std::vector<int32_t> goff{0};
    for (int i = 0; i < N; ++i) {
        for (size_t j = 0; j < optData_.guidePoints[i].size(); ++j) {
            for (int q = 0; q < 3; ++q) gpv.push_back(optData_.guidePoints[i][j](q));
            for (int q = 0; q < 3; ++q) gpv.push_back(optData_.guideDirections[i][j](q));
        }
        goff.push_back((int32_t)(gpv.size() / 6));
    }
    for (size_t j = 0; j < optData_.dynamicObstaclesPos.size(); ++j) {
        for (int q = 0; q < 3; ++q) obs.push_back(optData_.dynamicObstaclesPos[j](q));
        for (int q = 0; q < 3; ++q) obs.push_back(optData_.dynamicObstaclesVel[j](q));
        for (int q = 0; q < 3; ++q) obs.push_back(optData_.dynamicObstaclesSize[j](q));
    }
    const size_t G = gpv.size() / 6;
    const int n = 3 * (N - 2 * bsplineDegree);
    DevBuf dCtrl, dGoff, dGpv, dGunk, dObs, dW, dCost, dGrad;
    if (!dCtrl.upload(controlPoints.data(), 3 * N * 8) || !dGoff.upload(goff.data(), goff.size() * 4) || !dGpv.upload(gpv.data(), gpv.size() * 8) ||
        !dGunk.alloc(G) || !dObs.upload(obs.data(), obs.size() * 8) || !dW.upload(w.data(), 32) || !dCost.alloc(8) || !dGrad.alloc((size_t)n * 8))
        return false;
    if (G && vigo_guides_unknown(dev_, (int64_t)G, (const double*)dGpv.p, (uint8_t*)dGunk.p) != VIGO_OK) return false;
    if (vigo_cost_grad(dev_, 1, N, (const double*)dCtrl.p, (const int32_t*)dGoff.p, G ? (const double*)dGpv.p : nullptr,
                       G ? (const uint8_t*)dGunk.p : nullptr, nullptr, obs.empty() ? nullptr : (const double*)dObs.p, (int)(obs.size() / 9),
                       (const double*)dW.p, (double*)dCost.p, (double*)dGrad.p, nullptr) != VIGO_OK)
        return false;
    std::vector<double> g(n);
    if (!vigo_host::threadSync() || !dCost.download(&cost, 8) || !dGrad.download(g.data(), (size_t)n * 8)) return false;
    std::memcpy(gradient.data() + 3 * bsplineDegree, g.data(), (size_t)n * 8);
    return true;
}
// BT.cpp:823-932, :934-950, :952-999, :1001-1064
void bsplineTraj::getDistanceCost(const Eigen::MatrixXd& controlPoints, double& cost, Eigen::MatrixXd& gradient) { termCost(0, controlPoints, cost, gradient); }
void bsplineTraj::getSmoothnessCost(const Eigen::MatrixXd& controlPoints, double& cost, Eigen::MatrixXd& gradient) { termCost(1, controlPoints, cost, gradient); }
void bsplineTraj::getFeasibilityCost(const Eigen::MatrixXd& controlPoints, double& cost, Eigen::MatrixXd& gradient) { termCost(2, controlPoints, cost, gradient); }
void bsplineTraj::getDynamicObstacleCost(const Eigen::MatrixXd& controlPoints, double& cost, Eigen::MatrixXd& gradient) { termCost(3, controlPoints, cost, gradient); }

// BT.cpp:1464-1496: velocity / acceleration samples of the current trajectory as text files, raw and
// with the linear feasibility re-parameterisation.  (The reference fills acc_adjusted_info.txt from the
// VELOCITY spline scaled by factor^2, BT.cpp:1488 — reproduced.)
void bsplineTraj::writeCurrentTrajInfo(const std::string& filePath, double dt) {
    if (!(dt > 0)) return;
    std::ofstream velInfo(filePath + "/vel_info.txt"), accInfo(filePath + "/acc_info.txt");
    std::ofstream velAdj(filePath + "/vel_adjusted_info.txt"), accAdj(filePath + "/acc_adjusted_info.txt");
    trajPlanner::bspline vel = this->bspline_.getDerivative();
    trajPlanner::bspline acc = vel.getDerivative();
    for (double t = 0.0; t <= this->bspline_.getDuration(); t += dt) {
        const Eigen::Vector3d v = vel.at(t), a = acc.at(t);
        velInfo << t << " " << v(0) << " " << v(1) << " " << v(2) << "\n";
        accInfo << t << " " << a(0) << " " << a(1) << " " << a(2) << "\n";
    }
    const double f = this->getLinearFactor();
    for (double t = 0.0; this->getLinearReparamTime(t) <= this->bspline_.getDuration(); t += dt) {
        const Eigen::Vector3d v = vel.at(this->getLinearReparamTime(t));
        velAdj << t << " " << v(0) * f << " " << v(1) * f << " " << v(2) * f << "\n";
        accAdj << t << " " << v(0) * f * f << " " << v(1) * f * f << " " << v(2) * f * f << "\n";
    }
}

// BT.h:307-325 / :344-368 through the device gates
bool bsplineTraj::hasCollisionTrajectory(const Eigen::MatrixXd& controlPoints) {
    Eigen::MatrixXd keep = optData_.controlPoints;
    optData_.controlPoints = controlPoints;
    std::vector<bsplineTraj*> one{this};
    std::vector<uint8_t> col, dyn;
    gateBatch(one, col, dyn);
    optData_.controlPoints = keep;
    return col[0] != 0;
}

bool bsplineTraj::hasDynamicCollisionTrajectory(const Eigen::MatrixXd& controlPoints) {
    Eigen::MatrixXd keep = optData_.controlPoints;
    optData_.controlPoints = controlPoints;
    std::vector<bsplineTraj*> one{this};
    std::vector<uint8_t> col, dyn;
    gateBatch(one, col, dyn);
    optData_.controlPoints = keep;
    return dyn[0] != 0;
}

void bsplineTraj::reboundBegin(Rebound& r) {
    r = Rebound();
    r.w0 = this->weightDistance_;
    r.wo0 = this->weightDynamicObstacle_;
}

// leaving the loop: the weights the loop doubled are restored (BT.cpp:629-630, :635-636, :651-652)
void bsplineTraj::reboundFinish(Rebound& r, bool ok) {
    this->weightDistance_ = r.w0;
    this->weightDynamicObstacle_ = r.wo0;
    r.done = true;
    r.ok = ok;
}

// the body of the while loop of BT.cpp:619-681, one pass
void bsplineTraj::reboundStep(Rebound& r, bool hasCollision, bool hasDynamicCollision, bool timedOut) {
    r.needOptimize = false;
    auto finish = [&](bool ok) { this->reboundFinish(r, ok); };
    if (!hasCollision && !hasDynamicCollision) { finish(true); return; }
    if (timedOut) { cout << "[BsplineTraj]: Optimization timeout." << endl; finish(false); return; }
    std::vector<std::vector<Eigen::Vector3d>> tempAstarPaths;
    if (r.failCount >= 4) {
        std::vector<std::pair<int, int>> collisionSeg;
        this->findCollisionSeg(this->optData_.controlPoints, collisionSeg);
        if (this->pathSearch(collisionSeg, tempAstarPaths)) {
            this->astarPaths_ = tempAstarPaths;
            this->assignGuidePointsSemiCircle(tempAstarPaths, collisionSeg);
        }
    }
    if (r.failCount >= 8) { finish(false); return; }
    if (hasCollision) {
        std::vector<std::pair<int, int>> reguideCollisionSeg;
        if (this->isReguideRequired(reguideCollisionSeg)) {
            if (this->pathSearch(reguideCollisionSeg, tempAstarPaths)) {
                this->astarPaths_ = tempAstarPaths;
                this->assignGuidePointsSemiCircle(tempAstarPaths, reguideCollisionSeg);
            } else {
                this->weightDistance_ *= 2.0;
                ++r.failCount;
            }
        } else {
            this->weightDistance_ *= 2.0;
            ++r.failCount;
        }
    }
    if (hasDynamicCollision) this->weightDynamicObstacle_ *= 2.0;
    r.needOptimize = true;
}

// BT.cpp:611-685 for one planner
bool bsplineTraj::optimizeTrajectory() {
    std::vector<bsplineTraj*> one{this};
    // makePlanBatch's inner loop without the A* prologue
    Rebound r;
    reboundBegin(r);
    solveBatch(one);
    const double t0 = wallSeconds();
    while (!r.done) {
        std::vector<uint8_t> col, dyn;
        gateBatch(one, col, dyn);
        reboundStep(r, col[0] != 0, dyn[0] != 0, wallSeconds() - t0 > 0.03);
        if (!r.done && r.needOptimize) solveBatch(one);
    }
    return r.ok;
}

namespace {
using vigo_host::Companion;
std::atomic<size_t> g_pipelineThreshold{2048};
thread_local bool t_insidePipeline = false;
}  // namespace
void bsplineTraj::setBatchPipelineThreshold(size_t planners) { g_pipelineThreshold.store(planners); }

// BT.cpp:333-385 for many planners at once: host prologue per planner, then the rebound loops in
// lock-step so each optimize() round is one launch over all still-active planners.
std::vector<bool> bsplineTraj::makePlanBatch(const std::vector<bsplineTraj*>& planners) {
    const size_t P = planners.size();
    // A large batch runs as two to four pipelined parts of >= threshold / 2 planners, all but the first on companion
    // host threads with their own handles and HIP streams: while one part waits for its device rounds (a chain of
    // single-wave solves, ~10 ms per 1024 planners) the others' host work (A*, guide assignment) and device rounds
    // proceed — what a caller otherwise gets only by planning from several threads of its own.  Planners are independent
    // (per-trajectory results do not depend on which batch carries them), so the plans are those of the unsplit call.
    const size_t threshold = g_pipelineThreshold.load();
    if (!t_insidePipeline && threshold > 0 && P >= threshold && P >= 2) {
        constexpr size_t kMaxParts = 4;
        static thread_local Companion companions[kMaxParts - 1];
        const size_t per = threshold / 2 > 0 ? threshold / 2 : 1;
        const size_t parts = std::min(kMaxParts, std::max<size_t>(2, P / per));
        std::vector<std::vector<bsplineTraj*>> piece(parts);
        std::vector<std::vector<bool>> res(parts);
        for (size_t k = 0; k < parts; ++k) {
            const size_t lo = P * k / parts, hi = P * (k + 1) / parts;
            piece[k].assign(planners.begin() + lo, planners.begin() + hi);
            res[k].assign(hi - lo, false);
        }
        for (size_t k = 1; k < parts; ++k) {
            companions[k - 1].start([&piece, &res, k]() {
                t_insidePipeline = true;
                res[k] = bsplineTraj::makePlanBatch(piece[k]);
            });
        }
        t_insidePipeline = true;
        try {
            res[0] = bsplineTraj::makePlanBatch(piece[0]);
        } catch (...) {
            t_insidePipeline = false;
            for (size_t k = 1; k < parts; ++k) companions[k - 1].wait();
            throw;
        }
        t_insidePipeline = false;
        for (size_t k = 1; k < parts; ++k) companions[k - 1].wait();
        std::vector<bool> all;
        for (size_t k = 0; k < parts; ++k) all.insert(all.end(), res[k].begin(), res[k].end());
        return all;
    }
    std::vector<bool> result(P, false);
    std::vector<Rebound> rb(P);
    std::vector<bsplineTraj*> active;
    std::vector<size_t> activeIdx;
    // steps 1-3 (collision segments, A*, guide assignment) touch only the planner's own state and the
    // read-only map: the planners are spread over the host cores
    const double tp0 = wallSeconds();
    std::vector<uint8_t> prepared(P, 0);
    std::atomic<long long> nsSeg{0}, nsAstar{0}, nsGuide{0};   // (summed over the worker threads; printed with VIGO_FACADE_TIMING)
    parallelFor(P, [&](size_t i) {
        bsplineTraj* p = planners[i];
        if (!p->init_ || !p->map_) return;
        const double t0 = wallSeconds();
        p->findCollisionSeg(p->optData_.controlPoints, p->collisionSeg_);           // step 1
        const double t1 = wallSeconds();
        const bool found = p->pathSearch(p->collisionSeg_, p->astarPaths_);         // step 2
        const double t2 = wallSeconds();
        nsSeg += (long long)((t1 - t0) * 1e9);
        nsAstar += (long long)((t2 - t1) * 1e9);
        if (!found) return;
        p->assignGuidePointsSemiCircle(p->astarPaths_, p->collisionSeg_);           // step 3
        nsGuide += (long long)((wallSeconds() - t2) * 1e9);
        prepared[i] = 1;
    });
    for (size_t i = 0; i < P; ++i) {
        bsplineTraj* p = planners[i];
        if (!p->init_ || !p->map_) continue;
        if (!prepared[i]) {
            cout << "[BsplineTraj]: Fail because of A* failure." << endl;
            continue;
        }
        p->reboundBegin(rb[i]);
        active.push_back(p);
        activeIdx.push_back(i);
    }
    const double tp1 = wallSeconds();
    // step 4: rebound loops.  The 30 ms budget of BT.cpp:633 is per makePlan() call in the
    // reference; a batch keeps it per round so one slow planner cannot starve the others.
    const bool timing = getenv("VIGO_FACADE_TIMING") != nullptr;
    if (deviceResidentRebound()) {
        // The loop runs on the device between two A* calls (vigo_rebound_rounds): gates, success exit, isReguideRequired,
        // weight doubling and re-solve are queued for up to kRounds rounds without a host round trip; the host only sees
        // the planners that are done, need A* (re-guide, or failCount >= 4) or ran out of queued rounds.
        const int kRounds = 4;   // failCount reaches 4 after at most four device rounds: then every round needs A*
        const double t0 = wallSeconds();
        const double budget = 0.03 * std::max<size_t>(1, active.size());
        while (!active.empty()) {
            const double tr0 = wallSeconds();
            // one device batch per group of planners the lead's handle state fits
            std::vector<bool> grouped(active.size(), false);
            std::vector<uint8_t> devOk(active.size(), 0);
            for (size_t a = 0; a < active.size(); ++a) {
                if (grouped[a]) continue;
                std::vector<bsplineTraj*> grp;
                std::vector<Rebound*> grb;
                std::vector<size_t> members;
                for (size_t b = a; b < active.size(); ++b)
                    if (!grouped[b] && active[a]->sameBatchKey(*active[b])) {
                        grp.push_back(active[b]);
                        grb.push_back(&rb[activeIdx[b]]);
                        members.push_back(b);
                        grouped[b] = true;
                    }
                const bool ok = deviceRounds(grp, grb, kRounds);
                for (size_t m : members) devOk[m] = ok ? 1 : 0;
            }
            const double tr1 = wallSeconds();
            const bool timedOut = wallSeconds() - t0 > budget;
            if (timedOut) {
                // BT.cpp:633-637: out of time.  The reference tests the budget right after the gates: the planners still
                // in the loop get one more gate (a trajectory that is collision free by now succeeds), the rest fail.
                std::vector<bool> g2(active.size(), false);
                for (size_t a = 0; a < active.size(); ++a) {
                    if (g2[a]) continue;
                    std::vector<bsplineTraj*> grp;
                    std::vector<size_t> members;
                    for (size_t b = a; b < active.size(); ++b)
                        if (!g2[b] && active[a]->sameBatchKey(*active[b])) { grp.push_back(active[b]); members.push_back(b); g2[b] = true; }
                    std::vector<uint8_t> col, dyn;
                    gateBatch(grp, col, dyn);
                    for (size_t m = 0; m < members.size(); ++m) {
                        Rebound& r = rb[activeIdx[members[m]]];
                        if (r.devStatus == VIGO_RB_DONE) continue;
                        r.devStatus = (!col[m] && !dyn[m] && devOk[members[m]]) ? VIGO_RB_DONE : VIGO_RB_NEEDS_HOST;
                        r.gateStatic = col[m] != 0;
                        r.gateDynamic = dyn[m] != 0;
                    }
                }
            }
            size_t nHost = 0;
            for (size_t a = 0; a < active.size(); ++a) nHost += rb[activeIdx[a]].devStatus == VIGO_RB_NEEDS_HOST ? 1 : 0;
            parallelFor(active.size(), [&](size_t a) {
                Rebound& r = rb[activeIdx[a]];
                bsplineTraj* p = active[a];
                if (!devOk[a]) { p->reboundFinish(r, false); return; }             // no device: nothing to plan with
                if (r.devStatus == VIGO_RB_DONE) p->reboundFinish(r, true);          // BT.cpp:628-631
                else if (r.devStatus == VIGO_RB_NEEDS_HOST) p->reboundStep(r, r.gateStatic, r.gateDynamic, timedOut);   // A* and the rest of the pass
                // (still active: its next step is the gate, or the optimize() the device left for the next call)
            });
            std::vector<bsplineTraj*> next;
            std::vector<size_t> nextIdx;
            for (size_t a = 0; a < active.size(); ++a) {
                Rebound& r = rb[activeIdx[a]];
                if (r.done) {
                    result[activeIdx[a]] = r.ok;
                    if (!r.ok) cout << "[BsplineTraj]: Fail because of optimizer not finding a solution." << endl;
                } else {
                    next.push_back(active[a]);
                    nextIdx.push_back(activeIdx[a]);
                }
            }
            if (timing)
                cout << "[BsplineTraj]:   device rounds (<= " << kRounds << ") of " << active.size() << ": " << (tr1 - tr0) * 1e3 << " ms, host step of "
                     << nHost << " planners " << (wallSeconds() - tr1) * 1e3 << " ms, " << next.size() << " continue" << endl;
            active.swap(next);
            activeIdx.swap(nextIdx);
        }
    } else {
    double tr0 = wallSeconds();
    solveBatch(active);
    if (timing) cout << "[BsplineTraj]:   first solve of " << active.size() << ": " << (wallSeconds() - tr0) * 1e3 << " ms" << endl;
    const double t0 = wallSeconds();
    const double budget = 0.03 * std::max<size_t>(1, active.size());
    while (!active.empty()) {
        std::vector<uint8_t> col, dyn;
        tr0 = wallSeconds();
        gateBatch(active, col, dyn);
        const double tr1 = wallSeconds();
        const bool timedOut = wallSeconds() - t0 > budget;
        std::vector<bsplineTraj*> next, solve;
        std::vector<size_t> nextIdx;
        // one pass of the loop body per planner (validation outcome -> re-guide / weight doubling, BT.cpp:619-681):
        // planner-local, incl. the A* of a re-guide, so spread over the host cores
        parallelFor(active.size(), [&](size_t a) { active[a]->reboundStep(rb[activeIdx[a]], col[a] != 0, dyn[a] != 0, timedOut); });
        for (size_t a = 0; a < active.size(); ++a) {
            Rebound& r = rb[activeIdx[a]];
            if (r.done) {
                result[activeIdx[a]] = r.ok;
                if (!r.ok) cout << "[BsplineTraj]: Fail because of optimizer not finding a solution." << endl;
            } else {
                next.push_back(active[a]);
                nextIdx.push_back(activeIdx[a]);
                if (r.needOptimize) solve.push_back(active[a]);
            }
        }
        const double tr2 = wallSeconds();
        solveBatch(solve);
        if (timing)
            cout << "[BsplineTraj]:   round of " << active.size() << ": gates " << (tr1 - tr0) * 1e3 << " ms, rebound step " << (tr2 - tr1) * 1e3
                 << " ms, solve of " << solve.size() << " " << (wallSeconds() - tr2) * 1e3 << " ms" << endl;
        active.swap(next);
        activeIdx.swap(nextIdx);
    }
    }
    const double tp2 = wallSeconds();
    std::vector<uint8_t> okv(result.begin(), result.end());
    parallelFor(P, [&](size_t i) {
        if (!okv[i]) return;
        bsplineTraj* p = planners[i];
        p->bspline_ = trajPlanner::bspline(bsplineDegree, p->optData_.controlPoints, p->controlPointsTs_);  // step 5
        p->linearFeasibilityReparam();                                                                  // step 6
    });
    if (getenv("VIGO_FACADE_TIMING"))
        cout << "[BsplineTraj]: prologue CPU time summed over the workers: findCollisionSeg " << nsSeg.load() * 1e-6 << " ms, A* " << nsAstar.load() * 1e-6
             << " ms, guide assignment " << nsGuide.load() * 1e-6 << " ms" << endl;
    if (getenv("VIGO_FACADE_TIMING"))
        cout << "[BsplineTraj]: makePlanBatch of " << P << ": prologue " << (tp1 - tp0) * 1e3 << " ms, rebound loop " << (tp2 - tp1) * 1e3
             << " ms, epilogue " << (wallSeconds() - tp2) * 1e3 << " ms" << endl;
    return result;
}

// BT.cpp:754-793 — including the function-static previous goal distance shared by all instances
void bsplineTraj::adjustPathLengthDirect(const std::vector<Eigen::Vector3d>& path, std::vector<Eigen::Vector3d>& adjustedPath) {
    double prevOut = 0.0;
    this->adjustPathLengthWith(path, adjustedPath, g_prevPathLength.load(), prevOut);
    g_prevPathLength.store(prevOut);
}

void bsplineTraj::adjustPathLengthWith(const std::vector<Eigen::Vector3d>& path, std::vector<Eigen::Vector3d>& adjustedPath, double prevIn,
                                       double& prevOut) {
    const double prevPathLength = prevIn;
    prevOut = prevIn;
    if (path.empty()) return;
    double totalLength = 0.0;
    bool exceedLength = false;
    double minLength = 0.0;
    Eigen::Vector3d pStart = path[0];
    for (size_t i = 0; i + 1 < path.size(); ++i) {
        Eigen::Vector3d p1 = path[i], p2 = path[i + 1];
        totalLength = (p2 - pStart).norm();
        if (totalLength >= std::max(prevPathLength, this->maxPathLength_)) exceedLength = true;
        adjustedPath.push_back(p1);
        if (exceedLength) {
            bool free = !this->map_->isInflatedOccupiedLine(p1, p2);
            if (free && minLength >= 1.5) {
                adjustedPath.push_back(p2);
                prevOut = totalLength;
                return;
            }
        }
        if (this->map_->isInflatedOccupiedLine(p1, p2)) minLength = 0.0;
        else minLength += (p2 - p1).norm();
    }
    adjustedPath.push_back(path.back());
    prevOut = totalLength;
}

// BT.cpp:1116-1137
void bsplineTraj::linearFeasibilityReparam() {
    double trajMaxVel = 0.0, trajMaxAcc = 0.0;
    trajPlanner::bspline trajVel = this->bspline_.getDerivative();
    trajPlanner::bspline trajAcc = trajVel.getDerivative();
    for (double t = 0.0; t < this->bspline_.getDuration(); t += this->ts_) {
        trajMaxVel = std::max(trajMaxVel, trajVel.at(t).norm());
        trajMaxAcc = std::max(trajMaxAcc, trajAcc.at(t).norm());
    }
    double factorVel = this->maxVel_ / trajMaxVel;
    double factorAcc = std::sqrt(this->maxAcc_ / trajMaxAcc);
    this->linearFactor_ = std::min(factorVel, factorAcc);
}

double bsplineTraj::getLinearReparamTime(double t) { return this->linearFactor_ * t; }
double bsplineTraj::getLinearFactor() { return this->linearFactor_; }
double bsplineTraj::getInitTs() { return this->controlPointDistance_ / this->maxVel_; }
double bsplineTraj::getControlPointTs() { return this->controlPointsTs_; }
double bsplineTraj::getControlPointDist() { return this->controlPointDistance_; }
trajPlanner::bspline bsplineTraj::getTrajectory() { return this->bspline_; }

// BT.cpp:1402-1419
geometry_msgs::PoseStamped bsplineTraj::getPose(double t, bool yaw) {
    geometry_msgs::PoseStamped ps;
    Eigen::Vector3d p = this->bspline_.at(t);
    ps.header.frame_id = "map";
    ps.header.stamp = ros::Time::now();
    ps.pose.position.x = p(0);
    ps.pose.position.y = p(1);
    ps.pose.position.z = p(2);
    if (yaw) {
        trajPlanner::bspline velBspline = this->bspline_.getDerivative();
        Eigen::Vector3d vel = velBspline.at(t);
        ps.pose.orientation = trajPlanner::quaternion_from_rpy(0, 0, std::atan2(vel(1), vel(0)));
    }
    return ps;
}

double bsplineTraj::getDuration() { return this->bspline_.getDuration(); }
double bsplineTraj::getTimestep() { return this->ts_; }
Eigen::MatrixXd bsplineTraj::getControlPoints() { return this->optData_.controlPoints; }

std::vector<Eigen::Vector3d> bsplineTraj::evalTraj() { return this->evalTraj(this->map_->getRes() / this->maxVel_ / 2.0); }

std::vector<Eigen::Vector3d> bsplineTraj::evalTraj(double dt) {
    std::vector<Eigen::Vector3d> traj;
    trajPlanner::bspline sp(bsplineDegree, this->optData_.controlPoints, this->controlPointsTs_);
    for (double t = 0; t <= sp.getDuration(); t += dt) traj.push_back(sp.at(t));
    return traj;
}

bool bsplineTraj::isCurrTrajValid() {
    if (!this->init_) return false;
    return !this->hasCollisionTrajectory(this->optData_.controlPoints);
}

// BT.h:327-342
bool bsplineTraj::isCurrTrajValid(Eigen::Vector3d& firstCollisionPos) {
    if (!this->init_) return false;
    std::vector<Eigen::Vector3d> trajectory = this->evalTraj();
    for (int i = 0; i < (1.0 - this->notCheckRatio_) * int(trajectory.size()); ++i) {
        if (this->map_->isInflatedOccupied(trajectory[i])) {
            firstCollisionPos = trajectory[i];
            return false;
        }
    }
    return true;
}

nav_msgs::Path bsplineTraj::evalTrajToMsg(bool yaw) { return this->evalTrajToMsg(this->ts_, yaw); }

// BT.cpp:1502-1518
nav_msgs::Path bsplineTraj::evalTrajToMsg(double dt, bool yaw) {
    std::vector<Eigen::Vector3d> trajTemp = this->evalTraj(dt);
    nav_msgs::Path traj;
    this->eigenPointsToPathMsg(trajTemp, traj);
    trajPlanner::bspline velBspline = this->bspline_.getDerivative();
    for (size_t i = 0; i < traj.poses.size(); ++i) {
        Eigen::Vector3d vel = velBspline.at((double)i * dt);
        if (yaw) traj.poses[i].pose.orientation = trajPlanner::quaternion_from_rpy(0, 0, std::atan2(vel(1), vel(0)));
    }
    return traj;
}

void bsplineTraj::pathMsgToEigenPoints(const nav_msgs::Path& path, std::vector<Eigen::Vector3d>& points) {
    for (const auto& ps : path.poses) points.push_back(Eigen::Vector3d(ps.pose.position.x, ps.pose.position.y, ps.pose.position.z));
}

void bsplineTraj::eigenPointsToPathMsg(const std::vector<Eigen::Vector3d>& points, nav_msgs::Path& path) {
    path.poses.clear();
    for (const auto& q : points) {
        geometry_msgs::PoseStamped p;
        p.pose.position.x = q(0);
        p.pose.position.y = q(1);
        p.pose.position.z = q(2);
        path.poses.push_back(p);
    }
    path.header.frame_id = "map";
}

}  // namespace trajPlanner
