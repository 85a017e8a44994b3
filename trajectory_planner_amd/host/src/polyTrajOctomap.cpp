// polyTrajOctomap.cpp — corridor-collision facade (see the header for scope).  Behaviour follows
// polyTrajOctomap.cpp:547-689 and polyTrajSolver.cpp:1026-1137; the box sweep runs on the device
// (vigo_box_collision_points, include/vigo.h).
#include <trajectory_planner/polyTrajOctomap.h>

#include <hip/hip_runtime_api.h>

#include <cmath>
#include <iostream>

#include "../../../include/vigo.h"

using std::cout;
using std::endl;

namespace trajPlanner {

polyTrajOctomap::polyTrajOctomap() : polyTrajOctomap(ros::NodeHandle()) {}

polyTrajOctomap::polyTrajOctomap(const ros::NodeHandle& nh) : nh_(nh) {
    // PO.cpp:14-25, :45-60 keys (un-namespaced, as in the reference)
    if (!nh_.getParam("collision_box", collisionBox_) || collisionBox_.size() < 3) collisionBox_ = {0.5, 0.5, 0.5};
    if (!nh_.getParam("map_resolution", mapRes_)) mapRes_ = 0.2;
    if (!nh_.getParam("sample_delta_time", delT_)) delT_ = 0.1;
    if (!nh_.getParam("polynomial_degree", polyDegree_)) polyDegree_ = 7;
}

polyTrajOctomap::~polyTrajOctomap() {
    if (dev_) vigo_destroy(dev_);
}

void polyTrajOctomap::setMap(const std::shared_ptr<mapManager::occMap>& map) {
    map_ = map;
    mapVersion_ = 0;
}

bool polyTrajOctomap::syncDevice() {
    if (!map_) return false;
    if (!dev_ && vigo_create(&dev_, 0) != VIGO_OK) {
        cout << "[Trajectory Planner INFO]: no HIP device for the corridor checker (no CPU fallback)." << endl;
        dev_ = nullptr;
        return false;
    }
    if (mapVersion_ != map_->version) {
        const double o[3] = {map_->origin()(0), map_->origin()(1), map_->origin()(2)};
        if (vigo_set_grid_host(dev_, map_->nx(), map_->ny(), map_->nz(), o, map_->getRes(), map_->voxels().data()) != VIGO_OK) return false;
        mapVersion_ = map_->version;
    }
    return true;
}

void polyTrajOctomap::updatePath(const nav_msgs::Path& path) {
    std::vector<pose> trajPath;
    for (const auto& p : path.poses) trajPath.push_back(pose(p.pose.position.x, p.pose.position.y, p.pose.position.z));
    this->updatePath(trajPath);
}

void polyTrajOctomap::updatePath(const std::vector<pose>& path) { this->path_ = path; }
void polyTrajOctomap::updateInitVel(double vx, double vy, double vz) { initVel_.x = vx; initVel_.y = vy; initVel_.z = vz; }
void polyTrajOctomap::updateInitAcc(double ax, double ay, double az) { initAcc_.x = ax; initAcc_.y = ay; initAcc_.z = az; }

void polyTrajOctomap::setSolution(int polyDegree, const std::vector<double>& xSol, const std::vector<double>& ySol,
                                  const std::vector<double>& zSol, const std::vector<double>& timeKnot) {
    polyDegree_ = polyDegree;
    xSol_ = xSol; ySol_ = ySol; zSol_ = zSol;
    desiredTime_ = timeKnot;
}

// PS.cpp:1026-1056 (including the t == 0 -> 0.01 nudge of the yaw derivative)
pose polyTrajOctomap::getPoseAt(double t) {
    pose p;
    for (size_t i = 0; i + 1 < desiredTime_.size(); ++i) {
        const double startTime = desiredTime_[i], endTime = desiredTime_[i + 1];
        if (t >= startTime && t <= endTime) {
            t = (double)(t - startTime);
            const int c0 = (polyDegree_ + 1) * (int)i;
            double x = 0, y = 0, z = 0;
            for (int d = 0; d < polyDegree_ + 1; ++d) {
                x += xSol_[c0 + d] * std::pow(t, d);
                y += ySol_[c0 + d] * std::pow(t, d);
                z += zSol_[c0 + d] * std::pow(t, d);
            }
            if (t == 0) t = 0.01;
            double dx = 0, dy = 0;
            for (int d = 0; d < polyDegree_ + 1; ++d) {
                dx += d * xSol_[c0 + d] * std::pow(t, d - 1);
                dy += d * ySol_[c0 + d] * std::pow(t, d - 1);
            }
            p.x = x; p.y = y; p.z = z; p.yaw = std::atan2(dy, dx);
            break;
        }
    }
    return p;
}

// PS.cpp:1125-1137
void polyTrajOctomap::getTrajectory(std::vector<pose>& trajectory, double delT) {
    trajectory.clear();
    if (desiredTime_.empty()) return;
    const double endTime = desiredTime_.back();
    for (double t = 0; t < endTime; t += delT) trajectory.push_back(this->getPoseAt(t));
    if (!path_.empty()) trajectory.push_back(path_.back());
}

bool polyTrajOctomap::sweepPoints(const std::vector<pose>& pts, std::vector<uint8_t>& flags) {
    flags.assign(pts.size(), 1);
    if (pts.empty()) return true;
    if (!syncDevice()) return false;
    std::vector<double> xyz(pts.size() * 3);
    for (size_t i = 0; i < pts.size(); ++i) { xyz[3 * i] = pts[i].x; xyz[3 * i + 1] = pts[i].y; xyz[3 * i + 2] = pts[i].z; }
    void *dP = nullptr, *dF = nullptr;
    bool ok = hipMalloc(&dP, xyz.size() * 8) == hipSuccess && hipMalloc(&dF, pts.size()) == hipSuccess &&
              hipMemcpy(dP, xyz.data(), xyz.size() * 8, hipMemcpyHostToDevice) == hipSuccess;
    const double box[3] = {collisionBox_[0], collisionBox_[1], collisionBox_[2]};
    ok = ok && vigo_box_collision_points(dev_, (int64_t)pts.size(), (const double*)dP, box, mapRes_, (uint8_t*)dF) == VIGO_OK;
    ok = ok && hipDeviceSynchronize() == hipSuccess && hipMemcpy(flags.data(), dF, pts.size(), hipMemcpyDeviceToHost) == hipSuccess;
    if (dP) (void)hipFree(dP);
    if (dF) (void)hipFree(dF);
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
    const float x = (float)p.x, y = (float)p.y, z = (float)p.z;
    const double res = map_->getRes();
    const Eigen::Vector3d o = map_->origin();
    if (x < o(0) || x > o(0) + map_->nx() * res || y < o(1) || y > o(1) + map_->ny() * res || z < o(2) || z > o(2) + map_->nz() * res) return true;
    const double rf = 1.0 / res;
    const int kx = (int)std::floor(rf * (double)x) - (int)std::floor(o(0) / res + 0.5);
    const int ky = (int)std::floor(rf * (double)y) - (int)std::floor(o(1) / res + 0.5);
    const int kz = (int)std::floor(rf * (double)z) - (int)std::floor(o(2) / res + 0.5);
    if (kx < 0 || ky < 0 || kz < 0 || kx >= map_->nx() || ky >= map_->ny() || kz >= map_->nz()) return !ignoreUnknown;
    const unsigned v = map_->voxels()[((size_t)kx * map_->ny() + ky) * map_->nz() + kz];
    if (v & 2u) return !ignoreUnknown;
    return (v & 4u) != 0;
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

// PO.cpp:634-656: t accumulates delT per sample; first time knot interval containing t (inclusive)
bool polyTrajOctomap::checkCollisionTraj(const std::vector<pose>& trajectory, double delT, std::set<int>& collisionSeg) {
    collisionSeg.clear();
    std::vector<uint8_t> f;
    sweepPoints(trajectory, f);
    double t = 0;
    bool has = false;
    for (size_t k = 0; k < trajectory.size(); ++k) {
        if (f[k]) {
            has = true;
            for (size_t i = 0; i + 1 < desiredTime_.size(); ++i) {
                if (t >= desiredTime_[i] && t <= desiredTime_[i + 1]) { collisionSeg.insert((int)i); break; }
            }
        }
        t += delT;
    }
    return has;
}

// One pass of the corridor loop body (PO.cpp:430-432 / :513-515) on the installed solution.
void polyTrajOctomap::makePlan(std::vector<pose>& trajectory, double delT) {
    this->findValidTraj_ = false;
    if (this->path_.size() == 1) { trajectory = this->path_; this->findValidTraj_ = true; return; }
    if (desiredTime_.size() < 2) {
        cout << "[Trajectory Planner INFO]: no min-snap solution installed (the QP is outside this round's scope)." << endl;
        return;
    }
    this->getTrajectory(trajectory, delT);
    std::set<int> collisionSeg;
    this->findValidTraj_ = !this->checkCollisionTraj(trajectory, delT, collisionSeg);
    if (this->findValidTraj_) cout << "[Trajectory Planner INFO]: Found valid trajectory!" << endl;
    else cout << "[Trajectory Planner INFO]: " << collisionSeg.size() << " colliding segment(s)." << endl;
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
    geometry_msgs::PoseStamped ps;
    pose p = this->getPoseAt(t);
    ps.pose.position.x = p.x; ps.pose.position.y = p.y; ps.pose.position.z = p.z;
    ps.pose.orientation = quaternion_from_rpy(0, 0, p.yaw);
    ps.header.frame_id = "map";
    return ps;
}

double polyTrajOctomap::getDuration() {
    if (this->path_.size() == 1 || desiredTime_.empty()) return 0.0;
    return desiredTime_.back();
}

}  // namespace trajPlanner
