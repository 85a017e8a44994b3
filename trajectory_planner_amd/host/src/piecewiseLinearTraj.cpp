// piecewiseLinearTraj.cpp — trajPlanner::pwlTraj (see the header; PW = the reference's piecewiseLinearTraj.cpp).
#include <trajectory_planner/piecewiseLinearTraj.h>

#include <cmath>

namespace trajPlanner {

pwlTraj::pwlTraj(const ros::NodeHandle& nh) : nh_(nh) {}

namespace {
// PW.cpp:12-27, :45-61: positions, and the caller's yaw only when it is to be kept
std::vector<pose> posesOf(const nav_msgs::Path& path, bool useYaw) {
    std::vector<pose> out;
    for (const geometry_msgs::PoseStamped& p : path.poses)
        out.push_back(useYaw ? pose(p.pose.position.x, p.pose.position.y, p.pose.position.z, rpy_from_quaternion(p.pose.orientation))
                             : pose(p.pose.position.x, p.pose.position.y, p.pose.position.z));
    return out;
}

// PW.cpp:31-41, :66-76: every point looks along the leg that leaves it, the last one along the leg that reaches it
void headAlongLegs(std::vector<pose>& path) {
    double yaw = 0.0;
    for (size_t i = 0; i + 1 < path.size(); ++i) {
        yaw = std::atan2(path[i + 1].y - path[i].y, path[i + 1].x - path[i].x);
        path[i].yaw = yaw;
    }
    if (!path.empty()) path.back().yaw = yaw;
}
}  // namespace

void pwlTraj::updatePath(const nav_msgs::Path& path, bool useYaw) { this->updatePath(posesOf(path, useYaw), useYaw); }
void pwlTraj::updatePath(const nav_msgs::Path& path, double desiredVel, bool useYaw) { this->updatePath(posesOf(path, useYaw), desiredVel, useYaw); }

void pwlTraj::updatePath(const std::vector<pose>& path, bool useYaw) {
    path_ = path;
    if (!useYaw) headAlongLegs(path_);
    this->avgTimeAllocation(useYaw);
}

void pwlTraj::updatePath(const std::vector<pose>& path, double desiredVel, bool useYaw) {
    path_ = path;
    if (!useYaw) headAlongLegs(path_);
    this->avgTimeAllocation(desiredVel, useYaw);
}

void pwlTraj::avgTimeAllocation(bool useYaw) { this->avgTimeAllocation(desiredVel_, useYaw); }   // PW.cpp:83-121 == :123-161 with the member

// PW.cpp:123-161: knots = [0, leg 0, turn 1, leg 1, turn 2, leg 2, ...(, final turn when the yaws are the caller's)]
void pwlTraj::avgTimeAllocation(double desiredVel, bool useYaw) {
    double totalTime = 0;
    desiredTime_.clear();
    for (size_t i = 0; i + 1 < path_.size(); ++i) {
        if (i != 0) totalTime += (double)getYawDistance(path_[i - 1], path_[i]) / desiredAngularVel_;   // turn, then move
        desiredTime_.push_back(totalTime);
        totalTime += (double)getPoseDistance(path_[i], path_[i + 1]) / desiredVel;
        desiredTime_.push_back(totalTime);
    }
    if (useYaw && path_.size() >= 2) {
        const size_t last = path_.size() - 1;
        totalTime += (double)getYawDistance(path_[last - 1], path_[last]) / desiredAngularVel_;
        desiredTime_.push_back(totalTime);
    }
}

// PW.cpp:163-173
void pwlTraj::makePlan(nav_msgs::Path& trajectory, double delT) {
    std::vector<geometry_msgs::PoseStamped> v;
    if (!desiredTime_.empty()) {
        for (double t = 0; t < desiredTime_.back(); t += delT) v.push_back(this->getPose(t));
        v.push_back(this->getPose(desiredTime_.back()));
    }
    trajectory.poses = v;
    trajectory.header.frame_id = "map";
}

// PW.cpp:175-197: the yaw goes through the quaternion and back (rpy_from_quaternion of quaternion_from_rpy)
void pwlTraj::makePlan(std::vector<pose>& trajectory, double delT) {
    trajectory.clear();
    if (desiredTime_.empty()) return;
    auto sample = [&](double t) {
        const geometry_msgs::PoseStamped ps = this->getPose(t);
        trajectory.push_back(pose(ps.pose.position.x, ps.pose.position.y, ps.pose.position.z, rpy_from_quaternion(ps.pose.orientation)));
    };
    for (double t = 0; t < desiredTime_.back(); t += delT) sample(t);
    sample(desiredTime_.back());
}

// PW.cpp:199-277
geometry_msgs::PoseStamped pwlTraj::getPose(double t) {
    geometry_msgs::PoseStamped ps;
    ps.header.frame_id = "map";
    ps.header.stamp = ros::Time::now();
    if (path_.empty()) return ps;
    if (t >= this->getDuration()) {
        const pose& lastP = path_.back();
        ps.pose.position.x = lastP.x; ps.pose.position.y = lastP.y; ps.pose.position.z = lastP.z;
        ps.pose.orientation = quaternion_from_rpy(0, 0, lastP.yaw);
        return ps;
    }
    for (size_t i = 0; i + 1 < desiredTime_.size(); ++i) {
        const double startTime = desiredTime_[i], endTime = desiredTime_[i + 1];
        if (!(t >= startTime && t <= endTime)) continue;
        if (i % 2 == 1) {   // a turn: standing on the point the next leg leaves from
            const size_t pointIdx = (i - 1) / 2;
            const pose pCurr = path_[pointIdx], pTarget = path_[pointIdx + 1];
            const double yawDiff = pTarget.yaw - pCurr.yaw;
            double direction = 1.0, yawDiffAbs = std::abs(yawDiff);
            if (yawDiffAbs <= PI_const && yawDiff >= 0) direction = 1.0;
            else if (yawDiffAbs <= PI_const && yawDiff < 0) direction = -1.0;
            else if (yawDiffAbs > PI_const && yawDiff >= 0) { direction = -1.0; yawDiffAbs = 2 * PI_const - yawDiffAbs; }
            else if (yawDiffAbs > PI_const && yawDiff < 0) { direction = 1.0; yawDiffAbs = 2 * PI_const - yawDiffAbs; }
            ps.pose.position.x = pTarget.x; ps.pose.position.y = pTarget.y; ps.pose.position.z = pTarget.z;
            const double currYaw = pCurr.yaw + direction * (t - startTime) / (endTime - startTime) * yawDiffAbs;
            ps.pose.orientation = quaternion_from_rpy(0, 0, currYaw);
        } else {            // a leg
            const size_t pointIdx = i / 2;
            const pose pCurr = path_[pointIdx], pTarget = path_[pointIdx + 1];
            if (endTime - startTime < 1e-3) {
                ps.pose.position.x = pCurr.x; ps.pose.position.y = pCurr.y; ps.pose.position.z = pCurr.z;
            } else {
                ps.pose.position.x = pCurr.x + (t - startTime) * (pTarget.x - pCurr.x) / (endTime - startTime);
                ps.pose.position.y = pCurr.y + (t - startTime) * (pTarget.y - pCurr.y) / (endTime - startTime);
                ps.pose.position.z = pCurr.z + (t - startTime) * (pTarget.z - pCurr.z) / (endTime - startTime);
            }
            ps.pose.orientation = quaternion_from_rpy(0, 0, pCurr.yaw);
        }
        break;
    }
    return ps;
}

std::vector<double> pwlTraj::getTimeKnot() { return desiredTime_; }
double pwlTraj::getDuration() { return desiredTime_.empty() ? -1.0 : desiredTime_.back(); }   // PW.cpp:283-290
double pwlTraj::getDesiredVel() { return desiredVel_; }
double pwlTraj::getDesiredAngularVel() { return desiredAngularVel_; }

geometry_msgs::PoseStamped pwlTraj::getFirstPose() {   // PW.cpp:300-314
    geometry_msgs::PoseStamped ps;
    if (path_.empty()) return ps;
    ps.pose.position.x = path_[0].x; ps.pose.position.y = path_[0].y; ps.pose.position.z = path_[0].z;
    ps.pose.orientation = quaternion_from_rpy(0, 0, path_[0].yaw);
    return ps;
}

}  // namespace trajPlanner
