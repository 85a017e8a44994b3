/*
 * compat.h — the few Eigen / ROS message / map_manager types the planner facades expose.
 *
 * When the real headers are on the include path (a catkin workspace) define VIGO_WITH_ROS and
 * they are used as they are: the facade classes then have exactly the reference's public
 * signatures (bsplineTraj.h:87-158, polyTrajOctomap.h:60-135).  This image and the GPU box have
 * neither Eigen nor ROS, so for the in-tree build and its tests the minimal value types below
 * stand in — same member names, only what the facades touch.  They are NOT used to build any
 * reference source.
 */
#ifndef TRAJECTORY_PLANNER_COMPAT_H
#define TRAJECTORY_PLANNER_COMPAT_H

#ifdef VIGO_WITH_ROS
#include <ros/ros.h>
#include <Eigen/Eigen>
#include <nav_msgs/Path.h>
#include <geometry_msgs/PoseStamped.h>
#include <map_manager/occupancyMap.h>
#else

#include <cmath>
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace Eigen {
struct Vector3d {
    double v[3];
    Vector3d() : v{0, 0, 0} {}
    Vector3d(double x, double y, double z) : v{x, y, z} {}
    double& operator()(int i) { return v[i]; }
    double operator()(int i) const { return v[i]; }
    double& operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
    Vector3d operator+(const Vector3d& o) const { return {v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]}; }
    Vector3d operator-(const Vector3d& o) const { return {v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]}; }
    Vector3d operator-() const { return {-v[0], -v[1], -v[2]}; }
    Vector3d operator*(double s) const { return {v[0] * s, v[1] * s, v[2] * s}; }
    Vector3d operator/(double s) const { return {v[0] / s, v[1] / s, v[2] / s}; }
    Vector3d& operator+=(const Vector3d& o) { v[0] += o.v[0]; v[1] += o.v[1]; v[2] += o.v[2]; return *this; }
    double dot(const Vector3d& o) const { return (v[0] * o.v[0] + v[1] * o.v[1]) + v[2] * o.v[2]; }
    double squaredNorm() const { return dot(*this); }
    double norm() const { return std::sqrt(squaredNorm()); }
    Vector3d cross(const Vector3d& o) const {
        return {v[1] * o.v[2] - v[2] * o.v[1], v[2] * o.v[0] - v[0] * o.v[2], v[0] * o.v[1] - v[1] * o.v[0]};
    }
    Vector3d normalized() const { return *this / norm(); }
};
inline Vector3d operator*(double s, const Vector3d& a) { return a * s; }

struct Vector3i {
    int v[3];
    Vector3i() : v{0, 0, 0} {}
    Vector3i(int x, int y, int z) : v{x, y, z} {}
    int& operator()(int i) { return v[i]; }
    int operator()(int i) const { return v[i]; }
};

/* dynamic column-major matrix; the facades only use 3 x N */
class MatrixXd {
public:
    MatrixXd() : r_(0), c_(0) {}
    MatrixXd(int r, int c) : r_(r), c_(c), d_((size_t)r * c, 0.0) {}
    void resize(int r, int c) { r_ = r; c_ = c; d_.assign((size_t)r * c, 0.0); }
    int rows() const { return r_; }
    int cols() const { return c_; }
    double& operator()(int r, int c) { return d_[(size_t)c * r_ + r]; }
    double operator()(int r, int c) const { return d_[(size_t)c * r_ + r]; }
    double* data() { return d_.data(); }
    const double* data() const { return d_.data(); }
    Vector3d col(int c) const { return Vector3d((*this)(0, c), (*this)(1, c), (*this)(2, c)); }
    void setCol(int c, const Vector3d& p) { (*this)(0, c) = p(0); (*this)(1, c) = p(1); (*this)(2, c) = p(2); }
private:
    int r_, c_;
    std::vector<double> d_;
};
}  // namespace Eigen

namespace ros {
struct Time {
    double sec = 0;
    static Time now();
    double toSec() const { return sec; }
    Time operator-(const Time& o) const { Time t; t.sec = sec - o.sec; return t; }
};
/* parameter-server stand-in: a flat key -> value map filled by the embedding program */
class NodeHandle {
public:
    std::shared_ptr<std::map<std::string, std::vector<double>>> params =
        std::make_shared<std::map<std::string, std::vector<double>>>();
    void setParam(const std::string& k, double v) { (*params)[k] = {v}; }
    void setParam(const std::string& k, const std::vector<double>& v) { (*params)[k] = v; }
    bool getParam(const std::string& k, double& out) const {
        auto it = params->find(k);
        if (it == params->end() || it->second.empty()) return false;
        out = it->second[0];
        return true;
    }
    bool getParam(const std::string& k, bool& out) const {
        double d;
        if (!getParam(k, d)) return false;
        out = d != 0.0;
        return true;
    }
    bool getParam(const std::string& k, int& out) const {
        double d;
        if (!getParam(k, d)) return false;
        out = (int)d;
        return true;
    }
    bool getParam(const std::string& k, std::vector<double>& out) const {
        auto it = params->find(k);
        if (it == params->end()) return false;
        out = it->second;
        return true;
    }
};
inline bool ok() { return true; }
}  // namespace ros

namespace std_msgs { struct Header { std::string frame_id; ros::Time stamp; }; }
namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
struct PoseStamped { std_msgs::Header header; Pose pose; };
}  // namespace geometry_msgs
namespace nav_msgs { struct Path { std_msgs::Header header; std::vector<geometry_msgs::PoseStamped> poses; }; }

namespace mapManager {
/*
 * Dense stand-in for mapManager::occMap (external package map_manager, not vendored by the
 * reference).  Contract = include/vigo.h "voxel map": byte per voxel, bit0 inflated-occupied,
 * bit1 unknown, bit2 occupied; index = floor((p - origin)/res); outside => occupied and unknown.
 */
class occMap {
public:
    occMap(int nx, int ny, int nz, const Eigen::Vector3d& origin, double res)
        : nx_(nx), ny_(ny), nz_(nz), origin_(origin), res_(res), vox_((size_t)nx * ny * nz, 0) {}
    double getRes() const { return res_; }
    int nx() const { return nx_; }
    int ny() const { return ny_; }
    int nz() const { return nz_; }
    const Eigen::Vector3d& origin() const { return origin_; }
    std::vector<uint8_t>& voxels() { return vox_; }
    const std::vector<uint8_t>& voxels() const { return vox_; }
    uint8_t& at(int ix, int iy, int iz) { return vox_[((size_t)ix * ny_ + iy) * nz_ + iz]; }
    unsigned byteAt(const Eigen::Vector3d& p) const {
        int ix = (int)std::floor((p(0) - origin_(0)) / res_);
        int iy = (int)std::floor((p(1) - origin_(1)) / res_);
        int iz = (int)std::floor((p(2) - origin_(2)) / res_);
        if (ix < 0 || iy < 0 || iz < 0 || ix >= nx_ || iy >= ny_ || iz >= nz_) return 0xFFu;
        return vox_[((size_t)ix * ny_ + iy) * nz_ + iz];
    }
    bool isInflatedOccupied(const Eigen::Vector3d& p) const { return byteAt(p) & 1u; }
    bool isUnknown(const Eigen::Vector3d& p) const { return (byteAt(p) >> 1) & 1u; }
    bool isInflatedOccupiedLine(const Eigen::Vector3d& p1, const Eigen::Vector3d& p2) const {
        if (isInflatedOccupied(p1) || isInflatedOccupied(p2)) return true;
        Eigen::Vector3d diff = p2 - p1;
        double dist = diff.norm();
        Eigen::Vector3d inc(diff(0) / dist * res_, diff(1) / dist * res_, diff(2) / dist * res_);
        int steps = (int)(dist / res_);
        for (int i = 1; i < steps; ++i) {
            Eigen::Vector3d q(p1(0) + i * inc(0), p1(1) + i * inc(1), p1(2) + i * inc(2));
            if (isInflatedOccupied(q)) return true;
        }
        return false;
    }
    /* bumped by the owner whenever voxels() changes, so planners re-snapshot */
    uint64_t version = 1;
private:
    int nx_, ny_, nz_;
    Eigen::Vector3d origin_;
    double res_;
    std::vector<uint8_t> vox_;
};
}  // namespace mapManager

#endif /* VIGO_WITH_ROS */
#endif
