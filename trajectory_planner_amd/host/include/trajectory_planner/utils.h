/* utils.h — the pieces of the reference's utils.h the facades need (utils.h:19-52, :83-86):
 * pose, the TRUNCATED PI_const, quaternion_from_rpy with its yaw wrap, angleBetweenVectors. */
#ifndef TRAJECTORYPLANNERUTILS_H
#define TRAJECTORYPLANNERUTILS_H
#include <trajectory_planner/compat.h>
#include <cmath>
#include <iostream>

namespace trajPlanner {
const double PI_const = 3.1415926;  /* utils.h:19 — truncated on purpose (parity) */

struct pose {
    double x, y, z, yaw;
    pose() : x(0), y(0), z(0), yaw(0) {}
    pose(double _x, double _y, double _z) : x(_x), y(_y), z(_z), yaw(0) {}
    pose(double _x, double _y, double _z, double _yaw) : x(_x), y(_y), z(_z), yaw(_yaw) {}
};

inline std::ostream& operator<<(std::ostream& os, pose& p) {
    os << "pose: (" << p.x << " " << p.y << " " << p.z << " " << p.yaw << ")";
    return os;
}

/* tf2::Quaternion::setRPY */
inline geometry_msgs::Quaternion quaternion_from_rpy(double roll, double pitch, double yaw) {
    if (yaw > PI_const) yaw = yaw - 2 * PI_const;
    const double hr = roll * 0.5, hp = pitch * 0.5, hy = yaw * 0.5;
    const double cr = std::cos(hr), sr = std::sin(hr), cp = std::cos(hp), sp = std::sin(hp), cy = std::cos(hy), sy = std::sin(hy);
    geometry_msgs::Quaternion q;
    q.x = sr * cp * cy - cr * sp * sy;
    q.y = cr * sp * cy + sr * cp * sy;
    q.z = cr * cp * sy - sr * sp * cy;
    q.w = cr * cp * cy + sr * sp * sy;
    return q;
}

/* utils.h:54-67: tf2::Matrix3x3(q).getRPY.  tf2 is an external dependency of the reference (not vendored): its published
 * Matrix3x3::setRotation + getEulerYPR (first solution) are restated — "parity unpinned" by reference outputs. */
inline void rpy_from_quaternion(const geometry_msgs::Quaternion& quat, double& roll, double& pitch, double& yaw) {
    const double d = quat.x * quat.x + quat.y * quat.y + quat.z * quat.z + quat.w * quat.w;
    const double s = 2.0 / d;
    const double xs = quat.x * s, ys = quat.y * s, zs = quat.z * s;
    const double wx = quat.w * xs, wy = quat.w * ys, wz = quat.w * zs;
    const double xx = quat.x * xs, xy = quat.x * ys, xz = quat.x * zs;
    const double yy = quat.y * ys, yz = quat.y * zs, zz = quat.z * zs;
    const double m00 = 1.0 - (yy + zz), m10 = xy + wz, m20 = xz - wy, m21 = yz + wx, m22 = 1.0 - (xx + yy);
    if (std::fabs(m20) >= 1) {          // gimbal lock (never reached by the planners' yaw-only quaternions): yaw = 0
        yaw = 0;
        roll = std::atan2(m21, m22);
        pitch = (m20 < 0 ? 1.0 : -1.0) * (3.14159265358979323846 / 2.0);
    } else {
        pitch = -std::asin(m20);
        roll = std::atan2(m21 / std::cos(pitch), m22 / std::cos(pitch));
        yaw = std::atan2(m10 / std::cos(pitch), m00 / std::cos(pitch));
    }
}
inline double rpy_from_quaternion(const geometry_msgs::Quaternion& quat) {
    double roll, pitch, yaw;
    rpy_from_quaternion(quat, roll, pitch, yaw);
    return yaw;
}

/* utils.h:69-82 */
inline double getPoseDistance(const pose& p1, const pose& p2) {
    return std::sqrt(std::pow((p1.x - p2.x), 2) + std::pow((p1.y - p2.y), 2) + std::pow((p1.z - p2.z), 2));
}
inline double getYawDistance(const pose& pStart, const pose& pTarget) {
    double delta = std::abs(pTarget.yaw - pStart.yaw);
    if (delta > PI_const) delta = 2 * PI_const - delta;
    return delta;
}

inline double angleBetweenVectors(const Eigen::Vector3d& a, const Eigen::Vector3d& b) {
    return std::atan2(a.cross(b).norm(), a.dot(b));
}
}  // namespace trajPlanner
#endif
