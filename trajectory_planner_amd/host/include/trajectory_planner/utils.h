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

inline double angleBetweenVectors(const Eigen::Vector3d& a, const Eigen::Vector3d& b) {
    return std::atan2(a.cross(b).norm(), a.dot(b));
}
}  // namespace trajPlanner
#endif
