/*
 * mini_ros.h — ros::NodeHandle (getParam / setParam on a flat key -> value map), ros::Time::now() and the message
 * structs the facades' signatures name (nav_msgs::Path, geometry_msgs::PoseStamped), for builds without ROS.
 * Same member names as the real types; only what the facades touch.
 */
#ifndef TRAJECTORY_PLANNER_MINI_ROS_H
#define TRAJECTORY_PLANNER_MINI_ROS_H
#include <map>
#include <memory>
#include <string>
#include <vector>

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
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Twist { Vector3 linear, angular; };
}  // namespace geometry_msgs
namespace nav_msgs { struct Path { std_msgs::Header header; std::vector<geometry_msgs::PoseStamped> poses; }; }
#endif
