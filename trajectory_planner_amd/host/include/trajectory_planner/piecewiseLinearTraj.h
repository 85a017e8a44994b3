/*
 * piecewiseLinearTraj.h — trajPlanner::pwlTraj, the rotate-then-move trajectory polyTrajOctomap falls back to when the
 * polynomial planner finds no collision-free plan (PO.cpp:308-318, :373-383, :528-541; getPose / getDuration :672-674,
 * :686-688).  Interface and behaviour of the reference's piecewiseLinearTraj.{h,cpp} (PW below): a rotation period in
 * front of every leg but the first at desiredAngularVel_, the leg itself at desiredVel_ — both fixed at the class
 * defaults 0.5 rad/s and 1.0 m/s, the constructor reads no parameter (PW.cpp:9) —, yaw of a leg = its heading unless
 * the caller's yaws are kept (useYaw).  Host only.  (adjustHeading is declared by the reference, PW.h:33-34, and
 * defined nowhere: not carried.)
 */
#ifndef PIECEWISELINEARTRAJ_H
#define PIECEWISELINEARTRAJ_H
#include <trajectory_planner/compat.h>
#include <trajectory_planner/utils.h>

#include <vector>

namespace trajPlanner {
class pwlTraj {
private:
    ros::NodeHandle nh_;
    double desiredVel_ = 1.0;
    double desiredAngularVel_ = 0.5;
    std::vector<pose> path_;
    std::vector<double> desiredTime_;

public:
    pwlTraj(const ros::NodeHandle& nh);
    void updatePath(const nav_msgs::Path& path, bool useYaw = false);
    void updatePath(const std::vector<pose>& path, bool useYaw = false);
    void updatePath(const nav_msgs::Path& path, double desiredVel, bool useYaw = false);
    void updatePath(const std::vector<pose>& path, double desiredVel, bool useYaw = false);
    void avgTimeAllocation(bool useYaw = false);
    void avgTimeAllocation(double desiredVel, bool useYaw = false);

    void makePlan(nav_msgs::Path& trajectory, double delT);
    void makePlan(std::vector<pose>& trajectory, double delT);

    geometry_msgs::PoseStamped getPose(double t);
    std::vector<double> getTimeKnot();
    double getDuration();
    double getDesiredVel();
    double getDesiredAngularVel();
    geometry_msgs::PoseStamped getFirstPose();
};
}  // namespace trajPlanner
#endif
