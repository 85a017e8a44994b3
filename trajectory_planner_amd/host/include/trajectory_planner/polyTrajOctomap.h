/*
 * polyTrajOctomap.h — trajPlanner::polyTrajOctomap: the corridor-collision side of the
 * reference's min-snap planner (include/trajectory_planner/polyTrajOctomap.h:60-135) over the
 * MI355X back-end.
 *
 * In scope: checkCollision / checkCollisionPoint / checkCollisionLine /
 * checkCollisionTraj (PO.cpp:547-656), the polynomial sampler getPose / getTrajectory
 * (PS.cpp:1026-1056, :1125-1137), getDuration, updatePath, updateInitVel/Acc.
 * NOT in this round: the min-snap QP itself (polyTrajSolver + OSQP, SURVEY.md §8f "next" #3) —
 * the piecewise polynomial is handed in with setSolution(); makePlan() then runs the reference's
 * corridor loop body on it (sample -> device box sweep -> colliding segments) and reports
 * whether the trajectory is valid.  The map comes from a dense mapManager::occMap stand-in
 * instead of the /octomap_binary service (PO.cpp:133-145).
 */
#ifndef POLYTRAJOCTOMAP_H
#define POLYTRAJOCTOMAP_H
#include <trajectory_planner/compat.h>
#include <trajectory_planner/utils.h>

#include <memory>
#include <set>
#include <vector>

struct vigo_context;

namespace trajPlanner {
class polyTrajOctomap {
private:
    ros::NodeHandle nh_;
    std::vector<double> collisionBox_;  // collision_box, PO.cpp:14-25
    double mapRes_;                     // map_resolution
    double delT_;                       // sample_delta_time
    int polyDegree_;
    std::vector<pose> path_;
    std::vector<double> xSol_, ySol_, zSol_;  // (polyDegree_+1) coefficients per segment, local time
    std::vector<double> desiredTime_;         // time knots
    bool findValidTraj_ = false;
    geometry_msgs::Point initVel_, initAcc_;
    std::shared_ptr<mapManager::occMap> map_;
    vigo_context* dev_ = nullptr;
    uint64_t mapVersion_ = 0;
    bool syncDevice();
    bool sweepPoints(const std::vector<pose>& pts, std::vector<uint8_t>& flags);

public:
    polyTrajOctomap();
    polyTrajOctomap(const ros::NodeHandle& nh);
    ~polyTrajOctomap();
    polyTrajOctomap(const polyTrajOctomap&) = delete;
    polyTrajOctomap& operator=(const polyTrajOctomap&) = delete;

    void setMap(const std::shared_ptr<mapManager::occMap>& map);   // replaces updateMap(), PO.cpp:133-145
    void updatePath(const nav_msgs::Path& path);
    void updatePath(const std::vector<pose>& path);
    void updateInitVel(double vx, double vy, double vz);
    void updateInitAcc(double ax, double ay, double az);
    /* the min-snap solution: coefficient blocks of (degree+1) per segment and axis, time knots */
    void setSolution(int polyDegree, const std::vector<double>& xSol, const std::vector<double>& ySol,
                     const std::vector<double>& zSol, const std::vector<double>& timeKnot);

    void makePlan();
    void makePlan(nav_msgs::Path& trajectory, double delT);
    void makePlan(std::vector<pose>& trajectory, double delT);

    bool checkCollision(const pose& p);                                         // box sweep, PO.cpp:547-568
    bool checkCollisionPoint(const pose& p, bool ignoreUnknown = false);        // PO.cpp:571-595
    bool checkCollisionTraj(const std::vector<pose>& trajectory, std::vector<int>& collisionIdx);           // PO.cpp:619-632
    bool checkCollisionTraj(const std::vector<pose>& trajectory, double delT, std::set<int>& collisionSeg);  // PO.cpp:634-656

    pose getPoseAt(double t);                                       // polyTrajSolver::getPose, PS.cpp:1026-1056
    void getTrajectory(std::vector<pose>& trajectory, double delT);  // PS.cpp:1125-1137
    geometry_msgs::PoseStamped getPose(double t);                   // PO.cpp:658-677
    double getDuration();                                           // PO.cpp:679-689
    bool isValid() const { return findValidTraj_; }
    void trajMsgConverter(const std::vector<pose>& trajectoryTemp, nav_msgs::Path& trajectory);
};
}  // namespace trajPlanner
#endif
