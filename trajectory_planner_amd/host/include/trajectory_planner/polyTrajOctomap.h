/*
 * polyTrajOctomap.h — trajPlanner::polyTrajOctomap with the reference's public interface
 * (include/trajectory_planner/polyTrajOctomap.h:60-135) over the MI355X back-end.
 *
 * makePlan(): waypoint path -> min-snap QP (polyTrajSolver, host) -> sampled trajectory -> box-sweep
 * collision check of every sample on the DEVICE (vigo_box_collision_points, the batched
 * checkCollision of PO.cpp:547-589) -> shrink the colliding segments' corridors or insert
 * waypoints -> repeat (PO.cpp:259-545); piecewise-linear fallback when no valid trajectory is found.
 * Differences to the reference: the map arrives through setMap() (a dense mapManager::occMap, e.g.
 * from loadOctomapBt) instead of the /octomap_binary service (PO.cpp:133-145); no RViz publisher
 * threads; in the adding-waypoint mode the solver's path IS refreshed after insertWaypoint().
 */
#ifndef POLYTRAJOCTOMAP_H
#define POLYTRAJOCTOMAP_H
#include <trajectory_planner/compat.h>
#include <trajectory_planner/mapAdapter.h>
#include <trajectory_planner/piecewiseLinearTraj.h>
#include <trajectory_planner/polyTrajSolver.h>
#include <trajectory_planner/utils.h>

#include <memory>
#include <ostream>
#include <set>
#include <vector>

struct vigo_context;

namespace trajPlanner {
class polyTrajOctomap {
private:
    ros::NodeHandle nh_;
    std::vector<double> collisionBox_;  // collision_box, PO.cpp:14-25
    double mapRes_, delT_, desiredVel_, timeout_, initR_, fs_, corridorRes_;
    bool softConstraint_ = false;             /* PO.cpp:98-107: yaml soft_constraint / constraint_radius */
    double softConstraintRadius_ = 0.5;
    int polyDegree_, diffDegree_, continuityDegree_, maxIter_;
    bool mode_;                         // true: adding waypoints, false: corridor constraint
    std::vector<pose> path_;
    std::unique_ptr<polyTrajSolver> trajSolver_;
    // an externally supplied piecewise polynomial (setSolution) or the PWL fallback
    int extDegree_ = 0;
    std::vector<double> xSol_, ySol_, zSol_, extKnots_;
    std::unique_ptr<pwlTraj> pwlTrajSolver_;   // the fallback of PO.cpp:308-318: rotate-then-move along the waypoints
    std::vector<double> pwlKnots_;             // its time knots (what timeKnots() hands out while it is the plan)
    bool findValidTraj_ = false;
    double initVel_[3] = {0, 0, 0}, initAcc_[3] = {0, 0, 0};
    std::shared_ptr<mapManager::occMap> map_;
    vigo_context* dev_ = nullptr;
    uint64_t mapStamp_ = 0;             // mapAdapter's memo of the snapshot the handle holds (0 = none)
    mapRegion mapRegion_;
    int deviceOrdinal_ = 0;             // HIP device of the handle (setDevice)
    int lastIterations_ = 0;
    bool syncDevice();
    bool sweepPoints(const std::vector<pose>& pts, std::vector<uint8_t>& flags);
    void pwlPlan(std::vector<pose>& trajectory, double delT);
    pose extPose(double t);
    const std::vector<double>& timeKnots();

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
    void updateInitVel(const geometry_msgs::Twist& v) { updateInitVel(v.linear.x, v.linear.y, v.linear.z); }   // PO.cpp:202-204
    void updateInitAcc(double ax, double ay, double az);
    void updateInitAcc(const geometry_msgs::Twist& a) { updateInitAcc(a.linear.x, a.linear.y, a.linear.z); }   // PO.cpp:215-217
    void setDefaultInit();
    /* bypass the QP with a given piecewise polynomial (coefficient blocks per segment, time knots) */
    void setSolution(int polyDegree, const std::vector<double>& xSol, const std::vector<double>& ySol,
                     const std::vector<double>& zSol, const std::vector<double>& timeKnot);

    /* makePlan() of many planners in lock-step (corridor-constraint mode, PO.cpp:388-545): per round ONE
     * vigo_minsnap launch solves every active planner's QP and ONE vigo_box_collision_points launch sweeps
     * every sample of every candidate trajectory; corridor bookkeeping and the PWL fallback stay per planner.
     * trajectories[i] receives planner i's samples (delT = its sample_delta_time). */
    static std::vector<bool> makePlanBatch(const std::vector<polyTrajOctomap*>& planners, std::vector<std::vector<pose>>& trajectories);
    void makePlan();
    void makePlan(nav_msgs::Path& trajectory, double delT = 0.1);
    void makePlan(std::vector<pose>& trajectory, double delT = 0.1);
    /* the two planning loops, public in the reference as well (PO.h:100-103) */
    void makePlanAddingWaypoint(std::vector<pose>& trajectory, double delT);
    void makePlanCorridorConstraint(std::vector<pose>& trajectory, double delT);
    void makePlanAddingWaypoint() { std::vector<pose> t; makePlanAddingWaypoint(t, delT_); }            // PO.cpp:259-322
    void makePlanCorridorConstraint() { std::vector<pose> t; makePlanCorridorConstraint(t, delT_); }    // PO.cpp:388-452
    void adjustCorridorSize(const std::set<int>& collisionSeg, std::vector<double>& corridorSizeVec) {   // PO.cpp:188-192
        for (int s : collisionSeg) corridorSizeVec[s] = corridorSizeVec[s] * fs_;
    }
    void insertWaypoint(const std::set<int>& seg);                     // PO.cpp:178-186
    /* re-snapshot the map on the next device call (the reference re-fetches /octomap_binary, PO.cpp:133-145) */
    void updateMap() { mapAdapter::bumpGeneration(map_.get()); mapStamp_ = 0; }
    /* not in the reference: the box of the map the device snapshot covers (mapAdapter.h; ignored by the dense map) */
    void setMapRegion(const Eigen::Vector3d& boxMin, const Eigen::Vector3d& boxMax) {
        mapRegion_.set = true; mapRegion_.boxMin = boxMin; mapRegion_.boxMax = boxMax; mapStamp_ = 0;
    }

    /* not in the reference: HIP device ordinal of this planner's back-end handle (default 0), see bsplineTraj::setDevice */
    void setDevice(int ordinal);

    bool checkCollision(const pose& p);                                         // box sweep, PO.cpp:547-568
    bool checkCollisionPoint(const pose& p, bool ignoreUnknown = false);        // PO.cpp:571-595
    bool checkCollisionTraj(const std::vector<pose>& trajectory, std::vector<int>& collisionIdx);           // PO.cpp:619-632
    bool checkCollisionTraj(const std::vector<pose>& trajectory, double delT, std::set<int>& collisionSeg);  // PO.cpp:634-656

    geometry_msgs::PoseStamped getPose(double t);                   // PO.cpp:658-677
    double getDuration();                                           // PO.cpp:679-689
    /* parameter getters the nodes' `cout << polyPlanner` prints (PO.h:118-123) */
    double getDegree() { return polyDegree_; }
    double getDiffDegree() { return diffDegree_; }
    double getContinuityDegree() { return continuityDegree_; }
    double getDesiredVel() { return desiredVel_; }
    double getInitialRadius() { return initR_; }
    double getShrinkFactor() { return fs_; }
    bool isValid() const { return findValidTraj_; }
    int getIterations() const { return lastIterations_; }
    const std::vector<pose>& getPath() const { return path_; }
    void trajMsgConverter(const std::vector<pose>& trajectoryTemp, nav_msgs::Path& trajectory);
};

/* `cout << polyPlanner` of the demo nodes (src/poly_RRT_node.cpp:68) */
inline std::ostream& operator<<(std::ostream& os, polyTrajOctomap& planner) {
    os << "[polyTrajOctomap on MI355X] min-snap + corridor planner: polynomial degree " << planner.getDegree()
       << ", minimised derivative " << planner.getDiffDegree() << ", continuity up to order " << planner.getContinuityDegree()
       << ", desired velocity " << planner.getDesiredVel() << " m/s, corridor r0 " << planner.getInitialRadius() << " m x "
       << planner.getShrinkFactor() << " per shrink";
    return os;
}
}  // namespace trajPlanner
#endif
