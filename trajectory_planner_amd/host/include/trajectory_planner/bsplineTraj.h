/*
 * bsplineTraj.h — trajPlanner::bsplineTraj with the reference's public interface
 * (include/trajectory_planner/bsplineTraj.h:87-158) over the MI355X back-end (include/vigo.h).
 *
 * Host side (this class): path conditioning, B-spline fit, collision-segment bookkeeping, A*,
 * guide assignment, the rebound loop's decisions, time re-parameterisation, getters.
 * Device side (libvigo_hip.so): optimize() = vigo_optimize, the rebound-loop gates
 * (vigo_traj_collision / vigo_traj_dynamic_collision), isUnknown(guide) = vigo_guides_unknown.
 *
 * makePlan() runs one planner (B = 1, link compatibility); makePlanBatch() runs the rebound loops
 * of many planners in lock-step so that every optimize() of the batch is ONE kernel launch —
 * that is the configuration the device is built for.
 */
#ifndef BSPLINETRAJ_H
#define BSPLINETRAJ_H
#include <trajectory_planner/bspline.h>
#include <trajectory_planner/compat.h>
#include <trajectory_planner/mapAdapter.h>
#include <trajectory_planner/path_search/astarOcc.h>
#include <trajectory_planner/utils.h>

#include <memory>
#include <utility>
#include <string>
#include <vector>

const int bsplineDegree = 3;
struct vigo_context;
struct vigo_params_s;

namespace trajPlanner {
struct optData {
    Eigen::MatrixXd controlPoints;
    std::vector<std::vector<Eigen::Vector3d>> guidePoints;
    std::vector<std::vector<Eigen::Vector3d>> guideDirections;
    std::vector<bool> findGuidePoint;
    std::vector<Eigen::Vector3d> dynamicObstaclesPos;
    std::vector<Eigen::Vector3d> dynamicObstaclesVel;
    std::vector<Eigen::Vector3d> dynamicObstaclesSize;
};

class bsplineTraj {
private:
    ros::NodeHandle nh_;
    double controlPointDistance_ = 0.25;  // bsplineTraj.h:46
    double controlPointsTs_ = 0.2;        // bsplineTraj.h:47
    trajPlanner::bspline bspline_;
    trajPlanner::optData optData_;
    double ts_, dthresh_, maxVel_, maxAcc_;
    double weightDistance_, weightSmoothness_, weightFeasibility_, weightDynamicObstacle_;
    double notCheckRatio_ = 0.0;
    bool planInZAxis_;
    double minHeight_, maxHeight_, uncertainAwareFactor_, predHorizon_, distThreshDynamic_, maxPathLength_;
    Eigen::Vector3d maxObstacleSize_;
    std::shared_ptr<mapManager::occMap> map_;
    std::shared_ptr<AStar> pathSearch_;
    std::vector<std::pair<int, int>> collisionSeg_;
    std::vector<std::vector<Eigen::Vector3d>> astarPaths_;
    bool init_ = false;
    double linearFactor_ = 1.0;
    std::vector<Eigen::Vector3d> inputPathVis_;

    // device
    vigo_context* dev_ = nullptr;
    uint64_t mapStamp_ = 0;            // mapAdapter's memo of the snapshot this planner's handle holds (0 = none)
    mapRegion mapRegion_;
    int deviceOrdinal_ = 0;            // HIP device of this planner's handle (setDevice)
    int lastStatus_ = 0;
    bool syncDevice();   // params + map snapshot -> handle; false when no GPU / HIP failure

    // per-planner state of the rebound loop (BT.cpp:611-685) so makePlanBatch can interleave planners
    struct Rebound {
        double w0 = 0, wo0 = 0;
        int failCount = 0;
        bool done = false, ok = false, needOptimize = true;
        // filled by the device-resident rounds (vigo_rebound_rounds) for the host's part of the loop
        int devStatus = 0;
        bool gateStatic = false, gateDynamic = false;
    };

public:
    bsplineTraj();
    bsplineTraj(const ros::NodeHandle& nh);
    ~bsplineTraj();
    bsplineTraj(const bsplineTraj&) = delete;
    bsplineTraj& operator=(const bsplineTraj&) = delete;
    void init(const ros::NodeHandle& nh);
    void initParam();
    void setMap(const std::shared_ptr<mapManager::occMap>& map);
    /* not in the reference: the box of the map the device snapshot covers (needed for a map type that offers no bulk
     * access, see mapAdapter.h; ignored by the in-tree dense map) and the request to re-snapshot a map that changed */
    void setMapRegion(const Eigen::Vector3d& boxMin, const Eigen::Vector3d& boxMax);
    void refreshMap();
    /* not in the reference: the HIP device ordinal this planner's back-end handle lives on (default 0); call it before
     * the first plan, or later to move the planner (the handle is re-created and the map uploaded again) */
    void setDevice(int ordinal);
    void updateMaxVel(double maxVel);
    void updateMaxAcc(double maxAcc);
    bool inputPathCheck(const nav_msgs::Path& path, nav_msgs::Path& adjustedPath, double dt, double& finalTime);
    bool fillPath(const nav_msgs::Path& path, nav_msgs::Path& adjustedPath);
    bool updatePath(const nav_msgs::Path& adjustedPath, const std::vector<Eigen::Vector3d>& startEndConditions);
    void updateDynamicObstacles(const std::vector<Eigen::Vector3d>& obstaclesPos, const std::vector<Eigen::Vector3d>& obstaclesVel,
                                const std::vector<Eigen::Vector3d>& obstaclesSize);

    bool makePlan();
    bool makePlan(nav_msgs::Path& trajectory, bool yaw = true);
    /* Planners are grouped by control-point count, map object and every hot-path parameter (the yaml values, maxVel):
     * each group is one batch on the device.  Returns per-planner success like makePlan(). */
    static std::vector<bool> makePlanBatch(const std::vector<bsplineTraj*>& planners);
    /* The rebound loop of BT.cpp:611-685 runs on the device between two A* calls (vigo_rebound_rounds, default) or
     * round by round from the host (the round-1 path, kept for comparison: identical control points). */
    /* makePlanBatch of at least this many planners runs as two to four pipelined parts of >= half this size (all but the
     * first on companion host threads with their own handles and HIP streams); 0 = never.  Default 2048.  Same plans. */
    static void setBatchPipelineThreshold(size_t planners);
    static void setDeviceResidentRebound(bool on);
    static bool deviceResidentRebound();
    /* updatePath() for many planners at once: the least-squares fits run as one device launch */
    static std::vector<bool> updatePathBatch(const std::vector<bsplineTraj*>& planners, const std::vector<nav_msgs::Path>& paths,
                                             const std::vector<std::vector<Eigen::Vector3d>>& startEndConditions);
    void clear();
    void findCollisionSeg(const Eigen::MatrixXd& controlPoints, std::vector<std::pair<int, int>>& collisionSeg);
    bool pathSearch(std::vector<std::pair<int, int>>& collisionSeg, std::vector<std::vector<Eigen::Vector3d>>& paths);
    void assignGuidePointsSemiCircle(const std::vector<std::vector<Eigen::Vector3d>>& paths,
                                     const std::vector<std::pair<int, int>>& collisionSeg);
    bool isReguideRequired(std::vector<std::pair<int, int>>& reguideCollisionSeg);
    bool optimizeTrajectory();
    int optimize();
    void adjustPathLengthDirect(const std::vector<Eigen::Vector3d>& path, std::vector<Eigen::Vector3d>& adjustedPath);

    /* the lbfgs_evaluate_t seam (BT.h:118-119) on the device: cost and gradient of x */
    double costFunction(const double* x, double* grad, const int n);
    static double solverCostFunction(void* func_data, const double* x, double* grad, const int n);   // BT.h:118
    /* the four terms on their own (BT.h:120-123); gradient is 3 x N with the fixed end columns left zero */
    void getDistanceCost(const Eigen::MatrixXd& controlPoints, double& cost, Eigen::MatrixXd& gradient);
    void getSmoothnessCost(const Eigen::MatrixXd& controlPoints, double& cost, Eigen::MatrixXd& gradient);
    void getFeasibilityCost(const Eigen::MatrixXd& controlPoints, double& cost, Eigen::MatrixXd& gradient);
    void getDynamicObstacleCost(const Eigen::MatrixXd& controlPoints, double& cost, Eigen::MatrixXd& gradient);
    void writeCurrentTrajInfo(const std::string& filePath, double dt);                               // BT.cpp:1464-1496

    void linearFeasibilityReparam();
    double getLinearReparamTime(double t);
    double getLinearFactor();

    double getInitTs();
    double getControlPointTs();
    double getControlPointDist();
    trajPlanner::bspline getTrajectory();
    geometry_msgs::PoseStamped getPose(double t, bool yaw = true);
    double getDuration();
    double getTimestep();
    Eigen::MatrixXd getControlPoints();
    const optData& getOptData() const { return optData_; }   /* added (tests): guide points / directions as the optimizer gets them */
    /* added (workload tools, cabi_host.cpp): take these control points as the planner's current ones with empty guide
     * lists — the state updatePath() leaves (BT.cpp:315-322) — so the host steps of the rebound loop can be replayed on them */
    void setControlPoints(const Eigen::MatrixXd& controlPoints) { installControlPoints(controlPoints, {}); }
    bool isCurrTrajValid();
    bool isCurrTrajValid(Eigen::Vector3d& firstCollisionPos);
    int getLastSolverStatus() const { return lastStatus_; }
    bool hasDynamicObstacles() const { return !optData_.dynamicObstaclesPos.empty(); }

    std::vector<Eigen::Vector3d> evalTraj();
    std::vector<Eigen::Vector3d> evalTraj(double dt);
    nav_msgs::Path evalTrajToMsg(bool yaw = true);
    nav_msgs::Path evalTrajToMsg(double dt, bool yaw = true);
    void pathMsgToEigenPoints(const nav_msgs::Path& path, std::vector<Eigen::Vector3d>& points);
    void eigenPointsToPathMsg(const std::vector<Eigen::Vector3d>& points, nav_msgs::Path& path);

    bool checkCollisionLine(const Eigen::Vector3d& p1, const Eigen::Vector3d& p2);
    void shortcutPath(const std::vector<Eigen::Vector3d>& path, std::vector<Eigen::Vector3d>& pathSC);
    bool findGuidePointSemiCircle(int controlPointIdx, const std::pair<int, int>& seg, const std::vector<Eigen::Vector3d>& path,
                                  Eigen::Vector3d& guidePoint);
    bool hasCollisionTrajectory(const Eigen::MatrixXd& controlPoints);
    bool hasDynamicCollisionTrajectory(const Eigen::MatrixXd& controlPoints);
    /* public helpers of the reference's class, BT.h:165-172 (its triangle / distance-field / polygon helpers, :173-177,
       are only called from commented-out code and are not carried) */
    void shortcutPaths(const std::vector<std::vector<Eigen::Vector3d>>& paths, std::vector<std::vector<Eigen::Vector3d>>& pathsSC);
    bool indexInCollisionSeg(const std::vector<std::pair<int, int>>& collisionSeg, int idx);
    void compareCollisionSeg(const std::vector<std::pair<int, int>>& prevCollisionSeg, const std::vector<std::pair<int, int>>& newCollisionSeg,
                             std::vector<int>& newCollisionPoints, std::vector<int>& overlappedCollisionPoints);
    int findCollisionSegIndex(const std::vector<std::pair<int, int>>& collisionSeg, int idx);
    bool isControlPointRequireNewGuide(int controlPointIdx);

private:
    void reboundBegin(Rebound& r);
    /* one pass of the loop body of BT.cpp:619-681 given the gate results; sets r.done/ok/needOptimize */
    void reboundStep(Rebound& r, bool hasCollision, bool hasDynamicCollision, bool timedOut);
    bool prepareFitPoints(const nav_msgs::Path& adjustedPath, std::vector<Eigen::Vector3d>& adjustedCurveFitPoints);
    /* the same with the previous path length (adjustPathLengthDirect's function-static, BT.cpp:755) passed in and out:
       *wrote says whether the call reached that function at all */
    bool prepareFitPointsWith(const nav_msgs::Path& adjustedPath, std::vector<Eigen::Vector3d>& adjustedCurveFitPoints, double prevIn,
                              double& prevOut, bool& wrote);
    void adjustPathLengthWith(const std::vector<Eigen::Vector3d>& path, std::vector<Eigen::Vector3d>& adjustedPath, double prevIn, double& prevOut);
    void installControlPoints(const Eigen::MatrixXd& controlPoints, const std::vector<Eigen::Vector3d>& adjustedCurveFitPoints);
    bool termCost(int term, const Eigen::MatrixXd& controlPoints, double& cost, Eigen::MatrixXd& gradient);
    void reboundFinish(Rebound& r, bool ok);
    bool sameBatchKey(const bsplineTraj& o) const;                  // may share a device batch with o
    void fillParams(vigo_params_s* P) const;
    static void solveBatch(const std::vector<bsplineTraj*>& ps);   // one vigo_optimize for all
    /* up to maxRounds rounds of the loop on the device for one group; fills rb[i]->devStatus / gate flags and the
     * planners' control points, weights, failCount, collisionSeg_.  false: device failure (nothing usable) */
    static bool deviceRounds(const std::vector<bsplineTraj*>& grp, const std::vector<Rebound*>& rb, int maxRounds);
    static void gateBatch(const std::vector<bsplineTraj*>& ps, std::vector<uint8_t>& col, std::vector<uint8_t>& dyn);
};
}  // namespace trajPlanner
#endif
