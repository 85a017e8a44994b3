/*
 * polyTrajSolver.h — min-snap piecewise-polynomial QP with the reference's public interface
 * (include/trajectory_planner/polyTrajSolver.h:25-147): per-axis QP in normalised segment time,
 * continuity up to `continuityDegree`, optional corridor boxes; coefficients are rescaled to
 * un-normalised local time after the solve (polyTrajSolver.cpp:874-878).
 *
 * The reference hands the three QPs to OSQP 0.6.2 through OsqpEigen (prebuilt third-party
 * binaries, never loaded here; ADMM stopped at eps 1e-3).  This build solves them exactly: the
 * equality rows (waypoints, continuity, end conditions: 6 per segment) are eliminated with an
 * orthonormal null-space basis, leaving 2 free coefficients per segment, and the corridor boxes
 * are handled by a Goldfarb-Idnani dual active-set iteration on that small strictly convex QP;
 * an infeasible corridor is detected and reported.  Host-side plumbing for BASELINE config 1
 * (SURVEY.md §8f "next" #3); parity vs OSQP is 1e-3-class by OSQP's own tolerance.
 */
#ifndef POLYTRAJSOLVER_H
#define POLYTRAJSOLVER_H
#include <trajectory_planner/compat.h>
#include <trajectory_planner/utils.h>

#include <vector>

namespace trajPlanner {

/* min 1/2 x'Px + q'x  s.t.  l <= Ax <= u  (dense, row-major; P positive definite on the null space
 * of the equality rows); returns active-set iterations >= 0, -1 numerical failure, -2 infeasible */
int solveDenseQP(int n, int m, const std::vector<double>& P, const std::vector<double>& q, const std::vector<double>& A,
                 const std::vector<double>& l, const std::vector<double>& u, std::vector<double>& x);

class polyTrajSolver {
private:
    int polyDegree_, diffDegree_, continuityDegree_;
    int paramDim_ = 0, constraintNum_ = 0;
    double desiredVel_;
    std::vector<pose> path_;
    std::vector<double> desiredTime_;
    double initVel_[3] = {0, 0, 0}, endVel_[3] = {0, 0, 0}, initAcc_[3] = {0, 0, 0}, endAcc_[3] = {0, 0, 0};
    std::vector<double> xSol_, ySol_, zSol_;
    bool corridorConstraint_ = false;
    bool softConstraint_ = false;            /* PS.cpp:19-22: the interior waypoints as boxes of half size scDeviation_ */
    double scDeviation_[3] = {0, 0, 0};
    double corridorRes_ = 5.0;
    std::vector<double> corridorSizeVec_;
    std::vector<std::vector<std::pair<double, pose>>> segToTimePose_;  // (normalised time, box centre) per segment
    bool solved_ = false;

    int getConstraintNum() const;
    void avgTimeAllocation();
    void updateCorridorParam();
    void constructP(std::vector<double>& P) const;
    void constructA(std::vector<double>& A) const;
    void constructBound(std::vector<double> (&l)[3], std::vector<double> (&u)[3]) const;

public:
    polyTrajSolver(int polyDegree, int diffDegree, int continuityDegree, double desiredVel);
    void updatePath(const std::vector<pose>& path);
    void updateInitVel(double vx, double vy, double vz);
    void updateEndVel(double vx, double vy, double vz);
    void updateInitAcc(double ax, double ay, double az);
    void updateEndAcc(double ax, double ay, double az);
    void setSoftConstraint(double r);                              /* PS.cpp:943-958 */
    void setSoftConstraint(double rx, double ry);
    void setSoftConstraint(double rx, double ry, double rz);
    void setCorridorConstraint(const std::vector<double>& corridorSizeVec, double corridorRes);
    void setCorridorConstraint(double corridorSize, double corridorRes);
    /* install coefficients computed elsewhere (vigo_minsnap on the device) for the current path;
     * un-normalised local time, (deg+1) per segment; knots must be getTimeKnot()'s */
    void installSolution(const std::vector<double>& x, const std::vector<double>& y, const std::vector<double>& z);
    bool solve();   // the reference's solve() is void and silently keeps a stale solution on failure
    /* all three axes hold a polynomial OF THE CURRENT PATH (false until the first success, and again after
     * updatePath() changed the number of segments: a stale polynomial of another path cannot be sampled on
     * this path's knots) */
    bool hasSolution() const {
        const size_t want = paramDim_ > 0 ? (size_t)paramDim_ : 0;
        return want > 0 && xSol_.size() == want && ySol_.size() == want && zSol_.size() == want;
    }
    pose getPose(double t);
    Eigen::Vector3d getPos(double t) { const pose p = getPose(t); return Eigen::Vector3d(p.x, p.y, p.z); }   /* PS.h:135 */
    Eigen::Vector3d getVel(double t);   /* PS.h:136 */
    Eigen::Vector3d getAcc(double t);   /* PS.h:137 — the x component with the reference's exponent, see the .cpp */
    void getTrajectory(std::vector<pose>& trajectory, double delT);
    std::vector<double>& getTimeKnot();
    const std::vector<double>& getSolution(int axis) const { return axis == 0 ? xSol_ : (axis == 1 ? ySol_ : zSol_); }
    int getPolyDegree() const { return polyDegree_; }
    void getCorridor(std::vector<std::vector<std::pair<double, pose>>>& segToTimePose, std::vector<double>& corridorSizeVec) const;
};
}  // namespace trajPlanner
#endif
