/*
 * bspline.h — host-side uniform B-spline with the reference's public interface
 * (include/trajectory_planner/bspline.h:14-35): knots (i - degree) * ts, de Boor evaluation,
 * derivative spline, least-squares fit of waypoints to control points.  Own implementation;
 * batched evaluation on the device is vigo_bspline_eval (include/vigo.h).
 */
#ifndef BSPLINE_H
#define BSPLINE_H
#include <trajectory_planner/compat.h>
#include <vector>

namespace trajPlanner {
class bspline {
private:
    int degree_ = 3;
    Eigen::MatrixXd controlPoints_;
    double ts_ = 0.1;
    double duration_ = 0.0;
    double knot(int i) const { return (i - degree_) * ts_; }

public:
    bspline();
    bspline(int degree, const Eigen::MatrixXd& controlPoints, double ts);
    void initKnots();
    Eigen::Vector3d at(double t);
    double getDuration();
    bspline getDerivative();
    /* bspline.cpp:74-138; returns false (instead of exit(0)) on malformed input */
    static bool parameterizeToBspline(double ts, const std::vector<Eigen::Vector3d>& points,
                                      const std::vector<Eigen::Vector3d>& startEndConditions,
                                      Eigen::MatrixXd& controlPoints);
    Eigen::MatrixXd getControlPoints();
};
}  // namespace trajPlanner
#endif
