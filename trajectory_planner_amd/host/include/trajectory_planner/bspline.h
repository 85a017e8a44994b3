/*
 * bspline.h — host-side uniform B-spline carrying the public method set of the reference's
 * trajPlanner::bspline (include/trajectory_planner/bspline.h:14-35), so callers of the planner
 * classes compile against it unchanged.  Own implementation: the knot vector is implicit
 * (knot i sits at (i - degree) * ts, bspline.cpp:19-28), evaluation is de Boor's scheme, the fit
 * is a Householder least squares.  The batched device counterparts are vigo_bspline_eval and
 * vigo_bspline_fit (include/vigo.h).
 */
#ifndef VIGO_HOST_BSPLINE_H
#define VIGO_HOST_BSPLINE_H
#include <trajectory_planner/compat.h>

#include <vector>

namespace trajPlanner {

class bspline {
public:
    /* construction: an empty cubic, or `degree` over the columns of `controlPoints` spaced ts apart */
    bspline();
    bspline(int degree, const Eigen::MatrixXd& controlPoints, double ts);

    /* least-squares control points through `points` with start/end velocity and acceleration
     * (bspline.cpp:74-138); false instead of the reference's exit(0) on malformed input */
    static bool parameterizeToBspline(double ts, const std::vector<Eigen::Vector3d>& points,
                                      const std::vector<Eigen::Vector3d>& startEndConditions,
                                      Eigen::MatrixXd& controlPoints);

    void initKnots();                 /* recomputes the duration (the knots themselves are implicit) */
    Eigen::Vector3d at(double t);     /* position at t, clamped to [0, duration] */
    bspline getDerivative();          /* the degree-1 spline of bspline.cpp:64-72 */
    double getDuration();
    Eigen::MatrixXd getControlPoints();

private:
    double knot(int i) const { return (i - degree_) * ts_; }

    Eigen::MatrixXd controlPoints_;   /* 3 x N, one column per control point */
    double ts_ = 0.1;
    double duration_ = 0.0;
    int degree_ = 3;
};

}  // namespace trajPlanner
#endif  /* VIGO_HOST_BSPLINE_H */
