// bspline.cpp — host B-spline (reference interface: bspline.h:14-35, behaviour: bspline.cpp:19-138)
#include <trajectory_planner/bspline.h>

#include <algorithm>
#include <cmath>
#include <iostream>

namespace trajPlanner {

bspline::bspline() {}

bspline::bspline(int degree, const Eigen::MatrixXd& controlPoints, double ts) {
    degree_ = degree;
    controlPoints_ = controlPoints;
    ts_ = ts;
    initKnots();
}

void bspline::initKnots() {
    const int knotsNum = controlPoints_.cols() - 1 + degree_ + 1 + 1;
    duration_ = knot(knotsNum - degree_ - 1);
}

// de Boor; at exact knots the lower span is used (bspline.cpp:37-42)
Eigen::Vector3d bspline::at(double t) {
    // an empty spline (a planner whose makePlan() never succeeded) evaluates to the origin instead of
    // reading past its control points; the reference has undefined behaviour there
    if (controlPoints_.cols() < degree_ + 1 || degree_ < 0 || degree_ > 7) return Eigen::Vector3d(0.0, 0.0, 0.0);
    const double tb = std::min(std::max(0.0, t), duration_);
    int k = degree_;
    while (!(knot(k + 1) >= tb)) ++k;
    Eigen::Vector3d d[8];
    for (int i = 0; i <= degree_; ++i) d[i] = controlPoints_.col(k - degree_ + i);
    for (int r = 1; r <= degree_; ++r) {
        for (int i = degree_; i >= r; --i) {
            const double alpha = (tb - knot(i + k - degree_)) / (knot(i + 1 + k - r) - knot(i + k - degree_));
            for (int a = 0; a < 3; ++a) d[i](a) = (1 - alpha) * d[i - 1](a) + alpha * d[i](a);
        }
    }
    return d[degree_];
}

double bspline::getDuration() { return duration_; }

bspline bspline::getDerivative() {
    if (controlPoints_.cols() < 2 || degree_ < 1) return bspline();
    Eigen::MatrixXd ctp(controlPoints_.rows(), controlPoints_.cols() - 1);
    for (int i = 0; i < ctp.cols(); ++i) {
        const double den = knot(i + degree_ + 1) - knot(i + 1);
        for (int a = 0; a < 3; ++a) ctp(a, i) = degree_ * (controlPoints_(a, i + 1) - controlPoints_(a, i)) / den;
    }
    return bspline(degree_ - 1, ctp, ts_);
}

Eigen::MatrixXd bspline::getControlPoints() { return controlPoints_; }

// Least squares of the (K+4) x (K+2) system of bspline.cpp:95-131 by Householder QR.
bool bspline::parameterizeToBspline(double ts, const std::vector<Eigen::Vector3d>& points,
                                    const std::vector<Eigen::Vector3d>& startEndConditions,
                                    Eigen::MatrixXd& controlPoints) {
    if (ts <= 0) { std::cout << "[Bspline]: Invalid timestep." << std::endl; return false; }
    if (points.size() <= 3) {
        std::cout << "[Bspline]: Point set only has " << points.size() << " points. At least need 4." << std::endl;
        return false;
    }
    if (startEndConditions.size() != 4) {
        std::cout << "[Bspline]: Please enter correct start and end acc/vel." << std::endl;
        return false;
    }
    const int K = (int)points.size();
    const int R = K + 4, C = K + 2;
    std::vector<double> A((size_t)R * C, 0.0);  // row-major
    std::vector<double> b((size_t)R * 3, 0.0);
    auto a = [&](int r, int c) -> double& { return A[(size_t)r * C + c]; };
    for (int i = 0; i < K; ++i) {
        a(i, i) = 1 / 6.0; a(i, i + 1) = 4 / 6.0; a(i, i + 2) = 1 / 6.0;
        for (int q = 0; q < 3; ++q) b[(size_t)i * 3 + q] = points[i](q);
    }
    a(K, 0) = -1 / 2.0 / ts; a(K, 2) = 1 / 2.0 / ts;
    a(K + 1, K - 1) = -1 / 2.0 / ts; a(K + 1, K + 1) = 1 / 2.0 / ts;
    a(K + 2, 0) = 1 / ts / ts; a(K + 2, 1) = -2 / ts / ts; a(K + 2, 2) = 1 / ts / ts;
    a(K + 3, K - 1) = 1 / ts / ts; a(K + 3, K) = -2 / ts / ts; a(K + 3, K + 1) = 1 / ts / ts;
    for (int i = 0; i < 4; ++i)
        for (int q = 0; q < 3; ++q) b[(size_t)(K + i) * 3 + q] = startEndConditions[i](q);

    // Householder QR, applied to the three right-hand sides on the fly
    for (int c = 0; c < C; ++c) {
        double nrm = 0;
        for (int r = c; r < R; ++r) nrm += a(r, c) * a(r, c);
        nrm = std::sqrt(nrm);
        if (nrm == 0) return false;
        const double alpha = a(c, c) > 0 ? -nrm : nrm;
        std::vector<double> v(R - c);
        for (int r = c; r < R; ++r) v[r - c] = a(r, c);
        v[0] -= alpha;
        double vn = 0;
        for (double x : v) vn += x * x;
        if (vn == 0) continue;
        for (int cc = c; cc < C; ++cc) {
            double s = 0;
            for (int r = c; r < R; ++r) s += v[r - c] * a(r, cc);
            s = 2 * s / vn;
            for (int r = c; r < R; ++r) a(r, cc) -= s * v[r - c];
        }
        for (int q = 0; q < 3; ++q) {
            double s = 0;
            for (int r = c; r < R; ++r) s += v[r - c] * b[(size_t)r * 3 + q];
            s = 2 * s / vn;
            for (int r = c; r < R; ++r) b[(size_t)r * 3 + q] -= s * v[r - c];
        }
    }
    controlPoints.resize(3, C);
    for (int q = 0; q < 3; ++q) {
        for (int r = C - 1; r >= 0; --r) {
            double s = b[(size_t)r * 3 + q];
            for (int cc = r + 1; cc < C; ++cc) s -= a(r, cc) * controlPoints(q, cc);
            controlPoints(q, r) = s / a(r, r);
        }
    }
    return true;
}

}  // namespace trajPlanner
