/*
 * mini_eigen.h — the subset of Eigen the planner facades use, for builds without Eigen (this image, the GPU box).
 * Every member below exists in real Eigen with the same meaning, so the facade sources compile unchanged against
 * <Eigen/Eigen> (-DVIGO_WITH_ROS); nothing Eigen does not have is offered here.  (resize() zero-fills, which real
 * Eigen does not promise: the sources never rely on it — they use MatrixXd::Zero where zeros are meant.)
 * NOT used to build any reference source.
 */
#ifndef TRAJECTORY_PLANNER_MINI_EIGEN_H
#define TRAJECTORY_PLANNER_MINI_EIGEN_H
#include <cmath>
#include <cstddef>
#include <vector>

namespace Eigen {
struct Vector3d {
    double v[3];
    Vector3d() : v{0, 0, 0} {}
    Vector3d(double x, double y, double z) : v{x, y, z} {}
    double& operator()(int i) { return v[i]; }
    double operator()(int i) const { return v[i]; }
    double& operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
    Vector3d operator+(const Vector3d& o) const { return {v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]}; }
    Vector3d operator-(const Vector3d& o) const { return {v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]}; }
    Vector3d operator-() const { return {-v[0], -v[1], -v[2]}; }
    Vector3d operator*(double s) const { return {v[0] * s, v[1] * s, v[2] * s}; }
    Vector3d operator/(double s) const { return {v[0] / s, v[1] / s, v[2] / s}; }
    Vector3d& operator+=(const Vector3d& o) { v[0] += o.v[0]; v[1] += o.v[1]; v[2] += o.v[2]; return *this; }
    double dot(const Vector3d& o) const { return (v[0] * o.v[0] + v[1] * o.v[1]) + v[2] * o.v[2]; }
    double squaredNorm() const { return dot(*this); }
    double norm() const { return std::sqrt(squaredNorm()); }
    Vector3d cross(const Vector3d& o) const {
        return {v[1] * o.v[2] - v[2] * o.v[1], v[2] * o.v[0] - v[0] * o.v[2], v[0] * o.v[1] - v[1] * o.v[0]};
    }
    Vector3d normalized() const { return *this / norm(); }
};
inline Vector3d operator*(double s, const Vector3d& a) { return a * s; }

struct Vector3i {
    int v[3];
    Vector3i() : v{0, 0, 0} {}
    Vector3i(int x, int y, int z) : v{x, y, z} {}
    int& operator()(int i) { return v[i]; }
    int operator()(int i) const { return v[i]; }
};

/* dynamic column-major matrix; the facades only use 3 x N */
class MatrixXd {
public:
    MatrixXd() : r_(0), c_(0) {}
    MatrixXd(int r, int c) : r_(r), c_(c), d_((size_t)r * c, 0.0) {}
    void resize(int r, int c) { r_ = r; c_ = c; d_.assign((size_t)r * c, 0.0); }
    int rows() const { return r_; }
    int cols() const { return c_; }
    double& operator()(int r, int c) { return d_[(size_t)c * r_ + r]; }
    double operator()(int r, int c) const { return d_[(size_t)c * r_ + r]; }
    double* data() { return d_.data(); }
    const double* data() const { return d_.data(); }
    Vector3d col(int c) const { return Vector3d((*this)(0, c), (*this)(1, c), (*this)(2, c)); }
    static MatrixXd Zero(int r, int c) { return MatrixXd(r, c); }
private:
    int r_, c_;
    std::vector<double> d_;
};
}  // namespace Eigen
#endif
