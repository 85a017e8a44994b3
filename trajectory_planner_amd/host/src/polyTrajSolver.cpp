// polyTrajSolver.cpp — min-snap QP (structure of polyTrajSolver.cpp:125-138, :156-160, :241-307,
// :314-584, :587-846, :874-878, :985-1012, :1026-1056, :1125-1137 of the reference) and a small
// dense ADMM QP solver.  Own implementation.
#include <trajectory_planner/polyTrajSolver.h>

#include <algorithm>
#include <cmath>
#include <iostream>

namespace trajPlanner {

// ---- dense Cholesky helpers -------------------------------------------------------------
static bool cholesky(std::vector<double>& M, int n) {  // in place, lower
    for (int j = 0; j < n; ++j) {
        double d = M[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= M[(size_t)j * n + k] * M[(size_t)j * n + k];
        if (!(d > 0)) return false;
        d = std::sqrt(d);
        M[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = M[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) s -= M[(size_t)i * n + k] * M[(size_t)j * n + k];
            M[(size_t)i * n + j] = s / d;
        }
    }
    return true;
}
static void cholSolve(const std::vector<double>& L, int n, std::vector<double>& b) {
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[(size_t)i * n + k] * b[k];
        b[i] = s / L[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < n; ++k) s -= L[(size_t)k * n + i] * b[k];
        b[i] = s / L[(size_t)i * n + i];
    }
}

// ADMM of OSQP (Stellato et al. 2020, Algorithm 1) in its reduced dense form:
//   (P + sigma I + A' R A) xt = sigma x - q + A'(R z - y),  zt = A xt,
//   x <- alpha xt + (1-alpha) x,  z <- clip(alpha zt + (1-alpha) z + y/R),  y <- y + R(alpha zt + (1-alpha) z_prev - z)
// with R = diag(rho_i), rho_i = 1e3 rho on equality rows.  rho is re-balanced from the residual
// ratio every 50 iterations.
int solveDenseQP(int n, int m, const std::vector<double>& P, const std::vector<double>& q, const std::vector<double>& A_in,
                 const std::vector<double>& l_in, const std::vector<double>& u_in, std::vector<double>& x, double eps, int maxIter) {
    const double sigma = 1e-6, alpha = 1.6;
    // row equilibration: every constraint row scaled to unit infinity norm (the continuity rows carry
    // dt^4 factors), so one tolerance means the same thing on every row
    std::vector<double> A(A_in), l(l_in), u(u_in);
    for (int r = 0; r < m; ++r) {
        double mx = 0;
        for (int i = 0; i < n; ++i) mx = std::max(mx, std::fabs(A[(size_t)r * n + i]));
        if (mx > 0) {
            const double e = 1.0 / mx;
            for (int i = 0; i < n; ++i) A[(size_t)r * n + i] *= e;
            l[r] *= e;
            u[r] *= e;
        }
    }
    double rho = 0.1;
    std::vector<double> rhoV(m), z(m, 0.0), y(m, 0.0), K, rhs(n), zt(m), zprev(m), Ax(m), Px(n), Aty(n);
    x.assign(n, 0.0);
    auto factor = [&]() -> bool {
        for (int i = 0; i < m; ++i) rhoV[i] = (l[i] == u[i]) ? 1e3 * rho : rho;
        K.assign((size_t)n * n, 0.0);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j <= i; ++j) K[(size_t)i * n + j] = P[(size_t)i * n + j];
        for (int i = 0; i < n; ++i) K[(size_t)i * n + i] += sigma;
        for (int r = 0; r < m; ++r) {
            const double* a = &A[(size_t)r * n];
            for (int i = 0; i < n; ++i) {
                if (a[i] == 0.0) continue;
                const double ai = rhoV[r] * a[i];
                for (int j = 0; j <= i; ++j) K[(size_t)i * n + j] += ai * a[j];
            }
        }
        return cholesky(K, n);
    };
    if (!factor()) return -1;
    for (int it = 1; it <= maxIter; ++it) {
        for (int i = 0; i < n; ++i) rhs[i] = sigma * x[i] - q[i];
        for (int r = 0; r < m; ++r) {
            const double c = rhoV[r] * z[r] - y[r];
            const double* a = &A[(size_t)r * n];
            for (int i = 0; i < n; ++i) rhs[i] += a[i] * c;
        }
        cholSolve(K, n, rhs);  // rhs = xt
        for (int r = 0; r < m; ++r) {
            const double* a = &A[(size_t)r * n];
            double s = 0;
            for (int i = 0; i < n; ++i) s += a[i] * rhs[i];
            zt[r] = s;
        }
        for (int i = 0; i < n; ++i) x[i] = alpha * rhs[i] + (1 - alpha) * x[i];
        zprev = z;
        for (int r = 0; r < m; ++r) {
            const double v = alpha * zt[r] + (1 - alpha) * zprev[r];
            z[r] = std::min(std::max(v + y[r] / rhoV[r], l[r]), u[r]);
            y[r] += rhoV[r] * (v - z[r]);
        }
        if (it % 10 == 0 || it == maxIter) {
            // residuals: primal ||Ax - z||inf, dual ||Px + q + A'y||inf
            double rp = 0, rd = 0, nAx = 0, nz = 0, nPx = 0, nAty = 0, nq = 0;
            for (int r = 0; r < m; ++r) {
                const double* a = &A[(size_t)r * n];
                double s = 0;
                for (int i = 0; i < n; ++i) s += a[i] * x[i];
                Ax[r] = s;
                rp = std::max(rp, std::fabs(s - z[r]));
                nAx = std::max(nAx, std::fabs(s));
                nz = std::max(nz, std::fabs(z[r]));
            }
            for (int i = 0; i < n; ++i) {
                double s = 0, t = 0;
                for (int j = 0; j < n; ++j) s += P[(size_t)i * n + j] * x[j];
                for (int r = 0; r < m; ++r) t += A[(size_t)r * n + i] * y[r];
                Px[i] = s; Aty[i] = t;
                rd = std::max(rd, std::fabs(s + q[i] + t));
                nPx = std::max(nPx, std::fabs(s));
                nAty = std::max(nAty, std::fabs(t));
                nq = std::max(nq, std::fabs(q[i]));
            }
            const double ep = eps + eps * std::max(nAx, nz), ed = eps + eps * std::max(std::max(nPx, nAty), nq);
            if (rp <= ep && rd <= ed) return it;
            if (it % 50 == 0) {
                const double num = rp / std::max(std::max(nAx, nz), 1e-12), den = rd / std::max(std::max(std::max(nPx, nAty), nq), 1e-12);
                const double ratio = std::sqrt(num / std::max(den, 1e-30));
                if (ratio > 5.0 || ratio < 0.2) {
                    rho = std::min(std::max(rho * ratio, 1e-6), 1e6);
                    if (!factor()) return -1;
                }
            }
        }
    }
    return maxIter;
}

// ---- the min-snap problem -----------------------------------------------------------------
polyTrajSolver::polyTrajSolver(int polyDegree, int diffDegree, int continuityDegree, double desiredVel)
    : polyDegree_(polyDegree), diffDegree_(diffDegree), continuityDegree_(continuityDegree), desiredVel_(desiredVel) {}

int polyTrajSolver::getConstraintNum() const {  // PS.cpp:156-160
    const int K = (int)path_.size() - 1;
    return (2 + K - 1 + K - 1) + (2 + K - 1) + (2 + K - 1) + (K - 1) * (continuityDegree_ - 2);
}

void polyTrajSolver::avgTimeAllocation() {  // PS.cpp:125-138
    double total = 0;
    desiredTime_.assign(1, 0.0);
    for (size_t i = 1; i < path_.size(); ++i) {
        const pose &a = path_[i], &b = path_[i - 1];
        const double dist = std::sqrt(std::pow(a.x - b.x, 2) + std::pow(a.y - b.y, 2) + std::pow(a.z - b.z, 2));
        total += dist / desiredVel_;
        desiredTime_.push_back(total);
    }
}

void polyTrajSolver::updatePath(const std::vector<pose>& path) {
    path_ = path;
    paramDim_ = (polyDegree_ + 1) * ((int)path_.size() - 1);
    if (continuityDegree_ - 2 < 0) continuityDegree_ = 2;
    constraintNum_ = getConstraintNum();
    avgTimeAllocation();
    solved_ = false;
}

void polyTrajSolver::updateInitVel(double vx, double vy, double vz) { initVel_[0] = vx; initVel_[1] = vy; initVel_[2] = vz; }
void polyTrajSolver::updateEndVel(double vx, double vy, double vz) { endVel_[0] = vx; endVel_[1] = vy; endVel_[2] = vz; }
void polyTrajSolver::updateInitAcc(double ax, double ay, double az) { initAcc_[0] = ax; initAcc_[1] = ay; initAcc_[2] = az; }
void polyTrajSolver::updateEndAcc(double ax, double ay, double az) { endAcc_[0] = ax; endAcc_[1] = ay; endAcc_[2] = az; }

void polyTrajSolver::setCorridorConstraint(const std::vector<double>& corridorSizeVec, double corridorRes) {
    if (path_.empty()) { std::cout << "[Trajectory Solver]: Invalid! Please load path first!!" << std::endl; return; }
    corridorConstraint_ = true;
    corridorSizeVec_ = corridorSizeVec;
    corridorRes_ = corridorRes;
    updateCorridorParam();
}

void polyTrajSolver::setCorridorConstraint(double corridorSize, double corridorRes) {
    setCorridorConstraint(std::vector<double>(path_.size() - 1, corridorSize), corridorRes);
}

// PS.cpp:985-1012: box centres on the straight segment at normalised times 0, dt, 2dt, ... <= 1
void polyTrajSolver::updateCorridorParam() {
    segToTimePose_.clear();
    int count = 0;
    for (size_t i = 0; i + 1 < path_.size(); ++i) {
        std::vector<std::pair<double, pose>> timeToPose;
        if (corridorSizeVec_[i] != 0.0) {
            const pose &ps = path_[i], &pe = path_[i + 1];
            const double duration = desiredTime_[i + 1] - desiredTime_[i];
            const int num = (int)std::ceil(duration * corridorRes_);
            const double dt = 1.0 / num;
            for (double t = 0; t <= 1.0; t += dt) {
                timeToPose.push_back({t, pose(ps.x + (pe.x - ps.x) * t, ps.y + (pe.y - ps.y) * t, ps.z + (pe.z - ps.z) * t)});
                ++count;
            }
        }
        segToTimePose_.push_back(timeToPose);
    }
    constraintNum_ = getConstraintNum() + count;
}

// PS.cpp:241-307: integral over normalised time of the squared diffDegree-th derivative
void polyTrajSolver::constructP(std::vector<double>& P) const {
    const int n = paramDim_, D = polyDegree_ + 1;
    P.assign((size_t)n * n, 0.0);
    for (size_t s = 0; s + 1 < path_.size(); ++s)
        for (int i = diffDegree_; i < D; ++i)
            for (int j = diffDegree_; j < D; ++j) {
                double f = 1.0;
                for (int d = 0; d < diffDegree_; ++d) f *= (double)(i - d) * (j - d);
                f /= (double)(i + j - diffDegree_ * 2 + 1);
                P[(size_t)(s * D + i) * n + (s * D + j)] = f;
            }
}

// PS.cpp:314-584, same row order
void polyTrajSolver::constructA(std::vector<double>& A) const {
    const int n = paramDim_, D = polyDegree_ + 1, K = (int)path_.size() - 1;
    A.assign((size_t)constraintNum_ * n, 0.0);
    int row = 0;
    auto at = [&](int r, int c) -> double& { return A[(size_t)r * n + c]; };
    auto deriv = [](int d, int order, double t) -> double {  // d/dt^order of t^d
        if (d < order) return 0.0;
        double f = 1.0;
        for (int k = 0; k < order; ++k) f *= (d - k);
        return f * std::pow(t, d - order);
    };
    const int last = (K - 1) * D;
    // position: endpoints, K-1 midpoints, K-1 continuity
    for (int d = 0; d < D; ++d) at(row, d) = deriv(d, 0, 0.0);
    ++row;
    for (int d = 0; d < D; ++d) at(row, last + d) = deriv(d, 0, 1.0);
    ++row;
    for (int i = 0; i < K - 1; ++i, ++row)
        for (int d = 0; d < D; ++d) at(row, i * D + d) = deriv(d, 0, 1.0);
    for (int i = 0; i < K - 1; ++i, ++row)
        for (int d = 0; d < D; ++d) { at(row, i * D + d) = deriv(d, 0, 1.0); at(row, (i + 1) * D + d) -= deriv(d, 0, 0.0); }
    // velocity / acceleration: endpoints (normalised-time derivative, as the reference) + continuity
    for (int order = 1; order <= 2; ++order) {
        for (int d = 0; d < D; ++d) at(row, d) = deriv(d, order, 0.0);
        ++row;
        for (int d = 0; d < D; ++d) at(row, last + d) = deriv(d, order, 1.0);
        ++row;
        for (int i = 0; i < K - 1; ++i, ++row) {
            const double dtL = desiredTime_[i + 1] - desiredTime_[i], dtR = desiredTime_[i + 2] - desiredTime_[i + 1];
            for (int d = 0; d < D; ++d) {
                at(row, i * D + d) = deriv(d, order, 1.0) * std::pow(dtR, order);
                at(row, (i + 1) * D + d) = -deriv(d, order, 0.0) * std::pow(dtL, order);
            }
        }
    }
    // jerk, snap continuity
    for (int order = 3; order <= continuityDegree_; ++order)
        for (int i = 0; i < K - 1; ++i, ++row) {
            const double dtL = desiredTime_[i + 1] - desiredTime_[i], dtR = desiredTime_[i + 2] - desiredTime_[i + 1];
            for (int d = 0; d < D; ++d) {
                at(row, i * D + d) = deriv(d, order, 1.0) * std::pow(dtR, order);
                at(row, (i + 1) * D + d) = -deriv(d, order, 0.0) * std::pow(dtL, order);
            }
        }
    // corridor boxes
    if (corridorConstraint_)
        for (int i = 0; i < K; ++i)
            for (const auto& tp : segToTimePose_[i]) {
                for (int d = 0; d < D; ++d) at(row, i * D + d) = std::pow(tp.first, d);
                ++row;
            }
}

// PS.cpp:587-846
void polyTrajSolver::constructBound(std::vector<double> (&l)[3], std::vector<double> (&u)[3]) const {
    const int K = (int)path_.size() - 1;
    for (int a = 0; a < 3; ++a) { l[a].assign(constraintNum_, 0.0); u[a].assign(constraintNum_, 0.0); }
    auto coord = [](const pose& p, int a) { return a == 0 ? p.x : (a == 1 ? p.y : p.z); };
    int row = 0;
    auto eq = [&](int r, const double v[3]) { for (int a = 0; a < 3; ++a) l[a][r] = u[a][r] = v[a]; };
    {
        const double s[3] = {path_.front().x, path_.front().y, path_.front().z}, e[3] = {path_.back().x, path_.back().y, path_.back().z};
        eq(row++, s);
        eq(row++, e);
    }
    for (int i = 0; i < K - 1; ++i) { const double w[3] = {path_[i + 1].x, path_[i + 1].y, path_[i + 1].z}; eq(row++, w); }
    row += K - 1;                               // position continuity: 0
    eq(row++, initVel_); eq(row++, endVel_);
    row += K - 1;
    eq(row++, initAcc_); eq(row++, endAcc_);
    row += K - 1;
    row += (K - 1) * (continuityDegree_ - 2);   // jerk / snap continuity: 0
    if (corridorConstraint_)
        for (int i = 0; i < K; ++i)
            for (const auto& tp : segToTimePose_[i]) {
                for (int a = 0; a < 3; ++a) { l[a][row] = coord(tp.second, a) - corridorSizeVec_[i]; u[a][row] = coord(tp.second, a) + corridorSizeVec_[i]; }
                ++row;
            }
}

bool polyTrajSolver::solve() {
    if (path_.size() < 2) return false;
    std::vector<double> P, A, q(paramDim_, 0.0), l[3], u[3];
    constructP(P);
    constructA(A);
    constructBound(l, u);
    std::vector<double>* sol[3] = {&xSol_, &ySol_, &zSol_};
    bool ok = true;
    for (int a = 0; a < 3; ++a) {
        std::vector<double> x;
        const int maxIter = 20000;
        const int it = solveDenseQP(paramDim_, constraintNum_, P, q, A, l[a], u[a], x, 1e-7, maxIter);
        if (it < 0) { ok = false; continue; }   // keep the stale solution, like the reference
        if (it >= maxIter) ok = false;          // not converged (e.g. infeasible corridor): best iterate kept
        // PS.cpp:874-878: back to un-normalised local time
        for (size_t s = 0; s + 1 < path_.size(); ++s)
            for (int d = 0; d <= polyDegree_; ++d) x[s * (polyDegree_ + 1) + d] /= std::pow(desiredTime_[s + 1] - desiredTime_[s], d);
        *sol[a] = x;
    }
    solved_ = ok;
    return ok;
}

// PS.cpp:1026-1056
pose polyTrajSolver::getPose(double t) {
    pose p;
    if (xSol_.empty()) return p;
    for (size_t i = 0; i + 1 < desiredTime_.size(); ++i) {
        const double startTime = desiredTime_[i], endTime = desiredTime_[i + 1];
        if (t >= startTime && t <= endTime) {
            t = (double)(t - startTime);
            const int c0 = (polyDegree_ + 1) * (int)i;
            double x = 0, y = 0, z = 0;
            for (int d = 0; d < polyDegree_ + 1; ++d) {
                x += xSol_[c0 + d] * std::pow(t, d);
                y += ySol_[c0 + d] * std::pow(t, d);
                z += zSol_[c0 + d] * std::pow(t, d);
            }
            if (t == 0) t = 0.01;
            double dx = 0, dy = 0;
            for (int d = 0; d < polyDegree_ + 1; ++d) {
                dx += d * xSol_[c0 + d] * std::pow(t, d - 1);
                dy += d * ySol_[c0 + d] * std::pow(t, d - 1);
            }
            p.x = x; p.y = y; p.z = z; p.yaw = std::atan2(dy, dx);
            break;
        }
    }
    return p;
}

// PS.cpp:1125-1137
void polyTrajSolver::getTrajectory(std::vector<pose>& trajectory, double delT) {
    trajectory.clear();
    const double endTime = desiredTime_.back();
    for (double t = 0; t < endTime; t += delT) trajectory.push_back(getPose(t));
    trajectory.push_back(path_.back());
}

std::vector<double>& polyTrajSolver::getTimeKnot() { return desiredTime_; }

void polyTrajSolver::getCorridor(std::vector<std::vector<std::pair<double, pose>>>& segToTimePose, std::vector<double>& corridorSizeVec) const {
    segToTimePose = segToTimePose_;
    corridorSizeVec = corridorSizeVec_;
}

}  // namespace trajPlanner
