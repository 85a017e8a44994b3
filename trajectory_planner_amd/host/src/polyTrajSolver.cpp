// polyTrajSolver.cpp — min-snap QP (structure of polyTrajSolver.cpp:125-138, :156-160, :241-307,
// :314-584, :587-846, :874-878, :985-1012, :1026-1056, :1125-1137 of the reference) and an exact
// dense QP solver (null-space elimination of the equalities + Goldfarb-Idnani dual active set)
// in place of the reference's OSQP.  Own implementation.
#include <trajectory_planner/polyTrajSolver.h>

#include <algorithm>
#include <cmath>
#include <iostream>

namespace trajPlanner {

// ---- small dense helpers (row-major) ------------------------------------------------------
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
static void cholSolve(const std::vector<double>& L, int n, double* b) {
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

// Strictly convex QP  min 1/2 w'Hw + c'w  s.t.  a_k'w >= b_k  by the dual active-set method of
// Goldfarb & Idnani (Math. Programming 27, 1983): start at the unconstrained minimum, add the most
// violated constraint, step in the primal/dual space, drop constraints whose multiplier reaches 0.
// The working-set systems are solved from H^-1 directly (dimension <= 2 x segments here), not with
// incremental factor updates.  Returns iterations, -1 numerical failure, -2 infeasible.
static int dualActiveSet(int nf, const std::vector<double>& Hinv, const std::vector<double>& c, int nc,
                         const std::vector<double>& Acon, const std::vector<double>& bcon, std::vector<double>& w) {
    w.assign(nf, 0.0);
    for (int i = 0; i < nf; ++i) {
        double s = 0;
        for (int j = 0; j < nf; ++j) s += Hinv[(size_t)i * nf + j] * c[j];
        w[i] = -s;
    }
    std::vector<int> act;          // working set (constraint indices)
    std::vector<double> u;         // their multipliers
    std::vector<char> inAct(nc, 0);
    std::vector<double> Hn(nf), z(nf), r, M, HN, rhs;
    auto slack = [&](int k) {
        double s = -bcon[k];
        for (int i = 0; i < nf; ++i) s += Acon[(size_t)k * nf + i] * w[i];
        return s;
    };
    const int maxIter = 50 * (nc + nf) + 100;
    int iter = 0;
    int polished = 0;
    // The primal point is carried along by increments t z.  On long corridors the working set walks from box
    // to neighbouring box through dozens of nearly degenerate partial steps and the increments leave w up to
    // 1e-5 (relative) off the face the working set defines — feasible, but measurably sub-optimal (found by
    // tools/fuzz_minsnap.py against scipy's trust-constr).  Once nothing is violated any more, w and the
    // multipliers are therefore recomputed from the working set itself: u = M^-1 (b_A + N H^-1 c),
    // w = H^-1 (N'u - c); should that uncover a violated box, the loop resumes from the polished point.
    auto polish = [&]() -> bool {
        const int q = (int)act.size();
        if (q == 0) return false;
        std::vector<double> hc(nf), HNp((size_t)nf * q), Mp((size_t)q * q), un(q);
        for (int i = 0; i < nf; ++i) {
            double s = 0;
            for (int j = 0; j < nf; ++j) s += Hinv[(size_t)i * nf + j] * c[j];
            hc[i] = s;
        }
        for (int j = 0; j < q; ++j) {
            const double* nj = &Acon[(size_t)act[j] * nf];
            for (int i = 0; i < nf; ++i) {
                double s = 0;
                for (int k = 0; k < nf; ++k) s += Hinv[(size_t)i * nf + k] * nj[k];
                HNp[(size_t)i * q + j] = s;
            }
        }
        for (int a = 0; a < q; ++a) {
            const double* na = &Acon[(size_t)act[a] * nf];
            for (int b = 0; b <= a; ++b) {
                double s = 0;
                for (int i = 0; i < nf; ++i) s += na[i] * HNp[(size_t)i * q + b];
                Mp[(size_t)a * q + b] = s;
            }
            double s = bcon[act[a]];
            for (int i = 0; i < nf; ++i) s += na[i] * hc[i];
            un[a] = s;
        }
        if (!cholesky(Mp, q)) return false;
        cholSolve(Mp, q, un.data());
        for (int a = 0; a < q; ++a)
            if (!(un[a] >= 0.0)) return false;              // not the optimal working set after all: keep the carried point
        for (int i = 0; i < nf; ++i) {
            double s = -hc[i];
            for (int a = 0; a < q; ++a) s += HNp[(size_t)i * q + a] * un[a];
            w[i] = s;
        }
        u = un;
        return true;
    };
    for (;;) {
        int ip = -1;
        double worst = 0;
        for (int k = 0; k < nc; ++k) {
            if (inAct[k]) continue;
            const double s = slack(k), tol = 1e-9 * (1.0 + std::fabs(bcon[k]));
            if (s < -tol && s < worst) { worst = s; ip = k; }
        }
        if (ip < 0) {
            if (polished < 2 && polish()) { ++polished; continue; }   // re-test the boxes at the polished point
            return iter;
        }
        const double* np = &Acon[(size_t)ip * nf];
        double uq = 0.0, sip = worst;
        for (;;) {
            if (++iter > maxIter) return -1;
            const int q = (int)act.size();
            for (int i = 0; i < nf; ++i) {
                double s = 0;
                for (int j = 0; j < nf; ++j) s += Hinv[(size_t)i * nf + j] * np[j];
                Hn[i] = s;
            }
            double npHn = 0;
            for (int i = 0; i < nf; ++i) npHn += np[i] * Hn[i];
            z = Hn;
            r.assign(q, 0.0);
            if (q > 0) {
                // HN = Hinv N (nf x q), M = N' Hinv N, rhs = N' Hn
                HN.assign((size_t)nf * q, 0.0);
                for (int j = 0; j < q; ++j) {
                    const double* nj = &Acon[(size_t)act[j] * nf];
                    for (int i = 0; i < nf; ++i) {
                        double s = 0;
                        for (int k = 0; k < nf; ++k) s += Hinv[(size_t)i * nf + k] * nj[k];
                        HN[(size_t)i * q + j] = s;
                    }
                }
                M.assign((size_t)q * q, 0.0);
                rhs.assign(q, 0.0);
                for (int a = 0; a < q; ++a) {
                    const double* na = &Acon[(size_t)act[a] * nf];
                    for (int b = 0; b <= a; ++b) {
                        double s = 0;
                        for (int i = 0; i < nf; ++i) s += na[i] * HN[(size_t)i * q + b];
                        M[(size_t)a * q + b] = s;
                    }
                    double s = 0;
                    for (int i = 0; i < nf; ++i) s += na[i] * Hn[i];
                    rhs[a] = s;
                }
                if (!cholesky(M, q)) return -1;
                cholSolve(M, q, rhs.data());
                r = rhs;
                for (int i = 0; i < nf; ++i) {
                    double s = 0;
                    for (int j = 0; j < q; ++j) s += HN[(size_t)i * q + j] * r[j];
                    z[i] -= s;
                }
            }
            double zn = 0;
            for (int i = 0; i < nf; ++i) zn += z[i] * np[i];
            double t1 = INFINITY, t2 = INFINITY;
            int drop = -1;
            for (int j = 0; j < q; ++j)
                if (r[j] > 0 && u[j] / r[j] < t1) { t1 = u[j] / r[j]; drop = j; }
            if (zn > 1e-11 * npHn) t2 = -sip / zn;
            const double t = t1 < t2 ? t1 : t2;
            if (!(t < INFINITY)) return -2;                     // no step possible: infeasible
            for (int j = 0; j < q; ++j) u[j] -= t * r[j];
            uq += t;
            if (t2 < INFINITY) for (int i = 0; i < nf; ++i) w[i] += t * z[i];
            if (t == t2) {                                       // full step: the constraint becomes active
                act.push_back(ip); u.push_back(uq); inAct[ip] = 1;
                break;
            }
            inAct[act[drop]] = 0;                                // partial step: drop the blocking constraint
            act.erase(act.begin() + drop);
            u.erase(u.begin() + drop);
            sip = slack(ip);
        }
    }
}

// min 1/2 x'Px + q'x  s.t.  l <= Ax <= u.  Rows with l == u are eliminated with an orthonormal
// null-space basis (Householder QR of A_eq'): for the min-snap problem that leaves 2 free
// coefficients per segment, a tiny strictly convex QP in which only the corridor boxes remain —
// solved exactly by the dual active-set method above.
//
// The three axes of a path share P, A and the split of the rows into equalities and boxes; only the
// right-hand sides differ.  Everything that depends on the matrices alone — row scaling, the QR, Z'PZ and
// its inverse, the box normals in the reduced space — is computed once (factor) and reused per axis (solve).
namespace {
struct ReducedQP {
    int n = 0, m = 0, me = 0, nf = 0;
    std::vector<int> eqRows, inRows;
    std::vector<double> mx;        // infinity norm of each equality row (the rows are scaled by it)
    std::vector<double> Rt;        // n x me: the R factor of A_eq' = Q R in its upper triangle
    std::vector<double> Q;         // n x n orthogonal; Z = Q[:, me:]
    std::vector<double> PZ, Hinv;  // n x nf, nf x nf
    std::vector<double> AZ;        // |inRows| x nf: box rows in the reduced space
    std::vector<double> Pm, Am;    // the matrices this factorisation belongs to (own copies: the cache key)
    const std::vector<double>* P = nullptr;
    const std::vector<double>* A = nullptr;

    double Zc(int i, int k) const { return Q[(size_t)i * n + me + k]; }

    // returns 0, or -1 on rank-deficient equalities / a reduced Hessian that is not positive definite
    int factor(int n_, int m_, const std::vector<double>& P_, const std::vector<double>& A_, const std::vector<double>& l,
               const std::vector<double>& u) {
        n = n_; m = m_;
        Pm = P_; Am = A_;
        P = &Pm; A = &Am;
        eqRows.clear(); inRows.clear();
        for (int r = 0; r < m; ++r) {
            if (l[r] == u[r]) eqRows.push_back(r);
            else if (l[r] > -INFINITY || u[r] < INFINITY) inRows.push_back(r);
        }
        me = (int)eqRows.size();
        if (me > n) return -1;
        nf = n - me;
        // Rt = A_eq' (n x me), rows scaled to unit infinity norm first (continuity rows carry dt^4)
        Rt.assign((size_t)n * me, 0.0);
        mx.assign(me, 0.0);
        for (int j = 0; j < me; ++j) {
            const double* a = &A_[(size_t)eqRows[j] * n];
            double mj = 0;
            for (int i = 0; i < n; ++i) mj = std::max(mj, std::fabs(a[i]));
            if (!(mj > 0)) return -1;
            for (int i = 0; i < n; ++i) Rt[(size_t)i * me + j] = a[i] / mj;
            mx[j] = mj;
        }
        // Householder QR: Rt = Q R; Q accumulated explicitly (n x n)
        Q.assign((size_t)n * n, 0.0);
        std::vector<double> v(n);
        for (int i = 0; i < n; ++i) Q[(size_t)i * n + i] = 1.0;
        for (int j = 0; j < me; ++j) {
            double nrm = 0;
            for (int i = j; i < n; ++i) nrm += Rt[(size_t)i * me + j] * Rt[(size_t)i * me + j];
            nrm = std::sqrt(nrm);
            if (!(nrm > 1e-10)) return -1;                          // dependent equality rows
            const double alpha = Rt[(size_t)j * me + j] > 0 ? -nrm : nrm;
            for (int i = 0; i < n; ++i) v[i] = i < j ? 0.0 : Rt[(size_t)i * me + j];
            v[j] -= alpha;
            double vv = 0;
            for (int i = j; i < n; ++i) vv += v[i] * v[i];
            if (vv > 0) {
                for (int c = j; c < me; ++c) {                      // Rt <- (I - 2vv'/v'v) Rt
                    double sc = 0;
                    for (int i = j; i < n; ++i) sc += v[i] * Rt[(size_t)i * me + c];
                    sc *= 2.0 / vv;
                    for (int i = j; i < n; ++i) Rt[(size_t)i * me + c] -= sc * v[i];
                }
                for (int rr = 0; rr < n; ++rr) {                    // Q <- Q (I - 2vv'/v'v)
                    double sc = 0;
                    for (int i = j; i < n; ++i) sc += Q[(size_t)rr * n + i] * v[i];
                    sc *= 2.0 / vv;
                    for (int i = j; i < n; ++i) Q[(size_t)rr * n + i] -= sc * v[i];
                }
            }
        }
        if (nf == 0) return 0;
        // reduced Hessian H = Z'PZ and its inverse
        PZ.assign((size_t)n * nf, 0.0);
        for (int i = 0; i < n; ++i)
            for (int jj = 0; jj < n; ++jj) {
                const double pij = P_[(size_t)i * n + jj];
                if (pij == 0.0) continue;
                for (int k = 0; k < nf; ++k) PZ[(size_t)i * nf + k] += pij * Zc(jj, k);
            }
        std::vector<double> H((size_t)nf * nf, 0.0);
        for (int a = 0; a < nf; ++a)
            for (int b = 0; b < nf; ++b) {
                double sc = 0;
                for (int i = 0; i < n; ++i) sc += Zc(i, a) * PZ[(size_t)i * nf + b];
                H[(size_t)a * nf + b] = sc;
            }
        for (int a = 0; a < nf; ++a)
            for (int b = 0; b < a; ++b) H[(size_t)a * nf + b] = H[(size_t)b * nf + a] = 0.5 * (H[(size_t)a * nf + b] + H[(size_t)b * nf + a]);
        if (!cholesky(H, nf)) return -1;
        Hinv.assign((size_t)nf * nf, 0.0);
        std::vector<double> col(nf);
        for (int k = 0; k < nf; ++k) {
            std::fill(col.begin(), col.end(), 0.0);
            col[k] = 1.0;
            cholSolve(H, nf, col.data());
            for (int i = 0; i < nf; ++i) Hinv[(size_t)i * nf + k] = col[i];
        }
        // box rows in the reduced space: A_r Z
        AZ.assign(inRows.size() * (size_t)nf, 0.0);
        for (size_t ri = 0; ri < inRows.size(); ++ri) {
            const double* a = &A_[(size_t)inRows[ri] * n];
            for (int i = 0; i < n; ++i) {
                if (a[i] == 0.0) continue;
                for (int k = 0; k < nf; ++k) AZ[ri * nf + k] += a[i] * Zc(i, k);
            }
        }
        return 0;
    }

    // made for exactly these matrices (bitwise) — then only the row split remains to be compared
    bool sameMatrices(int n_, int m_, const std::vector<double>& P_, const std::vector<double>& A_) const {
        return n_ == n && m_ == m && P_ == Pm && A_ == Am;
    }

    // the row split of (l, u) is the one this factorisation was made for
    bool sameSplit(const std::vector<double>& l, const std::vector<double>& u) const {
        size_t e = 0, q = 0;
        for (int r = 0; r < m; ++r) {
            if (l[r] == u[r]) { if (e >= eqRows.size() || eqRows[e++] != r) return false; }
            else if (l[r] > -INFINITY || u[r] < INFINITY) { if (q >= inRows.size() || inRows[q++] != r) return false; }
        }
        return e == eqRows.size() && q == inRows.size();
    }

    // iterations >= 0, -1 numerical failure, -2 infeasible
    int solve(const std::vector<double>& q, const std::vector<double>& l, const std::vector<double>& u, std::vector<double>& x) const {
        for (int r = 0; r < m; ++r)
            if (l[r] > u[r]) return -2;
        // particular solution x0 = Y R^-T beq  (A_eq = R'Q' => (Q'x)[:me] = R^-T beq)
        std::vector<double> y(me), x0(n, 0.0);
        for (int i = 0; i < me; ++i) {
            double sc = l[eqRows[i]] / mx[i];
            for (int k = 0; k < i; ++k) sc -= Rt[(size_t)k * me + i] * y[k];
            y[i] = sc / Rt[(size_t)i * me + i];
        }
        for (int i = 0; i < n; ++i) {
            double sc = 0;
            for (int k = 0; k < me; ++k) sc += Q[(size_t)i * n + k] * y[k];
            x0[i] = sc;
        }
        x = x0;
        if (nf == 0) {
            for (int r : inRows) {
                double sc = 0;
                for (int i = 0; i < n; ++i) sc += (*A)[(size_t)r * n + i] * x[i];
                if (sc < l[r] - 1e-9 || sc > u[r] + 1e-9) return -2;
            }
            return 0;
        }
        // reduced gradient c = Z'(P x0 + q)
        std::vector<double> g(n, 0.0), c(nf, 0.0);
        for (int i = 0; i < n; ++i) {
            double sc = q[i];
            for (int jj = 0; jj < n; ++jj) {
                const double pij = (*P)[(size_t)i * n + jj];
                if (pij != 0.0) sc += pij * x0[jj];
            }
            g[i] = sc;
        }
        for (int a = 0; a < nf; ++a) {
            double sc = 0;
            for (int i = 0; i < n; ++i) sc += Zc(i, a) * g[i];
            c[a] = sc;
        }
        // inequality rows in w: lo <= (A_r Z) w + A_r x0 <= hi, as a'w >= b pairs
        std::vector<double> Acon, bcon;
        Acon.reserve(2 * AZ.size());
        for (size_t ri = 0; ri < inRows.size(); ++ri) {
            const int r = inRows[ri];
            const double* a = &(*A)[(size_t)r * n];
            double d0 = 0;
            for (int i = 0; i < n; ++i)
                if (a[i] != 0.0) d0 += a[i] * x0[i];
            const double* az = &AZ[ri * nf];
            if (l[r] > -INFINITY) { Acon.insert(Acon.end(), az, az + nf); bcon.push_back(l[r] - d0); }
            if (u[r] < INFINITY) { for (int k = 0; k < nf; ++k) Acon.push_back(-az[k]); bcon.push_back(d0 - u[r]); }
        }
        std::vector<double> w;
        const int it = dualActiveSet(nf, Hinv, c, (int)bcon.size(), Acon, bcon, w);
        if (it < 0) return it;
        // The active-set loop only tests constraints outside its working set.  On an infeasible corridor
        // rounding can hide the vanishing step direction (z'n ~ 1e-11 n'H^-1 n) and let it "finish" on an
        // ill-conditioned working set: every box is verified at the end, a violation means infeasible.
        for (size_t k = 0; k < bcon.size(); ++k) {
            double sl = -bcon[k];
            for (int i = 0; i < nf; ++i) sl += Acon[k * nf + i] * w[i];
            if (sl < -1e-7 * (1.0 + std::fabs(bcon[k]))) return -2;
        }
        for (int i = 0; i < n; ++i) {
            double sc = 0;
            for (int k = 0; k < nf; ++k) sc += Zc(i, k) * w[k];
            x[i] = x0[i] + sc;
        }
        return it;
    }
};
}  // namespace

// one problem, factor + solve.  Returns iterations >= 0, -1 numerical failure (rank-deficient equalities),
// -2 infeasible.
int solveDenseQP(int n, int m, const std::vector<double>& P, const std::vector<double>& q, const std::vector<double>& A,
                 const std::vector<double>& l, const std::vector<double>& u, std::vector<double>& x) {
    for (int r = 0; r < m; ++r)
        if (l[r] > u[r]) return -2;
    ReducedQP qp;
    const int rc = qp.factor(n, m, P, A, l, u);
    if (rc < 0) return rc;
    return qp.solve(q, l, u, x);
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

// PS.cpp:943-958
void polyTrajSolver::setSoftConstraint(double r) { setSoftConstraint(r, r, r); }
void polyTrajSolver::setSoftConstraint(double rx, double ry) { setSoftConstraint(rx, ry, 0.0); }
void polyTrajSolver::setSoftConstraint(double rx, double ry, double rz) {
    softConstraint_ = true;
    scDeviation_[0] = rx; scDeviation_[1] = ry; scDeviation_[2] = rz;
}

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
    for (int i = 0; i < K - 1; ++i) {
        const double w[3] = {path_[i + 1].x, path_[i + 1].y, path_[i + 1].z};
        if (!softConstraint_) eq(row, w);
        else for (int a = 0; a < 3; ++a) { l[a][row] = w[a] - scDeviation_[a]; u[a][row] = w[a] + scDeviation_[a]; }   // PS.cpp:644-659
        ++row;
    }
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
    // the three axes share the matrices and the equality/box split of the rows: one factorisation — which
    // also survives from call to call while the path and the corridor rows stay the same (the corridor loop
    // only shrinks radii, i.e. changes l and u): kept per host thread, keyed by the matrices themselves
    static thread_local ReducedQP qp;
    static thread_local int frcKept = -1;
    int frc;
    if (frcKept == 0 && qp.sameMatrices(paramDim_, constraintNum_, P, A) && qp.sameSplit(l[0], u[0])) frc = 0;
    else frc = frcKept = qp.factor(paramDim_, constraintNum_, P, A, l[0], u[0]);
    for (int a = 0; a < 3; ++a) {
        std::vector<double> x;
        int it = frc;
        if (frc == 0) it = qp.sameSplit(l[a], u[a]) ? qp.solve(q, l[a], u[a], x) : solveDenseQP(paramDim_, constraintNum_, P, q, A, l[a], u[a], x);
        if (it < 0) { ok = false; continue; }   // infeasible corridor / degenerate path: keep the stale solution, like the reference
        // PS.cpp:874-878: back to un-normalised local time
        for (size_t s = 0; s + 1 < path_.size(); ++s)
            for (int d = 0; d <= polyDegree_; ++d) x[s * (polyDegree_ + 1) + d] /= std::pow(desiredTime_[s + 1] - desiredTime_[s], d);
        *sol[a] = x;
    }
    solved_ = ok;
    return ok;
}

void polyTrajSolver::installSolution(const std::vector<double>& x, const std::vector<double>& y, const std::vector<double>& z) {
    xSol_ = x; ySol_ = y; zSol_ = z;
    solved_ = true;
}

// PS.cpp:1026-1056
pose polyTrajSolver::getPose(double t) {
    pose p;
    if (!hasSolution()) return p;
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

// PS.cpp:1080-1100.  (Outside every segment the reference returns an uninitialised vector; here it is zero.)
Eigen::Vector3d polyTrajSolver::getVel(double t) {
    Eigen::Vector3d vel(0, 0, 0);
    if (!hasSolution()) return vel;
    for (size_t i = 0; i + 1 < desiredTime_.size(); ++i) {
        if (t >= desiredTime_[i] && t <= desiredTime_[i + 1]) {
            t = (double)(t - desiredTime_[i]);
            const int c0 = (polyDegree_ + 1) * (int)i;
            double vx = 0, vy = 0, vz = 0;
            for (int d = 1; d < polyDegree_ + 1; ++d) {
                vx += xSol_[c0 + d] * d * std::pow(t, d - 1);
                vy += ySol_[c0 + d] * d * std::pow(t, d - 1);
                vz += zSol_[c0 + d] * d * std::pow(t, d - 1);
            }
            vel = Eigen::Vector3d(vx, vy, vz);
            break;
        }
    }
    return vel;
}

// PS.cpp:1102-1122 as written: the x component multiplies by pow(t, d-1) where y and z use pow(t, d-2) (:1112) — a caller
// of the reference gets exactly this, so a drop-in returns it too.
Eigen::Vector3d polyTrajSolver::getAcc(double t) {
    Eigen::Vector3d acc(0, 0, 0);
    if (!hasSolution()) return acc;
    for (size_t i = 0; i + 1 < desiredTime_.size(); ++i) {
        if (t >= desiredTime_[i] && t <= desiredTime_[i + 1]) {
            t = (double)(t - desiredTime_[i]);
            const int c0 = (polyDegree_ + 1) * (int)i;
            double ax = 0, ay = 0, az = 0;
            for (int d = 2; d < polyDegree_ + 1; ++d) {
                ax += xSol_[c0 + d] * d * (d - 1) * std::pow(t, d - 1);
                ay += ySol_[c0 + d] * d * (d - 1) * std::pow(t, d - 2);
                az += zSol_[c0 + d] * d * (d - 1) * std::pow(t, d - 2);
            }
            acc = Eigen::Vector3d(ax, ay, az);
            break;
        }
    }
    return acc;
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
