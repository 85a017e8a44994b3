"""numpy statement of the min-snap QP of polyTrajSolver (PS.cpp:241-846) used by the tests: the
matrices in normalised segment time, polynomial evaluation, and an algorithm-independent optimality
check (KKT conditions) for a candidate solution."""
import numpy as np


def minsnap_matrices(wp, deg, diff, cont, vel):
    """numpy statement of the QP (normalised time) for the equality-constrained case"""
    K = len(wp) - 1
    D = deg + 1
    T = np.concatenate([[0], np.cumsum(np.linalg.norm(np.diff(wp, axis=0), axis=1) / vel)])
    n = K * D
    P = np.zeros((n, n))
    for s in range(K):
        for i in range(diff, D):
            for j in range(diff, D):
                f = 1.0
                for d in range(diff):
                    f *= (i - d) * (j - d)
                P[s * D + i, s * D + j] = f / (i + j - 2 * diff + 1)

    def dv(d, order, t):
        if d < order:
            return 0.0
        f = 1.0
        for k in range(order):
            f *= d - k
        return f * t ** (d - order)

    rows, rhs = [], []

    def row(entries, b):
        r = np.zeros(n)
        for c, v in entries:
            r[c] += v
        rows.append(r)
        rhs.append(b)

    last = (K - 1) * D
    row([(d, dv(d, 0, 0.0)) for d in range(D)], wp[0])
    row([(last + d, dv(d, 0, 1.0)) for d in range(D)], wp[-1])
    for i in range(K - 1):
        row([(i * D + d, dv(d, 0, 1.0)) for d in range(D)], wp[i + 1])
    for i in range(K - 1):
        row([(i * D + d, dv(d, 0, 1.0)) for d in range(D)] + [((i + 1) * D + d, -dv(d, 0, 0.0)) for d in range(D)], np.zeros(3))
    for order in (1, 2):
        row([(d, dv(d, order, 0.0)) for d in range(D)], np.zeros(3))
        row([(last + d, dv(d, order, 1.0)) for d in range(D)], np.zeros(3))
        for i in range(K - 1):
            dl, dr = T[i + 1] - T[i], T[i + 2] - T[i + 1]
            row([(i * D + d, dv(d, order, 1.0) * dr ** order) for d in range(D)] +
                [((i + 1) * D + d, -dv(d, order, 0.0) * dl ** order) for d in range(D)], np.zeros(3))
    for order in range(3, cont + 1):
        for i in range(K - 1):
            dl, dr = T[i + 1] - T[i], T[i + 2] - T[i + 1]
            row([(i * D + d, dv(d, order, 1.0) * dr ** order) for d in range(D)] +
                [((i + 1) * D + d, -dv(d, order, 0.0) * dl ** order) for d in range(D)], np.zeros(3))
    return P, np.array(rows), np.array(rhs), T


def evaluate(coeffs, knots, t, deg=7):
    i = min(np.searchsorted(knots, t, side="right") - 1, len(knots) - 2)
    i = max(i, 0)
    lt = t - knots[i]
    c = coeffs[:, i * (deg + 1):(i + 1) * (deg + 1)]
    return c @ (lt ** np.arange(deg + 1))



def corridor_rows(wp, T, corridor, cres, deg=7):
    """PS.cpp:985-1012 + :565-576, :823-837: (row, centre[3], radius) per box, rows in normalised time"""
    D = deg + 1
    K = len(wp) - 1
    rows, cen, rad = [], [], []
    for i in range(K):
        if corridor[i] == 0.0:
            continue
        num = int(np.ceil((T[i + 1] - T[i]) * cres))
        dt = 1.0 / num
        t = 0.0
        while t <= 1.0:
            r = np.zeros(K * D)
            r[i * D:(i + 1) * D] = t ** np.arange(D)
            rows.append(r)
            cen.append(wp[i] + (wp[i + 1] - wp[i]) * t)
            rad.append(corridor[i])
            t += dt
    return np.array(rows).reshape(-1, K * D), np.array(cen).reshape(-1, 3), np.array(rad)


def kkt_violation(P, Aeq, beq, C, lo, hi, x, active_tol=1e-7):
    """max violation of the KKT conditions of  min 1/2 x'Px  s.t.  Aeq x = beq, lo <= C x <= hi  at x:
    (primal infeasibility, stationarity residual with sign-feasible multipliers on the active boxes).
    Algorithm independent: multipliers are found by non-negative least squares on the active set."""
    from scipy.optimize import nnls
    prim = (np.abs(Aeq @ x - beq) / np.abs(Aeq).max(axis=1)).max()   # rows carry dt^order factors
    if len(C):
        cx = C @ x
        prim = max(prim, (lo - cx).max(), (cx - hi).max())
        act_lo = np.where(np.abs(cx - lo) <= active_tol)[0]
        act_hi = np.where(np.abs(cx - hi) <= active_tol)[0]
    else:
        act_lo = act_hi = np.array([], dtype=int)
    # stationarity: P x + Aeq' nu - C_lo' mu_lo + C_hi' mu_hi = 0, mu >= 0; nu free -> split nu = nu+ - nu-
    cols = [Aeq.T, -Aeq.T]
    if len(act_lo):
        cols.append(-C[act_lo].T)
    if len(act_hi):
        cols.append(C[act_hi].T)
    G = np.concatenate(cols, axis=1)
    scale = max(1.0, np.abs(P @ x).max())
    _, resid = nnls(G, -(P @ x), maxiter=50 * G.shape[1])
    return prim, resid / scale
