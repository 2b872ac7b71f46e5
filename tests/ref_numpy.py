"""Independent numpy restatement of the reference's QP FORMATION (not of the solver), written matrix-for-matrix after
S/ConvexMpc.cpp:110-245 and S/A1RobotControl.cpp:11-48,394-413 with dense numpy products.  It shares no code with
oracle/a1mpc_oracle.c or with the HIP kernel, so agreement of (P, q, A, l, u) and the KKT check of returned solutions
pin the formation of both.  TEST INFRASTRUCTURE."""
import numpy as np

INF = 1e30  # OsqpEigen::INFTY


def skew(v):  # S/utils/Utils.cpp:35-41
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def mpc_qp(params, h, x0, xref, R, foot, contact):
    """Dense (P, q, A, l, u) of one MPC tick.  R: (3,3), foot: (4,3) leg-major, contact: (4,)."""
    dt, mu, m = params["dt"], params["mu"], params["mass"]
    Ib = np.asarray(params["inertia"], float).reshape(3, 3)
    q = np.asarray(params["q"], float); r = np.asarray(params["r"], float)
    yaw = x0[2]
    c, s = np.cos(yaw), np.sin(yaw)
    Ac = np.zeros((13, 13))                       # calculate_A_mat_c :110-130
    Ac[0:3, 6:9] = [[c, s, 0], [-s, c, 0], [0, 0, 1]]
    Ac[3:6, 9:12] = np.eye(3)
    Ac[11, 12] = 1.0
    Iw = R @ Ib @ R.T                             # calculate_B_mat_c :132-143
    Bc = np.zeros((13, 12))
    for i in range(4):
        Bc[6:9, 3 * i:3 * i + 3] = np.linalg.inv(Iw) @ skew(foot[i])
        Bc[9:12, 3 * i:3 * i + 3] = np.eye(3) / m
    Ad = np.eye(13) + Ac * dt                     # state_space_discretization :145-156
    Bd = Bc * dt
    Aqp = np.zeros((13 * h, 13)); Bqp = np.zeros((13 * h, 12 * h))   # calculate_qp_mats :181-202
    for i in range(h):
        Aqp[13 * i:13 * i + 13] = np.linalg.matrix_power(Ad, i + 1)
        for j in range(i + 1):
            Bqp[13 * i:13 * i + 13, 12 * j:12 * j + 12] = np.linalg.matrix_power(Ad, i - j) @ Bd
    Q = np.diag(np.tile(2 * q, h)); Rm = np.diag(np.tile(2 * r, h))     # ctor :16-44
    P = Bqp.T @ Q @ Bqp + Rm                      # :207-210
    g = Bqp.T @ Q @ (Aqp @ x0 - xref)             # :215-217
    A = np.zeros((20 * h, 12 * h)); l = np.zeros(20 * h); u = np.zeros(20 * h)
    for i in range(4 * h):                        # ctor :46-58, bounds :223-245
        A[5 * i + 0, 3 * i + 0] = 1; A[5 * i + 0, 3 * i + 2] = mu
        A[5 * i + 1, 3 * i + 0] = 1; A[5 * i + 1, 3 * i + 2] = -mu
        A[5 * i + 2, 3 * i + 1] = 1; A[5 * i + 2, 3 * i + 2] = mu
        A[5 * i + 3, 3 * i + 1] = 1; A[5 * i + 3, 3 * i + 2] = -mu
        A[5 * i + 4, 3 * i + 2] = 1
        cf = float(contact[i % 4])
        l[5 * i:5 * i + 5] = [0, -INF, 0, -INF, params["fz_min"] * cf]
        u[5 * i:5 * i + 5] = [INF, 0, INF, 0, params["fz_max"] * cf]
    return P, g, A, l, u


def balance_qp(root_acc, Rz, foot, contact, Qw=(1, 1, 1, 400, 400, 100), Rw=1e-3, mu=0.7, Fmin=0.0, Fmax=180.0):
    """Dense (P, q, A, l, u) of the balance QP, S/A1RobotControl.cpp:11-48, 394-413."""
    M = np.zeros((6, 12))
    for i in range(4):
        M[0:3, 3 * i:3 * i + 3] = np.eye(3)
        M[3:6, 3 * i:3 * i + 3] = Rz.T @ skew(foot[i])
    Q = np.diag(np.asarray(Qw, float))
    P = Rw * np.eye(12) + M.T @ Q @ M
    q = -M.T @ Q @ root_acc
    A = np.zeros((20, 12)); l = np.zeros(20); u = np.zeros(20)
    for i in range(4):
        A[i, 2 + 3 * i] = 1
        l[i] = Fmin * float(contact[i]); u[i] = Fmax * float(contact[i])
        rr = 4 + 4 * i
        A[rr + 0, 3 * i] = 1; A[rr + 0, 3 * i + 2] = -mu
        A[rr + 1, 3 * i] = -1; A[rr + 1, 3 * i + 2] = -mu
        A[rr + 2, 3 * i + 1] = 1; A[rr + 2, 3 * i + 2] = -mu
        A[rr + 3, 3 * i + 1] = -1; A[rr + 3, 3 * i + 2] = -mu
        l[rr:rr + 4] = -INF; u[rr:rr + 4] = 0
    return P, q, A, l, u


def kkt_violation(P, q, A, l, u, x, tol_active=1e-6):
    """Certifies x as the QP optimum WITHOUT a dual from the solver: on the active set read off x, finds multipliers with
    the signs the KKT conditions demand (upper bound: lam >= 0, lower: lam <= 0, equality: free) by a bounded least
    squares fit of the stationarity equation.  Returns (stationarity residual, primal violation, 0.0)."""
    from scipy.optimize import lsq_linear
    ax = A @ x
    finite_l = l > -1e20; finite_u = u < 1e20
    at_l = finite_l & (np.abs(ax - l) <= tol_active); at_u = finite_u & (np.abs(ax - u) <= tol_active)
    act = at_l | at_u
    Aa = A[act]
    grad = P @ x + q
    if Aa.shape[0]:
        eq = (at_l & at_u)[act]
        lo = np.where(eq | at_l[act], -np.inf, 0.0)   # rows at the lower bound (or equalities) may have lam < 0
        hi = np.where(eq | at_u[act], np.inf, 0.0)    # rows at the upper bound (or equalities) may have lam > 0
        res = lsq_linear(Aa.T, -grad, bounds=(lo, hi), method="bvls", tol=1e-14)
        stat = np.abs(grad + Aa.T @ res.x).max()
    else:
        stat = np.abs(grad).max()
    viol = max(0.0, (l - ax).max(), (ax - u).max())
    return float(stat), float(viol), 0.0
