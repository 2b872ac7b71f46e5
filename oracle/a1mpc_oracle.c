/*
 * a1mpc_oracle.c -- CPU ORACLE for the convex-MPC QP hot path.
 *
 * >>> TEST INFRASTRUCTURE ONLY. <<<
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the timed CPU baseline.  The product path
 * (a1-qp-mpc-controller_amd/csrc) never links, loads or calls anything in oracle/.
 *
 * >>> PARITY: FORMATION AND CALLER-SIDE ROWS PINNED TO THE REFERENCE, THE OSQP SOLVE UNPINNED. <<<
 * Pinned (round 2): `make -C oracle ref` compiles the reference's own sources -- S/ConvexMpc.cpp, S/A1RobotControl.cpp,
 * S/A1BasicEKF.cpp, S/utils/Utils.cpp, S/utils/filter.hpp, S/legKinematics/A1Kinematics.cpp, S/test/test_mpc.cpp -- verbatim
 * from /root/reference into oracle/_ref/liba1ref*.so over the stand-in headers of oracle/ref_shim/ (a mini-Eigen, no-op ROS
 * names, an OsqpEigen::Solver that calls orc_osqp_solve below), and tests/test_ref_pin.py drives both with the same numbers:
 * (P, g, A, l, u) equal to <= 1e-15 relative at h = 10 / 16 / 20 (the latter two through the one-macro PLAN_HORIZON edit) incl.
 * per-step B_d; test_mpc.cpp as written; compute_grf's MPC branch over a warm-started sequence and its balance branch; the
 * update_plan -> swing legs -> contacts / filters / terrain -> MPC -> joint torques chain over 150 ticks (element-wise rows bit for
 * bit); leg kinematics; the EKF.  That pins the reference's SOURCE LOGIC.  It does not pin the rounding of Eigen's kernels (the
 * stand-in's products are plain ascending-k sums; Eigen is not installed here).
 * Unpinned: the SOLVE.  The reference pins no expected values (S/test/test_mpc.cpp:157-161 only prints) and its arithmetic for
 * the solve lives in third-party OSQP (oxfordcontrol/osqp, fetched UNPINNED at image build: docker/Dockerfile:77,91; install log
 * shows libOsqpEigen.so.0.6.3, docker/Dockerfile:98 => OSQP 0.6.x), absent from /root/reference and from this machine (no
 * network, no wheel).  This file therefore
 *   (1) restates the reference's own QP formation loop-for-loop in plain arrays
 *       (S/ConvexMpc.cpp:7-58,110-156,181-245; S/A1RobotControl.cpp:11-48,377-413,
 *        439-444,452-488,498-514,555-561; S/utils/Utils.cpp:35-41), and
 *   (2) restates the published OSQP 0.6 algorithm (Stellato et al., "OSQP: an operator
 *       splitting solver for quadratic programs", Math. Prog. Comp. 2020, and the 0.6.x
 *       source layout: scaling.c scale_data, auxil.c set_rho_vec / update_xz_tilde /
 *       update_x / update_z / update_y / compute_pri_res / compute_dua_res /
 *       compute_rho_estimate / adapt_rho / check_termination / is_*_infeasible,
 *       osqp.c osqp_solve) with OSQP's default settings.
 * The one documented deviation: OSQP's default adaptive_rho_interval=0 derives the
 * rho-update period from WALL-CLOCK setup time (osqp.c, PROFILING branch), i.e. the
 * reference is not run-to-run reproducible.  The oracle uses a fixed iteration period
 * (default 25 = the value OSQP's own rounding rule c_max(c_roundmultiple(iter,25),25)
 * yields whenever 0.4*setup_time is worth < 38 iterations).
 * Data handling follows OSQP's: only the upper triangle of P is used (mirrored), scaling multiplies rows first and then
 * columns (mat_premult_diag / mat_postmult_diag).  The linear system has two back ends (settings->linsys): 0 = its reduced
 * form (P+sigma*I+A' diag(rho) A) by dense Cholesky -- algebraically what QDLDL's AMD ordering does to this KKT matrix (the
 * degree-2 constraint rows are eliminated first), z_tilde recovered through nu as solve_linsys_qdldl does; 1 = LDL' of the full
 * quasi-definite KKT matrix in natural order.  Both give the same iteration count and status on every QP of the soak
 * (profiles/r02_linsys_soak.json: 100 352 QPs, h = 10 / 16 / 20, three weight sets, zero mismatches); the worst QP of a 2048-QP batch differs by 1e-5 N (median batch) to 1.1e-3 N (worst batch, isaac weight set) -- the resolution any "same answer as OSQP" claim has.
 *
 * Pinned instead (tests/test_oracle_*.py): KKT optimality of the tight mode, an
 * independent scipy solve, analytic stand cases, and golden vectors in tests/golden/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#ifdef ORC_EXTENDED
/* Accuracy yardstick (tests/tools/extended_check.py, make liba1mpc_oracle_x87.so): the same source with every `double` below widened to the
 * x87 80-bit long double (64-bit significand) -- the OSQP iterate sequence with 2048x less rounding error, against which the
 * double-precision oracle and the GPU engine are both measured.  Never used as the pass/fail checker. */
#include <tgmath.h>
#define double long double
#endif

#define NS 13 /* MPC_STATE_DIM      S/A1Params.h:27 */
#define NU 12 /* NUM_DOF            S/A1Params.h:34 */
#define NC 20 /* MPC_CONSTRAINT_DIM S/A1Params.h:28 */
#define NLEG 4

/* OSQP 0.6 constants (include/constants.h) */
#define OSQP_INFTY 1e30
#define OSQP_RHO_MIN 1e-6
#define OSQP_RHO_MAX 1e6
#define OSQP_RHO_EQ_OVER_RHO_INEQ 1e3
#define OSQP_RHO_TOL 1e-4
#define OSQP_MIN_SCALING 1e-4
#define OSQP_MAX_SCALING 1e4

#include "a1mpc_oracle_api.h" /* status codes, orc_settings, orc_info (shared with the oracle/_ref OsqpEigen stand-in) */

/* OSQP defaults (osqp/include/constants.h 0.6.x); warm_start as the MPC call site sets it. */
void orc_default_settings(orc_settings *s) {
    s->rho = 0.1; s->sigma = 1e-6; s->alpha = 1.6;
    s->eps_abs = 1e-3; s->eps_rel = 1e-3; s->eps_prim_inf = 1e-4; s->eps_dual_inf = 1e-4;
    s->adaptive_rho_tolerance = 5.0;
    s->max_iter = 4000; s->scaling = 10; s->check_termination = 25;
    s->adaptive_rho = 1; s->adaptive_rho_interval = 25; s->warm_start = 0; s->linsys = 0; s->reserved_ = 0;
}

/* ------------------------------------------------------------------------------------------
 * Generic OSQP-0.6 restatement: min 1/2 x'Px + q'x  s.t.  l <= Ax <= u
 * P dense symmetric n*n (row-major), A in CSR.
 * ---------------------------------------------------------------------------------------- */
static double vmaxabs(const double *v, int n) {
    double m = 0; for (int i = 0; i < n; ++i) { double a = fabs(v[i]); if (a > m) m = a; } return m;
}
static double limit_scaling1(double v) { /* scaling.c limit_scaling */
    v = v < OSQP_MIN_SCALING ? 1.0 : v;
    v = v > OSQP_MAX_SCALING ? OSQP_MAX_SCALING : v;
    return v;
}

/* dense lower Cholesky in place (row-major, lower part used); returns 0 ok */
static int chol_lower(double *K, int n) {
    for (int j = 0; j < n; ++j) {
        double d = K[j * n + j];
        for (int k = 0; k < j; ++k) d -= K[j * n + k] * K[j * n + k];
        if (!(d > 0.0)) return 1;
        d = sqrt(d);
        K[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = K[i * n + j];
            const double *ri = K + i * n, *rj = K + j * n;
            for (int k = 0; k < j; ++k) s -= ri[k] * rj[k];
            K[i * n + j] = s / d;
        }
    }
    return 0;
}
static void chol_solve(const double *L, int n, double *b) {
    for (int i = 0; i < n; ++i) {
        double s = b[i]; const double *ri = L + i * n;
        for (int k = 0; k < i; ++k) s -= ri[k] * b[k];
        b[i] = s / ri[i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * b[k];
        b[i] = s / L[i * n + i];
    }
}

static void csr_mv(int m, const int32_t *rp, const int32_t *ci, const double *av, const double *x, double *y) {
    for (int i = 0; i < m; ++i) { double s = 0; for (int k = rp[i]; k < rp[i + 1]; ++k) s += av[k] * x[ci[k]]; y[i] = s; }
}
static void csr_mtv(int m, int n, const int32_t *rp, const int32_t *ci, const double *av, const double *y, double *x) {
    for (int j = 0; j < n; ++j) x[j] = 0;
    for (int i = 0; i < m; ++i) for (int k = rp[i]; k < rp[i + 1]; ++k) x[ci[k]] += av[k] * y[i];
}
static void sym_mv(int n, const double *P, const double *x, double *y) {
    for (int i = 0; i < n; ++i) { double s = 0; const double *r = P + i * n; for (int j = 0; j < n; ++j) s += r[j] * x[j]; y[i] = s; }
}

typedef struct {
    int n, m, nnz;
    const int32_t *rp, *ci;
    double *P, *q, *av, *l, *u;           /* scaled data */
    double *D, *Dinv, *E, *Einv, c, cinv; /* scaling */
    double *rho_vec, *rho_inv_vec; int *ctype;
    double *K;                            /* Cholesky factor of reduced KKT */
    double *KK, *KKd, *kb;                /* linsys = 1: LDL' of the full KKT (lower, row-major (n+m)^2), its D, a right-hand side */
    double *x, *z, *y, *x_prev, *z_prev, *xt, *zt, *delta_x, *delta_y;
    double *Ax, *Px, *Aty, *tn, *tm;
    double rho;
    const orc_settings *st;
    orc_info *info;
} work_t;

/* scaling.c scale_data */
static void scale_data(work_t *w) {
    int n = w->n, m = w->m;
    double *Dt = w->tn, *Et = w->tm;
    w->c = 1.0;
    for (int i = 0; i < n; ++i) w->D[i] = 1.0;
    for (int i = 0; i < m; ++i) w->E[i] = 1.0;
    for (int it = 0; it < w->st->scaling; ++it) {
        /* compute_inf_norm_cols_KKT */
        for (int j = 0; j < n; ++j) Dt[j] = 0;
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double a = fabs(w->P[i * n + j]); if (a > Dt[j]) Dt[j] = a; }
        for (int i = 0; i < m; ++i) {
            double e = 0;
            for (int k = w->rp[i]; k < w->rp[i + 1]; ++k) { double a = fabs(w->av[k]); if (a > e) e = a; if (a > Dt[w->ci[k]]) Dt[w->ci[k]] = a; }
            Et[i] = e;
        }
        for (int j = 0; j < n; ++j) Dt[j] = 1.0 / sqrt(limit_scaling1(Dt[j]));
        for (int i = 0; i < m; ++i) Et[i] = 1.0 / sqrt(limit_scaling1(Et[i]));
        /* P <- D P D ; A <- E A D ; q <- D q */
        /* OSQP stores the upper triangle and scales it rows first, then columns (mat_premult_diag, mat_postmult_diag):
         * P_ij <- (P_ij * D_i) * D_j for i <= j; the lower triangle is its mirror.  A likewise: (A_ij * E_i) * D_j. */
        for (int i = 0; i < n; ++i) for (int j = i; j < n; ++j) { double v = (w->P[i * n + j] * Dt[i]) * Dt[j]; w->P[i * n + j] = v; w->P[j * n + i] = v; }
        for (int i = 0; i < m; ++i) for (int k = w->rp[i]; k < w->rp[i + 1]; ++k) w->av[k] = (w->av[k] * Et[i]) * Dt[w->ci[k]];
        for (int j = 0; j < n; ++j) { w->q[j] *= Dt[j]; w->D[j] *= Dt[j]; }
        for (int i = 0; i < m; ++i) w->E[i] *= Et[i];
        /* cost normalisation */
        double mean = 0;
        for (int j = 0; j < n; ++j) { double cm = 0; for (int i = 0; i < n; ++i) { double a = fabs(w->P[i * n + j]); if (a > cm) cm = a; } mean += cm; }
        mean /= n;
        double nq = limit_scaling1(vmaxabs(w->q, n));
        double ct = mean > nq ? mean : nq;
        ct = 1.0 / limit_scaling1(ct);
        for (int i = 0; i < n * n; ++i) w->P[i] *= ct;
        for (int j = 0; j < n; ++j) w->q[j] *= ct;
        w->c *= ct;
    }
    w->cinv = 1.0 / w->c;
    for (int j = 0; j < n; ++j) w->Dinv[j] = 1.0 / w->D[j];
    for (int i = 0; i < m; ++i) { w->Einv[i] = 1.0 / w->E[i]; w->l[i] *= w->E[i]; w->u[i] *= w->E[i]; }
}

/* auxil.c set_rho_vec */
static void set_rho_vec(work_t *w) {
    w->rho = fmin(fmax(w->rho, OSQP_RHO_MIN), OSQP_RHO_MAX);
    for (int i = 0; i < w->m; ++i) {
        if (w->l[i] < -OSQP_INFTY * OSQP_MIN_SCALING && w->u[i] > OSQP_INFTY * OSQP_MIN_SCALING) { w->ctype[i] = -1; w->rho_vec[i] = OSQP_RHO_MIN; }
        else if (w->u[i] - w->l[i] < OSQP_RHO_TOL) { w->ctype[i] = 1; w->rho_vec[i] = OSQP_RHO_EQ_OVER_RHO_INEQ * w->rho; }
        else { w->ctype[i] = 0; w->rho_vec[i] = w->rho; }
        w->rho_inv_vec[i] = 1.0 / w->rho_vec[i];
    }
}

/* Second back end (settings->linsys = 1): LDL' of the full quasi-definite KKT matrix [P + sigma I, A'; A, -diag(1/rho)] as OSQP's
 * QDLDL factors it, dense and in natural order (QDLDL's AMD permutation cannot be reproduced without AMD; a quasi-definite matrix
 * has an LDL' factorisation under every symmetric permutation, so only the rounding differs).  It exists to show that iteration
 * counts and statuses do not depend on which of the two algebraically equal solves is used (tests/tools/linsys_soak.py). */
static int kkt_factor(work_t *w) {
    const int n = w->n, m = w->m, N = n + m;
    double *K = w->KK, *d = w->KKd;
    memset(K, 0, sizeof(double) * (size_t)N * N);
    for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) K[(size_t)i * N + j] = w->P[i * n + j];
    for (int j = 0; j < n; ++j) K[(size_t)j * N + j] += w->st->sigma;
    for (int i = 0; i < m; ++i) {
        for (int k = w->rp[i]; k < w->rp[i + 1]; ++k) K[(size_t)(n + i) * N + w->ci[k]] = w->av[k];
        K[(size_t)(n + i) * N + n + i] = -w->rho_inv_vec[i];
    }
    for (int j = 0; j < N; ++j) {          /* row-oriented LDL', lower triangle in place, unit diagonal implied */
        double *rj = K + (size_t)j * N;
        double dj = rj[j];
        for (int k = 0; k < j; ++k) dj -= rj[k] * rj[k] * d[k];
        if (dj == 0.0 || isnan(dj)) return 1;
        d[j] = dj;
        for (int i = j + 1; i < N; ++i) {
            double *ri = K + (size_t)i * N;
            double sv = ri[j];
            for (int k = 0; k < j; ++k) sv -= ri[k] * rj[k] * d[k];
            ri[j] = sv / dj;
        }
    }
    for (int j = 0; j < n; ++j) if (!(d[j] > 0.0)) return 1;   /* OSQP: wrong number of positive pivots => non-convex */
    return 0;
}
static void kkt_solve(const work_t *w, double *b) {
    const int N = w->n + w->m; const double *K = w->KK, *d = w->KKd;
    for (int i = 0; i < N; ++i) { double sv = b[i]; const double *ri = K + (size_t)i * N; for (int k = 0; k < i; ++k) sv -= ri[k] * b[k]; b[i] = sv; }
    for (int i = 0; i < N; ++i) b[i] /= d[i];
    for (int i = N - 1; i >= 0; --i) { double sv = b[i]; for (int k = i + 1; k < N; ++k) sv -= K[(size_t)k * N + i] * b[k]; b[i] = sv; }
}

/* reduced KKT: K = P + sigma I + A' diag(rho) A, Cholesky */
static int factor(work_t *w) {
    int n = w->n;
    if (w->st->linsys == 1) { w->info->nfact++; return kkt_factor(w); }
    memcpy(w->K, w->P, sizeof(double) * n * n);
    for (int j = 0; j < n; ++j) w->K[j * n + j] += w->st->sigma;
    for (int i = 0; i < w->m; ++i)
        for (int k = w->rp[i]; k < w->rp[i + 1]; ++k)
            for (int k2 = w->rp[i]; k2 < w->rp[i + 1]; ++k2)
                w->K[w->ci[k] * n + w->ci[k2]] += w->rho_vec[i] * w->av[k] * w->av[k2];
    w->info->nfact++;
    return chol_lower(w->K, n);
}

/* auxil.c compute_pri_res / compute_dua_res (unscaled norms, scaled vectors kept in z_prev/x_prev) */
static double compute_pri_res(work_t *w) {
    csr_mv(w->m, w->rp, w->ci, w->av, w->x, w->Ax);
    double r = 0;
    for (int i = 0; i < w->m; ++i) { w->z_prev[i] = w->Ax[i] - w->z[i]; double a = fabs(w->Einv[i] * w->z_prev[i]); if (a > r) r = a; }
    return r;
}
static double compute_dua_res(work_t *w) {
    sym_mv(w->n, w->P, w->x, w->Px);
    csr_mtv(w->m, w->n, w->rp, w->ci, w->av, w->y, w->Aty);
    double r = 0;
    for (int j = 0; j < w->n; ++j) { w->x_prev[j] = w->q[j] + w->Px[j] + w->Aty[j]; double a = fabs(w->Dinv[j] * w->x_prev[j]); if (a > r) r = a; }
    return w->cinv * r;
}
static double compute_pri_tol(work_t *w, double ea, double er) {
    double a = 0, b = 0;
    for (int i = 0; i < w->m; ++i) { double t = fabs(w->Einv[i] * w->z[i]); if (t > a) a = t; t = fabs(w->Einv[i] * w->Ax[i]); if (t > b) b = t; }
    return ea + er * fmax(a, b);
}
static double compute_dua_tol(work_t *w, double ea, double er) {
    double a = 0, b = 0, c = 0;
    for (int j = 0; j < w->n; ++j) {
        double t = fabs(w->Dinv[j] * w->q[j]); if (t > a) a = t;
        t = fabs(w->Dinv[j] * w->Aty[j]); if (t > b) b = t;
        t = fabs(w->Dinv[j] * w->Px[j]); if (t > c) c = t;
    }
    return ea + er * w->cinv * fmax(fmax(a, b), c);
}
static int is_primal_infeasible(work_t *w, double eps) {
    double nd = 0, lhs = 0;
    for (int i = 0; i < w->m; ++i) {
        if (w->u[i] > OSQP_INFTY * OSQP_MIN_SCALING) {
            if (w->l[i] < -OSQP_INFTY * OSQP_MIN_SCALING) w->delta_y[i] = 0.0; else w->delta_y[i] = fmin(w->delta_y[i], 0.0);
        } else if (w->l[i] < -OSQP_INFTY * OSQP_MIN_SCALING) w->delta_y[i] = fmax(w->delta_y[i], 0.0);
        double a = fabs(w->E[i] * w->delta_y[i]); if (a > nd) nd = a;
    }
    if (nd > eps) {
        for (int i = 0; i < w->m; ++i) lhs += w->u[i] * fmax(w->delta_y[i], 0) + w->l[i] * fmin(w->delta_y[i], 0);
        if (lhs < -eps * nd) {
            csr_mtv(w->m, w->n, w->rp, w->ci, w->av, w->delta_y, w->tn);
            double r = 0; for (int j = 0; j < w->n; ++j) { double a = fabs(w->Dinv[j] * w->tn[j]); if (a > r) r = a; }
            return r < eps * nd;
        }
    }
    return 0;
}
static int is_dual_infeasible(work_t *w, double eps) {
    double nd = 0, qd = 0;
    for (int j = 0; j < w->n; ++j) { double a = fabs(w->D[j] * w->delta_x[j]); if (a > nd) nd = a; qd += w->q[j] * w->delta_x[j]; }
    if (nd > eps && qd < -w->c * eps * nd) {
        sym_mv(w->n, w->P, w->delta_x, w->tn);
        double r = 0; for (int j = 0; j < w->n; ++j) { double a = fabs(w->Dinv[j] * w->tn[j]); if (a > r) r = a; }
        if (r < w->c * eps * nd) {
            csr_mv(w->m, w->rp, w->ci, w->av, w->delta_x, w->tm);
            for (int i = 0; i < w->m; ++i) {
                double v = w->Einv[i] * w->tm[i];
                if ((w->u[i] < OSQP_INFTY * OSQP_MIN_SCALING && v > eps * nd) || (w->l[i] > -OSQP_INFTY * OSQP_MIN_SCALING && v < -eps * nd)) return 0;
            }
            return 1;
        }
    }
    return 0;
}
static int check_termination(work_t *w, int approximate) {
    double ea = w->st->eps_abs, er = w->st->eps_rel, epi = w->st->eps_prim_inf, edi = w->st->eps_dual_inf;
    int prc = 0, drc = 0, pic = 0, dic = 0;
    if (w->info->pri_res > OSQP_INFTY || w->info->dua_res > OSQP_INFTY || isnan(w->info->pri_res) || isnan(w->info->dua_res)) { w->info->status = ORC_NON_CVX; return 1; }
    if (approximate) { ea *= 10; er *= 10; epi *= 10; edi *= 10; }
    if (w->m == 0) prc = 1;
    else { if (w->info->pri_res < compute_pri_tol(w, ea, er)) prc = 1; else pic = is_primal_infeasible(w, epi); }
    if (w->info->dua_res < compute_dua_tol(w, ea, er)) drc = 1; else dic = is_dual_infeasible(w, edi);
    if (prc && drc) { w->info->status = approximate ? ORC_SOLVED_INACCURATE : ORC_SOLVED; return 1; }
    if (pic) { w->info->status = ORC_PRIMAL_INFEASIBLE; return 1; }
    if (dic) { w->info->status = ORC_DUAL_INFEASIBLE; return 1; }
    return 0;
}
static void update_info(work_t *w) { w->info->pri_res = compute_pri_res(w); w->info->dua_res = compute_dua_res(w); }

/* auxil.c compute_rho_estimate (all SCALED quantities) */
static double compute_rho_estimate(work_t *w) {
    double pr = vmaxabs(w->z_prev, w->m), dr = vmaxabs(w->x_prev, w->n);
    double pn = fmax(vmaxabs(w->z, w->m), vmaxabs(w->Ax, w->m));
    pr /= (pn + 1e-10);
    double dn = fmax(fmax(vmaxabs(w->q, w->n), vmaxabs(w->Aty, w->n)), vmaxabs(w->Px, w->n));
    dr /= (dn + 1e-10);
    double re = w->rho * sqrt(pr / (dr + 1e-10));
    return fmin(fmax(re, OSQP_RHO_MIN), OSQP_RHO_MAX);
}
static int adapt_rho(work_t *w) {
    double rn = compute_rho_estimate(w);
    if (rn > w->rho * w->st->adaptive_rho_tolerance || rn < w->rho / w->st->adaptive_rho_tolerance) {
        w->rho = fmin(fmax(rn, OSQP_RHO_MIN), OSQP_RHO_MAX);
        for (int i = 0; i < w->m; ++i) {
            if (w->ctype[i] == 0) { w->rho_vec[i] = w->rho; w->rho_inv_vec[i] = 1.0 / w->rho; }
            else if (w->ctype[i] == 1) { w->rho_vec[i] = OSQP_RHO_EQ_OVER_RHO_INEQ * w->rho; w->rho_inv_vec[i] = 1.0 / w->rho_vec[i]; }
        }
        w->info->rho_updates++;
        return factor(w);
    }
    return 0;
}

/*
 * x (n), y (m): unscaled; read as the warm start when settings->warm_start != 0 (osqp_warm_start
 * semantics: x_s = Dinv x, z_s = A_s x_s, y_s = c Einv y), always written with the solution
 * (store_solution: x = D x_s, y = cinv E y_s; NaN on infeasible status as OSQP does).
 * rho_io: optional in/out carried rho (settings->rho persists across OSQP solves).
 */
/* auxil.c update_rho_vec: constraint types after a bound update; returns 1 when a type (and with it rho_vec) changed, i.e. when OSQP refactors */
static int update_rho_vec(work_t *w) {
    int changed = 0;
    for (int i = 0; i < w->m; ++i) {
        int t = (w->l[i] < -OSQP_INFTY * OSQP_MIN_SCALING && w->u[i] > OSQP_INFTY * OSQP_MIN_SCALING) ? -1 : (w->u[i] - w->l[i] < OSQP_RHO_TOL ? 1 : 0);
        if (t != w->ctype[i]) {
            w->ctype[i] = t;
            w->rho_vec[i] = t == -1 ? OSQP_RHO_MIN : (t == 1 ? OSQP_RHO_EQ_OVER_RHO_INEQ * w->rho : w->rho);
            w->rho_inv_vec[i] = 1.0 / w->rho_vec[i];
            changed = 1;
        }
    }
    return changed;
}

/* Per-thread scratch, reused from solve to solve.  Every solve used to calloc / malloc ~0.5 MB in pieces above glibc's mmap threshold: each one a fresh mapping,
 * page faults and the process-wide mmap lock -- the 128 OpenMP threads of the CPU baseline ran 9.8x one thread.  Buffers live as long as their thread. */
typedef struct { void *p; size_t cap; } orc_scratch;
static __thread orc_scratch tl_scratch[9];   /* slot 8: orc_last_z() only (it used to share slot 2 with the formation's B_mat_d_list) */
static void *scratch_get(int slot, size_t bytes, int zero) {
    orc_scratch *sc = &tl_scratch[slot];
    if (sc->cap < bytes) { free(sc->p); sc->p = malloc(bytes); sc->cap = sc->p ? bytes : 0; }
    if (sc->p && zero) memset(sc->p, 0, bytes);
    return sc->p;
}

/*
 * carry (optional): the persistent OSQP workspace of the reference's `OsqpEigen::Solver solver` member between ticks, for the UPDATE PATH the
 * reference takes on every tick after the first (S/A1RobotControl.cpp:533-538: updateHessianMatrix, updateGradient, updateLowerBound,
 * updateUpperBound, then solve() with warm start) -- restated from OSQP 0.6 (osqp.c osqp_update_P / osqp_update_lin_cost / osqp_update_lower_bound /
 * osqp_update_upper_bound, auxil.c update_rho_vec), unpinned like the rest of the solve:
 *   osqp_update_P        unscale_data, new P, scale_data AGAIN FROM D = E = c = 1 -- with the PREVIOUS tick's q, l, u still in the workspace (the gradient
 *                        is only replaced by the next call), so the cost scaling c sees the old gradient --, refactor with the carried rho_vec
 *   osqp_update_lin_cost q = c (D q_new)
 *   osqp_update_*_bound  l = E l_new, then u = E u_new; after each, update_rho_vec: a constraint whose type changed gets its rho and the KKT matrix is refactored
 *   osqp_solve           warm start: the SCALED iterates (x, z, y) of the previous solve are used as they are (they were scaled with the previous tick's D, E, c),
 *                        rho = the previous solve's adapted value
 * Layout: carry[0] = valid flag, [1] = rho, then x_s (n), z_s (m), y_s (m), q_prev (n), l_prev (m), u_prev (m): 2 + 2n + 4m doubles, zero-initialised by the caller.
 * A first tick (valid = 0) is osqp_setup + a cold solve (the workspace's iterates are zero whatever warm_start says).
 */
static int osqp_solve_impl(int n, int m, const double *P, const double *q, const int32_t *rp, const int32_t *ci, const double *av,
                   const double *l, const double *u, const orc_settings *st, double *x, double *y, double *rho_io,
                   orc_info *info, double *carry, int pattern_changed) {
    const int upd = carry && carry[0] != 0.0;
    const int reinit = upd && pattern_changed;   /* osqp-eigen's updateHessianMatrix with a changed sparsity pattern: see below */
    double *c_xs = carry ? carry + 2 : 0, *c_zs = carry ? c_xs + n : 0, *c_ys = carry ? c_zs + m : 0, *c_q = carry ? c_ys + m : 0, *c_l = carry ? c_q + n : 0, *c_u = carry ? c_l + m : 0;
    work_t w; memset(&w, 0, sizeof w);
    int nnz = rp[m];
    size_t tot = (size_t)2 * n * n + 16 * (size_t)n + 16 * (size_t)m + nnz + 64 + (st->linsys == 1 ? (size_t)(n + m) * (n + m) + 2 * (size_t)(n + m) : 0);
    double *buf = (double *)scratch_get(0, tot * sizeof(double), 1);
    int *ctype = (int *)scratch_get(1, (size_t)(m + 1) * sizeof(int), 1);
    if (!buf || !ctype) return -1;
    double *p = buf;
#define TAKE(k) (p += (k), p - (k))
    w.n = n; w.m = m; w.nnz = nnz; w.rp = rp; w.ci = ci; w.st = st; w.info = info; w.ctype = ctype;
    w.P = TAKE(n * n); w.K = TAKE(n * n); w.q = TAKE(n); w.av = TAKE(nnz); w.l = TAKE(m); w.u = TAKE(m);
    w.D = TAKE(n); w.Dinv = TAKE(n); w.E = TAKE(m); w.Einv = TAKE(m); w.rho_vec = TAKE(m); w.rho_inv_vec = TAKE(m);
    w.x = TAKE(n); w.z = TAKE(m); w.y = TAKE(m); w.x_prev = TAKE(n); w.z_prev = TAKE(m); w.xt = TAKE(n); w.zt = TAKE(m);
    w.delta_x = TAKE(n); w.delta_y = TAKE(m); w.Ax = TAKE(m); w.Px = TAKE(n); w.Aty = TAKE(n); w.tn = TAKE(n); w.tm = TAKE(m);
    if (st->linsys == 1) { w.KK = TAKE((n + m) * (n + m)); w.KKd = TAKE(n + m); w.kb = TAKE(n + m); }
#undef TAKE
    memcpy(w.P, P, sizeof(double) * n * n); memcpy(w.q, q, sizeof(double) * n); memcpy(w.av, av, sizeof(double) * nnz);
    /* OSQP is handed the UPPER triangle of P only (osqp-eigen: setHessianMatrix -> triangularView<Upper>); the reference's dense
     * B_qp'QB_qp is symmetric only up to rounding, so mirror the upper triangle instead of trusting the lower one */
    for (int i = 1; i < n; ++i) for (int j = 0; j < i; ++j) w.P[i * n + j] = w.P[j * n + i];
    memcpy(w.l, l, sizeof(double) * m); memcpy(w.u, u, sizeof(double) * m);
    memset(info, 0, sizeof *info); info->status = ORC_UNSOLVED;
    w.rho = (rho_io && st->warm_start && *rho_io > 0) ? *rho_io : st->rho;
    if (upd && !reinit) {  /* osqp_update_P re-scales with the previous tick's q, l, u in the workspace; settings->rho is the previous solve's */
        memcpy(w.q, c_q, sizeof(double) * n); memcpy(w.l, c_l, sizeof(double) * m); memcpy(w.u, c_u, sizeof(double) * m);
        w.rho = carry[1];
    } else if (carry) w.rho = st->rho;   /* first tick, or reinit: osqp_setup with the solver's own settings (rho back to settings->rho) and the current q, l, u */

    /* osqp_setup */
    if (st->scaling) scale_data(&w);
    else { w.c = w.cinv = 1; for (int j = 0; j < n; ++j) w.D[j] = w.Dinv[j] = 1; for (int i = 0; i < m; ++i) w.E[i] = w.Einv[i] = 1; }
    set_rho_vec(&w);
    int rc = factor(&w);
    if (rc) { info->status = ORC_NON_CVX; goto done; }

    if (reinit) {
        /* osqp-eigen 0.6.3, Solver::updateHessianMatrix when the triplets of hessian.sparseView() no longer match the workspace's P (the reference's dense
         * B_qp'QB_qp gains or loses exact zeros, S/ConvexMpc.cpp:211; level stance, yaw = 0, symmetric feet -- fixture T is such a state): osqp_update_P cannot
         * change a pattern, so osqp-eigen does  getPrimalVariable / getDualVariable -- which map the WORKSPACE iterates work->x, work->y, i.e. the previous solve's
         * SCALED x_s = D'^-1 x, y_s = c' E'^-1 y --, clearSolver, initSolver (= the osqp_setup above: fresh scaling with the current data, rho = settings->rho),
         * setPrimalVariable / setDualVariable = osqp_warm_start_x / _y, which treat what they are given as unscaled:  x <- D^-1 x_s,  z <- A x,  y <- c E^-1 y_s.
         * The updateGradient / update*Bound calls that follow in compute_grf hand over the data osqp_setup has just seen: nothing changes.  (Restated from memory of
         * osqp-eigen's Solver.tpp, unpinned like the rest of the solve.  initSolver re-reads q, l, u through the pointers setGradient / set*Bound stored at the first
         * tick -- the members of that tick's local ConvexMpc object; every tick's object lives at the same stack address, so it finds the current tick's values.) */
        for (int j = 0; j < n; ++j) w.x[j] = w.Dinv[j] * c_xs[j];
        csr_mv(m, rp, ci, w.av, w.x, w.z);
        for (int i = 0; i < m; ++i) w.y[i] = w.c * w.Einv[i] * c_ys[i];
    } else if (upd) {
        for (int j = 0; j < n; ++j) w.q[j] = (w.D[j] * q[j]) * w.c;                 /* osqp_update_lin_cost: vec_ew_prod(D, q), vec_mult_scalar(q, c) */
        for (int i = 0; i < m; ++i) w.l[i] = w.E[i] * l[i];                        /* osqp_update_lower_bound */
        if (update_rho_vec(&w)) { rc = factor(&w); if (rc) { info->status = ORC_NON_CVX; goto done; } }
        for (int i = 0; i < m; ++i) w.u[i] = w.E[i] * u[i];                        /* osqp_update_upper_bound */
        if (update_rho_vec(&w)) { rc = factor(&w); if (rc) { info->status = ORC_NON_CVX; goto done; } }
        memcpy(w.x, c_xs, sizeof(double) * n); memcpy(w.z, c_zs, sizeof(double) * m); memcpy(w.y, c_ys, sizeof(double) * m);
    }
    /* cold / warm start */
    if (st->warm_start && !carry) {
        for (int j = 0; j < n; ++j) w.x[j] = w.Dinv[j] * x[j];
        csr_mv(m, rp, ci, w.av, w.x, w.z);
        for (int i = 0; i < m; ++i) w.y[i] = w.c * w.Einv[i] * y[i];
    }

    {
        int iter, can_check = 0;
        const double alpha = st->alpha, sigma = st->sigma;
        /* adaptive_rho_interval = 0 (OSQP's automatic, wall-clock based rule): resolved to the outcome the rule has for these QP sizes,
         * c_max(c_roundmultiple(iter, check_termination), check_termination) = check_termination (see the header of this file) */
        const int rho_every = st->adaptive_rho_interval > 0 ? st->adaptive_rho_interval : (st->check_termination > 0 ? st->check_termination : 25);
        for (iter = 1; iter <= st->max_iter; ++iter) {
            double *t;
            t = w.x; w.x = w.x_prev; w.x_prev = t;
            t = w.z; w.z = w.z_prev; w.z_prev = t;
            /* update_xz_tilde: rhs, reduced solve, nu, z_tilde */
            for (int i = 0; i < m; ++i) w.tm[i] = w.z_prev[i] - w.rho_inv_vec[i] * w.y[i]; /* rhs_z */
            for (int j = 0; j < n; ++j) w.xt[j] = sigma * w.x_prev[j] - w.q[j];
            for (int i = 0; i < m; ++i) { double s = w.rho_vec[i] * w.tm[i]; for (int k = rp[i]; k < rp[i + 1]; ++k) w.xt[ci[k]] += w.av[k] * s; }
            if (st->linsys == 1) {
                /* auxil.c update_xz_tilde + lin_sys/qdldl solve_linsys_qdldl: [x_tilde; nu] = K^-1 [sigma x_prev - q; z_prev - rho^-1 y],
                 * z_tilde = rhs_z + rho^-1 nu  (rhs_z is what the solve was given in the lower block) */
                for (int j = 0; j < n; ++j) w.kb[j] = sigma * w.x_prev[j] - w.q[j];
                for (int i = 0; i < m; ++i) w.kb[n + i] = w.tm[i];
                kkt_solve(&w, w.kb);
                for (int j = 0; j < n; ++j) w.xt[j] = w.kb[j];
                for (int i = 0; i < m; ++i) w.zt[i] = w.tm[i] + w.rho_inv_vec[i] * w.kb[n + i];
            } else {
            chol_solve(w.K, n, w.xt);
            csr_mv(m, rp, ci, w.av, w.xt, w.zt); /* A x_tilde */
            for (int i = 0; i < m; ++i) {
                double nu = w.rho_vec[i] * (w.zt[i] - w.tm[i]);
                w.zt[i] = w.tm[i] + w.rho_inv_vec[i] * nu; /* b[n+j] += rho_inv*nu (solve_linsys_qdldl) */
            }
            }
            /* update_x, update_z, update_y */
            for (int j = 0; j < n; ++j) { w.x[j] = alpha * w.xt[j] + (1.0 - alpha) * w.x_prev[j]; w.delta_x[j] = w.x[j] - w.x_prev[j]; }
            for (int i = 0; i < m; ++i) {
                double v = alpha * w.zt[i] + (1.0 - alpha) * w.z_prev[i] + w.rho_inv_vec[i] * w.y[i];
                w.z[i] = fmin(fmax(v, w.l[i]), w.u[i]);
            }
            for (int i = 0; i < m; ++i) {
                w.delta_y[i] = w.rho_vec[i] * (alpha * w.zt[i] + (1.0 - alpha) * w.z_prev[i] - w.z[i]);
                w.y[i] += w.delta_y[i];
            }
            can_check = st->check_termination && (iter % st->check_termination == 0);
            if (can_check) { update_info(&w); if (check_termination(&w, 0)) break; }
            if (st->adaptive_rho && (iter % rho_every == 0)) {
                if (!can_check) update_info(&w);
                if (adapt_rho(&w)) { info->status = ORC_NON_CVX; break; }
            }
        }
        if (iter > st->max_iter) iter = st->max_iter;
        info->iters = iter;
        if (info->status == ORC_UNSOLVED) {
            if (!can_check) { update_info(&w); check_termination(&w, 0); }
            if (info->status == ORC_UNSOLVED && !check_termination(&w, 1)) info->status = ORC_MAX_ITER_REACHED;
        }
    }
done:
    /* A non-finite iterate (NaN / Inf inputs) is reported as NON_CVX whatever the residual tests concluded: OSQP's max-norms skip NaNs,
     * so its own termination test can "converge" on a NaN iterate.  (Same rule as the engine's write_outputs; a deviation from OSQP that
     * only concerns inputs the reference never produces.) */
    for (int j = 0; j < n; ++j) if (!isfinite(w.x[j])) { info->status = ORC_NON_CVX; break; }
    info->rho_final = w.rho;
    if (rho_io) *rho_io = w.rho;
    {   /* the unscaled z = Einv z_s of this solve, for orc_last_z() (tests: OSQP's termination test needs it beside x and y) */
        double *lz = (double *)scratch_get(8, (size_t)(m + 1) * sizeof(double), 0);
        if (lz) { lz[0] = (double)m; for (int i = 0; i < m; ++i) lz[1 + i] = w.Einv[i] * w.z[i]; }
    }
    if (carry) {  /* what stays in the reference's workspace for the next tick's update calls */
        const int failed = info->status == ORC_PRIMAL_INFEASIBLE || info->status == ORC_DUAL_INFEASIBLE || info->status == ORC_NON_CVX;
        carry[0] = 1.0; carry[1] = w.rho;   /* (also after a failed solve: cold_start() zeroes the iterates, settings->rho keeps what adapt_rho left there) */
        for (int j = 0; j < n; ++j) c_xs[j] = failed ? 0.0 : w.x[j];               /* store_solution: cold_start after a failed solve */
        for (int i = 0; i < m; ++i) { c_zs[i] = failed ? 0.0 : w.z[i]; c_ys[i] = failed ? 0.0 : w.y[i]; }
        memcpy(c_q, q, sizeof(double) * n); memcpy(c_l, l, sizeof(double) * m); memcpy(c_u, u, sizeof(double) * m);
    }
    if (info->status == ORC_PRIMAL_INFEASIBLE || info->status == ORC_DUAL_INFEASIBLE || info->status == ORC_NON_CVX) {
        for (int j = 0; j < n; ++j) x[j] = NAN;
        for (int i = 0; i < m; ++i) y[i] = NAN;
    } else {
        for (int j = 0; j < n; ++j) x[j] = w.D[j] * w.x[j];
        for (int i = 0; i < m; ++i) y[i] = w.cinv * w.E[i] * w.y[i];
    }
    return 0;
}
int orc_osqp_solve(int n, int m, const double *P, const double *q, const int32_t *rp, const int32_t *ci, const double *av,
                   const double *l, const double *u, const orc_settings *st, double *x, double *y, double *rho_io,
                   orc_info *info) {
    return osqp_solve_impl(n, m, P, q, rp, ci, av, l, u, st, x, y, rho_io, info, 0, 0);
}
int orc_osqp_solve_update(int n, int m, const double *P, const double *q, const int32_t *rp, const int32_t *ci, const double *av,
                          const double *l, const double *u, const orc_settings *st, double *x, double *y, double *carry, orc_info *info) {
    return osqp_solve_impl(n, m, P, q, rp, ci, av, l, u, st, x, y, 0, info, carry, 0);
}
/* the same when the caller (osqp-eigen's updateHessianMatrix) has found the sparsity pattern of P changed since the previous tick: re-initialisation + warm start */
int orc_osqp_solve_update_ex(int n, int m, const double *P, const double *q, const int32_t *rp, const int32_t *ci, const double *av,
                             const double *l, const double *u, const orc_settings *st, double *x, double *y, double *carry, int pattern_changed, orc_info *info) {
    return osqp_solve_impl(n, m, P, q, rp, ci, av, l, u, st, x, y, 0, info, carry, pattern_changed);
}
/* unscaled z of the calling thread's most recent solve (m doubles); returns m, or -1 when there was none or the size differs */
int orc_last_z(int m, double *z_out) {
    const double *lz = (const double *)tl_scratch[8].p;
    if (!lz || (int)lz[0] != m) return -1;
    for (int i = 0; i < m; ++i) z_out[i] = lz[1 + i];
    return m;
}
/* OSQP's termination test (auxil.c check_termination, scaled_termination = 0) evaluated on a GIVEN unscaled iterate (x, z, y): is this a point OSQP itself would
 * have stopped at?  Test infrastructure for the ticks where the engine and this oracle stop at different iterations (tests/test_gpu_parity.py, the 10 000-tick
 * update-path test).  The unscaled norms need no scaling data:  Einv (A_s x_s - z_s) = A x - z  and  cinv Dinv (P_s x_s + q_s + A_s' y_s) = P x + q + A' y,
 * likewise the norms of the tolerances (compute_pri_tol / compute_dua_tol above).  P: the upper triangle is used (what OSQP is handed).
 * out[0..3] = pri_res, pri_tol, dua_res, dua_tol.  Returns 1 when both tests pass (strict inequalities, as OSQP's). */
int orc_check_termination(int n, int m, const double *P, const double *q, const int32_t *rp, const int32_t *ci, const double *av,
                          const double *x, const double *z, const double *y, double eps_abs, double eps_rel, double *out) {
    double pri = 0, nz = 0, nAx = 0, dua = 0, nq = 0, nAty = 0, nPx = 0;
    double *Aty = (double *)calloc((size_t)n, sizeof(double));
    if (!Aty) return -1;
    for (int i = 0; i < m; ++i) {
        double ax = 0;
        for (int k = rp[i]; k < rp[i + 1]; ++k) { ax += av[k] * x[ci[k]]; Aty[ci[k]] += av[k] * y[i]; }
        double a = fabs(ax - z[i]); if (a > pri) pri = a;
        a = fabs(z[i]); if (a > nz) nz = a;
        a = fabs(ax); if (a > nAx) nAx = a;
    }
    for (int j = 0; j < n; ++j) {
        double px = 0;
        for (int k = 0; k < n; ++k) px += (k >= j ? P[(size_t)j * n + k] : P[(size_t)k * n + j]) * x[k];
        double a = fabs(px + q[j] + Aty[j]); if (a > dua) dua = a;
        a = fabs(q[j]); if (a > nq) nq = a;
        a = fabs(Aty[j]); if (a > nAty) nAty = a;
        a = fabs(px); if (a > nPx) nPx = a;
    }
    free(Aty);
    const double ptol = eps_abs + eps_rel * fmax(nz, nAx), dtol = eps_abs + eps_rel * fmax(fmax(nq, nAty), nPx);
    if (out) { out[0] = pri; out[1] = ptol; out[2] = dua; out[3] = dtol; }
    return pri < ptol && dua < dtol;
}
/* sparsity pattern of the upper triangle of a dense symmetric P as hessian.sparseView() sees it (exact zeros dropped): FNV-1a over the non-zero flags, plus the count */
void orc_pattern_signature(int n, const double *P, double *sig2) {
    uint64_t hsh = 1469598103934665603ULL; int64_t nnz = 0;
    for (int i = 0; i < n; ++i) for (int j = i; j < n; ++j) {
        const int nz = P[(size_t)i * n + j] != 0.0;
        nnz += nz; hsh = (hsh ^ (uint64_t)nz) * 1099511628211ULL;
    }
    sig2[0] = (double)(hsh >> 12);   /* 52 bits: exact in a double */
    sig2[1] = (double)nnz;
}

/* ------------------------------------------------------------------------------------------
 * Formation: MPC (S/ConvexMpc.cpp) -- literal restatement, runtime horizon
 * ---------------------------------------------------------------------------------------- */
typedef struct orc_mpc_params {
    int32_t horizon;
    double dt, mu, fz_min, fz_max;     /* S/ConvexMpc.cpp:8,223-224; S/A1RobotControl.cpp:462 */
    double q[NS], r[NU];               /* S/A1CtrlStates.h:365-366 */
    double mass, inertia[9];           /* S/A1CtrlStates.h:358-360, row-major 3x3 */
} orc_mpc_params;

static void mat3_mul(const double *a, const double *b, double *c) { /* row-major */
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += a[i * 3 + k] * b[k * 3 + j]; c[i * 3 + j] = s; }
}
static void mat3_inv(const double *a, double *inv) { /* cofactor formula, as Eigen's fixed 3x3 inverse */
    double c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
    double det = a[0] * c00 + a[1] * c01 + a[2] * c02, id = 1.0 / det;
    inv[0] = c00 * id; inv[1] = (a[2] * a[7] - a[1] * a[8]) * id; inv[2] = (a[1] * a[5] - a[2] * a[4]) * id;
    inv[3] = c01 * id; inv[4] = (a[0] * a[8] - a[2] * a[6]) * id; inv[5] = (a[2] * a[3] - a[0] * a[5]) * id;
    inv[6] = c02 * id; inv[7] = (a[1] * a[6] - a[0] * a[7]) * id; inv[8] = (a[0] * a[4] - a[1] * a[3]) * id;
}
static void skew3(const double *v, double *s) { /* S/utils/Utils.cpp:35-41 */
    s[0] = 0; s[1] = -v[2]; s[2] = v[1]; s[3] = v[2]; s[4] = 0; s[5] = -v[0]; s[6] = -v[1]; s[7] = v[0]; s[8] = 0;
}

/* S/ConvexMpc.cpp:110-130  (A_c 13x13 row-major) */
void orc_A_mat_c(double yaw, double *Ac) {
    memset(Ac, 0, sizeof(double) * NS * NS);
    double c = cos(yaw), s = sin(yaw);
    Ac[0 * NS + 6] = c; Ac[0 * NS + 7] = s; Ac[1 * NS + 6] = -s; Ac[1 * NS + 7] = c; Ac[2 * NS + 8] = 1;
    Ac[3 * NS + 9] = 1; Ac[4 * NS + 10] = 1; Ac[5 * NS + 11] = 1;
    Ac[11 * NS + NU] = 1;
}
/* S/ConvexMpc.cpp:132-143  (B_c 13x12 row-major); foot: 3x4 column-major (Eigen), Rw row-major */
void orc_B_mat_c(double mass, const double *Ib, const double *Rw, const double *foot, double *Bc) {
    double Rt[9], t[9], Iw[9], Iwi[9], sk[9], blk[9];
    memset(Bc, 0, sizeof(double) * NS * NU);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rt[i * 3 + j] = Rw[j * 3 + i];
    mat3_mul(Rw, Ib, t); mat3_mul(t, Rt, Iw);
    for (int leg = 0; leg < NLEG; ++leg) {
        mat3_inv(Iw, Iwi); skew3(foot + 3 * leg, sk); mat3_mul(Iwi, sk, blk);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Bc[(6 + i) * NU + 3 * leg + j] = blk[i * 3 + j];
        for (int i = 0; i < 3; ++i) Bc[(9 + i) * NU + 3 * leg + i] = 1.0 / mass;
    }
}

/*
 * Full MPC QP data.  All outputs dense row-major; n = 12h, m = 20h.
 *   x0[13], xref[13h], yaw (for A_c), Rw[9] row-major, foot: 3x4 col-major (stride 0 => same feet every
 *   step, S/A1RobotControl.cpp:498-514; stride 12 => per-step feet as S/test/test_mpc.cpp:106-122),
 *   contact[4] (+ stride 4 for per-step schedules; reference broadcasts, S/ConvexMpc.cpp:228-245).
 * P (n*n), g (n), A CSR (rp[m+1], ci[36h], av[36h]), l (m), u (m).
 */
void orc_mpc_form(const orc_mpc_params *pr, const double *x0, const double *xref, double yaw, const double *Rw,
                  const double *foot, int foot_stride, const uint8_t *contact, int contact_stride,
                  double *P, double *g, int32_t *rp, int32_t *ci, double *av, double *l, double *u) {
    const int h = pr->horizon, n = NU * h, ns = NS * h;
    double Ac[NS * NS], Ad[NS * NS], Bc[NS * NU];
    double *Bdl = (double *)scratch_get(2, sizeof(double) * (size_t)ns * NU, 1);      /* B_mat_d_list */
    double *Aqp = (double *)scratch_get(3, sizeof(double) * (size_t)ns * NS, 1);
    double *Bqp = (double *)scratch_get(4, sizeof(double) * (size_t)ns * n, 1);
    double *Qd = (double *)scratch_get(5, sizeof(double) * (size_t)(2 * ns + n), 0), *Rd = Qd + ns;
    double *tmp = Rd + n;
    /* ConvexMpc ctor :16-44 */
    for (int i = 0; i < h; ++i) { for (int k = 0; k < NS; ++k) Qd[i * NS + k] = 2 * pr->q[k]; for (int k = 0; k < NU; ++k) Rd[i * NU + k] = 2 * pr->r[k]; }
    orc_A_mat_c(yaw, Ac);
    for (int i = 0; i < NS; ++i) for (int j = 0; j < NS; ++j) Ad[i * NS + j] = (i == j ? 1.0 : 0.0) + Ac[i * NS + j] * pr->dt; /* :150 */
    for (int i = 0; i < h; ++i) {
        orc_B_mat_c(pr->mass, pr->inertia, Rw, foot + (size_t)i * foot_stride, Bc);
        for (int k = 0; k < NS * NU; ++k) Bdl[(size_t)i * NS * NU + k] = Bc[k] * pr->dt; /* :151, A1RobotControl.cpp:513 */
    }
    /* :184-202 */
    for (int i = 0; i < h; ++i) {
        double *Ai = Aqp + (size_t)i * NS * NS;
        if (i == 0) memcpy(Ai, Ad, sizeof Ad);
        else {
            const double *Ap = Aqp + (size_t)(i - 1) * NS * NS;
            for (int r = 0; r < NS; ++r) for (int c = 0; c < NS; ++c) { double s = 0; for (int k = 0; k < NS; ++k) s += Ap[r * NS + k] * Ad[k * NS + c]; Ai[r * NS + c] = s; }
        }
        for (int j = 0; j < i + 1; ++j) {
            const double *Bj = Bdl + (size_t)j * NS * NU;
            if (i - j == 0) {
                for (int r = 0; r < NS; ++r) for (int c = 0; c < NU; ++c) Bqp[(size_t)(i * NS + r) * n + j * NU + c] = Bj[r * NU + c];
            } else {
                const double *Ap = Aqp + (size_t)(i - j - 1) * NS * NS;
                for (int r = 0; r < NS; ++r) for (int c = 0; c < NU; ++c) { double s = 0; for (int k = 0; k < NS; ++k) s += Ap[r * NS + k] * Bj[k * NU + c]; Bqp[(size_t)(i * NS + r) * n + j * NU + c] = s; }
            }
        }
    }
    /* :207-210  dense_hessian = B_qp' Q B_qp + R */
    for (int a = 0; a < n; ++a) for (int b = 0; b < n; ++b) P[(size_t)a * n + b] = 0;
    for (int k = 0; k < ns; ++k) {
        const double *row = Bqp + (size_t)k * n; double qk = Qd[k];
        if (qk == 0.0) continue;
        for (int a = 0; a < n; ++a) { double v = row[a] * qk; if (v == 0.0) continue; double *Pa = P + (size_t)a * n; for (int b = 0; b < n; ++b) Pa[b] += v * row[b]; }
    }
    for (int a = 0; a < n; ++a) P[(size_t)a * n + a] += Rd[a];
    /* :215-217 gradient */
    for (int i = 0; i < h; ++i) for (int r = 0; r < NS; ++r) { double s = 0; for (int k = 0; k < NS; ++k) s += Aqp[(size_t)i * NS * NS + r * NS + k] * x0[k]; tmp[i * NS + r] = (s - xref[i * NS + r]) * Qd[i * NS + r]; }
    for (int a = 0; a < n; ++a) { double s = 0; for (int k = 0; k < ns; ++k) s += Bqp[(size_t)k * n + a] * tmp[k]; g[a] = s; }
    /* ctor :46-58 constraint stencil, and bounds :223-245 */
    int nz = 0;
    for (int i = 0; i < NLEG * h; ++i) {
        const uint8_t *cs = contact + (size_t)(i / NLEG) * contact_stride;
        double cf = cs[i % NLEG] ? 1.0 : 0.0;
        int r0 = 5 * i, c0 = 3 * i;
        rp[r0 + 0] = nz; ci[nz] = c0 + 0; av[nz++] = 1; ci[nz] = c0 + 2; av[nz++] = pr->mu;
        rp[r0 + 1] = nz; ci[nz] = c0 + 0; av[nz++] = 1; ci[nz] = c0 + 2; av[nz++] = -pr->mu;
        rp[r0 + 2] = nz; ci[nz] = c0 + 1; av[nz++] = 1; ci[nz] = c0 + 2; av[nz++] = pr->mu;
        rp[r0 + 3] = nz; ci[nz] = c0 + 1; av[nz++] = 1; ci[nz] = c0 + 2; av[nz++] = -pr->mu;
        rp[r0 + 4] = nz; ci[nz] = c0 + 2; av[nz++] = 1;
        l[r0 + 0] = 0; u[r0 + 0] = OSQP_INFTY;
        l[r0 + 1] = -OSQP_INFTY; u[r0 + 1] = 0;
        l[r0 + 2] = 0; u[r0 + 2] = OSQP_INFTY;
        l[r0 + 3] = -OSQP_INFTY; u[r0 + 3] = 0;
        l[r0 + 4] = pr->fz_min * cf; u[r0 + 4] = pr->fz_max * cf;
    }
    rp[NC * h] = nz;
}

/* S/A1RobotControl.cpp:470-488 -- reference trajectory from the compact command */
void orc_mpc_reference(int h, double dt, const double *euler, const double *pos, const double *Rw,
                       const double *euler_d, const double *lin_vel_d_body, const double *ang_vel_d, double pos_z_d,
                       double *xref) {
    double vw[3];
    for (int i = 0; i < 3; ++i) vw[i] = Rw[i * 3 + 0] * lin_vel_d_body[0] + Rw[i * 3 + 1] * lin_vel_d_body[1] + Rw[i * 3 + 2] * lin_vel_d_body[2];
    for (int i = 0; i < h; ++i) {
        double *x = xref + i * NS;
        x[0] = euler_d[0]; x[1] = euler_d[1]; x[2] = euler[2] + ang_vel_d[2] * dt * (i + 1);
        x[3] = pos[0] + vw[0] * dt * (i + 1); x[4] = pos[1] + vw[1] * dt * (i + 1); x[5] = pos_z_d;
        x[6] = ang_vel_d[0]; x[7] = ang_vel_d[1]; x[8] = ang_vel_d[2];
        x[9] = vw[0]; x[10] = vw[1]; x[11] = 0; x[12] = -9.8;
    }
}

/*
 * One MPC tick, S/A1RobotControl.cpp:446-562 minus ROS: form, OSQP solve, grf_body = R' f (Q9).
 * u_full (12h, optional), warm x/y (optional, n and m), returns info.
 * grf_out: 3x4 column-major.  NaN solution => zeros + status (replaces quirk Q8).
 */
int orc_mpc_solve(const orc_mpc_params *pr, const orc_settings *st, const double *x0, const double *xref, const double *Rw,
                  const double *foot, int foot_stride, const uint8_t *contact, int contact_stride,
                  double *grf_out, double *u_full, double *warm_x, double *warm_y, double *warm_rho, orc_info *info) {
    const int h = pr->horizon, n = NU * h, m = NC * h;
    double *P = (double *)scratch_get(6, sizeof(double) * ((size_t)n * n + n + 2 * m + 36 * h + n + m), 0);
    double *g = P + (size_t)n * n, *l = g + n, *u = l + m, *av = u + m, *x = av + 36 * h, *y = x + n;
    int32_t *rp = (int32_t *)scratch_get(7, sizeof(int32_t) * (m + 1 + 36 * h), 0), *ci = rp + m + 1;
    orc_mpc_form(pr, x0, xref, x0[2], Rw, foot, foot_stride, contact, contact_stride, P, g, rp, ci, av, l, u);
    if (warm_x && st->warm_start) { memcpy(x, warm_x, sizeof(double) * n); memcpy(y, warm_y, sizeof(double) * m); }
    else { memset(x, 0, sizeof(double) * n); memset(y, 0, sizeof(double) * m); }
    int rc = orc_osqp_solve(n, m, P, g, rp, ci, av, l, u, st, x, y, warm_rho, info);
    for (int leg = 0; leg < NLEG; ++leg) {
        const double *f = x + 3 * leg;
        int bad = isnan(f[0]) || isnan(f[1]) || isnan(f[2]);
        for (int i = 0; i < 3; ++i) grf_out[3 * leg + i] = bad ? 0.0 : Rw[0 * 3 + i] * f[0] + Rw[1 * 3 + i] * f[1] + Rw[2 * 3 + i] * f[2];
    }
    if (u_full) memcpy(u_full, x, sizeof(double) * n);
    if (warm_x) {
        /* A failed solve (NaN solution) must not poison the carried workspace: OSQP's store_solution() cold-starts the iterates in
         * that case (x = y = 0); the rho the solver had reached stays, as OSQP leaves it in settings->rho (round 4). */
        int failed = info->status == ORC_PRIMAL_INFEASIBLE || info->status == ORC_DUAL_INFEASIBLE || info->status == ORC_NON_CVX;
        if (failed) { memset(warm_x, 0, sizeof(double) * n); memset(warm_y, 0, sizeof(double) * m); }   /* (warm_rho: what the solve left, like OSQP's settings->rho) */
        else { memcpy(warm_x, x, sizeof(double) * n); memcpy(warm_y, y, sizeof(double) * m); }
    }
    return rc;
}

/* One MPC tick on the reference's UPDATE PATH (see osqp_solve_impl): carry = 2 + 2n + 4m doubles + 2 for the sparsity pattern of the previous tick's P
 * (osqp-eigen compares it in updateHessianMatrix), zero before the first tick of a robot. */
/* the general case of the reference's INTERFACE on the update path (per-step B_d: S/ConvexMpc.h:74 B_mat_d_list; a per-step contact schedule; A_c from another yaw,
 * S/test/test_mpc.cpp:94-122): the same persistent-solver semantics on the QP those inputs form */
int orc_mpc_solve_update_strided(const orc_mpc_params *pr, const orc_settings *st, const double *x0, const double *xref, double yaw_A, const double *Rw,
                                 const double *foot, int foot_stride, const uint8_t *contact, int contact_stride, double *grf_out, double *u_full, double *carry,
                                 orc_info *info) {
    const int h = pr->horizon, n = NU * h, m = NC * h;
    double *P = (double *)scratch_get(6, sizeof(double) * ((size_t)n * n + n + 2 * m + 36 * h + n + m), 0);
    double *g = P + (size_t)n * n, *l = g + n, *u = l + m, *av = u + m, *x = av + 36 * h, *y = x + n;
    int32_t *rp = (int32_t *)scratch_get(7, sizeof(int32_t) * (m + 1 + 36 * h), 0), *ci = rp + m + 1;
    orc_mpc_form(pr, x0, xref, yaw_A, Rw, foot, foot_stride, contact, contact_stride, P, g, rp, ci, av, l, u);
    memset(x, 0, sizeof(double) * n); memset(y, 0, sizeof(double) * m);
    double sig[2], *csig = carry + 2 + 2 * n + 4 * m;
    orc_pattern_signature(n, P, sig);
    const int changed = carry[0] != 0.0 && (sig[0] != csig[0] || sig[1] != csig[1]);
    csig[0] = sig[0]; csig[1] = sig[1];
    int rc = orc_osqp_solve_update_ex(n, m, P, g, rp, ci, av, l, u, st, x, y, carry, changed, info);
    info->reinit = changed;
    for (int leg = 0; leg < NLEG; ++leg) {
        const double *f = x + 3 * leg;
        int bad = isnan(f[0]) || isnan(f[1]) || isnan(f[2]);
        for (int i = 0; i < 3; ++i) grf_out[3 * leg + i] = bad ? 0.0 : Rw[0 * 3 + i] * f[0] + Rw[1 * 3 + i] * f[1] + Rw[2 * 3 + i] * f[2];
    }
    if (u_full) memcpy(u_full, x, sizeof(double) * n);
    return rc;
}

int orc_mpc_solve_update(const orc_mpc_params *pr, const orc_settings *st, const double *x0, const double *xref, const double *Rw,
                         const double *foot, const uint8_t *contact, double *grf_out, double *u_full, double *carry, orc_info *info) {
    return orc_mpc_solve_update_strided(pr, st, x0, xref, x0[2], Rw, foot, 0, contact, 0, grf_out, u_full, carry, info);
}

/* batch driver (CPU baseline): one problem per OpenMP thread, static partition */
int orc_mpc_solve_batch(const orc_mpc_params *pr, const orc_settings *st, int nb, const double *x0, const double *xref,
                        const double *Rw, const double *foot, const uint8_t *contact, double *grf_out, double *u_full,
                        int32_t *iters, int32_t *status, int32_t *nfact, int nthreads) {
    const int h = pr->horizon;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static)
#endif
    for (int b = 0; b < nb; ++b) {
        orc_info info;
        orc_mpc_solve(pr, st, x0 + (size_t)b * NS, xref + (size_t)b * NS * h, Rw + (size_t)b * 9, foot + (size_t)b * 12, 0,
                      contact + (size_t)b * 4, 0, grf_out + (size_t)b * 12, u_full ? u_full + (size_t)b * NU * h : 0, 0, 0, 0, &info);
        if (iters) iters[b] = info.iters;
        if (status) status[b] = info.status;
        if (nfact) nfact[b] = info.nfact;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Balance QP (S/A1RobotControl.cpp:11-48, 377-444)
 * ---------------------------------------------------------------------------------------- */
typedef struct orc_qp_params {
    double Qw[6];   /* :11  diag(1,1,1,400,400,100) */
    double R;       /* :12  1e-3 */
    double mu;      /* :13  0.7 */
    double F_min, F_max; /* :14-15 */
} orc_qp_params;

void orc_default_qp_params(orc_qp_params *p) {
    double q[6] = {1.0, 1.0, 1.0, 400.0, 400.0, 100.0};
    memcpy(p->Qw, q, sizeof q); p->R = 1e-3; p->mu = 0.7; p->F_min = 0; p->F_max = 180;
}

/* :379-391  desired wrench from the PD law (all 3-vectors; Rw row-major) */
void orc_balance_root_acc(const double *kp_lin, const double *kd_lin, const double *kp_ang, const double *kd_ang,
                          const double *pos_d, const double *pos, const double *lin_vel_d_body, const double *lin_vel_world,
                          const double *euler_d, const double *euler, const double *ang_vel_d_body, const double *ang_vel_world,
                          const double *Rw, double mass, double *root_acc) {
    double ee[3], t[3], t2[3];
    for (int i = 0; i < 3; ++i) ee[i] = euler_d[i] - euler[i];
    if (ee[2] > 3.1415926 * 1.5) ee[2] = euler_d[2] - 3.1415926 * 2 - euler[2];        /* :328-332 */
    else if (ee[2] < -3.1415926 * 1.5) ee[2] = euler_d[2] + 3.1415926 * 2 - euler[2];
    for (int i = 0; i < 3; ++i) root_acc[i] = kp_lin[i] * (pos_d[i] - pos[i]);
    for (int i = 0; i < 3; ++i) t[i] = lin_vel_d_body[i] - (Rw[0 * 3 + i] * lin_vel_world[0] + Rw[1 * 3 + i] * lin_vel_world[1] + Rw[2 * 3 + i] * lin_vel_world[2]);
    for (int i = 0; i < 3; ++i) t2[i] = kd_lin[i] * t[i];
    for (int i = 0; i < 3; ++i) root_acc[i] += Rw[i * 3 + 0] * t2[0] + Rw[i * 3 + 1] * t2[1] + Rw[i * 3 + 2] * t2[2];
    for (int i = 0; i < 3; ++i) root_acc[3 + i] = kp_ang[i] * ee[i];
    for (int i = 0; i < 3; ++i) root_acc[3 + i] += kd_ang[i] * (ang_vel_d_body[i] - (Rw[0 * 3 + i] * ang_vel_world[0] + Rw[1 * 3 + i] * ang_vel_world[1] + Rw[2 * 3 + i] * ang_vel_world[2]));
    root_acc[2] += mass * 9.8;
}

/* P (12x12), g(12), A CSR (20 rows, 36 nnz), l, u -- row order exactly as the ctor :28-48 */
void orc_balance_form(const orc_qp_params *qp, const double *root_acc, const double *Rz, const double *foot,
                      const uint8_t *contact, double *P, double *g, int32_t *rp, int32_t *ci, double *av, double *l, double *u) {
    double M[6 * 12], sk[9];
    memset(M, 0, sizeof M);
    for (int leg = 0; leg < NLEG; ++leg) {
        for (int i = 0; i < 3; ++i) M[i * 12 + 3 * leg + i] = 1.0;
        skew3(foot + 3 * leg, sk);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += Rz[k * 3 + i] * sk[k * 3 + j]; M[(3 + i) * 12 + 3 * leg + j] = s; } /* Rz' * skew */
    }
    for (int a = 0; a < 12; ++a) for (int b = 0; b < 12; ++b) { double s = (a == b) ? qp->R : 0.0; for (int k = 0; k < 6; ++k) s += M[k * 12 + a] * qp->Qw[k] * M[k * 12 + b]; P[a * 12 + b] = s; }
    for (int a = 0; a < 12; ++a) { double s = 0; for (int k = 0; k < 6; ++k) s += M[k * 12 + a] * qp->Qw[k] * root_acc[k]; g[a] = -s; }
    int nz = 0;
    for (int i = 0; i < NLEG; ++i) { rp[i] = nz; ci[nz] = 2 + 3 * i; av[nz++] = 1; double cf = contact[i] ? 1.0 : 0.0; l[i] = cf * qp->F_min; u[i] = cf * qp->F_max; }
    for (int i = 0; i < NLEG; ++i) {
        int r = NLEG + 4 * i;
        rp[r + 0] = nz; ci[nz] = 3 * i; av[nz++] = 1; ci[nz] = 3 * i + 2; av[nz++] = -qp->mu;
        rp[r + 1] = nz; ci[nz] = 3 * i; av[nz++] = -1; ci[nz] = 3 * i + 2; av[nz++] = -qp->mu;
        rp[r + 2] = nz; ci[nz] = 3 * i + 1; av[nz++] = 1; ci[nz] = 3 * i + 2; av[nz++] = -qp->mu;
        rp[r + 3] = nz; ci[nz] = 3 * i + 1; av[nz++] = -1; ci[nz] = 3 * i + 2; av[nz++] = -qp->mu;
        for (int k = 0; k < 4; ++k) { l[r + k] = -OSQP_INFTY; u[r + k] = 0; }
    }
    rp[20] = nz;
}

/* one balance-QP tick: cold OSQP (warm start OFF, :419), grf_body = R' f (:439-444) */
int orc_balance_solve(const orc_qp_params *qp, const orc_settings *st, const double *root_acc, const double *Rw, const double *Rz,
                      const double *foot, const uint8_t *contact, double *grf_out, double *f_world, orc_info *info) {
    double P[144], g[12], av[36], l[20], u[20], x[12], y[20];
    int32_t rp[21], ci[36];
    orc_balance_form(qp, root_acc, Rz, foot, contact, P, g, rp, ci, av, l, u);
    memset(x, 0, sizeof x); memset(y, 0, sizeof y);
    orc_settings s2 = *st; s2.warm_start = 0;
    int rc = orc_osqp_solve(12, 20, P, g, rp, ci, av, l, u, &s2, x, y, 0, info);
    for (int leg = 0; leg < NLEG; ++leg) {
        const double *f = x + 3 * leg;
        int bad = isnan(f[0]) || isnan(f[1]) || isnan(f[2]);
        for (int i = 0; i < 3; ++i) grf_out[3 * leg + i] = bad ? 0.0 : Rw[0 * 3 + i] * f[0] + Rw[1 * 3 + i] * f[1] + Rw[2 * 3 + i] * f[2];
    }
    if (f_world) memcpy(f_world, x, sizeof x);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * N2a: A1RobotControl::update_plan, S/A1RobotControl.cpp:148-202 -- gait counters, planned contacts,
 * Raibert foothold.  One robot; matrices 3x4 column-major (Eigen), rotation matrices row-major.
 * ---------------------------------------------------------------------------------------- */
typedef struct orc_gait_params {
    double counter_per_gait, counter_per_swing;   /* S/A1CtrlStates.h:24-25 */
    double control_dt;                            /* S/A1CtrlStates.h:332 */
    double foot_delta_x_limit, foot_delta_y_limit;/* S/A1Params.h:44-45 */
    double default_foot_pos[12];                  /* 3x4 column-major, S/A1CtrlStates.h:45 */
    double gait_counter_reset[4];                 /* S/A1CtrlStates.h:322-326 (gait_type 1: 0,120,120,0) */
} orc_gait_params;

void orc_update_plan(const orc_gait_params *gp, int movement_mode, double *gait_counter, const double *gait_counter_speed,
                     const double *root_lin_vel, const double *Rz, const double *Rw, const double *root_pos,
                     const double *root_lin_vel_d, uint8_t *plan_contacts, double *foot_pos_target_rel,
                     double *foot_pos_target_abs, double *foot_pos_target_world) {
    if (!movement_mode) {                                             /* :150-153 */
        for (int i = 0; i < NLEG; ++i) { plan_contacts[i] = 1; gait_counter[i] = gp->gait_counter_reset[i]; }
    } else {                                                          /* :155-165 */
        for (int i = 0; i < NLEG; ++i) {
            gait_counter[i] = gait_counter[i] + gait_counter_speed[i];
            gait_counter[i] = fmod(gait_counter[i], gp->counter_per_gait);
            plan_contacts[i] = gait_counter[i] <= gp->counter_per_swing ? 1 : 0;
        }
    }
    double lin_vel_rel[3];                                            /* :168-169  Rz' * v_world */
    for (int i = 0; i < 3; ++i) lin_vel_rel[i] = Rz[0 * 3 + i] * root_lin_vel[0] + Rz[1 * 3 + i] * root_lin_vel[1] + Rz[2 * 3 + i] * root_lin_vel[2];
    for (int k = 0; k < 12; ++k) foot_pos_target_rel[k] = gp->default_foot_pos[k];   /* :172 */
    for (int i = 0; i < NLEG; ++i) {                                  /* :173-201 */
        /* default_foot_pos(2) in the reference is LINEAR index 2 of the 3x4 matrix = z of leg 0 */
        double delta_x = sqrt(fabs(gp->default_foot_pos[2]) / 9.8) * (lin_vel_rel[0] - root_lin_vel_d[0]) +
                         ((gp->counter_per_swing / gait_counter_speed[i]) * gp->control_dt) / 2.0 * root_lin_vel_d[0];
        double delta_y = sqrt(fabs(gp->default_foot_pos[2]) / 9.8) * (lin_vel_rel[1] - root_lin_vel_d[1]) +
                         ((gp->counter_per_swing / gait_counter_speed[i]) * gp->control_dt) / 2.0 * root_lin_vel_d[1];
        if (delta_x < -gp->foot_delta_x_limit) delta_x = -gp->foot_delta_x_limit;
        if (delta_x > gp->foot_delta_x_limit) delta_x = gp->foot_delta_x_limit;
        if (delta_y < -gp->foot_delta_y_limit) delta_y = -gp->foot_delta_y_limit;
        if (delta_y > gp->foot_delta_y_limit) delta_y = gp->foot_delta_y_limit;
        foot_pos_target_rel[3 * i + 0] += delta_x;
        foot_pos_target_rel[3 * i + 1] += delta_y;
        for (int r = 0; r < 3; ++r) {
            double a = Rw[r * 3 + 0] * foot_pos_target_rel[3 * i + 0] + Rw[r * 3 + 1] * foot_pos_target_rel[3 * i + 1] +
                       Rw[r * 3 + 2] * foot_pos_target_rel[3 * i + 2];
            foot_pos_target_abs[3 * i + r] = a;
            foot_pos_target_world[3 * i + r] = a + root_pos[r];
        }
    }
}

/* ---- N3: A1RobotControl::compute_joint_torques, S/A1RobotControl.cpp:289-319 ------------------------------------------------
 * stance leg: tau = J' (-f_grf)                     (:303)
 * swing leg : tau = J^-1 (km .* f_kin)              (:306-307, Eigen jac.lu().solve(): PartialPivLU of a fixed 3x3 =
 *             Eigen/src/LU/PartialPivLU.h unblocked_lu -- pivot = first largest |a_ik|, row swap, column scaled by 1/pivot through
 *             a division per entry, rank-1 update -- then P b, unit-lower forward and upper backward substitution)
 * then + torques_gravity (:311) and the NaN guard (:314-317); all zeros while mpc_init_counter < 10 (:294-295, `active` = 0).
 * Jb: the four diagonal 3x3 blocks of j_foot, column-major each.  Eigen is not installed here: the LU order is restated from its
 * published algorithm, bit-level parity with a live Eigen build is unpinned (agreement with LAPACK-style solves is tested). */
void orc_joint_torques(int active, const uint8_t *contacts, const double *Jb, const double *grf, const double *f_kin, const double *km,
                       const double *torques_gravity, double *joint_torques) {
    if (!active) { for (int k = 0; k < 12; ++k) joint_torques[k] = 0.0; return; }
    for (int i = 0; i < NLEG; ++i) {
        const double *J = Jb + 9 * i;   /* J[r + 3 c] */
        double tau[3];
        if (contacts[i]) {
            for (int c = 0; c < 3; ++c) {     /* row c of J' times (-f): sum_r J(r,c) * (-f_r), left to right */
                tau[c] = J[0 + 3 * c] * -grf[3 * i + 0] + J[1 + 3 * c] * -grf[3 * i + 1] + J[2 + 3 * c] * -grf[3 * i + 2];
            }
        } else {
            double a[9], b[3];
            int perm[3];
            for (int k = 0; k < 9; ++k) a[k] = J[k];
            for (int k = 0; k < 3; ++k) b[k] = km[k] * f_kin[3 * i + k];
            for (int k = 0; k < 3; ++k) {
                int row = k; double big = fabs(a[k + 3 * k]);
                for (int r = k + 1; r < 3; ++r) if (fabs(a[r + 3 * k]) > big) { big = fabs(a[r + 3 * k]); row = r; }
                perm[k] = row;
                if (big != 0.0) {
                    if (row != k) for (int c = 0; c < 3; ++c) { double t = a[k + 3 * c]; a[k + 3 * c] = a[row + 3 * c]; a[row + 3 * c] = t; }
                    for (int r = k + 1; r < 3; ++r) a[r + 3 * k] /= a[k + 3 * k];
                }
                for (int r = k + 1; r < 3; ++r) for (int c = k + 1; c < 3; ++c) a[r + 3 * c] -= a[r + 3 * k] * a[k + 3 * c];
            }
            for (int k = 0; k < 3; ++k) if (perm[k] != k) { double t = b[k]; b[k] = b[perm[k]]; b[perm[k]] = t; }   /* P b */
            b[1] -= a[1 + 0] * b[0];                                   /* L y = P b, unit diagonal */
            b[2] -= a[2 + 0] * b[0] + a[2 + 3] * b[1];
            b[2] /= a[2 + 6];                                          /* U x = y */
            b[1] -= a[1 + 6] * b[2]; b[1] /= a[1 + 3];
            b[0] -= a[0 + 3] * b[1] + a[0 + 6] * b[2]; b[0] /= a[0];
            for (int k = 0; k < 3; ++k) tau[k] = b[k];
        }
        for (int k = 0; k < 3; ++k) {
            const double v = tau[k] + torques_gravity[3 * i + k];
            if (!isnan(v)) joint_torques[3 * i + k] = v;               /* :314-317 */
        }
    }
}

/* ---- N2b: contact logic + recent-contact filters (S/A1RobotControl.cpp:256-282), walking-surface fit (:566-582) and terrain pitch
 * (:335-376), with the stateful MovingWindowFilters (S/utils/filter.hpp:14-66; windows 60 and 100, S/A1RobotControl.cpp:52-56) -------
 * State of one robot = ORC_CT_STATE doubles: 13 filters (12 = [leg][x,y,z], 13th = terrain angle), each ORC_MWF doubles
 * [count, head, sum, correction, ring[100]], then early_contacts[4], foot_pos_recent_contact[12] (3x4 column-major).
 * The pseudo-inverse of the 3x3 SPD normal matrix (Utils::pseudo_inverse, S/utils/Utils.cpp:44-52, JacobiSVD + relative threshold) is
 * restated through a cyclic Jacobi eigen-decomposition (SVD = EVD for a symmetric PSD matrix); agreement with numpy's SVD-based pinv is
 * tested, bit parity with Eigen's JacobiSVD is not claimed. */
#define ORC_MWF 104
#define ORC_CT_STATE (13 * ORC_MWF + 4 + 12)
static double mwf_update(double *f, int window, double v) {        /* filter.hpp:26-39,53-66 */
    int count = (int)f[0], head = (int)f[1];
    double sum = f[2], corr = f[3];
    double *ring = f + 4;
#define NEUMAIER(val) do { const double v_ = (val); const double ns = sum + v_; \
        if (fabs(sum) >= fabs(v_)) corr += (sum - ns) + v_; else corr += (v_ - ns) + sum; sum = ns; } while (0)
    if (count >= window) { NEUMAIER(-ring[head]); } else { count += 1; }
    NEUMAIER(v);
    ring[head] = v;                         /* with a full window the oldest slot is the one just vacated */
    head = (head + 1) % window;
    f[0] = count; f[1] = head; f[2] = sum; f[3] = corr;
    return (sum + corr) / (double)window;
}
/* pinv of a symmetric PSD 3x3 (row-major m): V diag(1/l_i if l_i > eps*3*l_max else 0) V' */
static void sym3_pinv(const double *m, double *out) {
    double a[9], v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int k = 0; k < 9; ++k) a[k] = m[k];
    for (int sweep = 0; sweep < 8; ++sweep) {
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            const double apq = a[3 * p + q];
            if (apq == 0.0) continue;
            const double th = (a[3 * q + q] - a[3 * p + p]) / (2.0 * apq);
            const double t = (th >= 0.0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
            for (int k = 0; k < 3; ++k) { const double akp = a[3 * k + p], akq = a[3 * k + q]; a[3 * k + p] = c * akp - sn * akq; a[3 * k + q] = sn * akp + c * akq; }
            for (int k = 0; k < 3; ++k) { const double apk = a[3 * p + k], aqk = a[3 * q + k]; a[3 * p + k] = c * apk - sn * aqk; a[3 * q + k] = sn * apk + c * aqk; }
            for (int k = 0; k < 3; ++k) { const double vkp = v[3 * k + p], vkq = v[3 * k + q]; v[3 * k + p] = c * vkp - sn * vkq; v[3 * k + q] = sn * vkp + c * vkq; }
        }
    }
    double lmax = fabs(a[0]);
    if (fabs(a[4]) > lmax) lmax = fabs(a[4]);
    if (fabs(a[8]) > lmax) lmax = fabs(a[8]);
    const double tol = 2.220446049250313e-16 * 3.0 * lmax;
    double inv[3];
    for (int k = 0; k < 3; ++k) inv[k] = fabs(a[4 * k]) > tol ? 1.0 / a[4 * k] : 0.0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
        out[3 * i + j] = v[3 * i + 0] * inv[0] * v[3 * j + 0] + v[3 * i + 1] * inv[1] * v[3 * j + 1] + v[3 * i + 2] * inv[2] * v[3 * j + 2];
}
void orc_contact_terrain_step(double counter_per_swing, double foot_force_low, int use_terrain_adapt, double *state,
                              const double *gait_counter, const uint8_t *plan_contacts, const double *foot_force, const double *foot_pos_abs,
                              double root_pos_z, uint8_t *contacts, double *foot_pos_recent_contact, double *terrain_angle_out,
                              double *root_euler_d_pitch) {
    double *early = state + 13 * ORC_MWF, *recent = early + 4;
    for (int i = 0; i < NLEG; ++i) {                                              /* :259-282 */
        if (gait_counter[i] <= counter_per_swing * 1.5) early[i] = 0.0;
        if (!plan_contacts[i] && gait_counter[i] > counter_per_swing * 1.5 && foot_force[i] > foot_force_low) early[i] = 1.0;
        contacts[i] = (plan_contacts[i] || early[i] != 0.0) ? 1 : 0;
        if (contacts[i])
            for (int k = 0; k < 3; ++k) recent[3 * i + k] = mwf_update(state + (3 * i + k) * ORC_MWF, 60, foot_pos_abs[3 * i + k]);
    }
    for (int k = 0; k < 12; ++k) foot_pos_recent_contact[k] = recent[k];
    /* compute_walking_surface :566-582: a = pinv(W'W) W' z, W = [1 x y] */
    double M[9] = {0}, rhs[3] = {0}, P3[9], a[3];
    for (int i = 0; i < NLEG; ++i) {
        const double w[3] = {1.0, recent[3 * i + 0], recent[3 * i + 1]};
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) M[3 * r + c] += w[r] * w[c]; rhs[r] += w[r] * recent[3 * i + 2]; }
    }
    sym3_pinv(M, P3);
    for (int r = 0; r < 3; ++r) a[r] = P3[3 * r + 0] * rhs[0] + P3[3 * r + 1] * rhs[1] + P3[3 * r + 2] * rhs[2];
    const double s0 = a[1], s1 = a[2], s2 = -1.0;                                 /* surf_coef :580 */
    double terrain_angle = 0.0;                                                   /* :339-352 */
    if (root_pos_z > 0.1) {
        const double angle_cos = fabs(0.0 * s0 + 0.0 * s1 + 1.0 * s2) / (sqrt(0.0 * 0.0 + 0.0 * 0.0 + 1.0 * 1.0) * sqrt(s0 * s0 + s1 * s1 + s2 * s2));
        terrain_angle = mwf_update(state + 12 * ORC_MWF, 100, acos(angle_cos));
    }
    if (terrain_angle > 0.5) terrain_angle = 0.5;
    if (terrain_angle < -0.5) terrain_angle = -0.5;
    const double F_R_diff = recent[2] + recent[5] - recent[8] - recent[11];       /* :355 */
    if (use_terrain_adapt) *root_euler_d_pitch = F_R_diff > 0.05 ? -terrain_angle : terrain_angle;   /* :358-364 */
    *terrain_angle_out = terrain_angle;
}
int orc_contact_state_doubles(void) { return ORC_CT_STATE; }

/* ---- N4a: swing-leg targets and the foot PD force, the first block of generate_swing_legs_ctrl (S/A1RobotControl.cpp:204-254) with
 * BezierUtils::get_foot_pos_curve / bezier_curve (S/utils/Utils.cpp:64-104; FOOT_SWING_CLEARANCE1/2 = 0.0f / 0.4f, S/A1Params.h:41-42).
 * In/out per robot: foot_pos_start, foot_pos_rel_last_time, foot_pos_target_last_time (3x4 column-major each). */
static double bezier_curve4(double t, const double *P) {                           /* Utils.cpp:97-104 */
    static const double coefficients[5] = {1, 4, 6, 4, 1};
    const float bezier_degree = 4;                                                 /* Utils.h:29,42 */
    double y = 0;
    for (int i = 0; i <= bezier_degree; i++) y += coefficients[i] * pow(t, i) * pow(1 - t, bezier_degree - i) * P[i];
    return y;
}
void orc_swing_legs(double counter_per_swing, double dt, const double *Rz, const double *foot_pos_abs, const double *gait_counter,
                    const double *foot_pos_target_rel, const double *kp_foot, const double *kd_foot, double *foot_pos_start,
                    double *foot_pos_rel_last_time, double *foot_pos_target_last_time, double *foot_pos_cur_out, double *foot_forces_kin) {
    for (int i = 0; i < NLEG; ++i) {
        double cur[3], tgt[3];
        for (int r = 0; r < 3; ++r)                                                /* :224  Rz' * foot_pos_abs */
            cur[r] = Rz[0 * 3 + r] * foot_pos_abs[3 * i + 0] + Rz[1 * 3 + r] * foot_pos_abs[3 * i + 1] + Rz[2 * 3 + r] * foot_pos_abs[3 * i + 2];
        float spline_time = 0.0f;
        if (gait_counter[i] <= counter_per_swing) {                               /* :227-232 stance: keep refreshing the start point */
            for (int r = 0; r < 3; ++r) foot_pos_start[3 * i + r] = cur[r];
        } else {                                                                   /* :233-236 */
            spline_time = (float)(gait_counter[i] - counter_per_swing) / (float)counter_per_swing;
        }
        for (int r = 0; r < 3; ++r) {                                              /* get_foot_pos_curve, terrain_pitch_angle = 0.0 (:238-241) */
            double P[5] = {foot_pos_start[3 * i + r], foot_pos_start[3 * i + r], foot_pos_target_rel[3 * i + r], foot_pos_target_rel[3 * i + r],
                           foot_pos_target_rel[3 * i + r]};
            if (r == 2) { P[1] += 0.0f; P[2] += 0.4f + 0.5 * sin(0.0); }
            tgt[r] = bezier_curve4(spline_time, P);
        }
        for (int r = 0; r < 3; ++r) {                                              /* :243-252 */
            const int k = 3 * i + r;
            const double vel_cur = (cur[r] - foot_pos_rel_last_time[k]) / dt;
            foot_pos_rel_last_time[k] = cur[r];
            const double vel_tgt = (tgt[r] - foot_pos_target_last_time[k]) / dt;
            foot_pos_target_last_time[k] = tgt[r];
            const double pos_err = tgt[r] - cur[r], vel_err = vel_tgt - vel_cur;
            foot_forces_kin[k] = pos_err * kp_foot[r] + vel_err * kd_foot[r];
            foot_pos_cur_out[k] = cur[r];
        }
    }
}

/* ---- N4b: leg kinematics of one robot, the per-leg block of the joint-state callback (S/GazeboA1ROS.cpp:264-279) with
 * A1Kinematics::fk / jac (S/legKinematics/A1Kinematics.cpp:7-17; their bodies are MATLAB-generated trigonometric polynomials).
 * Restated from the leg model those polynomials expand: hip at (ox, oy), abduction q0 about x with lateral offset L = d + rho1,
 * thigh lt (q1) and calf lc (q2) in the sagittal plane, contact-point offsets rho0 (along the calf's x) and rho2 (shortening the calf):
 *   X  = ox - lt sin q1 - (lc - rho2) sin(q1+q2) + rho0 cos(q1+q2)
 *   Zp =    - lt cos q1 - (lc - rho2) cos(q1+q2) - rho0 sin(q1+q2)
 *   y  = oy + L cos q0 - Zp sin q0,   z = L sin q0 + Zp cos q0
 * (term-by-term identical to autoFunc_fk_derive after expanding the angle sums; the Jacobian is the analytic derivative = autoFunc_d_fk_dq).
 * rho_fix = [ox, oy, d, lt, lc] (:92-93), rho_opt = [rho0, rho1, rho2] (:94-95), Jb column-major per leg. */
void orc_leg_state(const double *joint_pos, const double *joint_vel, const double *Rw, const double *root_pos, const double *root_lin_vel,
                   const double *rho_fix /* 4x5 */, const double *rho_opt /* 4x3 */, double *foot_pos_rel, double *Jb, double *foot_vel_rel,
                   double *foot_pos_abs, double *foot_vel_abs, double *foot_pos_world, double *foot_vel_world) {
    for (int i = 0; i < NLEG; ++i) {
        const double *q = joint_pos + 3 * i, *qd = joint_vel + 3 * i, *f = rho_fix + 5 * i, *o = rho_opt + 3 * i;
        const double ox = f[0], oy = f[1], L = f[2] + o[1], lt = f[3], a = f[4] - o[2], r0 = o[0];
        const double s0 = sin(q[0]), c0 = cos(q[0]), s1 = sin(q[1]), c1 = cos(q[1]), s12 = sin(q[1] + q[2]), c12 = cos(q[1] + q[2]);
        const double Xq = r0 * c12 - a * s12, Zq = -(a * c12) - r0 * s12;      /* calf part */
        const double Xr = Xq - lt * s1, Zp = Zq - lt * c1;                      /* X - ox, Zp */
        double p[3] = {ox + Xr, oy + (L * c0 - Zp * s0), L * s0 + Zp * c0};
        double *J = Jb + 9 * i;
        J[0] = 0.0;        J[1] = -p[2];       J[2] = p[1] - oy;                /* d/dq0 */
        J[3] = Zp;         J[4] = s0 * Xr;     J[5] = -(c0 * Xr);               /* d/dq1 */
        J[6] = Zq;         J[7] = s0 * Xq;     J[8] = -(c0 * Xq);               /* d/dq2 */
        for (int r = 0; r < 3; ++r) {
            foot_pos_rel[3 * i + r] = p[r];
            foot_vel_rel[3 * i + r] = J[r] * qd[0] + J[3 + r] * qd[1] + J[6 + r] * qd[2];                         /* :273 */
        }
        for (int r = 0; r < 3; ++r) {                                                                            /* :275-279 */
            const double pa = Rw[3 * r] * foot_pos_rel[3 * i] + Rw[3 * r + 1] * foot_pos_rel[3 * i + 1] + Rw[3 * r + 2] * foot_pos_rel[3 * i + 2];
            const double va = Rw[3 * r] * foot_vel_rel[3 * i] + Rw[3 * r + 1] * foot_vel_rel[3 * i + 1] + Rw[3 * r + 2] * foot_vel_rel[3 * i + 2];
            foot_pos_abs[3 * i + r] = pa; foot_vel_abs[3 * i + r] = va;
            foot_pos_world[3 * i + r] = pa + root_pos[r]; foot_vel_world[3 * i + r] = va + root_lin_vel[r];
        }
    }
}

/* ---- N4c: A1BasicEKF (S/A1BasicEKF.cpp:7-163), the 18-state / 28-measurement Kalman filter for base position and velocity -----------
 * Dense restatement, matrix product by matrix product in the reference's evaluation order (left to right, inner index ascending).
 * state = [x 18 | P 18x18 row-major | initialised flag] = ORC_EKF_STATE doubles.  The two S.fullPivHouseholderQr().solve() calls (:134,138)
 * are restated as products with S^-1, computed ONCE by symmetric sweeps (in-place Gauss-Jordan without pivoting; S is symmetric positive definite; until round 3
 * the 47-wide tableau [S | error_y | C] was eliminated instead: 40 % more work for the same two solutions): same solution, rounding differs from a
 * Householder QR (agreement tested against LAPACK and against the reference's own source, 1e-9). */
#define EKF_NS 18
#define EKF_NM 28
#define ORC_EKF_STATE (EKF_NS + EKF_NS * EKF_NS + 1)
static void ekf_C(double *C) {                                       /* :10-17 */
    memset(C, 0, sizeof(double) * EKF_NM * EKF_NS);
    for (int i = 0; i < NLEG; ++i)
        for (int k = 0; k < 3; ++k) {
            C[(i * 3 + k) * EKF_NS + k] = -1.0;
            C[(i * 3 + k) * EKF_NS + 6 + i * 3 + k] = 1.0;
            C[(NLEG * 3 + i * 3 + k) * EKF_NS + 3 + k] = 1.0;
        }
    for (int i = 0; i < NLEG; ++i) C[(NLEG * 6 + i) * EKF_NS + 6 + i * 3 + 2] = 1.0;
}
/* device = 0: the PINNED restatement -- every product of S/A1BasicEKF.cpp:70-164 as a multiply followed by an add, the arithmetic the reference's source
 * states (compiled with -ffp-contract=off), the two solves as products with an explicit S^-1; tests/test_ref_pin.py holds THIS variant to the verbatim-compiled
 * reference.
 * device = 1: the device kernel's arithmetic for :134-139 (round 6: L D L' of S with the right-hand sides [C Pbar | error_y] riding along, see below; until then
 * the same explicit inverse with fma() accumulation in the four dense products).  It is not a second ground truth: tests hold it to the pinned variant within a
 * stated bound (the oracle must not follow the kernel). */
static void ekf_step_impl(int device, double *state, double dt, int assume_flat_ground, int movement_mode, const double *foot_force, const double *Rw,
                          const double *imu_acc, const double *imu_ang_vel, const double *foot_pos_rel, const double *foot_vel_rel,
                          double *root_pos, double *root_lin_vel, uint8_t *estimated_contacts_out) {
    double *x = state, *P = state + EKF_NS, *inited = state + EKF_NS + EKF_NS * EKF_NS;
    if (*inited == 0.0) {                                            /* init_state :54-68 */
        for (int i = 0; i < EKF_NS * EKF_NS; ++i) P[i] = 0.0;
        for (int i = 0; i < EKF_NS; ++i) P[i * EKF_NS + i] = 1.0 * 3;
        for (int i = 0; i < EKF_NS; ++i) x[i] = 0.0;
        x[2] = 0.09;
        for (int i = 0; i < NLEG; ++i)
            for (int r = 0; r < 3; ++r)
                x[6 + i * 3 + r] = (Rw[3 * r] * foot_pos_rel[3 * i] + Rw[3 * r + 1] * foot_pos_rel[3 * i + 1] + Rw[3 * r + 2] * foot_pos_rel[3 * i + 2]) + x[r];
        *inited = 1.0;
        for (int r = 0; r < 3; ++r) { root_pos[r] = x[r]; root_lin_vel[r] = x[3 + r]; }   /* the caller's state keeps its previous values in the reference; the filter state is returned here */
        for (int i = 0; i < NLEG; ++i) estimated_contacts_out[i] = 0;
        return;
    }
    static const double PIMU = 0.01, VIMU = 0.01, PFOOT = 0.01, S_PIMU_REL = 0.001, S_VIMU_REL = 0.1, S_ZFOOT = 0.001;   /* A1BasicEKF.h:15-20 */
    double A[EKF_NS * EKF_NS] = {0}, B[EKF_NS * 3] = {0}, C[EKF_NM * EKF_NS], Q[EKF_NS] , Rd[EKF_NM], ec[NLEG];
    for (int i = 0; i < EKF_NS; ++i) A[i * EKF_NS + i] = 1.0;
    for (int k = 0; k < 3; ++k) { A[k * EKF_NS + 3 + k] = dt; B[(3 + k) * 3 + k] = dt; }            /* :72-73 */
    ekf_C(C);
    double u[3];                                                                                      /* :76 */
    for (int r = 0; r < 3; ++r) u[r] = (Rw[3 * r] * imu_acc[0] + Rw[3 * r + 1] * imu_acc[1] + Rw[3 * r + 2] * imu_acc[2]) + (r == 2 ? -9.81 : 0.0);
    for (int i = 0; i < NLEG; ++i)                                                                    /* :79-86 */
        ec[i] = movement_mode == 0 ? 1.0 : fmin(fmax(foot_force[i] / (100.0 - 0.0), 0.0), 1.0);
    for (int k = 0; k < 3; ++k) { Q[k] = PIMU * dt / 20.0; Q[3 + k] = VIMU * dt * 9.8 / 20.0; }     /* :88-89 */
    for (int i = 0; i < EKF_NM; ++i) Rd[i] = 1.0;                                                     /* R.setIdentity(), blocks below overwrite every entry */
    for (int i = 0; i < NLEG; ++i) {                                                                  /* :91-107 */
        const double w = 1 + (1 - ec[i]) * 1e3;
        for (int k = 0; k < 3; ++k) {
            Q[6 + i * 3 + k] = w * dt * PFOOT;
            Rd[i * 3 + k] = w * S_PIMU_REL;
            Rd[NLEG * 3 + i * 3 + k] = w * S_VIMU_REL;
        }
        Rd[NLEG * 6 + i] = assume_flat_ground ? w * S_ZFOOT : 1e5;                                   /* :42-53, :103-106 */
    }
    double xbar[EKF_NS], T[EKF_NS * EKF_NS], Pbar[EKF_NS * EKF_NS];
    for (int i = 0; i < EKF_NS; ++i) {                                                                /* :111 */
        double a = 0, b = 0;
        for (int k = 0; k < EKF_NS; ++k) a += A[i * EKF_NS + k] * x[k];
        for (int k = 0; k < 3; ++k) b += B[i * 3 + k] * u[k];
        xbar[i] = a + b;
    }
    for (int i = 0; i < EKF_NS; ++i) for (int j = 0; j < EKF_NS; ++j) { double a = 0; for (int k = 0; k < EKF_NS; ++k) a += A[i * EKF_NS + k] * P[k * EKF_NS + j]; T[i * EKF_NS + j] = a; }
    for (int i = 0; i < EKF_NS; ++i) for (int j = 0; j < EKF_NS; ++j) {                               /* :112 */
        double a = 0; for (int k = 0; k < EKF_NS; ++k) a += T[i * EKF_NS + k] * A[j * EKF_NS + k];
        Pbar[i * EKF_NS + j] = a + (i == j ? Q[i] : 0.0);
    }
    double yhat[EKF_NM], y[EKF_NM];
    for (int r = 0; r < EKF_NM; ++r) { double a = 0; for (int k = 0; k < EKF_NS; ++k) a += C[r * EKF_NS + k] * xbar[k]; yhat[r] = a; }   /* :115 */
    const double wx = imu_ang_vel[0], wy = imu_ang_vel[1], wz = imu_ang_vel[2];
    for (int i = 0; i < NLEG; ++i) {                                                                  /* :119-128 */
        const double *fk = foot_pos_rel + 3 * i, *fv = foot_vel_rel + 3 * i;
        const double sk[3] = {0.0 * fk[0] + -wz * fk[1] + wy * fk[2], wz * fk[0] + 0.0 * fk[1] + -wx * fk[2], -wy * fk[0] + wx * fk[1] + 0.0 * fk[2]};  /* skew(w) fk */
        const double lv[3] = {-fv[0] - sk[0], -fv[1] - sk[1], -fv[2] - sk[2]};
        for (int r = 0; r < 3; ++r) {
            y[i * 3 + r] = Rw[3 * r] * fk[0] + Rw[3 * r + 1] * fk[1] + Rw[3 * r + 2] * fk[2];
            const double rl = Rw[3 * r] * lv[0] + Rw[3 * r + 1] * lv[1] + Rw[3 * r + 2] * lv[2];
            y[NLEG * 3 + i * 3 + r] = (1.0 - ec[i]) * x[3 + r] + ec[i] * rl;
        }
        y[NLEG * 6 + i] = (1.0 - ec[i]) * (x[2] + fk[2]) + ec[i] * 0;
    }
    double CP[EKF_NM * EKF_NS], M[EKF_NM * EKF_NM], err[EKF_NM], Serr[EKF_NM], SC[EKF_NM * EKF_NS];
    for (int r = 0; r < EKF_NM; ++r) for (int j = 0; j < EKF_NS; ++j) { double a = 0; for (int k = 0; k < EKF_NS; ++k) a += C[r * EKF_NS + k] * Pbar[k * EKF_NS + j]; CP[r * EKF_NS + j] = a; }
    for (int r = 0; r < EKF_NM; ++r) for (int c = 0; c < EKF_NM; ++c) {                               /* :130 */
        double a = 0; for (int k = 0; k < EKF_NS; ++k) a += CP[r * EKF_NS + k] * C[c * EKF_NS + k];
        M[r * EKF_NM + c] = a + (r == c ? Rd[r] : 0.0);
    }
    for (int r = 0; r < EKF_NM; ++r) for (int c = r + 1; c < EKF_NM; ++c) {                           /* :131 */
        const double v = 0.5 * (M[r * EKF_NM + c] + M[c * EKF_NM + r]); M[r * EKF_NM + c] = v; M[c * EKF_NM + r] = v;
    }
    for (int r = 0; r < EKF_NM; ++r) { M[r * EKF_NM + r] = 0.5 * (M[r * EKF_NM + r] + M[r * EKF_NM + r]); err[r] = y[r] - yhat[r]; }   /* :133 */
    if (device) {
        /* The device kernel's arithmetic (round 6).  K = Pbar C' S^-1 never appears: with S = L D L' (forward elimination without pivoting; S is symmetric positive
         * definite) and Y = L^-1 [C Pbar | error_y] -- the right-hand sides ride along in the elimination -- the two updates (:136, :139) are
         *     x = xbar + Y_P' D^-1 y_e,      P = Pbar - Y_P' D^-1 Y_P
         * and nothing is solved backwards.  Step k, p = a_kk:  f_i = a_ik / p,  a_ij -= f_i a_jk (j > k);  t_c = b_kc / p,  b_ic -= a_ik t_c  for the rows i > k.  Row i takes the pivot row
         * from COLUMN k as the other rows hold it (a_jk for a_kj: on the device row i lives in lane i and one word per lane is what a step exchanges); only entries of the
         * lower triangle ever feed another entry, so this IS the standard right-looking L D L' -- the upper triangle a lane drags along is never read.  A third of the
         * explicit inverse's arithmetic (28^3 / 3 + 28^2 19 / 2 multiply-adds against 2 28^3) and four dense products fewer behind it. */
        double Bm[EKF_NM][EKF_NS + 1], dinv[EKF_NM];
        for (int r = 0; r < EKF_NM; ++r) { for (int c = 0; c < EKF_NS; ++c) Bm[r][c] = CP[r * EKF_NS + c]; Bm[r][EKF_NS] = err[r]; }
        for (int k = 0; k < EKF_NM; ++k) {
            double col[EKF_NM];
            for (int j = 0; j < EKF_NM; ++j) col[j] = M[j * EKF_NM + k];
            const double pinv = 1.0 / col[k];
            dinv[k] = pinv;
            double t[EKF_NS + 1];
            for (int c = 0; c <= EKF_NS; ++c) t[c] = Bm[k][c] * pinv;
            for (int i = k + 1; i < EKF_NM; ++i) {
                const double f = col[i] * pinv;
                for (int j = k + 1; j < EKF_NM; ++j) M[i * EKF_NM + j] = fma(-f, col[j], M[i * EKF_NM + j]);
                for (int c = 0; c <= EKF_NS; ++c) Bm[i][c] = fma(-col[i], t[c], Bm[i][c]);
            }
        }
        for (int a_ = 0; a_ < EKF_NS; ++a_) {                                                         /* :136, :139 */
            double acc = 0, G[EKF_NS] = {0};
            for (int r = 0; r < EKF_NM; ++r) {
                const double w = Bm[r][a_] * dinv[r];
                acc = fma(w, Bm[r][EKF_NS], acc);
                for (int j = 0; j < EKF_NS; ++j) G[j] = fma(w, Bm[r][j], G[j]);
            }
            x[a_] = xbar[a_] + acc;
            for (int j = 0; j < EKF_NS; ++j) T[a_ * EKF_NS + j] = Pbar[a_ * EKF_NS + j] - G[j];
        }
    } else {
        /* S^-1 by the symmetric sweep operator (in-place Gauss-Jordan without pivoting; S is symmetric positive definite); the two solves (:134, :138) are then
         * products with it.  Sweep k with p = a_kk:  a_ij -= (a_ik / p) a_kj,  a_ik = a_ik / p,  a_kj = a_kj / p,  a_kk = -1 / p; the matrix stays symmetric and ends
         * as -S^-1.  Row i takes the pivot row from COLUMN k as the other rows hold it (a_jk for a_kj): on the device row i lives in lane i, and one word per lane is
         * all a sweep has to exchange.  That is exact because the update is evaluated as fma(-(a_ik a_jk), 1/p, a_ij): the product commutes, rows i and j compute
         * the same bits for a_ij and a_ji, and the matrix stays symmetric to the last bit.  (With f = a_ik / p first -- one operation fewer -- a_ij and a_ji drift
         * apart by a rounding per sweep and the inverse loses a digit and a half componentwise: tried, 8.6e-13 against 2.8e-14 on the filter's graded S.) */
        for (int k = 0; k < EKF_NM; ++k) {
            double col[EKF_NM];
            for (int j = 0; j < EKF_NM; ++j) col[j] = M[j * EKF_NM + k];
            const double pinv = 1.0 / col[k];
            for (int i = 0; i < EKF_NM; ++i) {
                if (i == k) { for (int j = 0; j < EKF_NM; ++j) M[k * EKF_NM + j] = j == k ? -pinv : M[k * EKF_NM + j] * pinv; continue; }
                const double aik = M[i * EKF_NM + k];
                for (int j = 0; j < EKF_NM; ++j) M[i * EKF_NM + j] = j == k ? aik * pinv : fma(-(aik * col[j]), pinv, M[i * EKF_NM + j]);
            }
        }
        for (int i = 0; i < EKF_NM * EKF_NM; ++i) M[i] = -M[i];
        for (int r = 0; r < EKF_NM; ++r) { double a = 0; for (int c = 0; c < EKF_NM; ++c) a += M[r * EKF_NM + c] * err[c]; Serr[r] = a; }                        /* :134 */
        for (int r = 0; r < EKF_NM; ++r) for (int j = 0; j < EKF_NS; ++j) { double a = 0; for (int c = 0; c < EKF_NM; ++c) a += M[r * EKF_NM + c] * C[c * EKF_NS + j]; SC[r * EKF_NS + j] = a; }   /* :138 */
        double G1[EKF_NS * EKF_NM], G2[EKF_NS * EKF_NS];
        for (int a_ = 0; a_ < EKF_NS; ++a_) for (int r = 0; r < EKF_NM; ++r) { double a = 0; for (int k = 0; k < EKF_NS; ++k) a += Pbar[a_ * EKF_NS + k] * C[r * EKF_NS + k]; G1[a_ * EKF_NM + r] = a; }
        for (int a_ = 0; a_ < EKF_NS; ++a_) {                                                             /* :136 */
            double a = 0; for (int r = 0; r < EKF_NM; ++r) a += G1[a_ * EKF_NM + r] * Serr[r];
            x[a_] = xbar[a_] + a;
        }
        for (int a_ = 0; a_ < EKF_NS; ++a_) for (int j = 0; j < EKF_NS; ++j) { double a = 0; for (int r = 0; r < EKF_NM; ++r) a += G1[a_ * EKF_NM + r] * SC[r * EKF_NS + j]; G2[a_ * EKF_NS + j] = a; }
        for (int a_ = 0; a_ < EKF_NS; ++a_) for (int j = 0; j < EKF_NS; ++j) {                            /* :139 */
            double a = 0; for (int k = 0; k < EKF_NS; ++k) a += G2[a_ * EKF_NS + k] * Pbar[k * EKF_NS + j];
            T[a_ * EKF_NS + j] = Pbar[a_ * EKF_NS + j] - a;
        }
    }
    for (int i = 0; i < EKF_NS; ++i) for (int j = 0; j < EKF_NS; ++j) P[i * EKF_NS + j] = 0.5 * (T[i * EKF_NS + j] + T[j * EKF_NS + i]);   /* :140 */
    if (P[0] * P[EKF_NS + 1] - P[1] * P[EKF_NS] > 1e-6) {                                             /* :143-147 */
        for (int i = 0; i < 2; ++i) for (int j = 2; j < EKF_NS; ++j) { P[i * EKF_NS + j] = 0.0; P[j * EKF_NS + i] = 0.0; }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) P[i * EKF_NS + j] /= 10.0;
    }
    for (int i = 0; i < NLEG; ++i) estimated_contacts_out[i] = ec[i] < 0.5 ? 0 : 1;                   /* :151-157 */
    for (int r = 0; r < 3; ++r) { root_pos[r] = x[r]; root_lin_vel[r] = x[3 + r]; }                   /* :159-163 */
}
/* the pinned restatement (multiply + add, as the reference's source states it) */
void orc_ekf_step(double *state, double dt, int assume_flat_ground, int movement_mode, const double *foot_force, const double *Rw,
                  const double *imu_acc, const double *imu_ang_vel, const double *foot_pos_rel, const double *foot_vel_rel,
                  double *root_pos, double *root_lin_vel, uint8_t *estimated_contacts_out) {
    ekf_step_impl(0, state, dt, assume_flat_ground, movement_mode, foot_force, Rw, imu_acc, imu_ang_vel, foot_pos_rel, foot_vel_rel, root_pos, root_lin_vel, estimated_contacts_out);
}
/* the device kernel's arithmetic (L D L' with the right-hand sides riding along, no explicit S^-1): bit-comparable with a1mpc_ekf_update_batch, held to orc_ekf_step by the tests */
void orc_ekf_step_device(double *state, double dt, int assume_flat_ground, int movement_mode, const double *foot_force, const double *Rw,
                      const double *imu_acc, const double *imu_ang_vel, const double *foot_pos_rel, const double *foot_vel_rel,
                      double *root_pos, double *root_lin_vel, uint8_t *estimated_contacts_out) {
    ekf_step_impl(1, state, dt, assume_flat_ground, movement_mode, foot_force, Rw, imu_acc, imu_ang_vel, foot_pos_rel, foot_vel_rel, root_pos, root_lin_vel, estimated_contacts_out);
}
int orc_ekf_state_doubles(void) { return ORC_EKF_STATE; }

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
