/*
 * a1mpc_oracle_api.h -- TEST INFRASTRUCTURE ONLY (see a1mpc_oracle.c).
 * The part of the CPU oracle's interface that C/C++ test code links against: the OSQP-0.6 restatement
 * (orc_osqp_solve) with its settings / info records.  Included by a1mpc_oracle.c itself (after its optional
 * `#define double long double` of the x87 yardstick build) and by oracle/ref_shim/OsqpEigen/OsqpEigen.h.
 */
#ifndef A1MPC_ORACLE_API_H
#define A1MPC_ORACLE_API_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* status values mirror OSQP's */
#define ORC_SOLVED 1
#define ORC_SOLVED_INACCURATE 2
#define ORC_MAX_ITER_REACHED (-2)
#define ORC_PRIMAL_INFEASIBLE (-3)
#define ORC_DUAL_INFEASIBLE (-4)
#define ORC_NON_CVX (-7)
#define ORC_UNSOLVED (-10)

typedef struct orc_settings {
    double rho, sigma, alpha, eps_abs, eps_rel, eps_prim_inf, eps_dual_inf, adaptive_rho_tolerance;
    int32_t max_iter, scaling, check_termination, adaptive_rho, adaptive_rho_interval, warm_start;
    int32_t linsys;     /* 0 = reduced system by dense Cholesky (default), 1 = LDL' of the full quasi-definite KKT matrix (QDLDL-style) */
    int32_t reserved_;
} orc_settings;

typedef struct orc_info {
    int32_t iters, status, rho_updates, nfact;
    double pri_res, dua_res, rho_final;
    int32_t reinit, pad_;   /* update path: this tick re-initialised the solver (osqp-eigen: the sparsity pattern of P changed) */
} orc_info;

void orc_default_settings(orc_settings *s);
int orc_osqp_solve(int n, int m, const double *P, const double *q, const int32_t *rp, const int32_t *ci, const double *av,
                   const double *l, const double *u, const orc_settings *st, double *x, double *y, double *rho_io,
                   orc_info *info);
/* the same solve on OSQP's UPDATE path (a persistent workspace between ticks): carry = 2 + 2n + 4m doubles, zero before the first tick (a1mpc_oracle.c, osqp_solve_impl) */
int orc_osqp_solve_update(int n, int m, const double *P, const double *q, const int32_t *rp, const int32_t *ci, const double *av,
                          const double *l, const double *u, const orc_settings *st, double *x, double *y, double *carry, orc_info *info);
/* ... when the caller has found the sparsity pattern of P changed since the previous tick (osqp-eigen's updateHessianMatrix): re-initialisation + warm start */
int orc_osqp_solve_update_ex(int n, int m, const double *P, const double *q, const int32_t *rp, const int32_t *ci, const double *av,
                             const double *l, const double *u, const orc_settings *st, double *x, double *y, double *carry, int pattern_changed, orc_info *info);
#ifdef __cplusplus
}
#endif
#endif
