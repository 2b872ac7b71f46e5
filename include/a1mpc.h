/*
 * a1mpc.h -- C ABI of the MI355X batched convex-MPC QP engine (liba1mpc.so).
 *
 * Drop-in boundary for the ONE hot path of ShuoYangRobotics/A1-QP-MPC-Controller: QP formation +
 * OSQP solve inside A1RobotControl::compute_grf.  Citations are reference file:line with
 * S/ = src/a1_cpp/src/.  The reference has no FFI; the C++ call sites this ABI replaces are
 *
 *   a1mpc_config                <- compile-time dims S/A1Params.h:26-34, ConvexMpc ctor constants
 *                                  S/ConvexMpc.cpp:8-44,223-224, A1CtrlStates fields S/A1CtrlStates.h:358-366,
 *                                  mpc_dt S/A1RobotControl.cpp:462, OSQP settings S/A1RobotControl.cpp:523-524
 *   a1mpc_create / _destroy     <- `OsqpEigen::Solver solver` member (S/A1RobotControl.h:67) + per-tick
 *                                  `ConvexMpc mpc_solver(q, r); reset()` (S/A1RobotControl.cpp:447-448)
 *   a1mpc_solve_batch           <- MPC branch of compute_grf, S/A1RobotControl.cpp:446-562:
 *                                  calculate_A_mat_c / calculate_B_mat_c / state_space_discretization /
 *                                  calculate_qp_mats (S/ConvexMpc.cpp:110-260), solver.update*()/solve()
 *                                  (:522-540), R' * solution[0:12] (:555-561)          -- n ticks at once
 *   a1mpc_balance_solve_batch   <- balance-QP branch of compute_grf, S/A1RobotControl.cpp:377-444 (+ ctor :11-48)
 *   a1mpc_reset_warm_start      <- destroying / re-creating the persistent OSQP workspace
 *
 * Plain pointers and sizes only.  All matrices use the reference's (Eigen) storage: 3x4 GRF / foot
 * matrices are column-major (leg-major, 12 doubles), root_rot_mat is passed ROW-major (9 doubles).
 * Streams: the *_device entry points launch on the stream the caller passes (NULL = the handle's own).  The handle owns scratch that
 * every launch uses, so consecutive calls on DIFFERENT streams are ordered by the library (the later call waits on the device for the
 * earlier one; calls on the same stream are ordered by the stream) -- a handle never runs two launches concurrently; use one handle
 * per stream for that.
 * Caller owns every host array (pageable is fine; the library snapshots inputs at entry because the
 * surrounding control program mutates A1CtrlStates without locks, S/MainGazebo.cpp:47-121).  A handle is
 * used by one thread at a time (the reference's thread 1); different handles are independent.
 * Nothing throws across this boundary.  There is NO CPU fallback: without a working HIP device every
 * entry point returns an error.
 */
#ifndef A1MPC_H_
#define A1MPC_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define A1MPC_STATE_DIM 13      /* MPC_STATE_DIM      S/A1Params.h:27 */
#define A1MPC_NUM_DOF 12        /* NUM_DOF            S/A1Params.h:34 */
#define A1MPC_CONSTRAINT_DIM 20 /* MPC_CONSTRAINT_DIM S/A1Params.h:28 */
#define A1MPC_NUM_LEG 4         /* NUM_LEG            S/A1Params.h:31 */

typedef enum a1mpc_status {
    A1MPC_OK = 0,
    A1MPC_ERR_INVALID_ARGUMENT = 1,
    A1MPC_ERR_UNSUPPORTED_HORIZON = 2, /* horizons compiled in: 1, 4, 6, 8, 10, 12, 14, 16, 20 (a1mpc_config.horizon) */
    A1MPC_ERR_NO_DEVICE = 3,
    A1MPC_ERR_HIP = 4,
    A1MPC_ERR_BATCH_TOO_LARGE = 5
} a1mpc_status;

/* per-problem solver status written to status_out[] -- OSQP's numbering, which the reference ignores
 * (S/A1RobotControl.cpp:540); on A1MPC_QP_NON_CVX the GRFs are zeros (replaces the reference's
 * uninitialised-matrix behaviour, S/A1RobotControl.cpp:322,559) */
#define A1MPC_QP_SOLVED 1
#define A1MPC_QP_SOLVED_INACCURATE 2
#define A1MPC_QP_MAX_ITER_REACHED (-2)
#define A1MPC_QP_NON_CVX (-7)
/* OSQP's other outcomes -- PRIMAL_INFEASIBLE (-3), PRIMAL_INFEASIBLE_INACCURATE (3), DUAL_INFEASIBLE (-4), DUAL_INFEASIBLE_INACCURATE (4) -- are UNREACHABLE here by
 * construction, and the kernels do not evaluate OSQP's infeasibility certificates (auxil.c is_primal_infeasible / is_dual_infeasible):
 *   * every configuration on which they could fire is refused with A1MPC_ERR_INVALID_ARGUMENT by a1mpc_create / a1mpc_update_config / a1mpc_sharded_create /
 *     a1mpc_pipeline_create (and the balance-QP constants by a1mpc_balance_solve_batch): mu < 0, fz_min > fz_max, fz_max < 0, a negative or non-finite weight,
 *     mass / dt <= 0, a singular body inertia, and every OSQP setting osqp_setup itself refuses (auxil.c validate_settings: rho, sigma <= 0, alpha outside (0, 2),
 *     negative tolerances, ...).  OSQP refuses the same data in osqp_setup (validate_data: l <= u); the reference ignores that return code
 *     (S/A1RobotControl.cpp:532,540) and goes on to read an uninitialised solution (:559) -- a deliberate deviation: the bad configuration is reported where it is given.
 *   * on every accepted configuration the per-problem inputs (states, feet, contacts) cannot make the QP infeasible or unbounded: the bounds come from the
 *     configuration and the contact flags only, (0, 0, max(fz_min, 0) c) is always feasible, the feasible set is compact and P = B'QB + R is positive semi-definite.
 * Non-finite INPUTS surface as A1MPC_QP_NON_CVX + zero forces (OSQP: a NaN residual ends the solve as OSQP_NON_CVX). */

typedef struct a1mpc_config {
    int32_t horizon; /* PLAN_HORIZON (S/A1Params.h:26 fixes 10; a run-time value here).  1, 10, 16, 20: the tuned horizons (the balance-QP analogue, the reference's own,
                        BASELINE's h = 16 / 20 configurations).  4, 6, 8, 12, 14: the same kernel families -- fast path and general path -- as they instantiate for them
                        (every entry point: all warm-start modes, contact schedules, tick records, per-step feet / yaw_A, pipelines, sharding), not tuned beyond that.  Anything
                        else -- odd horizons, 2, 18, > 20 -- is refused by a1mpc_create with A1MPC_ERR_UNSUPPORTED_HORIZON */
    double dt;       /* mpc_dt, S/A1RobotControl.cpp:462 */
    double mu;       /* S/ConvexMpc.cpp:8 */
    double fz_min;   /* S/ConvexMpc.cpp:223 */
    double fz_max;   /* S/ConvexMpc.cpp:224 */
    double q[A1MPC_STATE_DIM]; /* q_weights, S/A1CtrlStates.h:365 */
    double r[A1MPC_NUM_DOF];   /* r_weights, S/A1CtrlStates.h:366 */
    double mass;               /* robot_mass, S/A1CtrlStates.h:358 */
    double inertia_body[9];    /* a1_trunk_inertia, row-major, S/A1CtrlStates.h:359-360 */
    /* OSQP 0.6 settings; the reference only sets verbosity and warm start, everything else is the default */
    double rho, sigma, alpha, eps_abs, eps_rel, adaptive_rho_tolerance;
    int32_t max_iter;
    int32_t check_termination;     /* iterations between termination checks (25) */
    int32_t adaptive_rho;          /* 1 */
    int32_t adaptive_rho_interval; /* > 0: rho is re-estimated every that many iterations.  0 = OSQP's default "automatic", the
                                      setting the reference runs with: OSQP derives the period from measured wall-clock set-up time
                                      (non-reproducible); here 0 resolves deterministically to the outcome of OSQP's rule for these QP
                                      sizes, max(check_termination, 25-multiple) = check_termination (25).  a1mpc_default_config: 25 */
    int32_t scaling;               /* Ruiz passes (10) */
    int32_t warm_start;            /* 0: cold start every solve (the reference's balance-QP path and S/test/test_mpc.cpp).
                                      1: every tick is a fresh OSQP set-up warm-started from the previous tick's (x, y, rho) as osqp_warm_start defines it.
                                      2: the reference's UPDATE path on ticks >= 2 (S/A1RobotControl.cpp:533-538: updateHessianMatrix / updateGradient /
                                         update*Bound on the persistent OsqpEigen workspace, then solve()): OSQP re-equilibrates with the PREVIOUS tick's
                                         gradient still in the workspace and starts from the previous solve's SCALED (x, z, y) as they are.  The handle then
                                         also carries the previous scalings, gradient and z of every problem.  Restated from OSQP 0.6's update functions
                                         (oracle: orc_mpc_solve_update); every horizon > 1, on the fast path and on (round 5; round 6: at every batch size) the general
                                         path (per-step feet / contact schedules / its own A_c yaw: a1mpc_solve_batch_strided; its pattern-change test reads the
                                         zero patterns of the per-step tables, a superset of the changes osqp-eigen sees).  Horizon 1 (the balance QP, which the
                                         reference cold-starts anyway) behaves like 1: a1mpc_last_warm_start_mode reports which semantics a solve actually ran.
                                         When the sparsity pattern of the reference's Hessian (dense.sparseView(), S/ConvexMpc.cpp:211: exact zeros are not stored)
                                         changes from one tick to the next, osqp-eigen cannot take osqp_update_P: updateHessianMatrix clears the solver, initialises
                                         it again (rho back to cfg.rho, fresh scaling) and warm-starts it with the workspace's scaled iterates -- reproduced per
                                         problem (the pattern is a function of the zero patterns of the two 12x12 blocks the Hessian is made of).  A tick solved
                                         in another mode or through the general path's split pipeline drops the carried workspace: the next tick is a fresh set-up
                                         warm-started from (x, y, rho).  (An a1mpc_warm_start injection does NOT drop it since round 4: see there.) */
} a1mpc_config;

/* balance-QP constants, A1RobotControl ctor S/A1RobotControl.cpp:11-15 */
typedef struct a1mpc_balance_config {
    double Q[6];         /* diag(1,1,1,400,400,100) */
    double R;            /* 1e-3 */
    double mu;           /* 0.7 */
    double F_min, F_max; /* 0, 180 */
} a1mpc_balance_config;

typedef struct a1mpc_handle_s* a1mpc_handle;

/* Reference constants (mu 0.3, fz in [0,180], dt 0.0025, horizon 10), OSQP defaults, warm start on;
 * q, r, mass, inertia are left zero for the caller (they come from the rosparam YAML). */
void a1mpc_default_config(a1mpc_config* cfg);
void a1mpc_default_balance_config(a1mpc_balance_config* cfg);

/* device: HIP device ordinal (>= 0).  max_batch: largest n any later call will pass.  The configuration is validated first (see the status
 * values above for what is refused and why): A1MPC_ERR_INVALID_ARGUMENT + a1mpc_last_error() naming the field, whether or not a GPU is present. */
a1mpc_status a1mpc_create(const a1mpc_config* cfg, int32_t max_batch, int32_t device, a1mpc_handle* out);
void a1mpc_destroy(a1mpc_handle h);

/*
 * n independent MPC ticks.  Host pointers.
 *   x0      n x 13   mpc_states                    (S/A1RobotControl.cpp:452-456)
 *   x_ref   n x 13H  mpc_states_d                  (S/A1RobotControl.cpp:470-488)
 *   R_world n x 9    root_rot_mat, row-major
 *   foot    n x 12   foot_pos_abs, 3x4 column-major (same feet for all H steps, S/A1RobotControl.cpp:498-514)
 *   contact n x 4    contacts[], broadcast over the horizon (S/ConvexMpc.cpp:228-245)
 * out:
 *   grf_body_out n x 12  3x4 column-major, body frame (the value compute_grf returns)
 *   u_full_out   n x 12H world-frame forces of every horizon step, or NULL
 *   iters_out, status_out  n each, or NULL
 * With cfg.warm_start the handle keeps (x, y, rho) of problem i between calls, like the reference's
 * persistent OSQP workspace.
 * A handful of QPs (n <= 8: the reference's own call is n = 1) are read from and written to the handle's pinned block by the
 * kernel itself, and the call returns as soon as the last output word has arrived (polled; round 6) -- the handle's stream may
 * still be retiring the launch for a few microseconds, which later calls and every a1mpc_last_* query order themselves behind.
 */
a1mpc_status a1mpc_solve_batch(a1mpc_handle h, int32_t n, const double* x0, const double* x_ref, const double* R_world,
                               const double* foot_abs, const uint8_t* contact, double* grf_body_out, double* u_full_out,
                               int32_t* iters_out, int32_t* status_out);

/* Same, but every pointer is DEVICE memory of the handle's device and the launch is asynchronous on
 * `hip_stream` (a hipStream_t, NULL = the handle's own stream).  No host synchronisation. */
a1mpc_status a1mpc_solve_batch_device(a1mpc_handle h, int32_t n, const double* d_x0, const double* d_x_ref,
                                      const double* d_R_world, const double* d_foot_abs, const uint8_t* d_contact,
                                      double* d_grf_body_out, double* d_u_full_out, int32_t* d_iters_out,
                                      int32_t* d_status_out, void* hip_stream);

/*
 * The general case of the reference's ConvexMpc INTERFACE: a different B_d at every horizon step (public member B_mat_d_list,
 * S/ConvexMpc.h:74, filled per step by S/test/test_mpc.cpp:106-122 -- feet shifted by root_lin_vel_d * dt -- and by the commented
 * lines S/A1RobotControl.cpp:504-507) and a per-step contact schedule (a superset: calculate_qp_mats broadcasts the current contacts,
 * S/ConvexMpc.cpp:228-245).
 *   foot_stride    0: foot is n x 12, the same feet at every step       12: foot is n x 12H, step t of problem i at foot[(i*H + t)*12]
 *   contact_stride 0: contact is n x 4, broadcast over the horizon       4: contact is n x 4H, step t of problem i at contact[(i*H + t)*4]
 *   yaw_A          NULL: A_c is built from mpc_states[2] (S/A1RobotControl.cpp:452,492)      else n yaws, one per problem
 *                  (calculate_A_mat_c takes its own euler argument, S/ConvexMpc.h:28; S/test/test_mpc.cpp:94-104 passes an average)
 * Everything else as a1mpc_solve_batch.  (0, 0, NULL) IS a1mpc_solve_batch (same kernels, same bits).  (0, 4, NULL) -- a gait schedule over the
 * horizon with step-invariant feet, the case a controller with a contact plan has -- also runs the fast kernels at their speed: contacts only
 * change the bounds and which rows are equalities (any horizon a1mpc_create accepts).  Per-step feet and / or a yaw_A run the general kernels:
 * same OSQP iterates as the reference's formation with those inputs, with a bigger LDS image per QP (B~_t of every step), i.e.
 * fewer QPs in flight -- 1.7-2.5x slower by design (round 6; with two batches in flight 3.6 M / 1.5 M / 1.07 M first solves/s at 4096 x h10 / 8192 x h16 / 8192 x h20).
 * Every horizon > 1 that a1mpc_create accepts (10, 16, 20 tuned; 4, 6, 8, 12, 14 as the kernels instantiate).
 */
a1mpc_status a1mpc_solve_batch_strided(a1mpc_handle h, int32_t n, const double* x0, const double* x_ref, const double* R_world,
                                       const double* foot_abs, int32_t foot_stride, const uint8_t* contact, int32_t contact_stride,
                                       const double* yaw_A, double* grf_body_out, double* u_full_out, int32_t* iters_out,
                                       int32_t* status_out);
a1mpc_status a1mpc_solve_batch_strided_device(a1mpc_handle h, int32_t n, const double* d_x0, const double* d_x_ref,
                                              const double* d_R_world, const double* d_foot_abs, int32_t foot_stride,
                                              const uint8_t* d_contact, int32_t contact_stride, const double* d_yaw_A,
                                              double* d_grf_body_out, double* d_u_full_out, int32_t* d_iters_out,
                                              int32_t* d_status_out, void* hip_stream);

/*
 * N1 (the caller side of the path, S/A1RobotControl.cpp:452-488): the same solve from the COMPACT tick record; x0 (mpc_states)
 * and x_ref (mpc_states_d) are built on the device exactly as compute_grf builds them.  tick is n x 22 doubles:
 *   [0:3] root_euler  [3:6] root_pos  [6:9] root_ang_vel  [9:12] root_lin_vel          (world frame, as in mpc_states)
 *   [12:15] root_euler_d  [15:18] root_lin_vel_d (BODY frame; rotated by R_world as at :470)  [18:21] root_ang_vel_d  [21] root_pos_d[2]
 * Input per QP drops from (13 + 13H) to 22 doubles.  Host pointers; a1mpc_solve_batch_ticks_device takes device pointers + a stream.
 * A handful of ticks (n <= 8; the drop-in's compute_grf is n = 1) take the pinned-block path of a1mpc_solve_batch (see there).
 */
a1mpc_status a1mpc_solve_batch_ticks(a1mpc_handle h, int32_t n, const double* tick, const double* R_world, const double* foot_abs,
                                     const uint8_t* contact, double* grf_body_out, double* u_full_out, int32_t* iters_out,
                                     int32_t* status_out);
a1mpc_status a1mpc_solve_batch_ticks_device(a1mpc_handle h, int32_t n, const double* d_tick, const double* d_R_world,
                                            const double* d_foot_abs, const uint8_t* d_contact, double* d_grf_body_out,
                                            double* d_u_full_out, int32_t* d_iters_out, int32_t* d_status_out, void* hip_stream);

/*
 * n independent balance-QP ticks (12 variables, 20 constraints), cold-started like the reference.
 *   root_acc n x 6   desired wrench incl. m*9.8 (S/A1RobotControl.cpp:379-391)
 *   R_world  n x 9   root_rot_mat (output rotation), R_z n x 9 root_rot_mat_z (lever arms, :397), row-major
 *   foot     n x 12, contact n x 4
 * The handle's horizon is irrelevant for this call; its OSQP settings are used with warm_start forced off.
 */
a1mpc_status a1mpc_balance_solve_batch(a1mpc_handle h, const a1mpc_balance_config* qp, int32_t n, const double* root_acc,
                                       const double* R_world, const double* R_z, const double* foot_abs,
                                       const uint8_t* contact, double* grf_body_out, double* f_world_out,
                                       int32_t* iters_out, int32_t* status_out);

/*
 * N2a (the caller side of the path): A1RobotControl::update_plan, S/A1RobotControl.cpp:148-202, for n robots -- gait counters,
 * planned contacts and the Raibert foothold.  Element-wise, HBM-bound; results are bit-identical to the reference arithmetic.
 *   movement_mode      n          0 = stand (all feet planned in contact, counters reset), 1 = walk
 *   gait_counter       n x 4      in/out
 *   gait_counter_speed n x 4
 *   root_lin_vel n x 3 (world), R_z n x 9 (root_rot_mat_z), R_world n x 9 (root_rot_mat), root_pos n x 3, root_lin_vel_d n x 3 (body)
 * out: plan_contacts n x 4, foot_pos_target_rel / _abs / _world n x 12 each (3x4 column-major); any of the three may be NULL.
 * Host pointers.
 */
typedef struct a1mpc_gait_config {
    double counter_per_gait, counter_per_swing; /* S/A1CtrlStates.h:24-25 */
    double control_dt;                          /* S/A1CtrlStates.h:332 */
    double foot_delta_x_limit, foot_delta_y_limit; /* S/A1Params.h:44-45 */
    double default_foot_pos[12];                /* 3x4 column-major, S/A1CtrlStates.h:45 */
    double gait_counter_reset[4];               /* S/A1CtrlStates.h:322-326 */
} a1mpc_gait_config;
void a1mpc_default_gait_config(a1mpc_gait_config* cfg);
a1mpc_status a1mpc_update_plan_batch(a1mpc_handle h, const a1mpc_gait_config* gait, int32_t n, const uint8_t* movement_mode,
                                     double* gait_counter, const double* gait_counter_speed, const double* root_lin_vel,
                                     const double* R_z, const double* R_world, const double* root_pos, const double* root_lin_vel_d,
                                     uint8_t* plan_contacts_out, double* foot_pos_target_rel_out, double* foot_pos_target_abs_out,
                                     double* foot_pos_target_world_out);

/*
 * N3 (the step right after the path): A1RobotControl::compute_joint_torques, S/A1RobotControl.cpp:289-319, for n robots.
 *   active          n        0 while the reference's mpc_init_counter < 10 (:294): all torques zero
 *   contacts        n x 4
 *   j_foot_blocks   n x 4 x 9  the four diagonal 3x3 blocks of j_foot (S/A1CtrlStates.h), column-major each
 *   grf, f_kin      n x 12     foot_forces_grf / foot_forces_kin, 3x4 column-major
 *   km_foot         3, torques_gravity n x 12
 *   joint_torques   n x 12     in/out: NaN results keep the previous value (:314-317)
 * stance: tau = J'(-f_grf); swing: tau = J^-1 (km .* f_kin) by partial-pivot LU in Eigen's operation order.  Host pointers.
 */
a1mpc_status a1mpc_joint_torques_batch(a1mpc_handle h, int32_t n, const uint8_t* active, const uint8_t* contacts, const double* j_foot_blocks,
                                       const double* grf, const double* f_kin, const double* km_foot, const double* torques_gravity,
                                       double* joint_torques);

/*
 * N2b (caller side of the path): actual contacts, recent-contact positions and terrain pitch for n robots, one tick --
 * the contact block of generate_swing_legs_ctrl (S/A1RobotControl.cpp:256-282), compute_walking_surface (:566-582) and the
 * terrain adaptation of compute_grf (:335-376), including their MovingWindowFilters (S/utils/filter.hpp; windows 60 / 100).
 * The filter state of every robot (13 filters, early_contacts, foot_pos_recent_contact) lives on the device inside the handle,
 * indexed by the robot's position in the batch; a1mpc_reset_contact_state zeroes it (= constructing A1RobotControl).
 *   gait_counter n x 4, plan_contacts n x 4, foot_force n x 4, foot_pos_abs n x 12 (3x4 column-major), root_pos_z n
 *   root_euler_d_pitch n   in/out: overwritten with +-terrain_angle when use_terrain_adapt (:358-364)
 * out: contacts n x 4, foot_pos_recent_contact n x 12, terrain_angle n (= state.terrain_pitch_angle).  Host pointers.
 * Filter arithmetic is bit-identical to the reference's; the plane fit uses a Jacobi eigen-decomposition in place of Eigen's
 * JacobiSVD (same pseudo-inverse up to rounding), acos comes from the device math library.
 */
typedef struct a1mpc_contact_config {
    double counter_per_swing;   /* S/A1CtrlStates.h:25 */
    double foot_force_low;      /* FOOT_FORCE_LOW, S/A1Params.h:38 */
    int32_t use_terrain_adapt;  /* S/A1CtrlStates.h:22 */
} a1mpc_contact_config;
void a1mpc_default_contact_config(a1mpc_contact_config* cfg);
a1mpc_status a1mpc_contact_terrain_batch(a1mpc_handle h, const a1mpc_contact_config* cfg, int32_t n, const double* gait_counter,
                                         const uint8_t* plan_contacts, const double* foot_force, const double* foot_pos_abs,
                                         const double* root_pos_z, double* root_euler_d_pitch, uint8_t* contacts_out,
                                         double* foot_pos_recent_contact_out, double* terrain_angle_out);
a1mpc_status a1mpc_reset_contact_state(a1mpc_handle h);
/* The terrain block of compute_grf on its own (S/A1RobotControl.cpp:335-376 with compute_walking_surface :566-582) for callers that keep
 * the reference's generate_swing_legs_ctrl: foot_pos_recent_contact (n x 12, 3x4 column-major) comes from the caller's A1CtrlStates,
 * the terrain-angle filter (window 100) of robot i lives in the handle (the same state a1mpc_contact_terrain_batch uses and
 * a1mpc_reset_contact_state clears).  root_euler_d_pitch n in/out, terrain_angle_out n (= state.terrain_pitch_angle).  Host pointers. */
a1mpc_status a1mpc_terrain_batch(a1mpc_handle h, int32_t use_terrain_adapt, int32_t n, const double* foot_pos_recent_contact,
                                 const double* root_pos_z, double* root_euler_d_pitch, double* terrain_angle_out);

/*
 * N4a (caller side): swing-leg targets and the foot PD force -- the first block of generate_swing_legs_ctrl,
 * S/A1RobotControl.cpp:204-254, with the Bezier curve of S/utils/Utils.cpp:64-104 -- for n robots.
 *   R_z n x 9 (root_rot_mat_z), foot_pos_abs / foot_pos_target_rel n x 12, gait_counter n x 4, kp_foot / kd_foot 3 (per axis)
 *   foot_pos_start, foot_pos_rel_last_time, foot_pos_target_last_time n x 12  in/out (state carried by the caller)
 * out: foot_pos_cur n x 12, foot_forces_kin n x 12.  Host pointers.  Integer powers of the curve are products here (std::pow there):
 * agreement to a few ulp, everything else bit-identical arithmetic.
 */
a1mpc_status a1mpc_swing_legs_batch(a1mpc_handle h, int32_t n, double counter_per_swing, double dt, const double* R_z,
                                    const double* foot_pos_abs, const double* gait_counter, const double* foot_pos_target_rel,
                                    const double* kp_foot, const double* kd_foot, double* foot_pos_start, double* foot_pos_rel_last_time,
                                    double* foot_pos_target_last_time, double* foot_pos_cur_out, double* foot_forces_kin_out);

/*
 * N4b (caller side): leg kinematics of n robots -- the per-leg block of the joint-state callback, S/GazeboA1ROS.cpp:264-279, with
 * A1Kinematics::fk / jac (S/legKinematics/A1Kinematics.cpp) restated from the leg model (hip offset, abduction, thigh, calf).
 *   joint_pos, joint_vel n x 12; R_world n x 9; root_pos, root_lin_vel n x 3; rho_fix 4 x 5 = [ox, oy, d, lt, lc] per leg, rho_opt 4 x 3
 * out (each may be NULL except foot_pos_rel and j_foot_blocks): foot_pos_rel n x 12, j_foot_blocks n x 4 x 9 (column-major 3x3 per leg,
 * the layout a1mpc_joint_torques_batch takes), foot_vel_rel, foot_pos_abs, foot_vel_abs, foot_pos_world, foot_vel_world n x 12.
 * Host pointers.  sin / cos come from the device math library: agreement with the reference to a few ulp.
 */
a1mpc_status a1mpc_leg_state_batch(a1mpc_handle h, int32_t n, const double* joint_pos, const double* joint_vel, const double* R_world,
                                   const double* root_pos, const double* root_lin_vel, const double* rho_fix, const double* rho_opt,
                                   double* foot_pos_rel_out, double* j_foot_blocks_out, double* foot_vel_rel_out, double* foot_pos_abs_out,
                                   double* foot_vel_abs_out, double* foot_pos_world_out, double* foot_vel_world_out);

/*
 * N4c (caller side, the step before the path): A1BasicEKF, S/A1BasicEKF.cpp:7-163 -- the 18-state / 28-measurement Kalman filter that
 * provides root_pos and root_lin_vel -- for n robots, one tick.  The filter state of every robot (x 18, P 18x18, initialised flag) lives
 * on the device inside the handle; the first call for a robot performs init_state (:54-68), later calls update_estimation (:70-163), like
 * the reference's callers (S/GazeboA1ROS.cpp:194-198).  a1mpc_reset_ekf_state = constructing the filter anew.
 *   movement_mode n, foot_force n x 4, R_world n x 9, imu_acc n x 3, imu_ang_vel n x 3, foot_pos_rel / foot_vel_rel n x 12
 * out: root_pos n x 3, root_lin_vel n x 3, estimated_contacts n x 4.  Host pointers.
 * The two fullPivHouseholderQr solves with S are one Gauss-Jordan elimination of [S | error_y | C] here (S is symmetric positive definite).
 */
a1mpc_status a1mpc_ekf_update_batch(a1mpc_handle h, int32_t n, double dt, int32_t assume_flat_ground, const uint8_t* movement_mode,
                                    const double* foot_force, const double* R_world, const double* imu_acc, const double* imu_ang_vel,
                                    const double* foot_pos_rel, const double* foot_vel_rel, double* root_pos_out, double* root_lin_vel_out,
                                    uint8_t* estimated_contacts_out);
a1mpc_status a1mpc_reset_ekf_state(a1mpc_handle h);

/*
 * Device-pointer variants of the caller-side entry points above: every array argument is a device pointer (the small per-call constant
 * arrays kp / kd / km / rho_fix / rho_opt stay host pointers, they travel as kernel arguments), the call is asynchronous on `hip_stream`
 * (NULL = the handle's stream).  Together with a1mpc_solve_batch_ticks_device a whole control tick chains on the GPU without a PCIe hop.
 */
a1mpc_status a1mpc_update_plan_batch_device(a1mpc_handle h, const a1mpc_gait_config* gait, int32_t n, const uint8_t* d_movement_mode,
                                            double* d_gait_counter, const double* d_gait_counter_speed, const double* d_root_lin_vel,
                                            const double* d_R_z, const double* d_R_world, const double* d_root_pos, const double* d_root_lin_vel_d,
                                            uint8_t* d_plan_contacts_out, double* d_rel_out, double* d_abs_out, double* d_world_out, void* hip_stream);
a1mpc_status a1mpc_swing_legs_batch_device(a1mpc_handle h, int32_t n, double counter_per_swing, double dt, const double* d_R_z,
                                           const double* d_foot_pos_abs, const double* d_gait_counter, const double* d_foot_pos_target_rel,
                                           const double* kp_foot, const double* kd_foot, double* d_foot_pos_start, double* d_foot_pos_rel_last_time,
                                           double* d_foot_pos_target_last_time, double* d_foot_pos_cur_out, double* d_foot_forces_kin_out, void* hip_stream);
a1mpc_status a1mpc_contact_terrain_batch_device(a1mpc_handle h, const a1mpc_contact_config* cfg, int32_t n, const double* d_gait_counter,
                                                const uint8_t* d_plan_contacts, const double* d_foot_force, const double* d_foot_pos_abs,
                                                const double* d_root_pos_z, double* d_root_euler_d_pitch, uint8_t* d_contacts_out,
                                                double* d_foot_pos_recent_contact_out, double* d_terrain_angle_out, void* hip_stream);
a1mpc_status a1mpc_leg_state_batch_device(a1mpc_handle h, int32_t n, const double* d_joint_pos, const double* d_joint_vel, const double* d_R_world,
                                          const double* d_root_pos, const double* d_root_lin_vel, const double* rho_fix, const double* rho_opt,
                                          double* d_foot_pos_rel_out, double* d_j_foot_blocks_out, double* d_foot_vel_rel_out, double* d_foot_pos_abs_out,
                                          double* d_foot_vel_abs_out, double* d_foot_pos_world_out, double* d_foot_vel_world_out, void* hip_stream);
a1mpc_status a1mpc_ekf_update_batch_device(a1mpc_handle h, int32_t n, double dt, int32_t assume_flat_ground, const uint8_t* d_movement_mode,
                                           const double* d_foot_force, const double* d_R_world, const double* d_imu_acc, const double* d_imu_ang_vel,
                                           const double* d_foot_pos_rel, const double* d_foot_vel_rel, double* d_root_pos_out, double* d_root_lin_vel_out,
                                           uint8_t* d_estimated_contacts_out, void* hip_stream);
a1mpc_status a1mpc_joint_torques_batch_device(a1mpc_handle h, int32_t n, const uint8_t* d_active, const uint8_t* d_contacts, const double* d_j_foot_blocks,
                                              const double* d_grf, const double* d_f_kin, const double* km_foot, const double* d_torques_gravity,
                                              double* d_joint_torques, void* hip_stream);

/*
 * One control tick of n robots in ONE call, device-resident (round 5; SURVEY 8(f) N1-N4 chained): what the reference runs per robot every 2.5 ms --
 *   joint-state callback: leg FK / Jacobians (S/GazeboA1ROS.cpp:264-279)  ->  A1BasicEKF::update_estimation (S/A1BasicEKF.cpp:70-163)  ->  update_plan
 *   (S/A1RobotControl.cpp:148-202)  ->  generate_swing_legs_ctrl (:204-287; its contact logic together with the terrain fit of compute_grf, :335-376, :566-582)  ->
 *   compute_grf (:446-562: state packing, reference trajectory, QP formation, OSQP)  ->  compute_joint_torques (:289-319), driven by S/MainGazebo.cpp:47-119
 * -- as six element-wise / filter kernels, a tick-record pack and the MPC launch back to back on `hip_stream` (NULL = the handle's), no host round trip.
 * compute_joint_torques (N3) runs INSIDE the MPC kernel's output stage whenever the tick goes through the fused / latency kernel (every warm-started tick of a
 * known batch, every batch of <= 256 robots): the lanes that have just written a leg's GRF evaluate tau = J'(-f) (stance) or J^-1 (km .* f_kin) (swing) for that leg;
 * otherwise (a first tick, a batch beyond the fused kernel's range) it is one more launch.  Results are bit-identical to chaining the seven *_device entry points
 * (a1mpc_leg_state_batch_device, a1mpc_ekf_update_batch_device, a1mpc_update_plan_batch_device, a1mpc_swing_legs_batch_device, a1mpc_contact_terrain_batch_device,
 * a1mpc_solve_batch_ticks_device, a1mpc_joint_torques_batch_device) with root_pos[.][2] as root_pos_z and root_euler_d[.][1] as the terrain pitch.
 * The filter states (EKF, contact / terrain windows) and the carried OSQP workspace live in the handle, indexed by the robot's position in the batch.
 */
typedef struct a1mpc_tick_params {
    a1mpc_gait_config gait;           /* update_plan + counter_per_swing of the swing legs */
    a1mpc_contact_config contact;     /* contact logic, terrain adaptation */
    double control_dt;                /* EKF dt and the swing legs' velocity dt (2.5 ms, S/A1CtrlStates.h:332) */
    int32_t assume_flat_ground;       /* A1BasicEKF ctor argument, S/A1BasicEKF.cpp:42-53 */
    double kp_foot[3], kd_foot[3];    /* swing-leg PD, per axis (S/A1CtrlStates.h: kp_foot / kd_foot) */
    double km_foot[3];                /* S/A1CtrlStates.h: km_foot */
    double rho_fix[20], rho_opt[12];  /* leg geometry: 4 x [ox, oy, d, lt, lc] and 4 x 3 (S/GazeboA1ROS.cpp:20-50) */
} a1mpc_tick_params;
typedef struct a1mpc_tick_buffers {   /* DEVICE pointers, n robots each; layouts as in the per-stage entry points above */
    /* inputs of this tick: sensors, attitude (the reference takes it from the IMU / simulator), commands */
    const double *joint_pos, *joint_vel;                 /* n x 12 */
    const double *R_world, *R_z;                         /* n x 9 row-major: root_rot_mat, root_rot_mat_z */
    const double *root_euler, *root_ang_vel;             /* n x 3 (world frame), the tick record's [0:3] and [6:9] */
    const double *imu_acc, *imu_ang_vel;                 /* n x 3 (the EKF's inputs) */
    const double* foot_force;                            /* n x 4 */
    const uint8_t* movement_mode;                        /* n */
    const uint8_t* mpc_active;                           /* n: 0 while the reference's mpc_init_counter < 10 (S/A1RobotControl.cpp:294) */
    const double *root_lin_vel_d, *root_ang_vel_d;       /* n x 3: commanded velocities (lin: body frame, rotated at S/A1RobotControl.cpp:470) */
    const double* root_pos_d_z;                          /* n: commanded body height, root_pos_d[2] */
    const double* gait_counter_speed;                    /* n x 4 */
    const double* torques_gravity;                       /* n x 12 */
    /* state carried from tick to tick by the caller (in/out) */
    double* gait_counter;                                /* n x 4 */
    double *foot_pos_start, *foot_pos_rel_last_time, *foot_pos_target_last_time;   /* n x 12 */
    double* root_euler_d;                                /* n x 3: [1] is overwritten with +-terrain_angle when use_terrain_adapt (S/A1RobotControl.cpp:358-364) */
    double* joint_torques;                               /* n x 12: NaN results keep the previous value (:314-317) */
    double *root_pos, *root_lin_vel;                     /* n x 3: the estimate (in: the previous tick's, read by the leg stage's world-frame outputs; out: this tick's) */
    /* outputs */
    uint8_t *estimated_contacts, *plan_contacts, *contacts;   /* n x 4 */
    double *foot_pos_rel, *j_foot_blocks, *foot_vel_rel, *foot_pos_abs;   /* n x 12 / n x 36 / n x 12 / n x 12 */
    double *foot_vel_abs, *foot_pos_world, *foot_vel_world;               /* n x 12, each may be NULL */
    double* foot_pos_target_rel;                                          /* n x 12 */
    double *foot_pos_target_abs, *foot_pos_target_world;                  /* n x 12, each may be NULL */
    double *foot_pos_cur, *foot_forces_kin;              /* n x 12 */
    double* foot_pos_recent_contact;                     /* n x 12 */
    double* terrain_angle;                               /* n */
    double* grf;                                         /* n x 12: foot_forces_grf, body frame */
    int32_t *iters, *status;                             /* n each, or NULL */
} a1mpc_tick_buffers;
void a1mpc_default_tick_params(a1mpc_tick_params* p);   /* the reference's Gazebo parameter set and the A1's leg geometry */
a1mpc_status a1mpc_control_tick_device(a1mpc_handle h, const a1mpc_tick_params* params, const a1mpc_tick_buffers* buffers, int32_t n, void* hip_stream);
/* duration of the handle's last a1mpc_control_tick_device on the device (HIP events around the whole tick; synchronises it) and whether its joint torques were
 * written by the MPC kernel's output stage (1) or by a launch of their own (0) */
a1mpc_status a1mpc_last_control_tick_ms(a1mpc_handle h, float* ms_out, int32_t* torques_fused_out);

/*
 * Debug / verification: the dense QP data the reference's ConvexMpc keeps in its public members after calculate_qp_mats
 * (hessian, gradient, lb, ub: S/ConvexMpc.h:84-93, S/ConvexMpc.cpp:158-245) for n problems, formed on the GPU from the same inputs as
 * a1mpc_solve_batch_strided.  P_out n x (12H)^2 (row-major, symmetric), g_out n x 12H, l_out / u_out n x 20H (the constraint matrix is
 * the constant stencil of S/ConvexMpc.cpp:46-58).  The solver itself never forms these; callers that poke the members
 * (S/A1RobotControl.cpp:527-537, S/test/test_mpc.cpp:136-140) get them through include/a1mpc_dropin.hpp.  Host pointers; allocates
 * and frees its own device buffers; not a per-tick call.
 */
a1mpc_status a1mpc_form_qp_batch(a1mpc_handle h, int32_t n, const double* x0, const double* x_ref, const double* R_world,
                                 const double* foot_abs, int32_t foot_stride, const uint8_t* contact, int32_t contact_stride,
                                 const double* yaw_A, double* P_out, double* g_out, double* l_out, double* u_out);

/* Work-queue order of batches larger than the resident set: history = 1 (default) issues the QPs longest-first by the cost
 * (iterations + factor passes) each one had in the previous solve of this handle with the same n -- the same robots tick after
 * tick; history = 0 is plain index order.  The first solve of a batch size, the solve after a1mpc_reset_warm_start and the solve after
 * this call have no history: their queue is ordered by a cost guess the set-up kernel derives from each QP's velocity error.
 * Scheduling only: no returned number depends on it.  Environment override at create: A1MPC_SCHEDULE=index. */
a1mpc_status a1mpc_set_schedule(a1mpc_handle h, int32_t history);

/* forget the carried (x, y, rho) of every problem: next solve is a cold start */
a1mpc_status a1mpc_reset_warm_start(a1mpc_handle h);

/* Read / write the carried OSQP workspace of problems 0..n-1 (the reference keeps it inside its persistent
 * OsqpEigen::Solver member, S/A1RobotControl.h:67): x n x 12H (world-frame forces), y n x 20H (reference row order), rho n
 * (0 = "start from settings.rho").  Any pointer may be NULL.  Host pointers; synchronises the handle's stream.  A solve whose
 * solution is not finite leaves x = y = 0 behind, i.e. the next tick of that problem starts from cold iterates (OSQP's store_solution()
 * cold-starts its iterates after a failed solve), so one bad tick cannot poison the ticks after it -- and, since round 4, the rho the
 * solver had reached, exactly as OSQP's cold_start() leaves the adapted rho in settings->rho (the reference ignores the solver's return
 * code, S/A1RobotControl.cpp:540, so its next tick does run with that rho; engine and oracle used to restart from the configured rho).
 * With warm_start = 2 an injected state is re-expressed on the update path's workspace the way osqp_warm_start_x / _y do it on the
 * reference's persistent solver (round 4): x and y replace the carried iterates, z becomes A x, the previous tick's scalings and gradient
 * stay, and the next tick follows the update path from there (until round 4 an injected state cleared the carry and the next tick was a
 * fresh set-up). */
a1mpc_status a1mpc_warm_start(a1mpc_handle h, int32_t n, const double* x, const double* y, const double* rho);
a1mpc_status a1mpc_get_warm_start(a1mpc_handle h, int32_t n, double* x_out, double* y_out, double* rho_out);
/* warm_start = 2 only: the unscaled z = Pi(w) of problems 0..n-1 that the update path keeps beside (x, y, rho) -- what OSQP leaves in work->z, n x 20H in the
 * reference's row order (S/ConvexMpc.cpp:46-58).  With x and y it is everything OSQP's own termination test reads (auxil.c check_termination: ||Ax - z||,
 * ||Px + q + A'y||), so a caller -- the parity tests do, on the ticks where engine and oracle stop at different iterations -- can verify that a returned point is
 * one OSQP would have stopped at.  A1MPC_ERR_INVALID_ARGUMENT when the handle has no update-path carry (another warm-start mode, horizon 1, no tick yet).
 * Host pointer; synchronises the handle's stream.  Problems without a previous update-path tick read as zeros. */
a1mpc_status a1mpc_get_workspace_z(a1mpc_handle h, int32_t n, double* z_out);
/* warm_start = 2 only: the equilibration of the handle's last update-path tick as OSQP's workspace holds it -- D (n x 12H, variable scaling), E (n x 20H,
 * constraint scaling, reference row order) and the cost scaling c (n; 0 = no previous tick).  Together with a1mpc_get_warm_start (unscaled x, y, rho) and
 * a1mpc_get_workspace_z this is the complete state the reference's persistent OsqpEigen solver carries from tick to tick (scaled iterates x_s = x / D,
 * z_s = E z, y_s = c y / E): a second implementation of the update path can be started from it (the parity tests re-seed the oracle that way).
 * Any pointer may be NULL.  Host pointers; synchronises the handle's stream. */
a1mpc_status a1mpc_get_workspace_scaling(a1mpc_handle h, int32_t n, double* D_out, double* E_out, double* c_out);
/* Stage counters of the solve (the reference brackets its tick with stopwatches t1..t6, S/A1RobotControl.cpp:491-553; a1mpc_last_stage_ms gives
 * set-up | solve by HIP events).  a1mpc_set_profiling(h, 1): batches that go through the split pipeline (more QPs than the resident rows; horizon > 1,
 * warm_start != 2, contacts broadcast) run a clock-stamped instantiation of the persistent ADMM kernel -- the same arithmetic, the same results bit for bit,
 * shader-clock stamps around every factor pass, iteration segment and residual check (outside the hot loop).  a1mpc_last_stage_cycles then returns the cycles
 * summed over the QPs of the last profiled solve: [0] Riccati factor passes, [1] ADMM iterations, [2] residual checks + rho updates; *qps_out = the number of
 * QPs summed (0: the last solve was not profiled).  A QP's cycles include the time its wavefront spent on its wave-mate's divergent stages; the three sums'
 * ratio splits a1mpc_last_stage_ms' solve stage.  Host call; synchronises the handle's stream. */
a1mpc_status a1mpc_set_profiling(a1mpc_handle h, int32_t on);
a1mpc_status a1mpc_last_stage_cycles(a1mpc_handle h, double* cycles3_out, int32_t* qps_out);
/* The same stopwatches for the tick the reference actually runs (round 5): with profiling on, a solve that goes through the fused or the latency kernel at horizon 10
 * -- every warm-started closed-loop tick of a known batch up to 8192 robots, every batch of <= 256 QPs, i.e. the batch-1 control tick, in all three warm-start
 * semantics -- runs a clock-stamped instantiation of that kernel (same arithmetic, same results bit for bit) and this call returns the shader-clock cycles summed over
 * its QPs, in the order of the reference's tick (S/A1RobotControl.cpp:446-562):
 *   [0] formation: inputs, B~, gradient roll-out + adjoint, the two 12x12 Hessian blocks        (calculate_A/B_mat_c .. calculate_qp_mats, :491-520)
 *   [1] the Ruiz equilibration passes + cost scaling                                            (osqp_setup / osqp_update_P: scale_data)
 *   [2] hot state (rho vector, warm-start iterates, update-path carry) + set-up -> solver hand-off
 *   [3] Riccati factor passes   [4] ADMM iterations (OSQP's first iteration included)   [5] residual checks + rho updates        (solver.solve(), :540)
 *   [6] outputs: R' f, the carried workspace (x, y, rho; update path: z, scalings)                (:555-561, store_solution)
 *   [7] the whole tick, entry to exit (= the sum of [0..6])
 * A wavefront's QPs share its instruction stream: a QP's cycles include its wave-mate's.  *qps_out = QPs summed (0: the last solve was not such a tick -- a
 * split-pipeline solve reports through a1mpc_last_stage_cycles).  Host call; synchronises the handle's stream. */
a1mpc_status a1mpc_last_tick_stage_cycles(a1mpc_handle h, double* cycles8_out, int32_t* qps_out);
/* Which warm-start semantics the handle's last MPC solve actually ran: 0 (cold), 1 (fresh set-up + osqp_warm_start) or 2 (the reference's update path).
 * warm_start = 2 exists at every horizon > 1 on the fast path and on the general path (per-step feet, a separate A_c yaw: its latency and fused kernels -- round 6:
 * at every batch size, an update-path batch beyond the resident rows runs the fused kernel in several rounds); a solve at horizon 1, or a general-path batch forced onto
 * its split pipeline by A1MPC_PIPELINE=split, runs mode 1 instead (documented at a1mpc_config.warm_start) -- this call makes that visible to the caller.  -1 before the
 * first solve. */
a1mpc_status a1mpc_last_warm_start_mode(a1mpc_handle h, int32_t* mode_out);

/* Replace the configuration of a live handle -- everything except the horizon: dt (the reference uses the measured loop dt when
 * use_sim_time is "true", S/A1RobotControl.cpp:465), weights, mass / inertia, friction and force limits, OSQP settings.  The constants
 * travel to the kernels by value with every launch, so this costs nothing, keeps the carried warm start and takes effect with the
 * next call.  Validated like a1mpc_create; a refused configuration leaves the handle's configuration as it was. */
a1mpc_status a1mpc_update_config(a1mpc_handle h, const a1mpc_config* cfg);

/* The handle's HIP timing events (around every launch: a1mpc_last_kernel_ms, a1mpc_last_stage_ms, a1mpc_last_control_tick_ms read them) on (default) / off.  An event
 * record is a packet of its own on the GPU's queue; a 400 Hz control loop that never reads the instrumentation can turn it off -- three records less per MPC tick, four
 * per control tick (batch-1 p50 -5 us; a chained control tick -15..-18 us at 4096 robots: each record is a marker packet in front of which the command processor drains
 * the queue, 5-6 us -- kernel trace and the round-5 "timing off is slower" artefact explained in profiles/r06_control_tick_timeline.md).  With timing off the three calls
 * above return A1MPC_ERR_INVALID_ARGUMENT ("no kernel has been launched ...").  No result depends on it. */
a1mpc_status a1mpc_set_timing(a1mpc_handle h, int32_t on);

/* instrumentation: duration of the last kernel launched through this handle (HIP events on its stream;
 * synchronises that stream), bytes of dynamic LDS per workgroup and QPs per workgroup of the horizon's kernel */
a1mpc_status a1mpc_last_kernel_ms(a1mpc_handle h, float* ms_out);
a1mpc_status a1mpc_kernel_info(a1mpc_handle h, int32_t* lds_bytes_per_workgroup, int32_t* qps_per_workgroup,
                               int32_t* threads_per_workgroup);

/* Stage split of the last MPC launch, the engine's counterpart of the reference's t1..t6 stopwatches (S/A1RobotControl.cpp:491-553;
 * "form" = calculate_A/B_mat_c + discretisation + calculate_qp_mats + OSQP set-up, "solve" = solver.solve()): form_ms = the set-up kernel
 * (formation + Ruiz equilibration, + the queue-order kernel), solve_ms = the persistent ADMM kernel (factorisations + iterations + output).
 * Batches that run the fused / latency kernel are one launch: form_ms = 0, solve_ms = the whole launch.  Per-QP iteration counts come with
 * every solve (iters_out), factorisation counts from a1mpc_last_nfact.  Synchronises the launch. */
a1mpc_status a1mpc_last_stage_ms(a1mpc_handle h, float* form_ms_out, float* solve_ms_out);

/* number of KKT (Riccati) factorisations each QP of the last MPC / balance launch performed (1 + rho updates);
 * synchronises the handle's stream; instrumentation for the work model of bench.py */
a1mpc_status a1mpc_last_nfact(a1mpc_handle h, int32_t n, int32_t* nfact_out);

/*
 * The batch sharded over the GPUs of one node behind ONE handle and one host thread (SURVEY 8b "device = -1 = all", 8e): shard g solves a
 * contiguous slice (sizes differ by at most one, remainder to the low shards) on devices[g] with its own engine handle and stream; the QPs
 * are independent, so there is no data-path collective -- only scatter-inputs / gather-results, by one of two transports:
 *   transport 0  pinned host memory, one asynchronous copy fan-out per device each way (every GPU over its own PCIe link)
 *   transport 1  RCCL over xGMI: one copy of the whole batch to devices[0], grouped ncclSend / ncclRecv to the other devices and back
 *                (librccl.so is dlopen()ed by this call; distinct devices required)
 * devices NULL or n_devices <= 0 = every visible device.  A device may be listed twice with transport 0 (two shards on one GPU).
 * max_batch is the TOTAL batch.  Warm start, if configured, is carried per shard, i.e. per position in the batch as long as n stays the same.
 * Host pointers; layouts as a1mpc_solve_batch.
 */
typedef struct a1mpc_sharded_s* a1mpc_sharded;
a1mpc_status a1mpc_sharded_create(const a1mpc_config* cfg, int32_t max_batch, const int32_t* devices, int32_t n_devices, int32_t transport,
                                  a1mpc_sharded* out);
a1mpc_status a1mpc_sharded_solve_batch(a1mpc_sharded s, int32_t n, const double* x0, const double* x_ref, const double* R_world,
                                       const double* foot_abs, const uint8_t* contact, double* grf_body_out, int32_t* iters_out,
                                       int32_t* status_out);
/* Round 6: the closed loop on all GPUs.  The same sharded solve for the compact tick records (a1mpc_solve_batch_ticks: S/A1RobotControl.cpp:452-488 on the device), and for a batch that
 * is RESIDENT ON SHARD 0's GPU (device pointers of that device; layouts of a1mpc_solve_batch_device / _ticks_device): the root feeds the other shards over xGMI -- peer copies
 * with transport 0, grouped ncclSend / ncclRecv with transport 1 -- solves its own shard in place and collects every shard's GRFs / iterations / status in the caller's output
 * arrays on the root ("RCCL over xGMI only to scatter inputs / gather GRFs", the north_star).  hip_stream: a stream of the root device the inputs were produced on (the shards
 * start behind it; NULL: the inputs are ready).  The calls return when the outputs are complete.  Every shard's engine handle carries its own warm start: with
 * cfg.warm_start = 1 or 2 and a constant n, tick after tick through any of these entries is the closed loop of the single-GPU handle, shard by shard (a1mpc_sharded_handle
 * gives a shard's handle for a1mpc_reset_warm_start / a1mpc_get_warm_start / the instrumentation calls).  a1mpc_sharded_last_transfer: the bytes the last solve moved from the
 * root to the other shards and back (1968 B + 104 B per QP at h = 16). */
a1mpc_status a1mpc_sharded_solve_batch_ticks(a1mpc_sharded s, int32_t n, const double* tick, const double* R_world, const double* foot_abs,
                                             const uint8_t* contact, double* grf_body_out, int32_t* iters_out, int32_t* status_out);
a1mpc_status a1mpc_sharded_solve_batch_device(a1mpc_sharded s, int32_t n, const double* d_x0, const double* d_x_ref, const double* d_R_world,
                                              const double* d_foot_abs, const uint8_t* d_contact, double* d_grf_body_out, int32_t* d_iters_out,
                                              int32_t* d_status_out, void* hip_stream);
a1mpc_status a1mpc_sharded_solve_batch_ticks_device(a1mpc_sharded s, int32_t n, const double* d_tick, const double* d_R_world, const double* d_foot_abs,
                                                    const uint8_t* d_contact, double* d_grf_body_out, int32_t* d_iters_out, int32_t* d_status_out,
                                                    void* hip_stream);
a1mpc_status a1mpc_sharded_handle(a1mpc_sharded s, int32_t shard, a1mpc_handle* out);
a1mpc_status a1mpc_sharded_last_transfer(a1mpc_sharded s, int64_t* scatter_bytes, int64_t* gather_bytes);
a1mpc_status a1mpc_sharded_info(a1mpc_sharded s, int32_t* n_shards, int32_t* devices_out, int32_t* transport);
void a1mpc_sharded_destroy(a1mpc_sharded s);

/*
 * Batch pipeline: consecutive batches in flight together on ONE device (north_star: "HIP streams"; the reference has no counterpart -- it
 * solves one QP per tick on one thread, S/MainGazebo.cpp:57-68).  A launch of a few thousand QPs ends in a tail: its few 150-225-iteration
 * QPs keep a handful of wavefronts busy while the rest of the chip idles (the last 0.15-0.25 ms of a 0.85 ms launch at 4096 x h10).  The
 * pipeline owns `depth` complete engine handles (0 = the default: 2) on `depth` HIP streams and hands batches to them round-robin, so the
 * next batch's set-up kernel and persistent rows are dispatched onto the SIMDs the tail has left.  Batches in flight share nothing
 * (prepared-state records, queue, warm start are per slot): results are bit-identical to a lone handle's.  Measured on one MI355X, first
 * solves of distinct batches (profiles/r03_bench_default_run.json, r03_bench_depth1_run.json): 4096 x h10 4.2-5.0 M -> 6.35-6.44 M solves/s,
 * 8192 x h16 2.23 -> 2.47 M; depth 3 no longer pays (profiles/r03_pcie_probe_4096_h10.json); 16 384 x h10 neutral; a batch of 65 536 QPs
 * fills the chip on its own and loses 7 % when pipelined -- submit those through a plain handle.
 * Round 6 (kernel traces of the slots, profiles/r06_setup_ahead.md): what is left between two slots and one 65 536-QP launch (0.60-0.61 against 0.57-0.58 ms per 4096 x h10
 * batch) is the chain set-up kernel -> queue-order kernel -> persistent kernel of every batch, each of which finds the chip full; a third slot is worth 3-4 % when the
 * caller submits in a free-running loop and loses 11-19 % when it starts its slots behind one event, so the default stays two.  A slot of a two-slot pipeline runs the
 * general path's one-wave persistent kernel at horizon 10 (six QPs per CU) instead of the CU-wide one (seven): the other slot's set-up then runs beside it (+6 % at
 * 4096 QPs, +12 % at 2560; same bits).
 *   submit   device pointers, layouts of a1mpc_solve_batch_device.  slot = -1: next slot round-robin (returned in *slot_out), or a fixed
 *            slot (a robot population that is warm-started must stay on its slot: the carried OSQP workspace lives there).
 *            fresh_batch != 0: these QPs are new to the slot, order its queue by the set-up kernel's cost guess instead of the slot's
 *            previous batch (a1mpc_set_schedule).  inputs_ready_stream: the slot starts after everything queued on that stream so far
 *            (NULL = the inputs are ready now).  Returns at once; the outputs of a slot are valid after a1mpc_pipeline_wait / _join, and
 *            its output buffers must not be handed to another submit before that.
 *   submit (host pointers)  a1mpc_pipeline_submit: the arrays of a1mpc_solve_batch -- what a caller on the reference's side of the boundary has
 *            (S/A1RobotControl.h:44: A1CtrlStates in, a 3x4 matrix out).  Inputs are snapshotted into the slot's pinned block before the call
 *            returns (the caller may overwrite them at once); H2D copy, launches and D2H copy are queued on the slot's stream.  The OUTPUT arrays
 *            are written by the a1mpc_pipeline_wait (or by the next submit) that retires the slot: they must stay valid until then.  The
 *            caller's thread snapshots batch k + 1 while the GPU solves batch k.
 *   wait     the host waits for the slot's last submit (slot -1 = every slot) and hands a host-pointer batch to its output arrays;
 *            join: a caller's stream waits for it instead (device-pointer submits only).
 *   handle   the slot's engine handle, for warm-start I/O, a1mpc_update_config and the instrumentation calls.
 * One host thread per pipeline.
 */
typedef struct a1mpc_pipeline_s* a1mpc_pipeline;
a1mpc_status a1mpc_pipeline_create(const a1mpc_config* cfg, int32_t max_batch, int32_t device, int32_t depth, a1mpc_pipeline* out);
a1mpc_status a1mpc_pipeline_submit_device(a1mpc_pipeline p, int32_t slot, int32_t fresh_batch, int32_t n, const double* d_x0,
                                          const double* d_x_ref, const double* d_R_world, const double* d_foot_abs, const uint8_t* d_contact,
                                          double* d_grf_body_out, double* d_u_full_out, int32_t* d_iters_out, int32_t* d_status_out,
                                          void* inputs_ready_stream, int32_t* slot_out);
a1mpc_status a1mpc_pipeline_submit(a1mpc_pipeline p, int32_t slot, int32_t fresh_batch, int32_t n, const double* x0, const double* x_ref,
                                   const double* R_world, const double* foot_abs, const uint8_t* contact, double* grf_body_out,
                                   double* u_full_out, int32_t* iters_out, int32_t* status_out, int32_t* slot_out);
/* Round 6: the other two input forms of the MPC entry with batches in flight together -- per-step feet / contact schedules / an A_c yaw of its own (the general path of
 * the reference's INTERFACE: B_mat_d_list, S/ConvexMpc.h:74, driven by S/test/test_mpc.cpp:106-122; arguments of a1mpc_solve_batch_strided(_device)) and the compact tick
 * records (S/A1RobotControl.cpp:452-488; arguments of a1mpc_solve_batch_ticks_device).  A first solve of the general path is bounded by its longest QPs and no feature of
 * the inputs predicts them (profiles/r05_general_path_order.txt): the tail is what a second batch in flight fills.  Slots, events, fresh_batch, inputs_ready_stream and
 * the life time of the output arrays are those of a1mpc_pipeline_submit_device / a1mpc_pipeline_submit; results are bit-identical to the lone handle's entry points.
 * (foot_stride, contact_stride, yaw_A) = (0, 0, NULL) IS a1mpc_pipeline_submit(_device). */
a1mpc_status a1mpc_pipeline_submit_strided_device(a1mpc_pipeline p, int32_t slot, int32_t fresh_batch, int32_t n, const double* d_x0,
                                                  const double* d_x_ref, const double* d_R_world, const double* d_foot_abs, int32_t foot_stride,
                                                  const uint8_t* d_contact, int32_t contact_stride, const double* d_yaw_A, double* d_grf_body_out,
                                                  double* d_u_full_out, int32_t* d_iters_out, int32_t* d_status_out, void* inputs_ready_stream,
                                                  int32_t* slot_out);
a1mpc_status a1mpc_pipeline_submit_strided(a1mpc_pipeline p, int32_t slot, int32_t fresh_batch, int32_t n, const double* x0, const double* x_ref,
                                           const double* R_world, const double* foot_abs, int32_t foot_stride, const uint8_t* contact,
                                           int32_t contact_stride, const double* yaw_A, double* grf_body_out, double* u_full_out,
                                           int32_t* iters_out, int32_t* status_out, int32_t* slot_out);
a1mpc_status a1mpc_pipeline_submit_ticks_device(a1mpc_pipeline p, int32_t slot, int32_t fresh_batch, int32_t n, const double* d_tick,
                                                const double* d_R_world, const double* d_foot_abs, const uint8_t* d_contact, double* d_grf_body_out,
                                                double* d_u_full_out, int32_t* d_iters_out, int32_t* d_status_out, void* inputs_ready_stream,
                                                int32_t* slot_out);
a1mpc_status a1mpc_pipeline_wait(a1mpc_pipeline p, int32_t slot);
a1mpc_status a1mpc_pipeline_join(a1mpc_pipeline p, int32_t slot, void* hip_stream);
/* a1mpc_pipeline_handle: a host-pointer batch still in flight on the slot is waited for and handed to its caller's output arrays first (its results live in
 * the handle's one pinned mirror, which any host-pointer call on the handle would overwrite).  a1mpc_pipeline_destroy does the same for every slot: the output
 * arrays of submitted batches must stay valid until wait / the next submit to the slot / destroy has returned. */
a1mpc_status a1mpc_pipeline_handle(a1mpc_pipeline p, int32_t slot, a1mpc_handle* out);
a1mpc_status a1mpc_pipeline_depth(a1mpc_pipeline p, int32_t* depth_out);
void a1mpc_pipeline_destroy(a1mpc_pipeline p);

const char* a1mpc_status_string(a1mpc_status s);
const char* a1mpc_last_error(void); /* thread-local detail of the last non-OK return (e.g. the HIP error string) */
const char* a1mpc_build_info(void); /* "sources <sha256/16 of csrc + this header> arch gfx950": which sources the library was compiled from */

#ifdef __cplusplus
}
#endif
#endif /* A1MPC_H_ */
