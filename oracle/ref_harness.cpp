// ref_harness.cpp -- TEST INFRASTRUCTURE ONLY: C entry points over the REFERENCE's own sources.
//
// `make -C oracle ref` compiles, from where they lie under /root/reference/src/a1_cpp/src (nothing is copied):
//   ConvexMpc.cpp, A1RobotControl.cpp, A1BasicEKF.cpp, utils/Utils.cpp, legKinematics/A1Kinematics.cpp, test/test_mpc.cpp
//   (its main renamed on the command line) and utils/filter.hpp (header-only, used as is)
// against oracle/ref_shim/ (mini-Eigen, an OsqpEigen::Solver backed by the oracle's OSQP restatement, no-op ROS names)
// and links them with this file into oracle/_ref/liba1ref[_hNN].so.  This file only moves numbers in and out of the
// reference's own classes; tests/test_ref_pin.py compares what comes back with oracle/a1mpc_oracle.c.
// The horizon is the reference's compile-time PLAN_HORIZON: _h16 / _h20 libraries are built from a copy of A1Params.h
// whose one PLAN_HORIZON line is edited at build time (generated into oracle/_ref/, never committed).
#include <cstring>
#include <map>
#include <sstream>
#include <string>

#include "A1BasicEKF.h"
#include "A1RobotControl.h"
#include "ConvexMpc.h"
#include "legKinematics/A1Kinematics.h"
#include "utils/Utils.h"
#include "utils/filter.hpp"

int ref_test_mpc_main(int, char **);   // S/test/test_mpc.cpp's main (-Dmain=ref_test_mpc_main on that one file)

namespace {
struct Ctx {
    A1CtrlStates state;
    A1RobotControl ctrl;
    A1BasicEKF *ekf = nullptr;
    ~Ctx() { delete ekf; }
};
struct Quiet {   // the reference prints from constructors and from every tick
    std::streambuf *old; std::ostringstream sink;
    Quiet() : old(std::cout.rdbuf(sink.rdbuf())) {}
    ~Quiet() { std::cout.rdbuf(old); }
};

struct Field { double *p; int n; bool *b; int *i; };
bool lookup(A1CtrlStates &s, const std::string &k, Field &f) {
    f = Field{nullptr, 0, nullptr, nullptr};
#define MAT(name) if (k == #name) { f.p = s.name.data(); f.n = (int)s.name.size(); return true; }
#define SCL(name) if (k == #name) { f.p = &s.name; f.n = 1; return true; }
#define INT(name) if (k == #name) { f.i = &s.name; f.n = 1; return true; }
#define BLN(name) if (k == #name) { f.b = s.name; f.n = NUM_LEG; return true; }
    MAT(gait_counter) MAT(gait_counter_speed) MAT(root_pos_d) MAT(root_euler_d) MAT(root_lin_vel_d) MAT(root_lin_vel_d_world)
    MAT(root_ang_vel_d) MAT(mpc_states) MAT(mpc_states_d) MAT(a1_trunk_inertia) MAT(default_foot_pos) MAT(q_weights) MAT(r_weights)
    MAT(root_pos) MAT(root_euler) MAT(root_rot_mat) MAT(root_rot_mat_z) MAT(root_lin_vel) MAT(root_ang_vel) MAT(root_acc)
    MAT(foot_force) MAT(foot_forces_kin) MAT(foot_forces_grf) MAT(joint_pos) MAT(joint_vel) MAT(foot_pos_target_world)
    MAT(foot_pos_target_abs) MAT(foot_pos_target_rel) MAT(foot_pos_start) MAT(foot_pos_world) MAT(foot_pos_abs) MAT(foot_pos_rel)
    MAT(foot_pos_abs_mpc) MAT(foot_pos_rel_last_time) MAT(foot_pos_target_last_time) MAT(foot_pos_cur) MAT(foot_pos_recent_contact)
    MAT(foot_vel_world) MAT(foot_vel_abs) MAT(foot_vel_rel) MAT(j_foot) MAT(kp_foot) MAT(kd_foot) MAT(km_foot) MAT(kp_linear)
    MAT(kd_linear) MAT(kp_angular) MAT(kd_angular) MAT(torques_gravity) MAT(joint_torques) MAT(imu_acc) MAT(imu_ang_vel)
    MAT(estimated_root_pos) MAT(estimated_root_vel)
    SCL(robot_mass) SCL(counter_per_gait) SCL(counter_per_swing) SCL(control_dt) SCL(terrain_pitch_angle)
    INT(stance_leg_control_type) INT(movement_mode) INT(use_terrain_adapt) INT(counter) INT(gait_type)
    BLN(contacts) BLN(plan_contacts) BLN(early_contacts) BLN(estimated_contacts)
#undef MAT
#undef SCL
#undef INT
#undef BLN
    return false;
}
}  // namespace

extern "C" {

int ref_plan_horizon(void) { return PLAN_HORIZON; }

void *ref_ctx_new(void) { Quiet q; return new Ctx(); }
void ref_ctx_free(void *c) { delete (Ctx *)c; }

// matrices travel in Eigen's (column-major) storage order; bools / ints as 0/1 and whole numbers
int ref_state_set(void *c, const char *name, const double *v, int n) {
    Field f; if (!lookup(((Ctx *)c)->state, name, f) || f.n != n) return -1;
    for (int k = 0; k < n; ++k) { if (f.p) f.p[k] = v[k]; else if (f.b) f.b[k] = v[k] != 0; else f.i[k] = (int)v[k]; }
    return 0;
}
int ref_state_get(void *c, const char *name, double *v, int n) {
    Field f; if (!lookup(((Ctx *)c)->state, name, f) || f.n != n) return -1;
    for (int k = 0; k < n; ++k) v[k] = f.p ? f.p[k] : f.b ? (f.b[k] ? 1.0 : 0.0) : (double)f.i[k];
    return 0;
}

// ---- the hot path: S/A1RobotControl.cpp:321 (both branches, terrain block included) ----
void ref_compute_grf(void *c, double dt, double *grf_out /* 3x4 column-major */) {
    Quiet q; Ctx *x = (Ctx *)c;
    Eigen::Matrix<double, 3, NUM_LEG> g = x->ctrl.compute_grf(x->state, dt);
    std::memcpy(grf_out, g.data(), sizeof(double) * 12);
}
// what the OsqpEigen stand-in was handed in the most recent solve() anywhere in this library
int ref_last_qp_dims(int *n, int *m) { auto &r = OsqpEigen::shim_last(); *n = r.n; *m = r.m; return r.solves; }
void ref_last_qp(double *P, double *q, double *A, double *l, double *u, double *x, double *y) {
    auto &r = OsqpEigen::shim_last();
    if (P) std::memcpy(P, r.P.data(), sizeof(double) * r.P.size());
    if (q) std::memcpy(q, r.q.data(), sizeof(double) * r.q.size());
    if (A) std::memcpy(A, r.A.data(), sizeof(double) * r.A.size());
    if (l) std::memcpy(l, r.l.data(), sizeof(double) * r.l.size());
    if (u) std::memcpy(u, r.u.data(), sizeof(double) * r.u.size());
    if (x) std::memcpy(x, r.x.data(), sizeof(double) * r.x.size());
    if (y) std::memcpy(y, r.y.data(), sizeof(double) * r.y.size());
}
// did the most recent solve() re-initialise the solver (updateHessianMatrix found another sparsity pattern)?  returns the total count of such updates so far
int ref_last_reinit(int *this_solve) { auto &r = OsqpEigen::shim_last(); *this_solve = r.info.reinit; return r.reinits; }
void ref_last_info(int *iters, int *status, int *nfact, double *rho_final) {
    auto &r = OsqpEigen::shim_last(); *iters = r.info.iters; *status = r.info.status; *nfact = r.info.nfact; *rho_final = r.info.rho_final;
}
// OSQP settings every Solver constructed AFTER this call starts from (the reference only sets verbosity and warm start on top)
void ref_set_base_settings(const orc_settings *s) { OsqpEigen::shim_base_settings() = *s; }

// ---- ConvexMpc driven exactly the way compute_grf / test_mpc drive it (S/A1RobotControl.cpp:447-518, S/test/test_mpc.cpp:61-125) ----
// foot: 3x4 column-major, `foot_stride` doubles between horizon steps (0 = same feet every step); Rw, inertia column-major 3x3.
// Outputs dense: P (n*n, symmetric), g (n), A (m*n row-major), l, u (m); n = 12*PLAN_HORIZON, m = 20*PLAN_HORIZON.
void ref_convex_mpc_form(const double *qw, const double *rw, const double *euler, double mass, const double *inertia, const double *Rw,
                         const double *foot, int foot_stride, const unsigned char *contacts, const double *x0, const double *xref, double dt,
                         double *P, double *g, double *A, double *l, double *u, double *A_qp_out, double *B_qp_out) {
    Quiet quiet;
    A1CtrlStates state;
    Eigen::VectorXd q_weights(13), r_weights(12);
    for (int i = 0; i < 13; ++i) q_weights(i) = qw[i];
    for (int i = 0; i < 12; ++i) r_weights(i) = rw[i];
    ConvexMpc mpc_solver = ConvexMpc(q_weights, r_weights);
    mpc_solver.reset();
    for (int i = 0; i < 13; ++i) state.mpc_states(i) = x0[i];
    for (int i = 0; i < 13 * PLAN_HORIZON; ++i) state.mpc_states_d(i) = xref[i];
    for (int i = 0; i < NUM_LEG; ++i) state.contacts[i] = contacts[i] != 0;
    Eigen::Vector3d eul(euler[0], euler[1], euler[2]);
    Eigen::Matrix3d I, R;
    std::memcpy(I.data(), inertia, sizeof(double) * 9); std::memcpy(R.data(), Rw, sizeof(double) * 9);
    mpc_solver.calculate_A_mat_c(eul);
    for (int i = 0; i < PLAN_HORIZON; i++) {
        Eigen::Matrix<double, 3, NUM_LEG> fp;
        std::memcpy(fp.data(), foot + (size_t)i * foot_stride, sizeof(double) * 12);
        mpc_solver.calculate_B_mat_c(mass, I, R, fp);
        mpc_solver.state_space_discretization(dt);
        mpc_solver.B_mat_d_list.block<13, 12>(i * 13, 0) = mpc_solver.B_mat_d;
    }
    mpc_solver.calculate_qp_mats(state);
    const int n = NUM_DOF * PLAN_HORIZON, m = MPC_CONSTRAINT_DIM * PLAN_HORIZON;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) P[(size_t)i * n + j] = mpc_solver.hessian.coeff(i, j);
    for (int i = 0; i < n; ++i) g[i] = mpc_solver.gradient(i);
    for (int i = 0; i < m; ++i) { for (int j = 0; j < n; ++j) A[(size_t)i * n + j] = mpc_solver.linear_constraints.coeff(i, j); l[i] = mpc_solver.lb(i); u[i] = mpc_solver.ub(i); }
    if (A_qp_out) for (int i = 0; i < 13 * PLAN_HORIZON; ++i) for (int j = 0; j < 13; ++j) A_qp_out[(size_t)i * 13 + j] = mpc_solver.A_qp(i, j);
    if (B_qp_out) for (int i = 0; i < 13 * PLAN_HORIZON; ++i) for (int j = 0; j < n; ++j) B_qp_out[(size_t)i * n + j] = mpc_solver.B_qp(i, j);
}

// ---- S/test/test_mpc.cpp, run as written; returns what it printed and (through ref_last_qp) what it solved ----
int ref_run_test_mpc(char *printed, int cap) {
    Quiet q;
    int rc = ref_test_mpc_main(0, nullptr);
    std::string s = q.sink.str();
    if (printed && cap > 0) { std::strncpy(printed, s.c_str(), (size_t)cap - 1); printed[cap - 1] = 0; }
    return rc;
}

// ---- caller-side rows ----
void ref_update_plan(void *c, double dt) { Quiet q; Ctx *x = (Ctx *)c; x->ctrl.update_plan(x->state, dt); }
void ref_generate_swing_legs_ctrl(void *c, double dt) { Quiet q; Ctx *x = (Ctx *)c; x->ctrl.generate_swing_legs_ctrl(x->state, dt); }
void ref_compute_joint_torques(void *c) { Quiet q; Ctx *x = (Ctx *)c; x->ctrl.compute_joint_torques(x->state); }
void ref_compute_walking_surface(void *c, double *coef) { Quiet q; Ctx *x = (Ctx *)c; Eigen::Vector3d s = x->ctrl.compute_walking_surface(x->state); std::memcpy(coef, s.data(), 24); }

void ref_ekf_new(void *c, int assume_flat_ground) { Quiet q; Ctx *x = (Ctx *)c; delete x->ekf; x->ekf = new A1BasicEKF(assume_flat_ground != 0); }
void ref_ekf_init_state(void *c) { Quiet q; Ctx *x = (Ctx *)c; x->ekf->init_state(x->state); }
void ref_ekf_update(void *c, double dt) { Quiet q; Ctx *x = (Ctx *)c; x->ekf->update_estimation(x->state, dt); }

// S/utils/filter.hpp MovingWindowFilter, as is
void ref_filter_run(int window, int n, const double *in, double *out) {
    MovingWindowFilter f(window);
    for (int k = 0; k < n; ++k) out[k] = f.CalculateAverage(in[k]);
}
// S/legKinematics/A1Kinematics.cpp fk / jac (jac returned column-major like Eigen::Matrix3d::data())
void ref_leg_fk(const double *q, const double *rho_opt, const double *rho_fix, double *p) {
    A1Kinematics kin; Eigen::VectorXd ro(3), rf(5);
    for (int i = 0; i < 3; ++i) ro(i) = rho_opt[i];
    for (int i = 0; i < 5; ++i) rf(i) = rho_fix[i];
    Eigen::Vector3d r = kin.fk(Eigen::Vector3d(q[0], q[1], q[2]), ro, rf); std::memcpy(p, r.data(), 24);
}
void ref_leg_jac(const double *q, const double *rho_opt, const double *rho_fix, double *J) {
    A1Kinematics kin; Eigen::VectorXd ro(3), rf(5);
    for (int i = 0; i < 3; ++i) ro(i) = rho_opt[i];
    for (int i = 0; i < 5; ++i) rf(i) = rho_fix[i];
    Eigen::Matrix3d r = kin.jac(Eigen::Vector3d(q[0], q[1], q[2]), ro, rf); std::memcpy(J, r.data(), 72);
}
// S/utils/Utils.cpp
void ref_bezier_foot_curve(float t, const double *start, const double *fin, double pitch, double *out) {
    BezierUtils b; Eigen::Vector3d r = b.get_foot_pos_curve(t, Eigen::Vector3d(start[0], start[1], start[2]), Eigen::Vector3d(fin[0], fin[1], fin[2]), pitch);
    std::memcpy(out, r.data(), 24);
}
void ref_skew(const double *v, double *S /* column-major */) { Eigen::Matrix3d r = Utils::skew(Eigen::Vector3d(v[0], v[1], v[2])); std::memcpy(S, r.data(), 72); }
void ref_quat_to_euler(double w, double x, double y, double z, double *e) { Eigen::Vector3d r = Utils::quat_to_euler(Eigen::Quaterniond(w, x, y, z)); std::memcpy(e, r.data(), 24); }
void ref_pseudo_inverse(const double *M /* column-major */, double *out) { Eigen::Matrix3d m; std::memcpy(m.data(), M, 72); Eigen::Matrix3d r = Utils::pseudo_inverse(m); std::memcpy(out, r.data(), 72); }
double ref_dihedral_angle(const double *a, const double *b) { return Utils::cal_dihedral_angle(Eigen::Vector3d(a[0], a[1], a[2]), Eigen::Vector3d(b[0], b[1], b[2])); }

}  // extern "C"
