// a1mpc_dropin.hpp -- the reference's OWN C++ interface for the hot path, over the C ABI (a1mpc.h).
//
// Source-compatible with the reference's types: it is written against the Eigen 3 API (<Eigen/Dense>, <Eigen/Sparse>) and the
// reference's A1CtrlStates / A1Params.h, so the ROS control loop switches by changing two type names.  (Eigen is not installed in the
// build container; the tests compile this header against the small Eigen stand-in of oracle/ref_shim/ and, beside it in the same
// binary, the reference's own A1RobotControl compiled verbatim -- tests/cpp/test_dropin.cpp.)
//
//   a1mpc::ConvexMpcGpu<H>       <-  class ConvexMpc               S/ConvexMpc.h:22-94
//       same constructor and five methods; the public members the reference's callers read / write (S/A1RobotControl.cpp:513,527-537;
//       S/test/test_mpc.cpp:121,136-140) exist with the reference's types:  A_mat_c, B_mat_c, A_mat_d, B_mat_d, B_mat_d_list,
//       linear_constraints, hessian, gradient, lb, ub.  The small ones are computed on the host exactly as the reference's methods
//       compute them (a dozen multiplications, and callers copy B_mat_d into B_mat_d_list).  hessian / gradient / lb / ub are
//       MATERIALISED ON THE GPU by calculate_qp_mats (a1mpc_form_qp_batch) while materialize_qp_members is true (the default:
//       source compatibility first) -- a drop-in that only wants forces sets it to false and calls solve().
//       New: solve(root_rot_mat) = what the reference does next with OsqpEigen (S/A1RobotControl.cpp:522-561).
//   a1mpc::ComputeGrfGpu<State>  <-  A1RobotControl::compute_grf   S/A1RobotControl.h:44, S/A1RobotControl.cpp:321-564
//       both branches (stance_leg_control_type 0: balance QP :377-444, 1: MPC :446-562) and the terrain block (:335-376).
//
// Per-step B_d: the reference's callers fill B_mat_d_list one block per calculate_B_mat_c / state_space_discretization pair
// (S/A1RobotControl.cpp:498-514, S/test/test_mpc.cpp:106-122).  This class records the feet of every such call; when they differ
// between steps the solve takes the general (per-step) kernels.  A B_mat_d_list edited in any other way cannot be expressed through
// foot positions: calculate_qp_mats detects it (the list no longer equals what the recorded calls produce) and throws.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include <Eigen/Dense>
#include <Eigen/Sparse>

#include "a1mpc.h"

namespace a1mpc {

inline void check_status(a1mpc_status rc, const char* what) {
    if (rc != A1MPC_OK) throw std::runtime_error(std::string(what) + ": " + a1mpc_status_string(rc) + ": " + a1mpc_last_error());
}
struct HandleBox {   // owns an a1mpc_handle; movable so that `ConvexMpc m = ConvexMpc(q, r);` (S/test/test_mpc.cpp:61) compiles as C++14
    a1mpc_handle h = nullptr;
    HandleBox() = default;
    HandleBox(HandleBox&& o) noexcept : h(o.h) { o.h = nullptr; }
    HandleBox(const HandleBox&) = delete;
    HandleBox& operator=(const HandleBox&) = delete;
    ~HandleBox() { if (h) a1mpc_destroy(h); }
};

template <int H = 10>
class ConvexMpcGpu {
  public:
    static constexpr int NS = A1MPC_STATE_DIM, NU = A1MPC_NUM_DOF, NC = A1MPC_CONSTRAINT_DIM, NLEG = A1MPC_NUM_LEG;

    ConvexMpcGpu(Eigen::VectorXd& q_weights_, Eigen::VectorXd& r_weights_, int device = 0) : device_(device) {
        mu = 0.3; fz_min = 0.0; fz_max = 0.0;                             // S/ConvexMpc.cpp:8-10
        a1mpc_default_config(&cfg_);
        cfg_.horizon = H;
        cfg_.warm_start = 2;   // a persistent solver that is updated and re-solved every tick = OSQP's update path (S/A1RobotControl.cpp:533-538); config().warm_start = 0 for one-off solves
        for (int i = 0; i < NS; ++i) cfg_.q[i] = q_weights_(i);
        for (int i = 0; i < NU; ++i) cfg_.r[i] = r_weights_(i);
        linear_constraints.resize(NC * H, NU * H);                        // S/ConvexMpc.cpp:46-58
        for (int i = 0; i < NLEG * H; ++i) {
            linear_constraints.insert(0 + 5 * i, 0 + 3 * i) = 1; linear_constraints.insert(1 + 5 * i, 0 + 3 * i) = 1;
            linear_constraints.insert(2 + 5 * i, 1 + 3 * i) = 1; linear_constraints.insert(3 + 5 * i, 1 + 3 * i) = 1;
            linear_constraints.insert(4 + 5 * i, 2 + 3 * i) = 1;
            linear_constraints.insert(0 + 5 * i, 2 + 3 * i) = mu; linear_constraints.insert(1 + 5 * i, 2 + 3 * i) = -mu;
            linear_constraints.insert(2 + 5 * i, 2 + 3 * i) = mu; linear_constraints.insert(3 + 5 * i, 2 + 3 * i) = -mu;
        }
        reset();
    }
    ConvexMpcGpu(ConvexMpcGpu&&) = default;
    ConvexMpcGpu(const ConvexMpcGpu&) = delete;
    ConvexMpcGpu& operator=(const ConvexMpcGpu&) = delete;

    void reset() {                                                        // S/ConvexMpc.cpp:70-108
        A_mat_c.setZero(); B_mat_c.setZero(); A_mat_d.setZero(); B_mat_d.setZero(); B_mat_d_list.setZero();
        gradient.setZero(); lb.setZero(); ub.setZero();
        feet_.clear(); recorded_B_.clear(); calls_since_discretization_ = 0;
    }

    void calculate_A_mat_c(Eigen::Vector3d root_euler) {                  // S/ConvexMpc.cpp:110-130
        const double cos_yaw = cos(root_euler[2]), sin_yaw = sin(root_euler[2]);
        Eigen::Matrix3d ang_vel_to_rpy_rate;
        ang_vel_to_rpy_rate << cos_yaw, sin_yaw, 0, -sin_yaw, cos_yaw, 0, 0, 0, 1;
        A_mat_c.template block<3, 3>(0, 6) = ang_vel_to_rpy_rate;
        A_mat_c.template block<3, 3>(3, 9) = Eigen::Matrix3d::Identity();
        A_mat_c(11, NU) = 1;
        yaw_ = root_euler[2];
    }

    void calculate_B_mat_c(double robot_mass, const Eigen::Matrix3d& a1_trunk_inertia, Eigen::Matrix3d root_rot_mat,
                           Eigen::Matrix<double, 3, NLEG> foot_pos) {     // S/ConvexMpc.cpp:132-143
        Eigen::Matrix3d a1_trunk_inertia_world = root_rot_mat * a1_trunk_inertia * root_rot_mat.transpose();
        Eigen::Matrix3d inv = a1_trunk_inertia_world.inverse();
        for (int i = 0; i < NLEG; ++i) {
            Eigen::Matrix3d sk;
            sk << 0, -foot_pos(2, i), foot_pos(1, i), foot_pos(2, i), 0, -foot_pos(0, i), -foot_pos(1, i), foot_pos(0, i), 0;
            B_mat_c.template block<3, 3>(6, 3 * i) = inv * sk;
            B_mat_c.template block<3, 3>(9, 3 * i) = (1 / robot_mass) * Eigen::Matrix3d::Identity();
        }
        cfg_.mass = robot_mass;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { cfg_.inertia_body[i * 3 + j] = a1_trunk_inertia(i, j); R_[i * 3 + j] = root_rot_mat(i, j); }
        pending_feet_.assign(foot_pos.data(), foot_pos.data() + 12);      // 3x4 column-major = the ABI's layout
        ++calls_since_discretization_;
    }

    void state_space_discretization(double dt) {                          // S/ConvexMpc.cpp:145-156
        A_mat_d = Eigen::Matrix<double, NS, NS>::Identity() + A_mat_c * dt;
        B_mat_d = B_mat_c * dt;
        cfg_.dt = dt;
        if (calls_since_discretization_ > 0) {                            // the B_d block the caller is about to store in B_mat_d_list
            feet_.insert(feet_.end(), pending_feet_.begin(), pending_feet_.end());
            recorded_B_.push_back(B_mat_d);
            calls_since_discretization_ = 0;
        }
    }

    // state: mpc_states (13), mpc_states_d (13 H), contacts[4]  (A1CtrlStates)
    template <class State>
    void calculate_qp_mats(State& state) {                                // S/ConvexMpc.cpp:158-260
        x0_.resize(NS); xref_.resize(static_cast<size_t>(NS) * H);
        for (int i = 0; i < NS; ++i) x0_[i] = state.mpc_states(i);
        for (int i = 0; i < NS * H; ++i) xref_[i] = state.mpc_states_d(i);
        for (int i = 0; i < NLEG; ++i) contact_[i] = state.contacts[i] ? 1 : 0;
        // which feet belong to which horizon step: one recorded (calculate_B_mat_c, state_space_discretization) pair per block of B_mat_d_list
        if (static_cast<int>(recorded_B_.size()) < H)
            throw std::logic_error("a1mpc::ConvexMpcGpu: expected one calculate_B_mat_c + state_space_discretization call per horizon step before calculate_qp_mats");
        for (int i = 0; i < H; ++i)
            for (int r = 0; r < NS; ++r) for (int c = 0; c < NU; ++c)
                if (B_mat_d_list(i * NS + r, c) != recorded_B_[recorded_B_.size() - H + i](r, c))
                    throw std::logic_error("a1mpc::ConvexMpcGpu: B_mat_d_list was edited by other means than storing B_mat_d after each calculate_B_mat_c / "
                                           "state_space_discretization call; arbitrary B_d blocks cannot be passed to the GPU path (it takes foot positions)");
        if (static_cast<int>(feet_.size()) > 12 * H) feet_.erase(feet_.begin(), feet_.end() - 12 * H);   // the last H recorded calls = the H blocks of B_mat_d_list
        if (static_cast<int>(recorded_B_.size()) > H) recorded_B_.erase(recorded_B_.begin(), recorded_B_.end() - H);   // (a persistent instance that is never reset() would grow by H matrices per tick)
        per_step_feet_ = false;
        for (int i = 1; i < H && !per_step_feet_; ++i) per_step_feet_ = std::memcmp(&feet_[12 * i], &feet_[0], 12 * sizeof(double)) != 0;
        fz_min = 0; fz_max = 180;                                         // S/ConvexMpc.cpp:223-224
        cfg_.mu = mu; cfg_.fz_min = fz_min; cfg_.fz_max = fz_max;
        have_qp_ = true;
        if (materialize_qp_members) materialize();
    }

    // hessian / gradient / lb / ub as the reference's calculate_qp_mats leaves them, formed on the GPU
    void materialize() {
        if (!have_qp_) throw std::logic_error("a1mpc::ConvexMpcGpu::materialize before calculate_qp_mats");
        ensure_handle();
        const int n = NU * H, m = NC * H;
        std::vector<double> P(static_cast<size_t>(n) * n), g(n), l(m), u(m);
        check_status(a1mpc_form_qp_batch(box_.h, 1, x0_.data(), xref_.data(), R_, feet_.data(), per_step_feet_ ? 12 : 0, contact_, 0, yaw_ptr(), P.data(), g.data(), l.data(), u.data()),
                     "a1mpc_form_qp_batch");
        Eigen::Matrix<double, NU * H, NU * H> dense;
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) dense(i, j) = P[static_cast<size_t>(i) * n + j];
        hessian = dense.sparseView();                                     // S/ConvexMpc.cpp:211
        for (int i = 0; i < n; ++i) gradient(i) = g[i];
        for (int i = 0; i < m; ++i) { lb(i) = l[i]; ub(i) = u[i]; }
    }

    // what the reference does next with OsqpEigen (S/A1RobotControl.cpp:522-561): warm-started OSQP-faithful ADMM on the GPU, then R' f per leg
    Eigen::Matrix<double, 3, NLEG> solve() {
        if (!have_qp_) throw std::logic_error("a1mpc::ConvexMpcGpu::solve before calculate_qp_mats");
        ensure_handle();
        double grf[12];
        solution.resize(NU * H);
        check_status(a1mpc_solve_batch_strided(box_.h, 1, x0_.data(), xref_.data(), R_, feet_.data(), per_step_feet_ ? 12 : 0, contact_, 0, yaw_ptr(), grf, solution.data(),
                                               &last_iterations, &last_status), "a1mpc_solve_batch_strided");
        Eigen::Matrix<double, 3, NLEG> out;
        std::memcpy(out.data(), grf, sizeof grf);
        return out;
    }

    a1mpc_config& config() { return cfg_; }      // OSQP settings etc. (picked up by the next solve)
    a1mpc_handle handle() { ensure_handle(); return box_.h; }

    // ---- the reference's public members (S/ConvexMpc.h:37-93) ----
    double mu, fz_min, fz_max;
    Eigen::Matrix<double, NS, NS> A_mat_c, A_mat_d;
    Eigen::Matrix<double, NS, NU> B_mat_c, B_mat_d;
    Eigen::Matrix<double, NS * H, NU> B_mat_d_list;
    Eigen::SparseMatrix<double> hessian, linear_constraints;
    Eigen::Matrix<double, NU * H, 1> gradient;
    Eigen::Matrix<double, NC * H, 1> lb, ub;
    // ---- additions ----
    bool materialize_qp_members = true;
    Eigen::VectorXd solution;       // world-frame forces of every horizon step (OsqpEigen::Solver::getSolution())
    int32_t last_iterations = 0, last_status = 0;

  private:
    void ensure_handle() {
        if (!box_.h) check_status(a1mpc_create(&cfg_, 1, device_, &box_.h), "a1mpc_create");
        else check_status(a1mpc_update_config(box_.h, &cfg_), "a1mpc_update_config");   // host-only: the constants travel with every launch
    }
    // A_c is built from mpc_states[2] unless calculate_A_mat_c was given another yaw (S/test/test_mpc.cpp:94-104 passes an average)
    const double* yaw_ptr() const { return yaw_ == x0_[2] ? nullptr : &yaw_; }
    a1mpc_config cfg_;
    HandleBox box_;
    int device_;
    double yaw_ = 0.0, R_[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::vector<double> feet_, pending_feet_, x0_, xref_;
    std::vector<Eigen::Matrix<double, NS, NU>> recorded_B_;
    int calls_since_discretization_ = 0;
    uint8_t contact_[4] = {1, 1, 1, 1};
    bool per_step_feet_ = false, have_qp_ = false;
};

// =====================================================================================================================
// Drop-in for  Eigen::Matrix<double, 3, NUM_LEG> A1RobotControl::compute_grf(A1CtrlStates& state, double dt)   (S/A1RobotControl.h:44)
// State = the reference's A1CtrlStates.  One object per robot (it owns the persistent warm-started solver and the terrain filter, like the
// reference's `solver` and `terrain_angle_filter` members); one caller thread (the reference's thread 1).
// =====================================================================================================================
template <class State, int H = 10>
class ComputeGrfGpu {
  public:
    static constexpr int NLEG = A1MPC_NUM_LEG;
    explicit ComputeGrfGpu(int device = 0) : device_(device) {
        a1mpc_default_config(&cfg_); cfg_.horizon = H; a1mpc_default_balance_config(&qp_);
        cfg_.warm_start = 2;   // the reference's persistent `solver` member: initSolver once, then update*() + solve() every tick = OSQP's update path (S/A1RobotControl.cpp:522-538)
    }
    ~ComputeGrfGpu() { if (h_) a1mpc_destroy(h_); }
    ComputeGrfGpu(const ComputeGrfGpu&) = delete;
    ComputeGrfGpu& operator=(const ComputeGrfGpu&) = delete;

    std::string use_sim_time;        // S/A1RobotControl.h:105: "true" => mpc_dt = dt (S/A1RobotControl.cpp:465); the reference never sets it
    a1mpc_config& config() { return cfg_; }
    a1mpc_balance_config& balance_config() { return qp_; }
    int32_t last_iterations = 0, last_status = 0;

    Eigen::Matrix<double, 3, NLEG> compute_grf(State& state, double dt) {
        Eigen::Matrix<double, 3, NLEG> foot_forces_grf;
        foot_forces_grf.setZero();   // the reference leaves it uninitialised (:322); zeros replace that (a1mpc.h, status codes)
        Eigen::Vector3d euler_error = state.root_euler_d - state.root_euler;                                  // :325
        if (euler_error(2) > 3.1415926 * 1.5) euler_error(2) = state.root_euler_d(2) - 3.1415926 * 2 - state.root_euler(2);          // :328-332
        else if (euler_error(2) < -3.1415926 * 1.5) euler_error(2) = state.root_euler_d(2) + 3.1415926 * 2 - state.root_euler(2);
        sync_config(state, state.stance_leg_control_type == 1 && use_sim_time == "true" ? dt : 0.0025);     // :462-467
        double Rrow[9];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rrow[i * 3 + j] = state.root_rot_mat(i, j);
        uint8_t contact[4];
        for (int i = 0; i < NLEG; ++i) contact[i] = state.contacts[i] ? 1 : 0;
        double grf[12];
        if (state.stance_leg_control_type == 1) {                                                            // terrain adaptation, :335-376
            double pitch = state.root_euler_d[1], angle = 0.0, z = state.root_pos[2];
            check_status(a1mpc_terrain_batch(h_, state.use_terrain_adapt, 1, state.foot_pos_recent_contact.data(), &z, &pitch, &angle), "a1mpc_terrain_batch");
            state.root_euler_d[1] = pitch;
            state.terrain_pitch_angle = angle;
        }
        if (state.stance_leg_control_type == 0) {                                                            // balance QP, :377-444
            Eigen::Matrix<double, 6, 1> root_acc;
            root_acc.setZero();
            root_acc.template block<3, 1>(0, 0) = state.kp_linear.cwiseProduct(state.root_pos_d - state.root_pos);
            root_acc.template block<3, 1>(0, 0) += state.root_rot_mat * state.kd_linear.cwiseProduct(state.root_lin_vel_d - state.root_rot_mat.transpose() * state.root_lin_vel);
            root_acc.template block<3, 1>(3, 0) = state.kp_angular.cwiseProduct(euler_error);
            root_acc.template block<3, 1>(3, 0) += state.kd_angular.cwiseProduct(state.root_ang_vel_d - state.root_rot_mat.transpose() * state.root_ang_vel);
            root_acc(2) += state.robot_mass * 9.8;
            double Rz[9];
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rz[i * 3 + j] = state.root_rot_mat_z(i, j);
            check_status(a1mpc_balance_solve_batch(h_, &qp_, 1, root_acc.data(), Rrow, Rz, state.foot_pos_abs.data(), contact, grf, nullptr, &last_iterations, &last_status),
                         "a1mpc_balance_solve_batch");
            std::memcpy(foot_forces_grf.data(), grf, sizeof grf);
        } else if (state.stance_leg_control_type == 1) {                                                     // MPC, :446-562
            const double mpc_dt = cfg_.dt;
            state.mpc_states << state.root_euler[0], state.root_euler[1], state.root_euler[2], state.root_pos[0], state.root_pos[1], state.root_pos[2],
                state.root_ang_vel[0], state.root_ang_vel[1], state.root_ang_vel[2], state.root_lin_vel[0], state.root_lin_vel[1], state.root_lin_vel[2], -9.8;   // :452-456
            state.root_lin_vel_d_world = state.root_rot_mat * state.root_lin_vel_d;                          // :470
            for (int i = 0; i < H; ++i) {                                                                    // :472-488 (kept in the state, as the reference does)
                state.mpc_states_d.segment(i * 13, 13) << state.root_euler_d[0], state.root_euler_d[1], state.root_euler[2] + state.root_ang_vel_d[2] * mpc_dt * (i + 1),
                    state.root_pos[0] + state.root_lin_vel_d_world[0] * mpc_dt * (i + 1), state.root_pos[1] + state.root_lin_vel_d_world[1] * mpc_dt * (i + 1),
                    state.root_pos_d[2], state.root_ang_vel_d[0], state.root_ang_vel_d[1], state.root_ang_vel_d[2], state.root_lin_vel_d_world[0],
                    state.root_lin_vel_d_world[1], 0, -9.8;
            }
            // the compact tick record: x0 and x_ref are rebuilt on the device exactly as above (a1mpc_solve_batch_ticks)
            double tick[22];
            for (int i = 0; i < 3; ++i) {
                tick[i] = state.root_euler[i]; tick[3 + i] = state.root_pos[i]; tick[6 + i] = state.root_ang_vel[i]; tick[9 + i] = state.root_lin_vel[i];
                tick[12 + i] = state.root_euler_d[i]; tick[15 + i] = state.root_lin_vel_d[i]; tick[18 + i] = state.root_ang_vel_d[i];
            }
            tick[21] = state.root_pos_d[2];
            check_status(a1mpc_solve_batch_ticks(h_, 1, tick, Rrow, state.foot_pos_abs.data(), contact, grf, nullptr, &last_iterations, &last_status), "a1mpc_solve_batch_ticks");
            std::memcpy(foot_forces_grf.data(), grf, sizeof grf);
        }
        return foot_forces_grf;
    }

  private:
    // robot constants and weights are read from the state every tick, like the reference's per-tick ConvexMpc construction (:447)
    void sync_config(State& state, double mpc_dt) {
        a1mpc_config c = cfg_;
        // use_sim_time == "true" hands the caller's dt over (S/A1RobotControl.cpp:465), and the first simulated tick may carry dt <= 0: the reference then solves a
        // degenerate QP (B_d = 0) and returns its matrix; the library rejects a non-positive dt, and an exception out of the control thread is worse than either --
        // such a tick runs on the last valid dt (0.0025 before there was one)
        if (!(mpc_dt > 0.0)) mpc_dt = cfg_.dt > 0.0 ? cfg_.dt : 0.0025;
        c.dt = mpc_dt; c.mass = state.robot_mass;
        for (int i = 0; i < A1MPC_STATE_DIM; ++i) c.q[i] = state.q_weights(i);
        for (int i = 0; i < A1MPC_NUM_DOF; ++i) c.r[i] = state.r_weights(i);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) c.inertia_body[i * 3 + j] = state.a1_trunk_inertia(i, j);
        c.mu = 0.3; c.fz_min = 0.0; c.fz_max = 180.0;   // S/ConvexMpc.cpp:8,223-224
        if (!h_) { check_status(a1mpc_create(&c, 1, device_, &h_), "a1mpc_create"); a1mpc_set_timing(h_, 0); }   // (the control loop never reads the handle's timing events)
        else check_status(a1mpc_update_config(h_, &c), "a1mpc_update_config");   // host-only, every tick
        cfg_ = c;
    }
    a1mpc_config cfg_;
    a1mpc_balance_config qp_;
    a1mpc_handle h_ = nullptr;
    int device_;
};

}  // namespace a1mpc
