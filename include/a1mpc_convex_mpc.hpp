// a1mpc_convex_mpc.hpp -- header-only C++ adapter above the C ABI (a1mpc.h) that keeps the reference's own interface
// for the hot path, so the ROS control loop can switch by changing a type name.
//
//   reference                                                    here
//   ConvexMpc(q_weights, r_weights)        S/ConvexMpc.h:24       a1mpc::ConvexMpc(q_weights, r_weights)
//   reset()                                S/ConvexMpc.h:26       reset()
//   calculate_A_mat_c(root_euler)          S/ConvexMpc.h:28       same name: records the yaw (A_c itself is built on the GPU)
//   calculate_B_mat_c(m, I, R, foot)       S/ConvexMpc.h:30-31    same name: records m, I_body, R, foot_pos_abs
//   state_space_discretization(dt)         S/ConvexMpc.h:33       same name: records dt
//   calculate_qp_mats(state)               S/ConvexMpc.h:35       same name: records mpc_states, mpc_states_d, contacts
//   OsqpEigen::Solver solve/getSolution    S/A1RobotControl.cpp:522-561   solve(): formation + OSQP-faithful ADMM on the GPU,
//                                                                 returns the 3x4 body-frame GRF matrix compute_grf returns
//   A1RobotControl::compute_grf (MPC branch, S/A1RobotControl.cpp:446-562)   a1mpc::compute_grf_mpc(mpc, state)
//
// The matrix / vector types are template parameters used only through operator()(i) / operator()(i,j), so the header
// compiles against Eigen 3 (the reference's types) without including it -- Eigen is not installed in the build container;
// tests/cpp/test_adapter.cpp instantiates it with a minimal fixed-size shim.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "a1mpc.h"

namespace a1mpc {

class ConvexMpc {
public:
    static constexpr int kStateDim = A1MPC_STATE_DIM, kNumDof = A1MPC_NUM_DOF, kNumLeg = A1MPC_NUM_LEG;

    // q_weights (13), r_weights (12): anything indexable with operator()(int)  (Eigen::VectorXd in the reference)
    template <class VecQ, class VecR>
    ConvexMpc(const VecQ& q_weights, const VecR& r_weights, int horizon = 10, int device = 0) : device_(device) {
        a1mpc_default_config(&cfg_);
        cfg_.horizon = horizon;
        for (int i = 0; i < kStateDim; ++i) cfg_.q[i] = q_weights(i);
        for (int i = 0; i < kNumDof; ++i) cfg_.r[i] = r_weights(i);
        x0_.assign(kStateDim, 0.0);
        xref_.assign(static_cast<size_t>(kStateDim) * horizon, 0.0);
    }
    ~ConvexMpc() { if (h_) a1mpc_destroy(h_); }
    ConvexMpc(const ConvexMpc&) = delete;
    ConvexMpc& operator=(const ConvexMpc&) = delete;

    a1mpc_config& config() { return cfg_; }  // mu, fz_max, OSQP settings ... before the first solve()

    void reset() {}  // S/ConvexMpc.cpp:70-108 zero-fills work matrices; nothing to clear here (kept for source compatibility)

    template <class Vec3>
    void calculate_A_mat_c(const Vec3& root_euler) { yaw_ = root_euler(2); }  // S/ConvexMpc.cpp:110-130 uses only the yaw

    template <class Mat3, class Mat34>
    void calculate_B_mat_c(double robot_mass, const Mat3& a1_trunk_inertia, const Mat3& root_rot_mat, const Mat34& foot_pos) {
        if (h_ && (robot_mass != cfg_.mass)) throw std::logic_error("a1mpc::ConvexMpc: robot_mass changed after the first solve()");
        cfg_.mass = robot_mass;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) { cfg_.inertia_body[i * 3 + j] = a1_trunk_inertia(i, j); R_[i * 3 + j] = root_rot_mat(i, j); }
        for (int leg = 0; leg < kNumLeg; ++leg)
            for (int i = 0; i < 3; ++i) foot_[leg * 3 + i] = foot_pos(i, leg);  // 3x4, column = leg
    }

    void state_space_discretization(double dt) {
        if (h_ && dt != cfg_.dt) throw std::logic_error("a1mpc::ConvexMpc: dt changed after the first solve()");
        cfg_.dt = dt;
    }

    // state: anything with mpc_states (13), mpc_states_d (13*H), contacts[4]  (A1CtrlStates in the reference)
    template <class State>
    void calculate_qp_mats(const State& state) {
        for (int i = 0; i < kStateDim; ++i) x0_[i] = state.mpc_states(i);
        for (size_t i = 0; i < xref_.size(); ++i) xref_[i] = state.mpc_states_d(static_cast<int>(i));
        for (int i = 0; i < kNumLeg; ++i) contact_[i] = state.contacts[i] ? 1 : 0;
        x0_[2] = x0_[2];  // A_c is built from mpc_states[2] == root_euler[2] (S/A1RobotControl.cpp:452,492)
        (void)yaw_;
    }

    // formation + solve on the GPU; out(i, leg) receives the body-frame GRFs (Eigen::Matrix<double,3,4> in the reference)
    template <class Mat34>
    int32_t solve(Mat34& out) {
        ensure_handle();
        double grf[12];
        int32_t iters = 0, status = 0;
        const a1mpc_status rc = a1mpc_solve_batch(h_, 1, x0_.data(), xref_.data(), R_, foot_, contact_, grf, nullptr, &iters, &status);
        if (rc != A1MPC_OK) throw std::runtime_error(std::string("a1mpc_solve_batch: ") + a1mpc_status_string(rc) + ": " + a1mpc_last_error());
        for (int leg = 0; leg < kNumLeg; ++leg)
            for (int i = 0; i < 3; ++i) out(i, leg) = grf[leg * 3 + i];
        last_iters_ = iters;
        return status;
    }
    int32_t last_iterations() const { return last_iters_; }

private:
    void ensure_handle() {
        if (h_) return;
        const a1mpc_status rc = a1mpc_create(&cfg_, 1, device_, &h_);
        if (rc != A1MPC_OK) throw std::runtime_error(std::string("a1mpc_create: ") + a1mpc_status_string(rc) + ": " + a1mpc_last_error());
    }
    a1mpc_config cfg_;
    a1mpc_handle h_ = nullptr;
    int device_;
    double yaw_ = 0.0, R_[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, foot_[12] = {};
    uint8_t contact_[4] = {1, 1, 1, 1};
    std::vector<double> x0_, xref_;
    int32_t last_iters_ = 0;
};

// The MPC branch of A1RobotControl::compute_grf (S/A1RobotControl.cpp:446-562) with the reference's statement order.
// `mpc` must outlive the ticks: it owns the persistent (warm-started) solver like the reference's `solver` member.
template <class State, class Mat34>
void compute_grf_mpc(ConvexMpc& mpc, State& state, Mat34& foot_forces_grf, double mpc_dt = 0.0025, int horizon = 10) {
    mpc.reset();
    for (int i = 0; i < 3; ++i) {                                   // :452-456
        state.mpc_states(i) = state.root_euler(i);
        state.mpc_states(3 + i) = state.root_pos(i);
        state.mpc_states(6 + i) = state.root_ang_vel(i);
        state.mpc_states(9 + i) = state.root_lin_vel(i);
    }
    state.mpc_states(12) = -9.8;
    for (int i = 0; i < 3; ++i) {                                   // :470
        double s = 0;
        for (int j = 0; j < 3; ++j) s += state.root_rot_mat(i, j) * state.root_lin_vel_d(j);
        state.root_lin_vel_d_world(i) = s;
    }
    for (int i = 0; i < horizon; ++i) {                              // :471-488
        const double k = mpc_dt * (i + 1);
        state.mpc_states_d(i * 13 + 0) = state.root_euler_d(0);
        state.mpc_states_d(i * 13 + 1) = state.root_euler_d(1);
        state.mpc_states_d(i * 13 + 2) = state.root_euler(2) + state.root_ang_vel_d(2) * k;
        state.mpc_states_d(i * 13 + 3) = state.root_pos(0) + state.root_lin_vel_d_world(0) * k;
        state.mpc_states_d(i * 13 + 4) = state.root_pos(1) + state.root_lin_vel_d_world(1) * k;
        state.mpc_states_d(i * 13 + 5) = state.root_pos_d(2);
        state.mpc_states_d(i * 13 + 6) = state.root_ang_vel_d(0);
        state.mpc_states_d(i * 13 + 7) = state.root_ang_vel_d(1);
        state.mpc_states_d(i * 13 + 8) = state.root_ang_vel_d(2);
        state.mpc_states_d(i * 13 + 9) = state.root_lin_vel_d_world(0);
        state.mpc_states_d(i * 13 + 10) = state.root_lin_vel_d_world(1);
        state.mpc_states_d(i * 13 + 11) = 0;
        state.mpc_states_d(i * 13 + 12) = -9.8;
    }
    mpc.calculate_A_mat_c(state.root_euler);                         // :492
    mpc.calculate_B_mat_c(state.robot_mass, state.a1_trunk_inertia, state.root_rot_mat, state.foot_pos_abs);  // :498-503
    mpc.state_space_discretization(mpc_dt);                          // :510
    mpc.calculate_qp_mats(state);                                    // :518
    mpc.solve(foot_forces_grf);                                      // :522-561
}

}  // namespace a1mpc
