// tests/cpp/test_adapter.cpp -- instantiates include/a1mpc_convex_mpc.hpp with a minimal fixed-size matrix shim (Eigen is not
// installed here) on the inputs of S/test/test_mpc.cpp:18-60 and prints the GRFs, like the reference's test does.
// Build: g++ -std=c++17 -I include tests/cpp/test_adapter.cpp -L a1-qp-mpc-controller_amd -la1mpc -Wl,-rpath,... ; runs on a GPU box only.
#include <cstdio>

#include "a1mpc_convex_mpc.hpp"

template <int R, int C>
struct Mat {
    double a[R * C] = {};
    double& operator()(int i, int j) { return a[j * R + i]; }
    double operator()(int i, int j) const { return a[j * R + i]; }
    double& operator()(int i) { return a[i]; }
    double operator()(int i) const { return a[i]; }
};
struct State {  // the subset of A1CtrlStates the path reads (S/A1CtrlStates.h:354-411)
    Mat<13, 1> mpc_states;
    Mat<130, 1> mpc_states_d;
    Mat<3, 1> root_euler, root_pos, root_ang_vel, root_lin_vel, root_euler_d, root_ang_vel_d, root_lin_vel_d, root_lin_vel_d_world, root_pos_d;
    Mat<3, 3> root_rot_mat, a1_trunk_inertia;
    Mat<3, 4> foot_pos_abs;
    double robot_mass = 15.0;
    bool contacts[4] = {true, false, true, false};
};

int main() {
    Mat<13, 1> q; Mat<12, 1> r;
    const double qv[13] = {1, 1, 1, 0, 0, 50, 0, 0, 1, 1, 1, 1, 0};
    for (int i = 0; i < 13; ++i) q(i) = qv[i];
    for (int i = 0; i < 12; ++i) r(i) = 1e-6;
    State s;
    s.root_pos(2) = 0.15; s.root_pos_d(2) = 0.15;
    for (int i = 0; i < 3; ++i) s.root_rot_mat(i, i) = 1.0;
    s.a1_trunk_inertia(0, 0) = 0.0158533; s.a1_trunk_inertia(1, 1) = 0.0377999; s.a1_trunk_inertia(2, 2) = 0.0456542;
    const double fx[4] = {0.17, 0.17, -0.17, -0.17}, fy[4] = {0.15, -0.15, 0.15, -0.15};
    for (int l = 0; l < 4; ++l) { s.foot_pos_abs(0, l) = fx[l]; s.foot_pos_abs(1, l) = fy[l]; s.foot_pos_abs(2, l) = -0.35; }
    a1mpc::ConvexMpc mpc(q, r);
    mpc.config().warm_start = 0;
    Mat<3, 4> grf;
    a1mpc::compute_grf_mpc(mpc, s, grf);
    for (int l = 0; l < 4; ++l) std::printf("leg %d: %.6f %.6f %.6f\n", l, grf(0, l), grf(1, l), grf(2, l));
    std::printf("iterations %d\n", mpc.last_iterations());
    // reference fixture T: FL = RL ~ (0, -12.78, 42.61) N at OSQP default tolerances, swing legs ~ 0
    const bool ok = grf(2, 0) > 42.0 && grf(2, 0) < 43.2 && grf(1, 0) < -12.5 && grf(1, 0) > -13.1 && grf(2, 1) < 1e-2 && grf(2, 1) > -1e-2;
    std::printf(ok ? "ADAPTER_OK\n" : "ADAPTER_MISMATCH\n");
    return ok ? 0 : 1;
}
