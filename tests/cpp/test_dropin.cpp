// tests/cpp/test_dropin.cpp -- TEST: include/a1mpc_dropin.hpp next to the REFERENCE's own classes, in one process, on the reference's own
// A1CtrlStates type.  Built here (where /root/reference exists) by tests/cpp/Makefile against oracle/ref_shim (the Eigen stand-in) and linked
// with oracle/_ref/liba1ref.so (the reference compiled verbatim) and liba1mpc.so; runs on the GPU box.
//   part A: a1mpc::ConvexMpcGpu driven like S/A1RobotControl.cpp:447-518 / S/test/test_mpc.cpp:61-125 -- every public member the reference's
//           callers touch equals the reference ConvexMpc's (hessian / gradient / lb / ub formed on the GPU), per-step feet included
//   part B: a1mpc::ComputeGrfGpu::compute_grf vs A1RobotControl::compute_grf, MPC branch with the terrain block over a warm-started
//           sequence, then the balance branch
#include <cstdio>
#include <algorithm>
#include <random>

#include "A1RobotControl.h"
#include "ConvexMpc.h"
#include "a1mpc_dropin.hpp"

static std::mt19937_64 rng(12345);
static double uni(double a, double b) { return std::uniform_real_distribution<double>(a, b)(rng); }
static double nrm(double s) { return std::normal_distribution<double>(0.0, s)(rng); }

static Eigen::Matrix3d rot_zyx(double roll, double pitch, double yaw) {
    const double cr = cos(roll), sr = sin(roll), cp = cos(pitch), sp = sin(pitch), cy = cos(yaw), sy = sin(yaw);
    Eigen::Matrix3d R;
    R << cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr, sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr, -sp, cp * sr, cp * cr;
    return R;
}
static void random_state(A1CtrlStates& s, int type) {
    s.stance_leg_control_type = type; s.use_terrain_adapt = 1;
    s.robot_mass = 12.0;
    s.q_weights << 20, 10, 1, 0, 0, 420, .05, .05, .05, 30, 30, 10, 0;
    for (int i = 0; i < 12; ++i) s.r_weights(i) = 1e-7;
    s.root_euler << nrm(0.03), 0.15 + nrm(0.02), uni(-3, 3);
    s.root_rot_mat = rot_zyx(s.root_euler[0], s.root_euler[1], s.root_euler[2]);
    s.root_rot_mat_z = rot_zyx(0, 0, s.root_euler[2]);
    s.root_pos << nrm(0.5), nrm(0.5), 0.3 + nrm(0.005);
    s.root_pos_d = s.root_pos; s.root_pos_d[2] = 0.3;
    s.root_ang_vel << nrm(0.1), nrm(0.1), nrm(0.1);
    s.root_lin_vel << 0.3 + nrm(0.05), nrm(0.05), nrm(0.05);
    s.root_lin_vel_d << 0.3, 0.0, 0.0; s.root_ang_vel_d << 0.0, 0.0, 0.1;
    s.root_euler_d << 0.0, s.root_euler_d[1], s.root_euler[2] + nrm(0.05);
    for (int l = 0; l < 4; ++l) {
        Eigen::Vector3d r(s.default_foot_pos(0, l) + nrm(0.02), s.default_foot_pos(1, l) + nrm(0.02), s.default_foot_pos(2, l) + nrm(0.02));
        Eigen::Vector3d a = s.root_rot_mat * r;
        a[2] += 0.25 * a[0];   // feet on a slope: the terrain block has something to find
        s.foot_pos_abs.block<3, 1>(0, l) = a;
        s.foot_pos_recent_contact.block<3, 1>(0, l) = a;
        s.contacts[l] = uni(0, 1) < 0.75;
    }
    if (!(s.contacts[0] || s.contacts[1] || s.contacts[2] || s.contacts[3])) s.contacts[0] = true;
}
static double maxdiff(const Eigen::Dyn<double>& a, const Eigen::Dyn<double>& b) {
    double m = 0; for (long k = 0; k < a.size(); ++k) m = std::max(m, std::abs(a.data()[k] - b.data()[k])); return m;
}
static double maxabs(const Eigen::Dyn<double>& a) { double m = 0; for (long k = 0; k < a.size(); ++k) m = std::max(m, std::abs(a.data()[k])); return m; }

int main() {
    std::cout.setstate(std::ios_base::failbit);   // the reference prints every tick
    int bad = 0;
    // ------------------------------------------------------------------------------------------------ part A
    for (int rep = 0; rep < 6; ++rep) {
        A1CtrlStates s; random_state(s, 1);
        const bool per_step = rep % 2 == 1;
        Eigen::Vector3d eul_A = s.root_euler; if (rep >= 4) eul_A[2] = s.root_euler[2] * 2.0 / 11.0;   // S/test/test_mpc.cpp:94-102 passes an "average" yaw
        ConvexMpc ref(s.q_weights, s.r_weights); ref.reset();
        a1mpc::ConvexMpcGpu<PLAN_HORIZON> gpu(s.q_weights, s.r_weights); gpu.reset();
        s.mpc_states << s.root_euler[0], s.root_euler[1], s.root_euler[2], s.root_pos[0], s.root_pos[1], s.root_pos[2], s.root_ang_vel[0], s.root_ang_vel[1],
            s.root_ang_vel[2], s.root_lin_vel[0], s.root_lin_vel[1], s.root_lin_vel[2], -9.8;
        for (int i = 0; i < 13 * PLAN_HORIZON; ++i) s.mpc_states_d(i) = s.mpc_states(i % 13) + nrm(0.01);
        ref.calculate_A_mat_c(eul_A); gpu.calculate_A_mat_c(eul_A);
        s.foot_pos_abs_mpc = s.foot_pos_abs;
        for (int i = 0; i < PLAN_HORIZON; i++) {   // S/test/test_mpc.cpp:106-122
            ref.calculate_B_mat_c(s.robot_mass, s.a1_trunk_inertia, s.root_rot_mat, s.foot_pos_abs_mpc);
            gpu.calculate_B_mat_c(s.robot_mass, s.a1_trunk_inertia, s.root_rot_mat, s.foot_pos_abs_mpc);
            if (per_step) for (int l = 0; l < 4; ++l) s.foot_pos_abs_mpc.block<3, 1>(0, l) = s.foot_pos_abs_mpc.block<3, 1>(0, l) - s.root_lin_vel_d * 0.1;
            ref.state_space_discretization(0.0025); gpu.state_space_discretization(0.0025);
            ref.B_mat_d_list.block<13, 12>(i * 13, 0) = ref.B_mat_d;
            gpu.B_mat_d_list.block<13, 12>(i * 13, 0) = gpu.B_mat_d;
        }
        ref.calculate_qp_mats(s); gpu.calculate_qp_mats(s);
        const double eA = maxdiff(ref.A_mat_d, gpu.A_mat_d), eB = maxdiff(ref.B_mat_d_list, gpu.B_mat_d_list);
        const double eP = maxdiff(ref.hessian.dense, gpu.hessian.dense) / maxabs(ref.hessian.dense), eg = maxdiff(ref.gradient, gpu.gradient) / std::max(maxabs(ref.gradient), 1e-300);
        const double el = maxdiff(ref.lb, gpu.lb), eu = maxdiff(ref.ub, gpu.ub), eC = maxdiff(ref.linear_constraints.dense, gpu.linear_constraints.dense);
        const bool ok = eA == 0 && eB == 0 && eP <= 1e-12 && eg <= 1e-10 && el == 0 && eu == 0 && eC == 0;
        std::fprintf(stderr, "A%d per_step=%d yawA=%d: A_d %.1e B_list %.1e hessian(rel) %.1e gradient(rel) %.1e lb %.1e ub %.1e Ac %.1e %s\n", rep, per_step, rep >= 4, eA, eB, eP, eg, el, eu, eC, ok ? "ok" : "MISMATCH");
        bad += !ok;
        // the solve the reference would do next: OsqpEigen on its own members vs the drop-in's solve()
        OsqpEigen::Solver solver;
        solver.settings()->setVerbosity(false); solver.settings()->setWarmStart(false);
        solver.data()->setNumberOfVariables(12 * PLAN_HORIZON); solver.data()->setNumberOfConstraints(20 * PLAN_HORIZON);
        solver.data()->setLinearConstraintsMatrix(ref.linear_constraints); solver.data()->setHessianMatrix(ref.hessian); solver.data()->setGradient(ref.gradient);
        solver.data()->setLowerBound(ref.lb); solver.data()->setUpperBound(ref.ub);
        solver.initSolver(); solver.solve();
        Eigen::VectorXd sol = solver.getSolution();
        gpu.config().warm_start = 0;
        Eigen::Matrix<double, 3, 4> grf = gpu.solve();
        double es = maxdiff(sol, gpu.solution), eR = 0;
        for (int l = 0; l < 4; ++l) { Eigen::Vector3d f = s.root_rot_mat.transpose() * sol.segment<3>(l * 3); for (int k = 0; k < 3; ++k) eR = std::max(eR, std::abs(f[k] - grf(k, l))); }
        const bool ok2 = es <= 1e-5 && eR <= 1e-5 && gpu.last_iterations == OsqpEigen::shim_last().info.iters;
        std::fprintf(stderr, "   solve: |du| %.1e |dgrf| %.1e iters %d vs %d %s\n", es, eR, gpu.last_iterations, OsqpEigen::shim_last().info.iters, ok2 ? "ok" : "MISMATCH");
        bad += !ok2;
    }
    // ------------------------------------------------------------------------------------------------ part B
    {
        A1RobotControl ref;
        a1mpc::ComputeGrfGpu<A1CtrlStates, PLAN_HORIZON> gpu;
        A1CtrlStates sr, sg;
        double worst = 0, worst_pitch = 0; int iter_mis = 0;
        for (int t = 0; t < 60; ++t) {
            A1CtrlStates s; random_state(s, 1);
            s.root_euler_d[1] = sr.root_euler_d[1];           // the terrain block's own output is carried from tick to tick
            const double pitch_g = sg.root_euler_d[1];
            sr = s; sg = s; sg.root_euler_d[1] = pitch_g;
            if (t == 30) { sr.root_pos[2] = sg.root_pos[2] = 0.05; }   // body low: terrain angle forced to zero (:341-345)
            Eigen::Matrix<double, 3, 4> a = ref.compute_grf(sr, 0.0025), b = gpu.compute_grf(sg, 0.0025);
            worst = std::max(worst, maxdiff(a, b));
            worst_pitch = std::max({worst_pitch, std::abs(sr.root_euler_d[1] - sg.root_euler_d[1]), std::abs(sr.terrain_pitch_angle - sg.terrain_pitch_angle)});
            iter_mis += OsqpEigen::shim_last().info.iters != gpu.last_iterations;
            if (maxdiff(sr.mpc_states, sg.mpc_states) != 0 || maxdiff(sr.mpc_states_d, sg.mpc_states_d) > 1e-12) { ++bad; std::fprintf(stderr, "B tick %d: mpc_states differ\n", t); }
        }
        const bool ok = worst <= 1e-6 && worst_pitch <= 1e-9 && iter_mis == 0 && std::abs(sg.root_euler_d[1]) > 0.05;
        std::fprintf(stderr, "B mpc+terrain, 60 warm-started ticks: |dGRF| %.2e N, |d pitch| %.1e, iteration mismatches %d, final desired pitch %.3f %s\n", worst, worst_pitch, iter_mis,
                     sg.root_euler_d[1], ok ? "ok" : "MISMATCH");
        bad += !ok;
        worst = 0; iter_mis = 0;
        for (int t = 0; t < 40; ++t) {
            A1CtrlStates s; random_state(s, 0);
            s.root_euler_d << s.root_euler[0] + nrm(0.05), s.root_euler[1] + nrm(0.05), s.root_euler[2] + (t == 7 ? 5.0 : nrm(0.05));   // t = 7: the yaw wrap (:328-332)
            sr = s; sg = s;
            Eigen::Matrix<double, 3, 4> a = ref.compute_grf(sr, 0.0025), b = gpu.compute_grf(sg, 0.0025);
            worst = std::max(worst, maxdiff(a, b));
            iter_mis += OsqpEigen::shim_last().info.iters != gpu.last_iterations;
        }
        const bool ok3 = worst <= 2e-4 && iter_mis == 0;
        std::fprintf(stderr, "B balance QP, 40 ticks: |dGRF| %.2e N, iteration mismatches %d %s\n", worst, iter_mis, ok3 ? "ok" : "MISMATCH");
        bad += !ok3;
    }
    std::fprintf(stderr, bad ? "DROPIN_MISMATCH\n" : "DROPIN_OK\n");
    return bad ? 1 : 0;
}
