// a1mpc_solver.hpp -- one convex-MPC QP per DPP row (16 lanes), two (one, four) QPs per wavefront.
//
// What it computes (reference behaviour, cited as S/ = src/a1_cpp/src/ of the reference tree):
//   * QP formation of ConvexMpc (S/ConvexMpc.cpp:110-156 A_c/B_c/Euler discretisation, :181-245
//     A_qp/B_qp, P = B_qp' Q B_qp + R, g = B_qp' Q (A_qp x0 - x_ref), friction-pyramid rows and bounds)
//     as driven by A1RobotControl::compute_grf (S/A1RobotControl.cpp:446-562: same feet / rotation for
//     every horizon step, contacts broadcast over the horizon, first-step forces rotated by R').
//   * The OSQP 0.6 ADMM iteration the reference delegates to (call sites S/A1RobotControl.cpp:522-555):
//     Ruiz equilibration (10 passes) + cost scaling, per-row rho (equality rows x1e3), over-relaxed
//     ADMM (alpha 1.6), unscaled residual termination every 25 iterations, rho adaptation with
//     re-factorisation -- the same iterates as oracle/a1mpc_oracle.c, to round-off.
//
// How (MI355X-first, nothing dense):
//   * The horizon structure is exploited instead of materialised.  With the reference's step-invariant
//     B_d and A_d = I + dt*A_c (A_c^2 B = 0) the Hessian is  P = alpha (x) U + beta (x) V + I (x) R  with two
//     12x12 matrices U, V per problem and two H x H integer tables alpha, beta shared by everybody, so
//     P is never formed: Ruiz column norms evaluate |gamma_st U_ab + V_ab| on the fly, P x and the
//     gradient are one forward roll-out + one adjoint sweep, and the ADMM linear system
//     (P + sigma I + A' rho A) x = b is an LQ problem solved by a Riccati recursion: per step a 12x12
//     feedback K_t and a 12x12 S_t^{-1} (234 doubles) live in LDS, 18.7 KB per QP at H = 10.
//   * Lane map inside a row: a 3-vector per quad.  Force layout: lane 4*leg+c holds f_{leg,c}; state
//     layout: quads hold (rpy, pos, omega, vel).  12x12 mat-vecs are 12 x v_fmac_f64_dpp row_newbcast (broadcast
//     and FMA in one instruction) with the matrix row read from LDS; the same LDS image serves K x (row read)
//     and K' r (column read).  The two Riccati sweeps of an ADMM iteration are monolithic instruction blocks
//     (csrc/gfx950/a1mpc_rowops.hpp: sweep_back_rhs / sweep_back_chains / sweep_fwd_gain / sweep_fwd_input).
//   * The ADMM state is carried unscaled in the w form (xh, wh0, wh1, rr0, rr1, sigma-term: 6 doubles per horizon
//     step and lane, + d_t between the sweeps) in VGPRs for the whole solve; all horizon loops over them are fully
//     unrolled (template H), the Riccati factor loop is not.  While rho is small the iterations also carry
//     G = c P x + c g through the x-update identity (RowSolver::careful).
//   * Drivers: solve_row (fused: set-up + solve, also the latency variant whose four rows share one QP's Ruiz
//     passes, RowSolver::coop_n), setup_row + admm_rows (split pipeline: persistent rows drain a work queue).
//
// Everything is IEEE double, like the reference (Eigen double, OSQP c_float = double).
#pragma once
#include <math.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

#include <a1mpc_rowops.hpp>  // csrc/gfx950/ (DPP) in the product build; tests/emu/ (host fibers) in the CPU test double

namespace a1mpc {

// ---- OSQP 0.6 constants (osqp/include/constants.h) ------------------------------------------------
constexpr double kInfty = 1e30;
constexpr double kRhoMin = 1e-6, kRhoMax = 1e6, kRhoEqOverIneq = 1e3, kRhoTol = 1e-4;
constexpr double kMinScaling = 1e-4, kMaxScaling = 1e4;

enum : int32_t {
    A1MPC_SOLVED = 1,
    A1MPC_SOLVED_INACCURATE = 2,
    A1MPC_MAX_ITER_REACHED = -2,
    A1MPC_NON_CVX = -7,
    A1MPC_UNSOLVED = -10,
};

struct DeviceParams {
    double dt, mu, fz_min, fz_max;
    double q2[12];  // 2*q[0..11]   (S/ConvexMpc.cpp:20; q[12], the gravity state, never reaches P or g)
    double r2[12];  // 2*r          (S/ConvexMpc.cpp:41)
    double mass, inertia[9];
    double rho0, sigma, alpha, eps_abs, eps_rel, adaptive_rho_tol;
    int32_t max_iter, check_every, adaptive_rho, adaptive_rho_every, scaling_iters, warm_start;
};

enum : int { kModeMpc = 0, kModeBalance = 1 };

struct ProblemIO {
    const double* root_acc;  // balance mode only: 6, desired wrench (S/A1RobotControl.cpp:379-391)
    const double* Rz;        // balance mode only: 9, row-major root_rot_mat_z
    const double* tick;      // 22 or null: compact tick record (N1) -- when set, x0 / xref are built on the device and the two pointers below are ignored
    const double* x0;        // 13
    const double* xref;      // 13*H
    const double* R;         // 9, row-major root_rot_mat
    const double* foot;      // 12, 3x4 column-major foot_pos_abs (world-aligned, CoM-relative); GEN: + foot_stride doubles per horizon step
    const uint8_t* contact;  // 4; + contact_stride bytes per horizon step
    int32_t foot_stride;     // GEN only: 0 = the same feet at every step (S/A1RobotControl.cpp:498-514), 12 = per-step feet (S/test/test_mpc.cpp:106-122)
    const double* yaw_A;     // GEN only, or null: the yaw A_c is built from when it is not mpc_states[2] (S/test/test_mpc.cpp:94-102 passes an average yaw)
    int32_t contact_stride;  // 0 = contacts broadcast over the horizon (S/ConvexMpc.cpp:228-245), 4 = a per-step contact schedule (read by every set-up: make_io_sched / make_io_gen)
    double* grf;             // 12 out: 3x4 column-major body-frame GRFs
    double* u_full;          // 12*H out (world frame, all steps) or null
    double* warm_x;          // 12*H in/out or null   (unscaled primal, OSQP workspace x)
    double* warm_y;          // 20*H in/out or null   (unscaled dual, reference row order)
    double* rho_io;          // 1 in/out or null      (carried rho)
    int32_t* iters;          // out or null
    int32_t* status;         // out or null
    int32_t* nfact;          // out or null
    double* carry;           // Carry<H>::STRIDE in/out or null: what the reference's persistent OSQP workspace still holds of the previous tick (warm_start = 2, the UPDATE path)
    int32_t* cost;           // fused kernel only, or null: this QP's cost record (iterations + 10 factor passes), the next tick's launch order (a1mpc_solve_kernel)
    // N3 in the output stage (a1mpc_control_tick_device; fused / latency kernels only): compute_joint_torques (S/A1RobotControl.cpp:289-319) of this robot, or tq_tau = null
    const uint8_t* tq_active;  // 1: 0 while the reference's mpc_init_counter < 10
    const double* tq_J;        // 36: the four diagonal 3x3 blocks of j_foot, column-major each
    const double* tq_fkin;     // 12: foot_forces_kin
    const double* tq_tg;       // 12: torques_gravity
    const double* tq_km;       // 3: km_foot (device copy inside the handle)
    double* tq_tau;            // 12 in/out: joint_torques (NaN results keep the previous value)
};

// compute_joint_torques for ONE leg (S/A1RobotControl.cpp:297-311): stance: tau = J'(-f_grf); swing: tau = J^-1 (km .* f_kin) by partial-pivot LU in Eigen's operation
// order (unblocked_lu).  No FMA contraction: bit-identical to the reference's C++ arithmetic (and to the oracle, which is compiled the same way).  Shared by the
// stand-alone torque kernel and the MPC kernels' output stage.
A1_DEV void leg_joint_torque(const double* __restrict__ Jp, bool contact, double g0, double g1, double g2, double k0, double k1, double k2, double (&t)[3]) {
#pragma clang fp contract(off)
    double J[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) J[k] = Jp[k];
    if (contact) {                                                             // :303  J' (-f)
        const double f0 = -g0, f1 = -g1, f2 = -g2;
        t[0] = J[0] * f0 + J[1] * f1 + J[2] * f2;
        t[1] = J[3] * f0 + J[4] * f1 + J[5] * f2;
        t[2] = J[6] * f0 + J[7] * f1 + J[8] * f2;
    } else {                                                                    // :306-307  PartialPivLU, Eigen's unblocked_lu order
        double b0 = k0, b1 = k1, b2 = k2;
        // k = 0: pivot = first largest |J(r,0)|
        int r0 = 0; double big = fabs(J[0]);
        if (fabs(J[1]) > big) { big = fabs(J[1]); r0 = 1; }
        if (fabs(J[2]) > big) { big = fabs(J[2]); r0 = 2; }
        if (big != 0.0) {
            if (r0 == 1) { double x; x = J[0]; J[0] = J[1]; J[1] = x; x = J[3]; J[3] = J[4]; J[4] = x; x = J[6]; J[6] = J[7]; J[7] = x; }
            if (r0 == 2) { double x; x = J[0]; J[0] = J[2]; J[2] = x; x = J[3]; J[3] = J[5]; J[5] = x; x = J[6]; J[6] = J[8]; J[8] = x; }
            J[1] /= J[0]; J[2] /= J[0];
        }
        J[4] -= J[1] * J[3]; J[7] -= J[1] * J[6]; J[5] -= J[2] * J[3]; J[8] -= J[2] * J[6];
        // k = 1
        int r1 = 1; big = fabs(J[4]);
        if (fabs(J[5]) > big) { big = fabs(J[5]); r1 = 2; }
        if (big != 0.0) {
            if (r1 == 2) { double x; x = J[1]; J[1] = J[2]; J[2] = x; x = J[4]; J[4] = J[5]; J[5] = x; x = J[7]; J[7] = J[8]; J[8] = x; }
            J[5] /= J[4];
        }
        J[8] -= J[5] * J[7];
        // P b, L y = P b, U x = y
        if (r0 == 1) { const double x = b0; b0 = b1; b1 = x; }
        if (r0 == 2) { const double x = b0; b0 = b2; b2 = x; }
        if (r1 == 2) { const double x = b1; b1 = b2; b2 = x; }
        b1 -= J[1] * b0;
        b2 -= J[2] * b0 + J[5] * b1;
        b2 /= J[8];
        b1 -= J[7] * b2; b1 /= J[4];
        b0 -= J[3] * b1 + J[6] * b2; b0 /= J[0];
        t[0] = b0; t[1] = b1; t[2] = b2;
    }
}

// ---- compile-time helpers ----------------------------------------------------------------------
#define A1_CV(x) (std::remove_cv_t<std::remove_reference_t<decltype(x)>>::value)  // value of an integral_constant argument
template <class F, int... I>
A1_DEV void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
A1_DEV void static_for(F&& f) {
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}
constexpr int lane_of(int j) { return 4 * (j / 3) + (j % 3); }  // compact 0..11 -> lane in row
template <int J>
A1_DEV double bc(double v) {
    return row_bcast<lane_of(J)>(v);
}
// init + sum_{j<N} m[j] * x[compact index OFF + j]  (two interleaved accumulator chains; x must be row_dpp_ready())
template <int OFF, int N>
A1_DEV double dot_bc(const double (&m)[N], double x, double init = 0.0) {
    static_assert(N >= 2 && N % 2 == 0, "even length");
    double a0 = init, a1 = 0.0;
    static_for<N / 2>([&](auto J) {
        constexpr int j = 2 * A1_CV(J);
        fma_bcast<lane_of(OFF + j)>(a0, m[j], x);
        fma_bcast<lane_of(OFF + j + 1)>(a1, m[j + 1], x);
    });
    return a0 + a1;
}
// below this rho the dual residual at a checkpoint is dominated by the x-update's backward error unless c P x + c g is carried (RowSolver::careful)
constexpr double kRhoCareful = 1e-3;
constexpr double kSigGeneralPath = 2251799813685248.0;   // 2^51: tag of a pattern signature (Carry<H>::SIG) written by the general path; its hash stays below 2^50
// gamma_st = alpha_st / beta_st and beta_st = H - max(s,t) never grow with t for fixed s (csrc/a1mpc_tables.hpp; the quotients are compared as integers,
// rounding them to double is monotone)
constexpr bool gamma_beta_monotone(int H) {
    for (int s = 0; s < H; ++s) {
        long ap = 0, bp = 0;
        for (int t = 0; t < H; ++t) {
            const int m = s > t ? s : t;
            long a = 0;
            for (int i = m; i < H; ++i) a += static_cast<long>(i - s) * (i - t);
            const long b = H - m;
            if (t > 0 && (b > bp || a * bp > ap * b)) return false;  // beta grew, or alpha / beta > alpha' / beta'
            ap = a; bp = b;
        }
    }
    return true;
}
constexpr int alpha_diag(int s, int H) {  // sum_{i=s}^{H-1} (i-s)^2
    int a = 0;
    for (int i = s; i < H; ++i) a += (i - s) * (i - s);
    return a;
}

A1_DEV int imin(int a, int b) { return a < b ? a : b; }
A1_DEV double limit_scaling(double v) {  // osqp scaling.c limit_scaling
    v = v < kMinScaling ? 1.0 : v;
    v = v > kMaxScaling ? kMaxScaling : v;
    return v;
}
A1_DEV double row_allmax(double v) {
    v = max_f64(v, row_ror<8>(v));
    v = max_f64(v, row_ror<4>(v));
    v = max_f64(v, row_ror<2>(v));
    v = max_f64(v, row_ror<1>(v));
    return v;
}
A1_DEV double row_allsum(double v) {
    v += row_ror<8>(v);
    v += row_ror<4>(v);
    v += row_ror<2>(v);
    v += row_ror<1>(v);
    return v;
}

// ---- packed S_t^-1: which of the 78 doubles of a step's S area holds entry (i, j), i >= j ------------------------------------
// Not row-major triangular: the backward sweep of a twin pair reads "cross" b -- lane ci takes entry (max(ci, b), min(ci, b)) -- with one
// ds_read_b64, which the LDS serves in two 32-lane groups on 64 dword banks (32 doubles), and the lane group of the twin rows holds TWO QPs
// whose images lie 16 doubles (mod 32) apart.  With the twelve entries of every cross on twelve different residues mod 16 the 24 reads of a
// group never share a bank (row-major packing: 21 LDS cycles per 12 reads; SQ_LDS_BANK_CONFLICT was 11.7 % of the ADMM kernel's LDS cycles).
// Found by tools/sinv_layout_search.py (which also minimises the clashes of the factor pass's stores); checked below at compile time.
constexpr unsigned char kSinvSlot[78] = {
    57,
    49, 15,
    66,  9, 74,
     7, 54, 75, 14,
    76, 26, 30, 50, 36,
    45, 11, 44, 19, 24,  0,
     6, 62, 47, 65, 59, 58, 56,
    42, 55, 64, 52, 38, 41, 77, 21,
     8, 48, 20, 69, 13, 22, 28, 67, 71,
    63,  4, 70, 40,  5, 39, 51, 17,  2, 29,
    68, 72,  3, 12, 16, 34, 53, 31, 25, 27, 33,
    46, 61,  1, 32, 35, 37, 23, 60, 43, 73, 10, 18,
};
constexpr int sinv_slot_ce(int i, int j) { return i >= j ? kSinvSlot[i * (i + 1) / 2 + j] : kSinvSlot[j * (j + 1) / 2 + i]; }
constexpr bool sinv_slots_ok() {
    bool used[78] = {};
    for (int k = 0; k < 78; ++k) {  // a bijection onto the S area
        if (kSinvSlot[k] >= 78 || used[kSinvSlot[k]]) return false;
        used[kSinvSlot[k]] = true;
    }
    for (int b = 0; b < 12; ++b) {  // every cross on twelve different residues mod 16
        bool bank[16] = {};
        for (int c = 0; c < 12; ++c) {
            const int r = sinv_slot_ce(c, b) % 16;
            if (bank[r]) return false;
            bank[r] = true;
        }
    }
    return true;
}
static_assert(sinv_slots_ok(), "packed S^-1 slot table");
// Slot of entry (max(ci, B), min(ci, B)) for a compile-time B and a lane's ci (B = -1: the diagonal entry (ci, ci)): the twelve slots of cross B are
// packed into two immediates and picked by shifts -- plain arithmetic on ci that the compiler computes once per kernel and keeps in the twelve address
// registers the triangular formula used to occupy (a table in memory put a load and a branch into every use).
template <int B>
A1_DEV int sinv_slot(int ci) {
    constexpr auto at = [](int c) { return static_cast<unsigned long long>(B < 0 ? sinv_slot_ce(c, c) : sinv_slot_ce(c, B)); };
    constexpr unsigned long long lo = at(0) | at(1) << 8 | at(2) << 16 | at(3) << 24 | at(4) << 32 | at(5) << 40 | at(6) << 48 | at(7) << 56;
    constexpr unsigned hi = static_cast<unsigned>(at(8) | at(9) << 8 | at(10) << 16 | at(11) << 24);
    return ci < 8 ? static_cast<int>((lo >> (8 * ci)) & 0xff) : static_cast<int>((hi >> (8 * (ci - 8))) & 0xff);
}

// ---- LDS image of one QP ------------------------------------------------------------------------
// GEN (per-step feet / per-step contact schedules, RowSolver<.., GEN = true>): three more tables per QP behind c*g --
// the omega rows of B~_t for every step, and the slot-0 bounds of every step.
template <int H, bool GEN = false>
struct Layout {
    static constexpr int KSTR = 13;          // padded row stride of K_t: conflict-free row and column reads
    static constexpr int K_SZ = 12 * KSTR;   // 156
    static constexpr int S_SZ = 78;          // packed S_t^{-1}: the 78 entries i >= j, placed by kSinvSlot
    static constexpr int SLOT = K_SZ + S_SZ; // 234 doubles per horizon step
    static constexpr int FAC = 0;
    static constexpr int GCOL = 12;          // the pad column of K_t's stride-13 rows carries G_t = (c P x + c g)_t, see RowSolver::careful
    static constexpr int BL = H * SLOT;      // B~ (6x12): rows 0-2 = dt*Iw^-1*skew(r), rows 3-5 = dt/m*I
    static constexpr int ZROW = 6;           // a seventh, all-zero row of B~: the row of every lane that owns no wrench state
    static constexpr int CG = BL + 84;       // c*g = D^-1 q_s, [t][12]
    static constexpr int BW = CG + 12 * H;   // GEN: [t][3][12] = dt*Iw^-1*skew(r_t), the omega rows of B~_t (rows 3-5 of B~ are step-invariant)
    static constexpr int RAW = BW + (GEN ? 36 * H : 0);   // (round 6: the per-step bounds of the general path live in registers like the fast path's -- slot_bounds from the contact bits --
                                                          // not in two more tables: 24 H doubles less per image, seven instead of six QPs per CU at H = 10 in the CU-wide workgroup)
    // set-up-only aliases inside the factor region (the factor is written after Ruiz is finished)
    static constexpr int TBL = 0;            // T*B~_omega (3x12)
    static constexpr int DL = 36;            // D table of the current Ruiz pass, [t][12]
    static constexpr int GQT = DL + 12 * H;  // GEN: [t][12] the gradient the cost normalisation of the Ruiz passes sees (this tick's; on the update path the previous tick's)
    static constexpr int FT = GQT + (GEN ? 12 * H : 0);    // GEN: [t][12] the foot positions of step t (3 x 4, column-major like the input): the Ruiz sweeps evaluate a block entry as a cross product with them
    static constexpr int COOP = FT + (GEN ? 12 * H : 0);   // [row][s][16 lanes] partial column maxima of a set-up shared by several rows of a wave (RowSolver::coop_n)
    static constexpr int ML = COOP;          // GEN: [t][12] the column maxima m_t of a Ruiz pass, each written by the row that owns horizon step t (RowSolver::setup, general path)
    static constexpr int E0X = COOP + 4 * H * 16;          // [t][16 lanes]: the rows of a shared set-up hand each other the E of their own horizon steps, once, behind the last Ruiz pass
    static constexpr int TAB = E0X + H * 16;               // [s][t][2] = (alpha_st / beta_st, beta_st): the fused / latency kernels stage the batch's table here (stage_table), where a
                                                           // column of it is one LDS round trip away instead of one global-memory round trip per column of every Ruiz sweep
    static_assert(TAB + 2 * H * H <= COOP + 8 * H * 16, "the table fits behind the E hand-over");
    static_assert(ML + 12 * H <= E0X, "the general path's column maxima sit in front of the E hand-over");
    static_assert(COOP + 8 * H * 16 <= H * SLOT || H == 1, "alias");
    // row stride mod 32 in {4,10,16,22,28}: the two QPs that share a 32-lane LDS phase then read the stride-13 rows of K_t
    // from disjoint banks; even, so that 16-byte alignment survives.  H = 10: 2544 doubles = 20,352 B per QP -> eight QPs
    // per CU (160 KiB): four workgroups of two rows.
    static constexpr int stride_for(int raw) {
        for (int s = raw;; ++s) {
            const int m = s % 32;
            if (s % 2 == 0 && (m == 4 || m == 10 || m == 16 || m == 22 || m == 28)) return s;
        }
    }
    static constexpr int ROW_STRIDE = stride_for(RAW);
};

// LDS image of the set-up kernel (formation + Ruiz only: no factor): 348 doubles per QP at H = 10; GEN (the general path's own set-up kernel): + the per-step
// table B~w_t, 600 doubles = 4.8 KB per QP at H = 10 (the per-step bounds are rebuilt from the contact bits by the ADMM kernel)
template <int H, bool GEN = false>
struct LayoutSetup {
    static constexpr int KSTR = 13, K_SZ = 12 * KSTR, S_SZ = 78, SLOT = K_SZ + S_SZ, FAC = 0, GCOL = 12;  // unused by the set-up code paths
    static constexpr int TBL = 0;
    static constexpr int DL = 36;
    static constexpr int FT = DL + 12 * H;                   // GEN: [t][12] the foot positions of step t (see Layout)
    static constexpr int GQT = 0;                            // (the general path's set-up kernel has no update-path instantiation: never read)
    static constexpr int ML = FT + (GEN ? 12 * H : 0);       // GEN: [t][12] the column maxima of a Ruiz pass, written by the row that owns step t
    static constexpr int E0X = ML;                           // GEN: [t][16 lanes] the E hand-over behind the last pass (the maxima are dead by then)
    static constexpr int COOP = 0, TAB = 0;           // never used by the set-up kernels (the fast path's: one row per QP; the kernels stage the table behind the rows' images themselves)
    static constexpr int CUV = 0;                            // (not written by a set-up-only solver)
    static constexpr int BL = ML + (GEN ? 16 * H : 0);
    static constexpr int ZROW = 6;
    static constexpr int CG = BL + 84;
    static constexpr int BW = 0;                             // (no table of the omega rows of B~_t: the general path's set-up kernel writes them into the record and recomputes them, RowSolver::setup)
    static constexpr int RAW = CG + 12 * H;                  // GEN: 120 + 52 H doubles per QP (1160 = 9.3 KB at H = 20)
    static constexpr int ROW_STRIDE = RAW + (RAW % 2);
};

// warm_start = 2 -- the reference's UPDATE path on ticks >= 2 (S/A1RobotControl.cpp:533-538: updateHessianMatrix / updateGradient / update*Bound on the persistent
// OsqpEigen workspace, then solve() with warm start).  OSQP 0.6 then (osqp_update_P) re-equilibrates from D = E = c = 1 with the PREVIOUS tick's gradient still in
// the workspace, (osqp_update_lin_cost / _bounds) scales the new q, l, u with the new D, E, c, and (osqp_solve, warm) starts from the previous solve's SCALED iterates
// as they are.  Per problem the engine therefore carries, next to the unscaled (x, y, rho) of warm_start = 1: the previous scalings, the previous unscaled gradient
// and the previous unscaled z = Pi(w).  Per-lane fields are [t][12 lanes] like the hand-off record; C = 0 marks "no previous tick".
template <int H>
struct Carry {
    static constexpr int C = 0, D = 1, E0 = D + 12 * H, E1 = E0 + 12 * H, G = E1 + 12 * H, Z0 = G + 12 * H, Z1 = Z0 + 12 * H;
    // SIG [12 lanes]: which entries of my rows of U and V were non-zero -- the sparsity pattern of the previous tick's Hessian as osqp-eigen's updateHessianMatrix
    // compares it (the reference's hessian is dense.sparseView(), S/ConvexMpc.cpp:211: exact zeros are not stored); see RowSolver::setup, "pattern change"
    static constexpr int SIG = Z1 + 12 * H;
    static constexpr int STRIDE = SIG + 12 + 1;  // (even)
};

// prepared state handed from the set-up kernel to the ADMM kernel: [field][12 active lanes] doubles per QP (pad lanes hold nothing).
// XH (the warm-start x) comes last and is only written / read when the solve is warm-started.
// Round 4: rr1 = E1^2 rho is not stored.  On the fx / fy lanes the Ruiz passes update E0 and E1 by the same operations on the same operands (the rows
// [f + mu fz] and [f - mu fz] have the same absolute entries: RowSolver::setup, ruiz_step), so E1 == E0 bit for bit, those lanes' first row is never an equality
// (eq: fz lanes only), hence rr1 == rr0 there; fz and pad lanes have no second row (E1 = 0: rr1 = 0).  The contact bits, the equality bits and the flags share
// one word (PK = cmask + 2^20 eqmask + 2^40 flags: integers below 2^43, exact in a double).  54 -> 41 fields at H = 10.
template <int H>
struct Prep {
    static_assert(H <= 20, "PK packs one bit per horizon step into 20-bit fields");
    static constexpr int RR0 = 0, DI2 = H, CG = 2 * H, BT = 3 * H;  // per-lane fields
    static constexpr int CSC = BT + 6, CY = CSC + 1, SY = CY + 1, RHO = SY + 1, PK = RHO + 1;
    static constexpr int XH = PK + 1;
    // update path only (FLAGS bit 2; H > 1): the first iteration's y-hat of my two rows and its c g - A'[(2 - alpha) rr delta] (RowSolver::setup, "update path")
    static constexpr int YW0 = XH + H, YW1 = YW0 + H, CGE = YW1 + H;
    static constexpr int FIELDS = XH + H + (H > 1 ? 3 * H : 0);
    static constexpr int STRIDE = FIELDS * 12;  // doubles per QP: 3.9 KB cold / 4.9 KB warm at H = 10 are written (round 3: 5.2 / 6.1)
    // general path, split pipeline: + my column of the omega rows of B~_t for every step, [t][3] (the record of a QP is then STRIDE_GEN doubles)
    static constexpr int BWF = FIELDS;
    static constexpr int STRIDE_GEN = (FIELDS + 3 * H) * 12;
};

// =================================================================================================
// One QP, executed by the 16 lanes of a DPP row.  `tab` = [s][t][2] = (alpha_st/beta_st, beta_st).
//
// MODE = kModeBalance (H = 1) is the 12-variable balance QP of compute_grf (S/A1RobotControl.cpp:377-444):
// P = R I + M' Q M, q = -M' Q b with M = [I; Rz' skew(r_i)] is the H = 1 member of the same family
// (alpha_00 = 0, beta_00 = 1, B~ := M with the torque rows first, dt := 0); its friction rows are the MPC
// rows with two signs flipped, which leaves every ADMM iterate of x unchanged.
//
// The object lives in registers (every array index is a compile-time constant after unrolling).  Two ways to drive it:
//   fused   setup(io); solve(); write_outputs(io);                               one kernel, small batches / latency path
//   split   K1: setup(io); save_prepared(p)        K2: load_prepared(p, io); { advance(); } ...; write_outputs(io)
//           K2 rows pull the next prepared QP as soon as theirs has converged (advance() = one checkpoint-aligned segment).
// =================================================================================================
// GEN = true is the general path of the reference's INTERFACE (S/ConvexMpc.h:74 B_mat_d_list, S/test/test_mpc.cpp:106-122): per-step
// foot positions (a different B_d at every horizon step) and per-step contact schedules.  With step-dependent B~_t the Hessian block
// (s,t) is still alpha_st U_st + beta_st V_st (A_c^2 B = 0 holds for every B_t), but U_st, V_st now depend on both steps: they are
// evaluated on the fly from the per-step tables (6 FMAs per entry instead of 1), the Riccati sweeps read B~_t and the bounds per step
// from LDS.  Same iterates as the oracle's strided formation; slower (bigger LDS image, more reads) and only in the fused kernel.
// TWIN = true (persistent ADMM kernel only): this row runs as one of a main / twin pair of DPP rows on the SAME QP (row_is_twin(),
// a1mpc_rowops.hpp).  The pair shares the control flow (every decision is taken on values both rows hold) and the LDS image; it splits the
// per-lane ADMM state by horizon step (main row: even steps, twin: odd steps) and the two products of a backward-sweep step
// (admm_iteration_twin); everything sequential over the steps is computed redundantly by both rows.
// UNI = true (persistent ADMM kernel, H >= 16): the caller guarantees contacts broadcast over the horizon (contact_stride = 0, what the reference's controller
// does, S/ConvexMpc.cpp:228-245), so ONE pair of bounds serves every slot and the per-slot pairs (2 HS doubles per lane) leave the register file -- at H = 16
// the hot loop loses its 8 scratch reloads and 14 of 72 AGPR moves per iteration (8192 x h16 first solve 3.93 -> 3.73 ms).  Same values, same bits.
// CLK = true (profiling instantiations: a1mpc_set_profiling): shader-clock stamps around the factor passes, the iteration segments and the
// residual checks of a QP -- outside the hot loop; the numbers behind a1mpc_last_stage_cycles (the reference's t1..t6 stopwatches, S/A1RobotControl.cpp:491-553).
// Round 5: the fused and the latency kernel (what every warm-started closed-loop tick and every batch-1 tick runs) have such an instantiation too, with stamps behind
// the formation and the Ruiz passes of the set-up as well (ckF, ckR; kTickStages below)
// QUAD = true (every kernel that holds ONE QP per wavefront at a horizon that is a multiple of 4 -- h = 16 / 20: persistent rows, fused and latency kernels, fast and general
// path): the four rows of the wavefront work on one QP -- rows 0 / 1 in the main role,
// rows 2 / 3 as twins, rows 1 / 3 bit-identical copies of rows 0 / 2 through everything sequential -- and the per-lane state is split four ways: slot k = step 4k + own,
// own = 0 / 1 / 2 / 3 on rows 0 / 2 / 1 / 3.  The chains, their operands and their order are the pair's: same bits; the element-wise part of an iteration, the residual
// norms and the state's registers are halved again.
template <int H, int MODE = kModeMpc, bool SETUP_ONLY = false, bool GEN = false, bool TWIN = false, bool UNI = false, bool CLK = false, bool QUAD = false>
struct RowSolver {
    static_assert(!QUAD || (TWIN && H % 4 == 0), "quads of rows: a twin pair doubled, horizon a multiple of 4");
    static_assert(!UNI || (!GEN && MODE == kModeMpc), "uniform bounds: the fast path with broadcast contacts");
    static_assert(!GEN || (MODE == kModeMpc && H > 1), "the general path is an MPC solve");
    static_assert(!TWIN || (MODE == kModeMpc && !SETUP_ONLY && H > 1), "twin rows: the iterations of an MPC solve");
    using L = std::conditional_t<SETUP_ONLY, LayoutSetup<H, GEN>, Layout<H, GEN>>;
    using PR = Prep<H>;
    const DeviceParams& P;
    const double* __restrict__ tab;
    double* __restrict__ lds;
    // lane identity
    int ln, quad, comp, ci, krow;
    int mofs[12];     // TWIN: element b of my backward-sweep read inside a step's slot -- K_t[b][ci] on a main row, entry (ci, b) of the packed S_t^-1 on a twin
    bool act, wl;
    bool twin, wr;    // TWIN: second row of the pair / this lane stores what both rows hold (act && !twin; QUAD: row 0 only)
    int tw;           // 0 / 1: main / twin role
    int own;          // my steps are NS k + own  (TWIN: own = tw; QUAD: own = tw + 2 (row & 1))
    double hm, gAm, gBm, gCm, gVm;  // TWIN: 1 (main) / 0 (twin) and the costate-seed multipliers masked by it (sweep_back_rhs_twin)
    const double* brow;
    double dt, mu;
    // per-lane constants of the problem
    double Bt[6];  // my column of B~ (force layout); zero on pad lanes
    static constexpr bool kBrowInRegs = ((H <= 10) || TWIN) && !GEN;  // beyond that the per-lane ADMM state alone (6H doubles; 3H for a twin pair) overflows the register file
    static constexpr int NS = TWIN ? (QUAD ? 4 : 2) : 1;  // rows that split the per-lane state of one QP
    static constexpr int HS = H / NS;  // slots of the per-lane state: slot k = step NS k + own
    double Brw[12];  // my row of B~ (state layout; zeros on lanes without a wrench state): step-invariant, so it stays in registers --
                     // an LDS read costs the wave ~12 issue cycles whatever its width (tools/ubench/issue_cost_ubench.hip)
    double cy, sy, fA, fB, fC, fP, gA, gB, gC, gV, q2s, r2a;
    double csc, cinv, qd, lo_u, hi_u, lb0, ub0;   // (lo_u .. ub0: step 0's)
    double lbk[HS], ubk[HS];  // bounds of my slot-0 row per slot: contacts may follow a per-step schedule (contact_stride = 4)
    unsigned eqmask;  // bit t: my slot-0 row at step t is an equality row (swing leg: l = u = 0; auxil.c set_rho_vec)
    unsigned cmask;   // bit t: my leg is in contact at step t
    int r0, r1;       // reference row numbers of my two rows inside a (step, leg) block
    // hot state (see setup()): 6 doubles per horizon step and lane
    double xh[HS], wh0[HS], wh1[HS], rr0[HS], rr1[HS], dI2[HS];
    double rho;
    bool warm, first_special;
    bool upd;         // warm_start = 2 and a previous tick in the carry: this solve follows the reference's update path
    double epsv[HS];  // set-up, update path: A'[(2 - alpha) rr delta] of my lane (the first iteration's correction of c g)
    int coop_id = 0, coop_n = 1;  // set-up only: row coop_id of coop_n rows of the wave that work on the SAME QP (batch-1 latency path), sharing its LDS image
    bool careful;  // rho is small: c P x + c g is carried through the x-update identity (G in LDS) instead of re-evaluated at the checkpoints
    // bookkeeping
    int iter, nfact;
    int32_t status;
    bool fac_ok, need_factor, done;
#ifdef A1X_CLK
    long long clkB = 0, clkF = 0, clkT = 0, clkU = 0, clkX = 0;
#endif
    long long pfX = 0, pfT = 0, pfU = 0;  // CLK: shader-clock cycles this QP spent in factor passes / iteration segments / residual checks (wave-mates' stalls included)
    long long ckF = 0, ckR = 0;           // CLK, set-up: shader clock behind the formation (inputs, B~, gradient, U / V) and behind the Ruiz passes (a1mpc_last_tick_stage_cycles)
    int pred_cost = 0;  // set-up's guess of this QP's cost (queue order of a first solve, see predict_cost)
    double* gen_rec = nullptr;   // general path, set-up kernel: this QP's hand-off record -- the per-step columns of B~w_t go straight into it (its LDS image has no table for them)
    struct Info {
        double pri_res, dua_res, nEz, nEAx, nDq, nDAty, nDPx;  // unscaled
        double s_pri, s_dua, s_z, s_Ax, s_q, s_Aty, s_Px;      // scaled (rho estimate)
    } info;

    A1_DEV RowSolver(const DeviceParams& P_, const double* tab_, double* lds_) : P(P_), tab(tab_), lds(lds_) {
        ln = row_lane();
        quad = ln >> 2; comp = ln & 3;
        act = comp < 3;
        twin = TWIN && row_is_twin();
        tw = twin ? 1 : 0;
        own = tw;
        wr = act && !twin;
        if constexpr (QUAD) { own = tw + 2 * row_sub(); wr = wr && row_sub() == 0; }
        hm = twin ? 0.0 : 1.0;
        ci = act ? 3 * quad + comp : 0;  // compact index (safe 0 on pad lanes)
        krow = ci * L::KSTR;
        // (opaque: twelve loop-invariant address registers with the horizon step in the read's immediate -- left to fold the selects the compiler re-derives
        // one of them per step, ten registers for one, and the hot loop pays for them in AGPR moves)
        static_for<12>([&](auto B) { constexpr int b = A1_CV(B); mofs[b] = TWIN ? row_opaque(twin ? L::K_SZ + sinv_slot<b>(ci) : b * L::KSTR + ci) : 0; });
        wl = act && quad >= 2;           // wrench lanes: state rows 6..11 = quads 2, 3
        brow = lds + L::BL + (wl ? ci - 6 : L::ZROW) * 12;
        dt = P.dt; mu = P.mu;
        q2s = act ? P.q2[ci] : 0.0;      // state-lane weight 2 q_i
        r2a = act ? P.r2[ci] : 0.0;      // force-lane weight 2 r_a
        r0 = comp == 0 ? 0 : (comp == 1 ? 2 : 4); r1 = comp == 0 ? 1 : 3;
        iter = 0; nfact = 0; status = A1MPC_UNSOLVED; fac_ok = true; need_factor = true; done = false;
        warm = false; first_special = false; eqmask = 0; careful = false; upd = false;
    }

    // the one lane that speaks for this QP (scalar outputs, the work queue)
    A1_DEV bool lead() const { return ln == 0 && own == 0; }
    // orders LDS traffic between the lanes that work on this QP: my row, or both rows of a twin pair
    A1_DEV void sync() const {
        if constexpr (TWIN) pair_sync();
        else row_sync();
    }
    // set-up: orders LDS traffic between the lanes that work on this QP's set-up -- my row, or every row that shares it (coop_n > 1: the rows of
    // the wavefront run in lock-step; the CPU test double has to be told)
    A1_DEV void set_sync() const {
        if (coop_n > 1) coop_sync();
        else row_sync();
    }
    // A_d = I + dt*A_c and its transpose as row operators on a state-layout vector (T = A_c(0:3,6:9), S/ConvexMpc.cpp:123-125)
    A1_DEV void set_rotation(double c, double s) {
        cy = c; sy = s;
        fA = ln == 0 ? dt * cy : (ln == 1 ? -dt * sy : 0.0);
        fB = ln == 0 ? dt * sy : (ln == 1 ? dt * cy : 0.0);
        fC = ln == 2 ? dt : 0.0;
        fP = (quad == 1 && act) ? dt : 0.0;
        gA = ln == 8 ? dt * cy : (ln == 9 ? dt * sy : 0.0);
        gB = ln == 8 ? -dt * sy : (ln == 9 ? dt * cy : 0.0);
        gC = ln == 10 ? dt : 0.0;
        gV = (quad == 3 && act) ? dt : 0.0;
        if constexpr (TWIN) { gAm = hm * gA; gBm = hm * gB; gCm = hm * gC; gVm = hm * gV; }
    }
    // NOTE: opA, opAT, BtT, Bu read their argument through DPP: it must have passed row_dpp_ready()
    A1_DEV double opA(double s) const {  // (A_d s): rpy += dt*T*omega, pos += dt*vel
        double a0 = s, a1 = fP * row_ror<8>(s);
        fma_bcast<8>(a0, fA, s); fma_bcast<9>(a1, fB, s); fma_bcast<10>(a0, fC, s);
        return a0 + a1;
    }
    A1_DEV double opAT(double p) const {  // (A_d' p): omega += dt*T'*rpy-part, vel += dt*pos-part
        double a0 = p, a1 = gV * row_ror<8>(p);
        fma_bcast<0>(a0, gA, p); fma_bcast<1>(a1, gB, p); fma_bcast<2>(a0, gC, p);
        return a0 + a1;
    }
    A1_DEV double BtT(double lam) const { return dot_bc<6>(Bt, lam); }  // (B~' lambda_{omega,v}) in force layout
    A1_DEV double Bu(double u) const {                                  // (B~ u) on the wrench lanes of a state-layout vector
        double Br[12];
#pragma unroll
        for (int b = 0; b < 12; ++b) Br[b] = brow[b];
        return dot_bc<0>(Br, u);  // lanes without a wrench state read the zero row
    }
    // ---- per-step accessors (GEN: tables in LDS; otherwise the step-invariant registers)
    template <int K>
    A1_DEV double lbs(int) const { return lbk[UNI ? 0 : K]; }  // slot K = horizon step t
    template <int K>
    A1_DEV double ubs(int) const { return ubk[UNI ? 0 : K]; }
    A1_DEV void slot_bounds(int k, int t) {  // from the contact bit of step t
        const double cf = (cmask >> t) & 1u ? 1.0 : 0.0;
        lbk[k] = comp == 2 ? P.fz_min * cf : 0.0;
        ubk[k] = comp == 2 ? P.fz_max * cf : kInfty;
    }
    A1_DEV void Bt_at(int t, double (&o)[6]) const {  // my column of B~_t (force layout)
#pragma unroll
        for (int k = 0; k < 6; ++k) o[k] = Bt[k];
        if constexpr (GEN) {
#pragma unroll
            for (int k = 0; k < 3; ++k) o[k] = act ? lds[L::BW + (t * 3 + k) * 12 + ci] : 0.0;
        }
    }
    A1_DEV const double* brow_at(int t) const {   // my row of B~_t (state layout): omega lanes read the step's table
        if constexpr (GEN) return (wl && quad == 2) ? lds + L::BW + (t * 3 + (ci - 6)) * 12 : brow;
        else return brow;
    }
    // my column (leg, comp) of the omega rows of B~ for a foot at r: dt * Iw^-1 * skew(r)[:, comp]  (S/ConvexMpc.cpp:138,151; Ii = Iw^-1 row-major).  No contraction left to the
    // compiler: the general path evaluates this in several places and every copy must round alike
    A1_DEV void bw_from_foot(const double (&Ii)[9], double rx, double ry, double rz, double (&o)[3]) const {
#pragma clang fp contract(off)
        const double k0 = comp == 0 ? 0.0 : (comp == 1 ? -rz : ry);   // column `comp` of skew(r) (S/utils/Utils.cpp:35-41)
        const double k1 = comp == 0 ? rz : (comp == 1 ? 0.0 : -rx);
        const double k2 = comp == 0 ? -ry : (comp == 1 ? rx : 0.0);
#pragma unroll
        for (int k = 0; k < 3; ++k) o[k] = (Ii[k * 3 + 0] * k0 + Ii[k * 3 + 1] * k1 + Ii[k * 3 + 2] * k2) * dt;
    }
    A1_DEV void bounds_from_contact(double cf) {
        lo_u = P.fz_min * cf; hi_u = P.fz_max * cf;
        // physical bounds of my two rows: slot 0 = [fx+mu fz >= 0 | fy+mu fz >= 0 | fz in [lo,hi]], slot 1 = [.. <= 0]
        lb0 = comp == 2 ? lo_u : 0.0; ub0 = comp == 2 ? hi_u : kInfty;
    }

    // ================================================================================ set-up: formation + Ruiz + hot state
    // UPD = false: an instantiation without the update path (warm_start = 2 never reaches it) -- the split pipeline's set-up kernel of every other mode keeps the
    // code it had before the update path existed (with it in: +9 % on that kernel, measured)
    // COOP (general path): the number of rows of the wavefront that share this set-up, as a compile-time constant (== coop_n; 0 = not given: one row).  The rows of a shared
    // general-path set-up split the ROWS of the Hessian -- horizon step s belongs to row s mod COOP -- so a lane carries the step-s factors of H / COOP steps only (round 6)
    template <bool UPD = false, int COOP = 0>
    A1_DEV void setup(const ProblemIO& io) {
        double Rm[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) Rm[i] = io.R[i];
        double c_ = 1.0, s_ = 0.0;
        if constexpr (MODE == kModeMpc) {
            double yaw = io.tick ? io.tick[2] : io.x0[2];
            if constexpr (GEN) { if (io.yaw_A) yaw = *io.yaw_A; }
            c_ = cos(yaw); s_ = sin(yaw);  // S/ConvexMpc.cpp:115-116
        }
        set_rotation(c_, s_);
        [[maybe_unused]] double Ii[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // inverse world inertia (GEN re-uses it for every step's feet)
        if constexpr (MODE == kModeBalance) {
            static_assert(MODE != kModeBalance || H == 1, "the balance QP is the H = 1 case");
            const double rx = io.foot[3 * quad + 0], ry = io.foot[3 * quad + 1], rz = io.foot[3 * quad + 2];
            const double k0 = comp == 0 ? 0.0 : (comp == 1 ? -rz : ry);
            const double k1 = comp == 0 ? rz : (comp == 1 ? 0.0 : -rx);
            const double k2 = comp == 0 ? -ry : (comp == 1 ? rx : 0.0);
#pragma unroll
            for (int k = 0; k < 3; ++k) {  // inertia_inv = [I; Rz' skew(r)] (S/A1RobotControl.cpp:394-399)
                Bt[k] = act ? io.Rz[0 * 3 + k] * k0 + io.Rz[1 * 3 + k] * k1 + io.Rz[2 * 3 + k] * k2 : 0.0;
                Bt[3 + k] = (act && comp == k) ? 1.0 : 0.0;
            }
        } else {
            // I_world = R I_b R', inverse by cofactors (S/ConvexMpc.cpp:136-141)
            double t9[9], Iw[9];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    double s = 0;
#pragma unroll
                    for (int k = 0; k < 3; ++k) s += Rm[i * 3 + k] * P.inertia[k * 3 + j];
                    t9[i * 3 + j] = s;
                }
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    double s = 0;
#pragma unroll
                    for (int k = 0; k < 3; ++k) s += t9[i * 3 + k] * Rm[j * 3 + k];
                    Iw[i * 3 + j] = s;
                }
            const double c00 = Iw[4] * Iw[8] - Iw[5] * Iw[7], c01 = Iw[5] * Iw[6] - Iw[3] * Iw[8],
                         c02 = Iw[3] * Iw[7] - Iw[4] * Iw[6];
            const double idet = 1.0 / (Iw[0] * c00 + Iw[1] * c01 + Iw[2] * c02);
            Ii[0] = c00 * idet; Ii[1] = (Iw[2] * Iw[7] - Iw[1] * Iw[8]) * idet; Ii[2] = (Iw[1] * Iw[5] - Iw[2] * Iw[4]) * idet;
            Ii[3] = c01 * idet; Ii[4] = (Iw[0] * Iw[8] - Iw[2] * Iw[6]) * idet; Ii[5] = (Iw[2] * Iw[3] - Iw[0] * Iw[5]) * idet;
            Ii[6] = c02 * idet; Ii[7] = (Iw[1] * Iw[6] - Iw[0] * Iw[7]) * idet; Ii[8] = (Iw[0] * Iw[4] - Iw[1] * Iw[3]) * idet;
            const double rx = io.foot[3 * quad + 0], ry = io.foot[3 * quad + 1], rz = io.foot[3 * quad + 2];
            // column `comp` of skew(r) (S/utils/Utils.cpp:35-41)
            const double k0 = comp == 0 ? 0.0 : (comp == 1 ? -rz : ry);
            const double k1 = comp == 0 ? rz : (comp == 1 ? 0.0 : -rx);
            const double k2 = comp == 0 ? -ry : (comp == 1 ? rx : 0.0);
            const double invm = 1.0 / P.mass;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                Bt[k] = act ? (Ii[k * 3 + 0] * k0 + Ii[k * 3 + 1] * k1 + Ii[k * 3 + 2] * k2) * dt : 0.0;  // :138,:151
                Bt[3 + k] = (act && comp == k) ? invm * dt : 0.0;                                           // :139,:151
            }
        }
        // my column of T*B~_omega
        double TB[3];
        TB[0] = cy * Bt[0] + sy * Bt[1];
        TB[1] = -sy * Bt[0] + cy * Bt[1];
        TB[2] = Bt[2];
        set_sync();
        if (act) {
#pragma unroll
            for (int k = 0; k < 6; ++k) lds[L::BL + k * 12 + ci] = Bt[k];
            lds[L::BL + L::ZROW * 12 + ci] = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) lds[L::TBL + k * 12 + ci] = TB[k];
        }
        if constexpr (GEN) {
            // per-step feet: the omega rows of B~_t = dt * Iw^-1 * skew(r_t) (S/ConvexMpc.cpp:138,151 with the step's foot_pos).  T * them (the rpy rows of B_d) is recomputed
            // from this table where it is needed (my own column of a few steps: step_factors below) -- until round 6 it had a table (fused kernels) or 3 H registers (set-up kernel) of its own
            // The set-up kernel of the general path's split pipeline (SETUP_ONLY) keeps no B~w table in LDS: the column goes straight into the hand-off record (gen_rec) and is
            // recomputed from the foot table where the set-up needs it again (bw_at: the same expression, the same bits) -- 36 H doubles less per QP, eight instead of seven
            // one-QP workgroups per CU at H = 20
            // Every step's foot position is loaded before the first column is stored (round 6, last session): written step by step, the set-up kernel's loop was load -> wait ->
            // compute -> three stores to the hand-off record -> wait for the stores (this part counts loads and stores in ONE vmcnt, and the compiler must assume that the record
            // and the feet overlap) -> next load: 2 H exposures to the memory latency per QP, a fifth of the kernel's wave cycles parked at s_waitcnt
            // (profiles/r06_general_setup_counters.txt).
            // (In chunks of at most eight steps: at H = 16 the 96 registers of all the feet pushed a spill of the set-up kernel into its Ruiz loops.)
            constexpr int CH = H <= 12 ? H : 8, NCH = (H + CH - 1) / CH;
            static_for<NCH>([&](auto C_) {
                constexpr int t0 = A1_CV(C_) * CH, tn = (t0 + CH < H ? t0 + CH : H) - t0;
                double rf[3 * CH];
                static_for<tn>([&](auto T) {
                    constexpr int t = t0 + A1_CV(T);
                    const double* fp = io.foot + static_cast<int64_t>(t) * io.foot_stride;
#pragma unroll
                    for (int k = 0; k < 3; ++k) rf[3 * A1_CV(T) + k] = fp[3 * quad + k];
                });
                static_for<tn>([&](auto T) {
                    constexpr int t = t0 + A1_CV(T);
                    const double rx = rf[3 * A1_CV(T) + 0], ry = rf[3 * A1_CV(T) + 1], rz = rf[3 * A1_CV(T) + 2];
                    double bw[3];
                    bw_from_foot(Ii, rx, ry, rz, bw);
                    if (act) {
                        if constexpr (SETUP_ONLY) {
                            if (gen_rec != nullptr && coop_id == 0) {
#pragma unroll
                                for (int k = 0; k < 3; ++k) gen_rec[(PR::BWF + 3 * t + k) * 12 + ci] = bw[k];
                            }
                        } else {
#pragma unroll
                            for (int k = 0; k < 3; ++k) lds[L::BW + (t * 3 + k) * 12 + ci] = bw[k];
                        }
                        lds[L::FT + t * 12 + ci] = comp == 0 ? rx : (comp == 1 ? ry : rz);
                    }
                });
            });
        }
        set_sync();
        // my column of the omega rows of B~_t during the set-up (general path): from the LDS table, or -- set-up kernel -- again from the step's foot position
        [[maybe_unused]] auto bw_at = [&](int t, double (&o)[3]) {
            if constexpr (SETUP_ONLY) {
                const double* ft = lds + L::FT + t * 12 + 3 * quad;
                bw_from_foot(Ii, ft[0], ft[1], ft[2], o);
#pragma unroll
                for (int k = 0; k < 3; ++k) o[k] = act ? o[k] : 0.0;
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) o[k] = act ? lds[L::BW + (t * 3 + k) * 12 + ci] : 0.0;
            }
        };

        // ---------------------------------------------------------------- gradient g = B_qp' Q (A_qp x0 - x_ref)
        double g[H];
        if constexpr (MODE == kModeBalance) {
            // q = -M' Q b (S/A1RobotControl.cpp:406); wrench order here: torque (root_acc[3:6]), force (root_acc[0:3])
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                a += Bt[k] * P.q2[6 + k] * io.root_acc[3 + k];
                a += Bt[3 + k] * P.q2[9 + k] * io.root_acc[k];
            }
            g[0] = -a;
        } else {
            double w[H];
            double xs, grav, xr_base = 0.0, xr_slope = 0.0;
            if (io.tick) {
                // N1: the tick record [euler, pos, ang_vel, lin_vel | euler_d, lin_vel_d (body), ang_vel_d, pos_z_d]; x0 and x_ref exactly as
                // S/A1RobotControl.cpp:452-456 and :470-488 build them: x_ref_i = base + (slope * dt) * (i + 1)
                const double* k = io.tick;
                xs = act ? k[ci] : 0.0;
                grav = -9.8;
                const double vwx = Rm[0] * k[15] + Rm[1] * k[16] + Rm[2] * k[17], vwy = Rm[3] * k[15] + Rm[4] * k[16] + Rm[5] * k[17];
                if (ci == 0) xr_base = k[12];
                else if (ci == 1) xr_base = k[13];
                else if (ci == 2) { xr_base = k[2]; xr_slope = k[20]; }
                else if (ci == 3) { xr_base = k[3]; xr_slope = vwx; }
                else if (ci == 4) { xr_base = k[4]; xr_slope = vwy; }
                else if (ci == 5) xr_base = k[21];
                else if (ci <= 8) xr_base = k[18 + ci - 6];
                else if (ci == 9) xr_base = vwx;
                else if (ci == 10) xr_base = vwy;
                if (!act) { xr_base = 0.0; xr_slope = 0.0; }
                xr_slope *= dt;
            } else {
                xs = act ? io.x0[ci] : 0.0;
                grav = io.x0[12];
            }
            const double x_now = xs;
            double err0 = 0.0;  // first reference state minus the current state, my component
            static_for<H>([&](auto T) {
                xs = opA(row_dpp_ready(xs));
                if (ln == 14) xs += dt * grav;  // A_c(11,12) = 1 (S/ConvexMpc.cpp:129)
                const double xr = io.tick ? xr_base + xr_slope * double(A1_CV(T) + 1) : (act ? io.xref[T * 13 + ci] : 0.0);
                if constexpr (A1_CV(T) == 0) err0 = xr - x_now;
                w[T] = q2s * (xs - xr);
            });
            predict_cost(row_dpp_ready(err0), io);
            double lam = row_dpp_ready(0.0);
            static_for<H>([&](auto TT) {
                constexpr int t = H - 1 - A1_CV(TT);
                lam = row_dpp_ready(w[t] + opAT(lam));
                if constexpr (GEN) {
                    double Bq[6], bq3[3];
                    bw_at(t, bq3);
#pragma unroll
                    for (int k = 0; k < 3; ++k) { Bq[k] = bq3[k]; Bq[3 + k] = Bt[3 + k]; }
                    const double gt = dot_bc<6>(Bq, lam);
                    if (act) lds[L::CG + t * 12 + ci] = gt;   // (general path: the gradient waits in its LDS table -- raw here, c g behind the Ruiz passes -- not in H registers)
                }
                else g[t] = BtT(lam);
            });
        }

        // ---------------------------------------------------------------- U, V rows (P = alpha(x)U + beta(x)V + I(x)R)
        double U[12], V[12], Ud = 0.0, Vd = 0.0;
        static_for<12>([&](auto B) {
            double u = 0.0, v = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                u += P.q2[k] * TB[k] * lds[L::TBL + k * 12 + B] + P.q2[3 + k] * Bt[3 + k] * lds[L::BL + (3 + k) * 12 + B];
                v += P.q2[6 + k] * Bt[k] * lds[L::BL + k * 12 + B] + P.q2[9 + k] * Bt[3 + k] * lds[L::BL + (3 + k) * 12 + B];
            }
            U[B] = u * dt * dt;
            V[B] = v;
            if (act && ci == B) { Ud = U[B]; Vd = V[B]; }
        });
        // GEN: block (s,t) of B_qp'QB_qp is beta_st (gamma_st U_st + V_st) with U_st[a][b] = dt^2 (sum_c q_c TB_s[c][a] TB_t[c][b] + q_{3+k} (dt/m)^2 [k = comp_a = comp_b]),
        // V_st[a][b] = sum_c q_{6+c} B~w_s[c][a] B~w_t[c][b] + q_{9+k} (dt/m)^2 [..]: my row's step-s factors live in registers (for the steps s this row of the set-up owns), the
        // step-t tables in LDS
        [[maybe_unused]] double Ucs[3], Vcs[3], Ucd[3];
        [[maybe_unused]] const double dt2 = dt * dt;
        if constexpr (GEN) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                Ucs[k] = (act && comp == k) ? P.q2[3 + k] * Bt[3 + k] * Bt[3 + k] : 0.0;
                Vcs[k] = (act && comp == k) ? P.q2[9 + k] * Bt[3 + k] * Bt[3 + k] : 0.0;
                Ucd[k] = Ucs[k] * dt2;
            }
        }

        if constexpr (CLK) ckF = row_clock();
        // ---------------------------------------------------------------- Ruiz equilibration (osqp scaling.c scale_data)
        // update path (warm_start = 2, a previous tick in the carry): osqp_update_P re-equilibrates while the workspace still holds the PREVIOUS tick's
        // gradient -- the only place the gradient enters scale_data is the cost normalisation below
        using CR = Carry<H>;
        upd = false;
        if constexpr (UPD && MODE == kModeMpc && H > 1) upd = P.warm_start == 2 && io.carry != nullptr && io.warm_x != nullptr && io.warm_y != nullptr && io.carry[CR::C] > 0.0;
        // Pattern change.  osqp-eigen's updateHessianMatrix takes osqp_update_P only while the upper-triangular triplets of hessian.sparseView() keep their pattern; when
        // exact zeros of the reference's dense B_qp'QB_qp appear or vanish it reads the workspace iterates, clears the solver, initialises it again (fresh scaling with
        // the CURRENT data, rho back to settings.rho) and warm-starts it with those iterates through osqp_warm_start_x / _y -- which scale what they are given, and
        // what they are given are the previous solve's SCALED iterates (oracle/a1mpc_oracle.c osqp_solve_impl, `reinit`).  Block (s,t) of the Hessian is alpha_st U + beta_st V with
        // alpha_st = 0 exactly where max(s,t) = H - 1, so the pattern of P is a function of the zero patterns of U and V: my rows' 24 flags are the signature.
        [[maybe_unused]] bool reinit = false;
        [[maybe_unused]] double sig = 0.0;
        if constexpr (UPD && MODE == kModeMpc && H > 1) {
            if constexpr (GEN) {
                // General path (round 5): every block (s,t) has its own U_st, V_st; an entry of the reference's dense Hessian is structurally zero exactly when the omega /
                // rpy rows of B~w_s[:,a] and B~w_t[:,b] have no common non-zero (and the velocity / position rows do not match), so the pattern is a function of the zero
                // patterns of my columns of the per-step tables B~w_s and T B~w_s: 6 flags per step and lane, folded into one exactly representable number (< 2^50).
                // A change of these flags is a SUPERSET of the changes osqp-eigen sees (a flag can change under a zero weight without touching the Hessian's pattern);
                // for inputs in general position neither ever changes.
                unsigned long long hsh = 0;
                static_for<H>([&](auto S) {
                    unsigned v = 0;
                    double bwS[3];
                    bw_at(A1_CV(S), bwS);
                    const double tbS[3] = {fma(cy, bwS[0], sy * bwS[1]), fma(-sy, bwS[0], cy * bwS[1]), bwS[2]};   // my column of T B~w_s, as step_factors forms it
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        v |= (bwS[c] != 0.0 ? 1u : 0u) << c;
                        v |= (tbS[c] != 0.0 ? 1u : 0u) << (3 + c);
                    }
                    hsh = (hsh * 67ull + v + 1ull) & ((1ull << 50) - 1ull);
                });
                sig = act ? static_cast<double>(hsh) + kSigGeneralPath : 0.0;   // (+ 2^51: "written by the general path", exact in a double)
            } else {
                unsigned bits = 0;
                static_for<12>([&](auto B) { bits |= (U[B] != 0.0 ? 1u : 0u) << A1_CV(B); bits |= (V[B] != 0.0 ? 1u : 0u) << (12 + A1_CV(B)); });
                sig = act ? static_cast<double>(bits) : 0.0;
            }
            if (upd) {
                // The two paths encode the pattern differently (24 flags of U / V | a hash of the per-step tables' flags, tagged 2^51): signatures are only comparable when
                // the same path wrote both.  A handle that alternates between the paths (foot_stride / yaw_A toggled between ticks) keeps the UPDATE path across the switch --
                // everything else in the carry (previous scalings, gradient, z) means the same on both, and the reference's persistent solver takes osqp_update_P whenever
                // the dense Hessian keeps its pattern, which inputs in general position do on either path (until round 6 every switch took the pattern-change branch)
                const double prev = act ? io.carry[CR::SIG + ci] : 0.0;
                const bool same_path = (prev >= kSigGeneralPath) == (sig >= kSigGeneralPath);
                reinit = row_allmax((same_path && prev != sig) ? 1.0 : 0.0) > 0.0;
            }
        }
        [[maybe_unused]] double gq[(UPD && MODE == kModeMpc && H > 1 && !GEN) ? H : 1];   // the gradient the cost normalisation sees: the previous tick's on the update path
        if constexpr (UPD && MODE == kModeMpc && H > 1 && !GEN) {
#pragma unroll
            for (int t = 0; t < H; ++t) gq[t] = (upd && !reinit) ? (act ? io.carry[CR::G + t * 12 + ci] : 0.0) : g[t];
        }
        double D[H], E0[H], E1[H];
        csc = 1.0;
#pragma unroll
        for (int t = 0; t < H; ++t) { D[t] = 1.0; E0[t] = act ? 1.0 : 0.0; }
        if constexpr (GEN) {
            // ================================================================ general path: Ruiz passes with the ROWS of the Hessian split over the rows of the set-up (round 6)
            // Every block (s,t) of the implicit Hessian is evaluated in every sweep (no pruning bound holds on this path: see the measurements named at the old column loop,
            // profiles/r05_general_path_early_stop.txt), so a lane needs its step-s factors (cu, cv: 6 doubles per step) for every s it evaluates.  Until round 6 a row
            // evaluated EVERY s for the columns t it was given: 6 H doubles of factors + D, E, m for all H steps per lane -- 330-390 doubles at H = 16 / 20 in a 256-double
            // register file, 390-750 VGPRs spilled to scratch and 145-500 scratch instructions inside the sweep loops.  Now row j of the N rows that share
            // the set-up OWNS the horizon steps s = N k + j: it evaluates the block rows s it owns against every column t, keeps factors / D / E / m of those H / N steps
            // only, and nothing is combined across rows any more (a column maximum m_s has ONE producer).  What all rows need of each other goes through three small LDS tables per
            // pass: D (DL, read by every sweep anyway), m (ML) and -- once, behind the last pass -- E (E0X).  The gradient waits in LDS (CG / GQT) instead of H registers.
            constexpr int N = COOP > 0 ? COOP : 1;
            constexpr int HSN = (H + N - 1) / N;
            const int cid = N > 1 ? coop_id : 0;
            if (act) {   // (the raw gradient is in the CG table already: written by the adjoint sweep above, by every row of a shared set-up alike)
#pragma unroll
                for (int t = 0; t < H; ++t) {
                    if constexpr (UPD && H > 1) lds[L::GQT + t * 12 + ci] = (upd && !reinit) ? io.carry[CR::G + t * 12 + ci] : lds[L::CG + t * 12 + ci];
                    lds[L::DL + t * 12 + ci] = 1.0;    // (every row of a shared set-up writes the same words)
                }
            }
            constexpr int GSRC = (UPD && H > 1) ? L::GQT : L::CG;   // the gradient the cost normalisation sees (the previous tick's on the update path)
            if (P.scaling_iters > 0) {
                int sk[HSN]; bool vk[HSN];
                double zuS[HSN][3], zvS[HSN][3], dgS[HSN], Dk[HSN], E0k[HSN], mk[HSN];
                set_sync();   // the per-step table B~w_t is complete
                static_for<HSN>([&](auto K) {
#pragma clang fp contract(off)   // (no contraction left to the compiler: every instantiation of this block -- set-up kernel, fused, latency -- must round alike; the FMAs are written out)
                    constexpr int k = A1_CV(K);
                    const int s = N * k + cid;
                    vk[k] = s < H; sk[k] = vk[k] ? s : H - 1;
                    double bw[3];
                    bw_at(sk[k], bw);
                    const double tb[3] = {fma(cy, bw[0], sy * bw[1]), fma(-sy, bw[0], cy * bw[1]), bw[2]};   // my column of T B~w_s
                    double ud = 0.0, vd = 0.0, cu[3], cv[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        cu[c] = P.q2[c] * tb[c]; cv[c] = P.q2[6 + c] * bw[c];
                        ud += cu[c] * tb[c]; vd += cv[c] * bw[c];
                        if (comp == c) { ud += Ucs[c]; vd += Vcs[c]; }
                    }
                    // the true diagonal entry carries R: alpha_ss U_ss + beta_ss V_ss + R  (alpha_ss = sum_{i >= s} (i - s)^2, beta_ss = H - s: exact in a double)
                    const int nn = H - sk[k];
                    const double ad = static_cast<double>((nn - 1) * nn * (2 * nn - 1) / 6), bd = static_cast<double>(nn);
                    dgS[k] = fma(ad, ud * dt2, fma(bd, vd, r2a));
                    // Entry (s,a),(t,b) = beta_st (y . B~w_t[:,b] + k_{b%3})  with  y = gamma_st dt^2 T'(q T B~w_s[:,a]) + q_w B~w_s[:,a]  and the velocity-row constants k.
                    // Column b = (leg, c) of B~w_t is dt Iw^-1 (r_t,leg x e_c), so  y . B~w_t[:,b] = (z x r_t,leg)_c  with  z = dt Iw^-T y = gamma_st zu + zv: the three entries
                    // of a leg are ONE cross product with the step's foot position (2 FMAs per entry seeded with k; round 6 -- until then 3 FMAs per entry against the 3 x 12
                    // table B~w_t) and a sweep reads 12 instead of 36 table words per column block
                    const double yu[3] = {(cu[0] * cy - cu[1] * sy) * dt2, (cu[0] * sy + cu[1] * cy) * dt2, cu[2] * dt2};   // (T B~w_t = T (B~w_t): the yaw rotation folded into my step-s factors)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        zuS[k][j] = fma(Ii[6 + j], yu[2], fma(Ii[3 + j], yu[1], Ii[j] * yu[0])) * dt;
                        zvS[k][j] = fma(Ii[6 + j], cv[2], fma(Ii[3 + j], cv[1], Ii[j] * cv[0])) * dt;
                    }
                    Dk[k] = 1.0; E0k[k] = act ? 1.0 : 0.0;
                });
                // m[s] = max_{t,b} D_tb |P_(s,a),(t,b)| for my rows (s, a), s owned: one pass over my block rows of the implicit Hessian
                // entry (s,a),(t,b) = beta_st (y . B~w_t[:,b] + k_{b%3})  with  y = gamma_st dt^2 T'(q T B~w_s[:,a]) + q_w B~w_s[:,a]  and the velocity-row constants k: 3 FMAs
                // per entry and one table.  Tried and measured slower here: per-block pruning bounds (round 2) and the fast path's column loop with a rounding-safe bound
                // (round 5: the blocks of the last horizon steps never tie with their bound, no sweep stops)
                auto sweep = [&]() {
                    static_for<HSN>([&](auto K) { mk[K] = dgS[K] * Dk[K]; });
#pragma unroll 1
                    for (int t = 0; t < H; ++t) {
                        double Dt[12], rt[12];
#pragma unroll
                        for (int b = 0; b < 12; ++b) { Dt[b] = lds[L::DL + t * 12 + b]; rt[b] = lds[L::FT + t * 12 + b]; }
                        double gnx = tab[(sk[0] * H + t) * 2], bnx = tab[(sk[0] * H + t) * 2 + 1];   // (gamma_st, beta_st) of my next step: read one step ahead of its use
                        static_for<HSN>([&](auto K) {
                            constexpr int k = A1_CV(K);
                            const double gam = gnx, bet = bnx;
                            if constexpr (k + 1 < HSN) { gnx = tab[(sk[k + 1] * H + t) * 2]; bnx = tab[(sk[k + 1] * H + t) * 2 + 1]; }
                            const double z0 = fma(gam, zuS[k][0], zvS[k][0]), z1 = fma(gam, zuS[k][1], zvS[k][1]), z2 = fma(gam, zuS[k][2], zvS[k][2]);
                            const double k3[3] = {fma(gam, Ucd[0], Vcs[0]), fma(gam, Ucd[1], Vcs[1]), fma(gam, Ucd[2], Vcs[2])};   // (Ucd = dt^2 Ucs: the velocity-row constants of U)
                            double a0, a1;
                            static_for<4>([&](auto Lg) {   // (z x r)_0 = z1 r2 - z2 r1, (z x r)_1 = z2 r0 - z0 r2, (z x r)_2 = z0 r1 - z1 r0
                                constexpr int l = 3 * A1_CV(Lg);
                                const double e0 = fma(z1, rt[l + 2], fma(-z2, rt[l + 1], k3[0]));
                                const double e1 = fma(z2, rt[l + 0], fma(-z0, rt[l + 2], k3[1]));
                                const double e2 = fma(z0, rt[l + 1], fma(-z1, rt[l + 0], k3[2]));
                                if constexpr (A1_CV(Lg) == 0) {   // the first leg seeds the two running maxima (|x| rides on the multiply's source modifiers: no move, no max)
                                    a0 = fabs(e0) * Dt[l]; a1 = fabs(e1) * Dt[l + 1];
                                    a0 = max_abs_f64(a0, e2 * Dt[l + 2]);
                                } else if constexpr (A1_CV(Lg) % 2 == 0) max_abs3_f64(a0, a1, e0 * Dt[l], e1 * Dt[l + 1], e2 * Dt[l + 2]);   // a0 = max(a0, |x0|, |x2|), a1 = max(a1, |x1|)
                                else max_abs3_f64(a1, a0, e2 * Dt[l + 2], e1 * Dt[l + 1], e0 * Dt[l]);
                            });
                            mk[k] = max_f64(mk[k], bet * max_f64(a0, a1));
                        });
                    }
                };
                // one Ruiz step of an owned horizon step (scaling.c scale_data: column norms of [P A'; A 0] -> D, row norms -> E); E1 == E0 on the fx / fy lanes: see the fast path below
                auto ruiz_step = [&](double& Dt_, double& E0t, double mt) {
                    const double Dz = quad_perm<2, 2, 2, 2>(Dt_);
                    const double mE = E0t;
                    const double mEx = quad_perm<0, 0, 0, 0>(mE), mEy = quad_perm<1, 1, 1, 1>(mE);
                    const double colA = Dt_ * (comp == 2 ? fmax(mu * fmax(mEx, mEy), E0t) : mE);
                    const double colP = csc * Dt_ * mt;
                    const double dtmp = row_rsqrt(limit_scaling(fmax(colP, colA)));
                    const double rowf = comp == 2 ? Dt_ : fmax(Dt_, mu * Dz);
                    const double e0 = row_rsqrt(limit_scaling(E0t * rowf));
                    Dt_ *= dtmp; E0t *= e0;
                };
                set_sync();   // the D = 1 table is complete
                sweep();
#pragma unroll 1
                for (int pass = 0; pass < P.scaling_iters; ++pass) {
                    static_for<HSN>([&](auto K) { ruiz_step(Dk[K], E0k[K], mk[K]); });
                    set_sync();   // every row is done with the D table of the sweep before
                    if (act) { static_for<HSN>([&](auto K) { if (vk[K]) lds[L::DL + sk[K] * 12 + ci] = Dk[K]; }); }
                    set_sync();
                    sweep();
                    if (act) { static_for<HSN>([&](auto K) { if (vk[K]) lds[L::ML + sk[K] * 12 + ci] = mk[K]; }); }
                    set_sync();
                    // cost normalisation: the mean column norm of c P and the norm of c q over ALL steps, summed in the fixed order every row (and the oracle) uses
                    double sum = 0.0, nq = 0.0;
                    {
#pragma clang fp contract(off)
#pragma unroll
                        for (int t = 0; t < H; ++t) {
                            const double cd = csc * (act ? lds[L::DL + t * 12 + ci] : 1.0);
                            sum += cd * (act ? lds[L::ML + t * 12 + ci] : 0.0);
                            nq = fmax(nq, fabs(cd * (act ? lds[GSRC + t * 12 + ci] : 0.0)));
                        }
                    }
                    const double mean = row_allsum(sum) / double(12 * H);
                    nq = limit_scaling(row_allmax(nq));
                    const double ct = 1.0 / limit_scaling(fmax(mean, nq));
                    csc *= ct;
                }
                // the E of every step (each row iterated its own steps' only); the final D is in its table.  The hot-state loop below reads D_t, E_t and g_t of a step from these
                // tables when it gets there (h20: as arrays they are 160 registers held across it)
                set_sync();   // the last pass's reads of ML are done (E0X may alias it)
                static_for<HSN>([&](auto K) { if (vk[K]) lds[L::E0X + sk[K] * 16 + ln] = E0k[K]; });
            } else {
#pragma unroll
                for (int t = 0; t < H; ++t) lds[L::E0X + t * 16 + ln] = act ? 1.0 : 0.0;
            }
        }
        // E1 (the scaling of my second constraint row: fx / fy lanes only) is not iterated (round 5): the rows [f + mu fz] and [f - mu fz] have the same absolute entries, so
        // every Ruiz pass would update E1 by the same operations on the same operands as E0 -- E1 == E0 bit for bit on the fx / fy lanes, 0 elsewhere (see Prep<H>) -- it
        // is copied from E0 behind the passes.  One of three rsqrt chains per horizon step and pass, and a third of the tables a shared set-up hands around, go.
        if constexpr (!GEN) { if (P.scaling_iters > 0) {
            double m[H];
            // m[s] = max_{t,b} D_tb |P_(s,a),(t,b)|  for my rows (s, a): one pass over the implicit Hessian.
            // Exact pruning (see the column loop below): starting from the diagonal entry, 60-75 % of the blocks provably cannot raise a maximum.
            double Umax = 0.0, Vmax = 0.0;
#pragma unroll
            for (int b = 0; b < 12; ++b) { Umax = fmax(Umax, fabs(U[b])); Vmax = fmax(Vmax, fabs(V[b])); }
            // sweep: the caller has published the current D of every step in the DL table (and synchronised)
            auto sweep = [&](double(&mm)[H]) {
                // the largest D of the whole QP (pad lanes hold 1.0 and do not count)
                double Dall = 0.0;
#pragma unroll
                for (int t = 0; t < H; ++t) Dall = max_f64(Dall, D[t]);
                Dall = row_allmax(act ? Dall : 0.0);
                static_for<H>([&](auto S) {  // the true diagonal entry carries R
                    constexpr double ad = alpha_diag(A1_CV(S), H), bd = H - A1_CV(S);
                    mm[S] = (ad * Ud + bd * Vd + r2a) * D[S];
                });
                // Blocks are visited column by column (t ascending; a row of a shared set-up takes every coop_n-th) and only while some block still to
                // come could raise a maximum.  Inside a column every s is evaluated, branch-free, on the column's pre-scaled rows u_b = U_ab D_tb,
                // v_b = V_ab D_tb: entry = |fma(gamma_st, u_b, v_b)| beta_st, 2 instructions per entry.  The bound of block (s,t),
                //     B_st = fma(gamma_st, Umax Dall, Vmax Dall) beta_st        (Umax = max_b |U_ab|, Dall = max over the whole D table),
                // is the SAME operation sequence on operands that dominate the entry's one by one, so by the monotonicity of IEEE rounding it dominates
                // every rounded entry of the block exactly -- no safety margin, ties prune (they are structural: with the reference's weights the
                // columns of A set D, the same at every step).  For fixed s both gamma_st and beta_st never grow with t (gamma_beta_monotone<H>, asserted),
                // hence neither does B_st: once B_st <= m[s] for every s, no later column can change a bit of the result.  Random SRBD states stop after
                // 2-4 of 10 columns (5 of 20 at H = 20).
                static_assert(gamma_beta_monotone(H), "the column loop of the Ruiz sweep stops early on the strength of this");
                const double UDall = Umax * Dall, VDall = Vmax * Dall;
                {
                    int t = coop_id;
#pragma unroll 1
                    while (true) {
                        const int tc = t < H ? t : H - 1;  // (rows of a shared set-up run out of columns at different times; they re-visit a real block, which is harmless)
                        double gb[2 * H];
#pragma unroll
                        for (int s2 = 0; s2 < H; ++s2) { gb[2 * s2] = tab[(s2 * H + tc) * 2]; gb[2 * s2 + 1] = tab[(s2 * H + tc) * 2 + 1]; }
                        bool need = false;
                        static_for<H>([&](auto S) { need |= fma(gb[2 * A1_CV(S)], UDall, VDall) * gb[2 * A1_CV(S) + 1] > mm[S]; });
#ifdef A1X_FULL_SWEEP  // test build (tests/test_emu_parity.py): every column is visited; the early stop must not change a bit of the result
                        need = true;
#endif
                        if (!row_wave_any(need && t < H)) break;
                        double UD[12], VD[12];
#pragma unroll
                        for (int b = 0; b < 12; ++b) { const double db = lds[L::DL + tc * 12 + b]; UD[b] = U[b] * db; VD[b] = V[b] * db; }
                        static_for<H>([&](auto S) {
                            const double gam = gb[2 * A1_CV(S)], bet = gb[2 * A1_CV(S) + 1];
                            double a0 = fabs(fma(gam, UD[0], VD[0])), a1 = fabs(fma(gam, UD[1], VD[1]));
                            static_for<5>([&](auto J) {
                                a0 = max_abs_f64(a0, fma(gam, UD[2 * J + 2], VD[2 * J + 2]));
                                a1 = max_abs_f64(a1, fma(gam, UD[2 * J + 3], VD[2 * J + 3]));
                            });
                            mm[S] = max_f64(mm[S], bet * max_f64(a0, a1));
                        });
                        t += coop_n;
                    }
                }
                if (coop_n > 1) {  // each row of the wave visited every coop_n-th t: the column maxima are the maxima over the rows (exact, order-free)
                    static_for<H>([&](auto S) { lds[L::COOP + (coop_id * H + A1_CV(S)) * 16 + ln] = mm[S]; });
                    set_sync();
                    for (int r = 0; r < coop_n; ++r)
                        static_for<H>([&](auto S) { mm[S] = fmax(mm[S], lds[L::COOP + (r * H + A1_CV(S)) * 16 + ln]); });
                }
            };
            set_sync();
            if (act) {   // D = 1: every row of a shared set-up writes the same words
#pragma unroll
                for (int t = 0; t < H; ++t) lds[L::DL + t * 12 + ci] = D[t];
            }
            set_sync();
            sweep(m);
#pragma unroll 1
            for (int pass = 0; pass < P.scaling_iters; ++pass) {
                // one Ruiz step of horizon step T (scaling.c scale_data: column norms of [P A'; A 0] -> D, row norms -> E)
                auto ruiz_step = [&](double& Dt_, double& E0t, double mt) {
                    const double Dz = quad_perm<2, 2, 2, 2>(Dt_);
                    const double mE = E0t;   // = max(E0, E1): my two rows share their scaling (above)
                    const double mEx = quad_perm<0, 0, 0, 0>(mE), mEy = quad_perm<1, 1, 1, 1>(mE);
                    const double colA = Dt_ * (comp == 2 ? fmax(mu * fmax(mEx, mEy), E0t) : mE);
                    const double colP = csc * Dt_ * mt;
                    const double dtmp = row_rsqrt(limit_scaling(fmax(colP, colA)));
                    const double rowf = comp == 2 ? Dt_ : fmax(Dt_, mu * Dz);
                    const double e0 = row_rsqrt(limit_scaling(E0t * rowf));
                    Dt_ *= dtmp; E0t *= e0;
                };
                // Shared set-up (coop_n rows of the wave on one QP): row j OWNS the horizon steps t = coop_n k + j.  A step's E is only ever needed by that step's own
                // update, so it stays in its owner's registers through all passes (handed around once, behind the last pass); the new D of a step goes straight into the
                // DL table the next sweep reads anyway, and every row reads the whole table back (the cost normalisation sums over all steps in a fixed order).
                // Round 5: until then the four tables (D, E0, E1, m) went through LDS twice per pass -- 105 LDS operations per pass at H = 10 against 25 now.
                auto shared_pass = [&](auto NN) {
                    constexpr int N = A1_CV(NN);
                    static_for<(H + N - 1) / N>([&](auto K) {
                        constexpr int k = A1_CV(K);
                        double Dt_ = D[N * k], E0t = E0[N * k], mt = m[N * k];
                        static_for<N - 1>([&](auto J) {
                            constexpr int t = N * k + A1_CV(J) + 1;
                            if constexpr (t < H) { if (coop_id == A1_CV(J) + 1) { Dt_ = D[t]; E0t = E0[t]; mt = m[t]; } }
                        });
                        ruiz_step(Dt_, E0t, mt);
                        static_for<N>([&](auto J) {
                            constexpr int t = N * k + A1_CV(J);
                            if constexpr (t < H) { if (coop_id == A1_CV(J)) { E0[t] = E0t; if (act) lds[L::DL + t * 12 + ci] = Dt_; } }
                        });
                    });
                };
                set_sync();   // every row is done with the DL table of the sweep before
                if (coop_n == 1) {
                    static_for<H>([&](auto T) { ruiz_step(D[T], E0[T], m[T]); });
                    if (act) {
#pragma unroll
                        for (int t = 0; t < H; ++t) lds[L::DL + t * 12 + ci] = D[t];
                    }
                    set_sync();
                } else {
                    if (coop_n == 2) shared_pass(std::integral_constant<int, 2>{});
                    else shared_pass(std::integral_constant<int, 4>{});
                    set_sync();
#pragma unroll
                    for (int t = 0; t < H; ++t) D[t] = act ? lds[L::DL + t * 12 + ci] : 1.0;   // (pad lanes: D stays 1, as every pass leaves it there)
                }
                sweep(m);
                double sum = 0.0, nq = 0.0;
#pragma unroll
                for (int t = 0; t < H; ++t) {
                    sum += csc * D[t] * m[t];
                    if constexpr (UPD && MODE == kModeMpc && H > 1) nq = fmax(nq, fabs(csc * D[t] * gq[t]));
                    else nq = fmax(nq, fabs(csc * D[t] * g[t]));
                }
                const double mean = row_allsum(sum) / double(12 * H);
                nq = limit_scaling(row_allmax(nq));
                const double ct = 1.0 / limit_scaling(fmax(mean, nq));
                csc *= ct;
            }
        } }
        if constexpr (!GEN) { if (P.scaling_iters > 0 && coop_n > 1) {   // the E of the other rows' horizon steps (each row wrote its own steps' only)
            set_sync();
            static_for<H>([&](auto T) { if ((A1_CV(T) & (coop_n - 1)) == coop_id) lds[L::E0X + A1_CV(T) * 16 + ln] = E0[T]; });   // (coop_n: 2 or 4)
            set_sync();
            static_for<H>([&](auto T) { E0[T] = lds[L::E0X + A1_CV(T) * 16 + ln]; });
        } }
        if constexpr (!GEN) {
#pragma unroll
            for (int t = 0; t < H; ++t) E1[t] = comp < 2 ? E0[t] : 0.0;
        }
        cinv = 1.0 / csc;
        qd = csc * q2s;
        if constexpr (CLK) ckR = row_clock();

        // ---------------------------------------------------------------- hot state of the ADMM loop
        // The iteration is carried in UNSCALED variables so that the Ruiz factors drop out of the hot loop:
        //   xh = D x_s (world-frame forces), wh = w_s / E with w_s = z_s + y_s / rho.  OSQP's pair (z, y) is a function
        //   of w alone after the first iteration:  z = Pi(w),  y = rho (w - z)  (update_z / update_y), and
        //   w+ = w + alpha (z~ - Pi(w)).  In unscaled variables the projection uses the constant physical bounds, the
        //   only scaling-dependent per-row datum is rr = E^2 rho_row, and the only per-variable one is sigma D^-2.
        // Per horizon step and lane: xh, wh0, wh1, rr0, rr1, dI2 (6 doubles) in VGPRs; c*g lives in LDS.
        bounds_from_contact((act && io.contact[quad]) ? 1.0 : 0.0);  // contacts broadcast over the horizon (S/ConvexMpc.cpp:228-245); GEN: step 0's
        rho = P.rho0;
        warm = P.warm_start && io.warm_x != nullptr && io.warm_y != nullptr;
        if (warm && io.rho_io != nullptr && *io.rho_io > 0.0) rho = *io.rho_io;
        if constexpr (UPD && MODE == kModeMpc && H > 1) { if (reinit) rho = P.rho0; }   // a re-initialised solver starts from its settings' rho
        rho = fmin(fmax(rho, kRhoMin), kRhoMax);
        // OSQP's first iteration starts from z0 = A x0 (not projected) and y0; with x0 = y0 = 0 and 0 inside the bounds it
        // coincides with the generic w-form iteration from w = 0
        first_special = warm || !(P.fz_min <= 0.0 && P.fz_max >= 0.0);
        eqmask = 0; cmask = 0;
        set_sync();  // the Ruiz D table (aliased into the factor region) is dead from here on
        // Every global load of the hot state is issued HERE, back to back, ahead of the per-step arithmetic (round 5): the carried iterates and, on the update path, what the
        // previous tick left in the workspace.  Inside the loop below they sat behind per-step branches (pattern change | update | plain), one exposed memory round trip per
        // horizon step of a lone wavefront -- ~20 k of the 43 k cycles the update path's hot state + hand-off took (profiles/r05_tick_stages_*.json).
        [[maybe_unused]] constexpr bool kUpdPath = UPD && MODE == kModeMpc && H > 1;
        [[maybe_unused]] double cDp[kUpdPath ? H : 1], cE0p[kUpdPath ? H : 1], cZ0[kUpdPath ? H : 1], cZ1[kUpdPath ? H : 1], ccp = 0.0;
        static_for<H>([&](auto T) {
            constexpr int t = A1_CV(T);
            xh[t] = (warm && act) ? io.warm_x[t * 12 + ci] : 0.0;  // x_s = D^-1 x  <=>  xh = x
            park_warm_y(io, t);
            if constexpr (kUpdPath) {
                cDp[t] = 1.0; cE0p[t] = 1.0; cZ0[t] = 0.0; cZ1[t] = 0.0;
                if (upd) {
                    const double* cr = io.carry;
                    cDp[t] = act ? cr[CR::D + t * 12 + ci] : 1.0; cE0p[t] = act ? cr[CR::E0 + t * 12 + ci] : 1.0;   // (E1' == E0' on the fx / fy lanes: see the Ruiz passes)
                    cZ0[t] = act ? cr[CR::Z0 + t * 12 + ci] : 0.0; cZ1[t] = (act && comp < 2) ? cr[CR::Z1 + t * 12 + ci] : 0.0;
                }
            }
        });
        if constexpr (kUpdPath) { if (upd) ccp = io.carry[CR::C]; }
        static_for<H>([&](auto T) {
            constexpr int t = A1_CV(T);
            // this step's scalings and gradient (general path: from the tables the Ruiz passes left in LDS)
            double Dt_, E0t_, gt_;
            if constexpr (GEN) { Dt_ = act ? lds[L::DL + t * 12 + ci] : 1.0; E0t_ = lds[L::E0X + t * 16 + ln]; gt_ = act ? lds[L::CG + t * 12 + ci] : 0.0; }
            else { Dt_ = D[t]; E0t_ = E0[t]; gt_ = g[t]; }
            const double E1t_ = comp < 2 ? E0t_ : 0.0;
            // the step's contact flags (a per-step schedule when contact_stride = 4): bounds of my slot-0 row at step t
            const bool ct = act && io.contact[static_cast<int64_t>(t) * io.contact_stride + quad];
            const double cf = ct ? 1.0 : 0.0;
            const double lo_t = P.fz_min * cf, hi_t = P.fz_max * cf;
            if (ct) cmask |= 1u << t;
            if constexpr (!TWIN) {   // (the solve object of every kernel rebuilds its slots' bounds from the contact bits of the hand-off record: load_prepared)
                lbk[t] = comp == 2 ? lo_t : 0.0; ubk[t] = comp == 2 ? hi_t : kInfty;
            }
            const bool eq = comp == 2 && (E0t_ * hi_t - E0t_ * lo_t < kRhoTol);
            if (eq) eqmask |= 1u << t;
            rr0[t] = E0t_ * E0t_ * (eq ? kRhoEqOverIneq * rho : rho);
            rr1[t] = E1t_ * E1t_ * rho;
            const double di = 1.0 / Dt_;
            dI2[t] = di * di;
            // D^-1 q_s = c g.  General path: the table holds the raw gradient until here, so ONE row of a shared set-up turns it into c g (a second one would scale it twice
            // wherever the rows are not in lock-step -- the CPU test double's fibers are not)
            if (act && (!GEN || coop_id == 0)) lds[L::CG + t * 12 + ci] = csc * gt_;
            if constexpr (GEN && UPD && MODE == kModeMpc && H > 1) {
                // what the next tick's update calls will find in the workspace: this tick's scalings and unscaled gradient (the reads of the previous tick's fields of this step are
                // above; every row that shares the set-up holds the same values, one writes)
                if (P.warm_start == 2 && io.carry != nullptr && coop_id == 0 && act) {
                    double* cw = io.carry;
                    cw[CR::D + t * 12 + ci] = Dt_; cw[CR::E0 + t * 12 + ci] = E0t_; cw[CR::G + t * 12 + ci] = gt_;
                    if (comp < 2) cw[CR::E1 + t * 12 + ci] = E1t_;
                }
            }
            if constexpr (UPD && MODE == kModeMpc && H > 1) {
                if (upd && reinit) {
                    // pattern change: the previous solve's SCALED x_s = x / D', y_s = c' y / E' go through osqp_warm_start_x / _y as if they were unscaled -- a plain
                    // warm start (the code of warm_start = 1 below and in the first iteration) from those values
                    const double cp = ccp, Dp = cDp[t], E0p = cE0p[t];
                    xh[t] = act ? xh[t] / Dp : 0.0;
                    const double ce = cp / E0p;
                    wh0[t] = act ? ce * wh0[t] : 0.0;
                    wh1[t] = (act && comp < 2) ? ce * wh1[t] : 0.0;
                    epsv[t] = 0.0;   // (they travel like the update path's y^: through the hand-off record, with an uncorrected c g)
                } else if (upd) {
                    // The carried SCALED iterates (x_s, z_s, y_s) are used as they are, i.e. read in the NEW scaling:  x0 = D (x / D'), z0 = (E' / E) z,
                    // y0 = (c' / c)(E / E') y  with the previous scalings D', E', c' and the previous unscaled x, z, y.  OSQP's first iteration uses z0 twice:
                    // E (rho z0 - y0) in the right-hand side and (1 - alpha) z0 + y0 / rho in the w update.  The kernels compute z0 = A x0 in both places
                    // (osqp_warm_start semantics); with delta = z0 - A x0 both are reproduced exactly by parking  y^ = y0 + (1 - alpha) rr delta / c  in the w
                    // registers and by handing the first iteration  c g - A'[(2 - alpha) rr delta]  in the slot of c g (admm_iteration<FIRST> is not touched;
                    // the true c g comes back after iteration 1, see load_prepared / advance).
                    const double cp = ccp, Dp = cDp[t], E0p = cE0p[t], zp0 = cZ0[t], zp1 = cZ1[t];
                    xh[t] = act ? Dt_ * (xh[t] / Dp) : 0.0;
                    const double cr_c = cp / csc;
                    // E1 == E0 and E1' == E0' bit for bit on the fx / fy lanes (my two rows share their scaling: see the Ruiz passes): one pair of quotients serves both rows
                    const double eup = E0t_ / E0p, edn = E0p / E0t_;
                    const double y0 = act ? cr_c * eup * wh0[t] : 0.0, y1 = (act && comp < 2) ? cr_c * eup * wh1[t] : 0.0;
                    const double zc0 = act ? edn * zp0 : 0.0, zc1 = (act && comp < 2) ? edn * zp1 : 0.0;
                    const double xz = quad_perm<2, 2, 2, 2>(xh[t]);
                    const double z00 = comp == 2 ? xh[t] : fma(mu, xz, xh[t]), z01 = fma(-mu, xz, xh[t]);
                    const double d0 = act ? zc0 - z00 : 0.0, d1 = (act && comp < 2) ? zc1 - z01 : 0.0;
                    const double oma = 1.0 - P.alpha;
                    wh0[t] = y0 + oma * rr0[t] * d0 / csc;
                    wh1[t] = y1 + oma * rr1[t] * d1 / csc;
                    const double s0 = (1.0 + oma) * rr0[t] * d0, s1 = (1.0 + oma) * rr1[t] * d1;
                    const double sm = s0 - s1;
                    const double smx = quad_perm<0, 0, 0, 0>(sm), smy = quad_perm<1, 1, 1, 1>(sm);
                    epsv[t] = fma(comp == 2 ? mu : 0.0, smx + smy, s0 + s1);
                }
            }
        });
        set_sync();
        if constexpr (UPD && MODE == kModeMpc && H > 1) {
            // what the next tick's update calls will find in the workspace: this tick's scalings and unscaled gradient (z follows in write_outputs).
            // Every row that shares this set-up holds the same values; the reads of the previous tick's fields above are complete (set_sync)
            if (P.warm_start == 2 && io.carry != nullptr && coop_id == 0) {
                double* cw = io.carry;
                if (act) {
                    if constexpr (!GEN) {   // (general path: written step by step in the loop above)
                        static_for<H>([&](auto T) {
                            constexpr int t = A1_CV(T);
                            cw[CR::D + t * 12 + ci] = D[t]; cw[CR::E0 + t * 12 + ci] = E0[t]; cw[CR::G + t * 12 + ci] = g[t];
                            if (comp < 2) cw[CR::E1 + t * 12 + ci] = E1[t];
                        });
                    }
                    cw[CR::SIG + ci] = sig;
                }
                if (ln == 0) cw[CR::C] = csc;
            }
        }
        iter = 0; nfact = 0; status = A1MPC_UNSOLVED; fac_ok = true; need_factor = true; done = false; careful = false;
    }

    // Queue-order heuristic for a batch without history (scheduling only, no result depends on it): ADMM needs more iterations the more
    // force the velocity error demands, and more when all four legs share the load.  A linear fit of (iterations + 10 factor passes) on random SRBD
    // states, 30 e_vz + 27 |e_vxy|, has rank correlation 0.4-0.6 with the true cost where states vary like that -- enough to start most long QPs early
    // (tools/wave_sim.py, tools/first_solve_probe.py) -- and orders like chance where they do not.  What decides the makespan of a batch of 1-4x the
    // resident rows is where its 20-30 longest QPs start, and 85-90 % of those have four stance legs (50 % of all QPs in the flat-ground batches): a flat
    // bonus for exactly that pattern moves them forward (timeline model over 8 flat-ground batches: ADMM kernel 0.87 -> 0.81 ms at 4096 x h10, -2 % at
    // 8192 x h16, +-0 on the mixed-contact h20 batches, where 7 % of the QPs have it and they are not harder).  A term PROPORTIONAL to the number of
    // stance legs was tried first and dropped: on mixed-contact batches it sends the hardest, one- and two-leg QPs to the back of the queue (+4-6 %).
    // Stored in 1/8 units of the real cost's scale, like the counting sort of a1mpc_order_kernel expects.
    A1_DEV void predict_cost(double err0_ready, const ProblemIO& io) {
        const double evx = bc<9>(err0_ready), evy = bc<10>(err0_ready), evz = bc<11>(err0_ready);
        const bool all_stance = io.contact[0] && io.contact[1] && io.contact[2] && io.contact[3];  // (step 0's pattern when there is a schedule)
        const double hard = 30.0 * evz + 27.0 * sqrt(evx * evx + evy * evy) + (all_stance ? 10.0 : 0.0);
        const double c = 8.0 * hard + 400.0;
        pred_cost = c > 0.0 ? (c < 2047.0 ? static_cast<int>(c) : 2047) : 0;  // (NaN -> 0)
    }

    // OSQP's first iteration needs the warm-start dual y0 twice (both sweeps).  All its loads are issued here, back to back, into the
    // w registers, which hold nothing before the first iteration (w = 0 when there is no warm start): no global-memory latency
    // inside the sweeps.
    A1_DEV void park_warm_y(const ProblemIO& io, int t) {
        wh0[t] = (warm && act) ? io.warm_y[t * 20 + 5 * quad + r0] : 0.0;
        wh1[t] = (warm && act && comp < 2) ? io.warm_y[t * 20 + 5 * quad + r1] : 0.0;
    }

    // ================================================================================ hand-off between the two kernels
    template <bool UPD = false>
    A1_DEV void save_prepared(double* __restrict__ p) const {  // p: this QP's Prep<H>::STRIDE doubles
        if (!act) return;
        static_for<H>([&](auto T) {
            constexpr int t = A1_CV(T);
            p[(PR::RR0 + t) * 12 + ci] = rr0[t];   // (rr1: rr0 again on the fx / fy lanes, zero elsewhere -- see Prep)
            p[(PR::DI2 + t) * 12 + ci] = dI2[t];
            p[(PR::CG + t) * 12 + ci] = lds[L::CG + t * 12 + ci];
            if (warm) p[(PR::XH + t) * 12 + ci] = xh[t];
            if constexpr (UPD && H > 1 && !TWIN && !(SETUP_ONLY && GEN)) {
                if (upd) {  // update path: y^ of my two rows and the first iteration's c g (see setup)
                    p[(PR::YW0 + t) * 12 + ci] = wh0[t]; p[(PR::YW1 + t) * 12 + ci] = wh1[t];
                    p[(PR::CGE + t) * 12 + ci] = lds[L::CG + t * 12 + ci] - epsv[t];
                }
            }
        });
#pragma unroll
        for (int k = 0; k < 6; ++k) p[(PR::BT + k) * 12 + ci] = Bt[k];
        p[PR::CSC * 12 + ci] = csc; p[PR::CY * 12 + ci] = cy; p[PR::SY * 12 + ci] = sy; p[PR::RHO * 12 + ci] = rho;
        unsigned fl = (warm ? 1u : 0u) + (first_special ? 2u : 0u);
        if constexpr (UPD) fl += upd ? 4u : 0u;
        p[PR::PK * 12 + ci] = static_cast<double>(static_cast<unsigned long long>(cmask) | static_cast<unsigned long long>(eqmask) << 20 | static_cast<unsigned long long>(fl) << 40);
        // (general path, set-up kernel: the per-step omega rows of B~_t -- fields BWF -- were written by setup() itself: gen_rec)
    }
    // tables (GEN): the record comes from the general path's set-up kernel -- B~w_t and the per-step bounds are rebuilt in this LDS image (in the fused
    // kernel they are still where the set-up wrote them)
    // UPD = false: an instantiation without the update path's hand-off (warm_start = 2 never reaches it): the persistent ADMM kernel of every other mode keeps
    // the code -- and with it the register allocation of its hot loop -- it had before the update path existed (with it: 4 more AGPR moves per iteration, +1 %)
    template <bool UPD = false>
    A1_DEV void load_prepared(const double* __restrict__ p, const ProblemIO& io, [[maybe_unused]] bool tables = false) {
        sync();  // the previous QP's LDS image is dead
        const double am = act ? 1.0 : 0.0;  // pad lanes read lane 0's record (ci == 0) and zero what must be zero
        const unsigned long long pk = static_cast<unsigned long long>(p[PR::PK * 12 + ci]);
        const int fl = static_cast<int>(pk >> 40);
        const bool two = act && comp < 2;   // lanes with a second constraint row
        warm = fl & 1; first_special = (fl & 2) != 0;
        [[maybe_unused]] const bool upd_ = UPD && (fl & 4) != 0;
        [[maybe_unused]] double cgk[HS];  // update path: the true c g of my steps, parked in the pad column of K_t until the first iteration is done
        if constexpr (TWIN) {
            const int to = own * 12;  // my step of slot k is NS k + own: `own` record rows further on
            static_for<HS>([&](auto K) {
                constexpr int k = A1_CV(K);
                rr0[k] = am * p[(PR::RR0 + NS * k) * 12 + ci + to];
                rr1[k] = two ? p[(PR::RR0 + NS * k) * 12 + ci + to] : 0.0;
                dI2[k] = p[(PR::DI2 + NS * k) * 12 + ci + to];
                xh[k] = warm ? am * p[(PR::XH + NS * k) * 12 + ci + to] : 0.0;
                const int t = NS * k + own;
                wh0[k] = (warm && act) ? io.warm_y[t * 20 + 5 * quad + r0] : 0.0;  // park_warm_y()
                wh1[k] = (warm && act && comp < 2) ? io.warm_y[t * 20 + 5 * quad + r1] : 0.0;
                if constexpr (UPD && H > 1) {
                    cgk[k] = p[(PR::CG + NS * k) * 12 + ci + to];
                    if (upd_) {  // update path: y^ and the first iteration's c g come from the set-up (see setup)
                        wh0[k] = am * p[(PR::YW0 + NS * k) * 12 + ci + to]; wh1[k] = am * p[(PR::YW1 + NS * k) * 12 + ci + to];
                        if (act) lds[L::CG + NS * k * 12 + ci + to] = p[(PR::CGE + NS * k) * 12 + ci + to];
                    }
                    if (act && !upd_) lds[L::CG + NS * k * 12 + ci + to] = cgk[k];
                } else {
                    if (act) lds[L::CG + NS * k * 12 + ci + to] = p[(PR::CG + NS * k) * 12 + ci + to];
                }
            });
        } else {
            static_for<H>([&](auto T) {
                constexpr int t = A1_CV(T);
                rr0[t] = am * p[(PR::RR0 + t) * 12 + ci];
                rr1[t] = two ? p[(PR::RR0 + t) * 12 + ci] : 0.0;
                dI2[t] = p[(PR::DI2 + t) * 12 + ci];
                xh[t] = warm ? am * p[(PR::XH + t) * 12 + ci] : 0.0;
                park_warm_y(io, t);
                if constexpr (UPD && H > 1) {
                    cgk[t] = p[(PR::CG + t) * 12 + ci];
                    if (upd_) {
                        wh0[t] = am * p[(PR::YW0 + t) * 12 + ci]; wh1[t] = am * p[(PR::YW1 + t) * 12 + ci];
                        if (act) lds[L::CG + t * 12 + ci] = p[(PR::CGE + t) * 12 + ci];
                    }
                    if (act && !upd_) lds[L::CG + t * 12 + ci] = cgk[t];
                } else {
                    if (act) lds[L::CG + t * 12 + ci] = p[(PR::CG + t) * 12 + ci];
                }
            });
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            Bt[k] = am * p[(PR::BT + k) * 12 + ci];
            if (wr) lds[L::BL + k * 12 + ci] = Bt[k];
        }
        if (wr) lds[L::BL + L::ZROW * 12 + ci] = 0.0;
        csc = p[PR::CSC * 12 + ci]; cinv = 1.0 / csc; qd = csc * q2s;
        set_rotation(p[PR::CY * 12 + ci], p[PR::SY * 12 + ci]);
        rho = p[PR::RHO * 12 + ci];
        cmask = act ? static_cast<unsigned>(pk & 0xfffffu) : 0u;
#pragma unroll
        for (int k = 0; k < HS; ++k) slot_bounds(k, NS * k + own);
        lo_u = P.fz_min * (cmask & 1u ? 1.0 : 0.0); hi_u = P.fz_max * (cmask & 1u ? 1.0 : 0.0);
        lb0 = comp == 2 ? lo_u : 0.0; ub0 = comp == 2 ? hi_u : kInfty;
        eqmask = act ? static_cast<unsigned>((pk >> 20) & 0xfffffu) : 0u;
        if constexpr (GEN) {
            if (tables && wr) {
                static_for<H>([&](auto T) {
                    constexpr int t = A1_CV(T);
#pragma unroll
                    for (int c = 0; c < 3; ++c) lds[L::BW + (t * 3 + c) * 12 + ci] = p[(PR::BWF + 3 * t + c) * 12 + ci];
                });
            }
        }
        sync();
        if constexpr (UPD && H > 1) {
            if (first_special) {  // the true c g waits in the pad column of K_t until iteration 1 is done (restore_cg; the update path's first iteration runs on a corrected one).
                                  // After the sync: in the fused kernels the record is staged in the factor region, which holds the pad columns
                if (act) {
#pragma unroll
                    for (int k = 0; k < HS; ++k) lds[L::FAC + (NS * k + own) * L::SLOT + ci * L::KSTR + L::GCOL] = cgk[k];
                }
                sync();
            }
        }
        iter = 0; nfact = 0; status = A1MPC_UNSOLVED; fac_ok = true; need_factor = true; done = false; careful = false;
        pfX = pfT = pfU = 0;
#ifdef A1X_CLK
        clkB = clkF = clkT = clkU = clkX = 0;
#endif
    }
    // update path: iteration 1 is done -- the true c g (parked in the pad column of K_t by load_prepared) replaces the first iteration's corrected one
    A1_DEV void restore_cg() {
        sync();
        if (act) {
#pragma unroll
            for (int k = 0; k < HS; ++k) {
                const int t = NS * k + own;
                lds[L::CG + t * 12 + ci] = lds[L::FAC + t * L::SLOT + ci * L::KSTR + L::GCOL];
            }
        }
        sync();
    }

    // ================================================================================ Riccati factorisation of
    //   M = c P + sigma D^-2 + A' (E^2 rho) A      (K_s = D M D is OSQP's reduced KKT matrix)
    A1_DEV void factorize() {
        ++nfact;
        const double sigma_f = row_opaque(P.sigma);
        // stage W_t = c R + sigma D^-2 + A_t' (E^2 rho) A_t  (block-diagonal, 3x3 per leg) into slot t: my row of my leg's block
        sync();
        static_for<HS>([&](auto T) {  // (a twin pair: each row stages the W_t of its own steps)
            const double a0 = rr0[T], a1 = rr1[T];
            const double sp = a0 + a1;
            const double spx = quad_perm<0, 0, 0, 0>(sp), spy = quad_perm<1, 1, 1, 1>(sp);
            const double base = csc * r2a + sigma_f * dI2[T];
            const double wd = act ? (comp == 2 ? base + a0 + mu * mu * (spx + spy) : base + sp) : 0.0;
            const double wo = comp < 2 ? mu * (a0 - a1) : 0.0;
            const double wox = quad_perm<0, 0, 0, 0>(wo), woy = quad_perm<1, 1, 1, 1>(wo);
            double* w = lds + L::FAC + (NS * A1_CV(T) + own) * L::SLOT + L::K_SZ + 3 * ln;  // staged in the S area of the slot (the K area's pad column carries G)
            w[0] = comp == 0 ? wd : (comp == 2 ? wox : 0.0);
            w[1] = comp == 1 ? wd : (comp == 2 ? woy : 0.0);
            w[2] = comp == 2 ? wd : wo;
        });
        sync();
        // lane constants of this pass: component indicators (1.0 / 0.0).  "Add on my diagonal entry only" is one v_fmac_f64_dpp with
        // the leg's bank mask, the component indicator (times my own value) as the own-lane factor and 1.0 = cm[0] of lane 0 as the broadcast one.
                double cm[3], qdc[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { cm[c] = comp == c ? 1.0 : 0.0; qdc[c] = comp == c ? qd : 0.0; }
        // the broadcast factor (1.0 on lane 0) is read through DPP: it must come out of row_dpp_ready(), which also keeps the
        // compiler from re-materialising it right in front of a use (VALU write -> DPP read hazard)
        const double one0 = row_dpp_ready(cm[0]);
        double Pn[12];  // row of P_{t+1} (state layout); terminal value c Q
#pragma unroll
        for (int j = 0; j < 12; ++j) Pn[j] = (act && ci == j) ? qd : 0.0;
        double Btr[6];  // my column of B~, as a DPP source
#pragma unroll
        for (int k = 0; k < 6; ++k) Btr[k] = row_dpp_ready(Bt[k]);
#pragma unroll 1
        for (int t = H - 1; t >= 0; --t) {
            double* slot = lds + L::FAC + t * L::SLOT;
            const double wv[3] = {slot[L::K_SZ + 3 * ln], slot[L::K_SZ + 3 * ln + 1], slot[L::K_SZ + 3 * ln + 2]};
            [[maybe_unused]] double Btt[6];
            if constexpr (GEN) {  // this step's B~_t
                Bt_at(t, Btt);
#pragma unroll
                for (int k = 0; k < 6; ++k) Btr[k] = row_dpp_ready(Btt[k]);
            }
            sync();  // everybody holds W_t before the slot is overwritten
            // G = A' P_{t+1}  (rows mixed across lanes), then GA = G A (columns, lane-local)
            double G[12];
            row_dpp_ready12(Pn);
            static_for<12>([&](auto J) { G[J] = opAT(Pn[J]); });
            // F' = G(:,6:12) B~  (state row-owner)  and  Y = P_{t+1}(6:12,6:12) B~  (valid on the wrench lanes).  B~[k][b] is register k
            // of force lane b (Bt), so every term is a DPP broadcast: no LDS traffic, no staging registers for B~'s rows
            double Ft[12], Y[12];
#pragma unroll
            for (int b = 0; b < 12; ++b) { Ft[b] = 0.0; Y[b] = 0.0; }
            static_for<6>([&](auto K) {
                static_for<12>([&](auto B) {
                    fma_bcast<lane_of(A1_CV(B))>(Ft[B], G[6 + K], Btr[K]);
                    fma_bcast<lane_of(A1_CV(B))>(Y[B], Pn[6 + K], Btr[K]);
                });
            });
            G[6] += dt * (cy * G[0] - sy * G[1]);
            G[7] += dt * (sy * G[0] + cy * G[1]);
            G[8] += dt * G[2];
            G[9] += dt * G[3];
            G[10] += dt * G[4];
            G[11] += dt * G[5];
            // S = W_t + B~' Y   (force row-owner): twelve interleaved chains, each seeded with its W entry
            double S[12];
#pragma unroll
            for (int b = 0; b < 12; ++b) S[b] = 0.0;
            row_dpp_ready12(Y);
            static_for<6>([&](auto K) {
                static_for<12>([&](auto B) {
                    if constexpr (GEN) fma_bcast<lane_of(6 + A1_CV(K))>(S[B], Btt[K], Y[B]);
                    else fma_bcast<lane_of(6 + A1_CV(K))>(S[B], Bt[K], Y[B]);
                });
            });
            static_for<12>([&](auto B) { fma_bcast_leg<0, A1_CV(B) / 3>(S[B], wv[A1_CV(B) % 3], one0); });  // + W on my leg's block (one0 of lane 0 = 1)
            // in-place Gauss-Jordan inverse of the SPD 12x12 (no pivoting).  Pivot k: row k is scaled by 1/p, every other row i
            // subtracts S_ik/p times row k.  Both are  S_ij += m_i * S_kj  with m_k = 1/p - 1 and m_i = -S_ik/p, i.e. ONE
            // v_fmac_f64_dpp per element whose DPP source is the element's own register read from lane k (gj_pivot); the
            // reciprocal of the next pivot is computed inside the block, between the eliminations that do not feed it.
            row_dpp_ready12(S);
            double piv = bc<0>(S[0]);
            double pinv = row_recip(piv);
            static_for<12>([&](auto K) {
                constexpr int k = A1_CV(K);
                if (!(piv > 0.0)) fac_ok = false;
                fnma_bcast_leg<0, k / 3>(S[k], cm[k % 3], one0);  // S_ik, minus one on the pivot's own row
                const double mlt = -(S[k] * pinv);
                S[k] = (act && ci == k) ? pinv : mlt;
                gj_pivot<k>(S, mlt, piv, pinv);
            });
            if (wr) {
                // packed lower triangle, branch-free: entries right of my diagonal go to my own diagonal slot first, in descending
                // column order, so that the diagonal entry itself is the last one written there (LDS stores of a wave stay in order)
                static_for<12>([&](auto Bd) {
                    constexpr int b = 11 - A1_CV(Bd);
                    slot[L::K_SZ + (b <= ci ? sinv_slot<b>(ci) : sinv_slot<-1>(ci))] = S[b];
                });
            }
            // K' = F' S^-1  (state row-owner): K'[i][a] = sum_b F'[i][b] S^-1[b][a]; S^-1[b][a] = register a of force lane b, so every
            // term is one v_fmac_f64_dpp (no LDS round trip)
            double Kt[12];
#pragma unroll
            for (int a = 0; a < 12; ++a) Kt[a] = 0.0;
            row_dpp_ready12(S);
            static_for<12>([&](auto B) {  // twelve independent accumulator chains
                static_for<12>([&](auto A_) { fma_bcast<lane_of(A1_CV(B))>(Kt[A_], Ft[B], S[A_]); });
            });
            if (wr && t > 0) {  // K_0 is never used (x_0 = 0)
#pragma unroll
                for (int a = 0; a < 12; ++a) slot[a * L::KSTR + ci] = Kt[a];
            }
            // P_t = c Q + A' P_{t+1} A - F' K:  K[a][j] = register a of state lane j
            if (t > 0) {
                static_for<12>([&](auto J) {
                    Pn[J] = G[J];
                    fma_bcast_leg<0, A1_CV(J) / 3>(Pn[J], qdc[A1_CV(J) % 3], one0);
                });
                row_dpp_ready12(Kt);
                static_for<12>([&](auto A_) {
                    static_for<12>([&](auto J) { fnma_bcast<lane_of(A1_CV(J))>(Pn[J], Ft[A_], Kt[A_]); });
                });
            }
            sync();  // K_t and S_t^-1 of this step are in LDS before the next step reuses the registers' sources
        }
#pragma unroll
        for (int b = 0; b < 12; ++b) Brw[b] = kBrowInRegs ? brow[b] : 0.0;
        need_factor = false;
        if (!fac_ok) { status = A1MPC_NON_CVX; done = true; }
    }

    // ================================================================================ one ADMM iteration
    // (osqp.c: update_xz_tilde, update_x, update_z, update_y).  The linear system M v = b is solved by the two Riccati
    // sweeps; the right-hand side is formed inside the backward sweep and the x / w updates consume v_t inside the
    // forward sweep, so only d_t crosses between the sweeps.
    // FIRST: OSQP's iteration 1 starts from z0 = A x0 (not projected) and y0 (warm start).
    template <bool FIRST, bool CAREFUL = false>
    A1_DEV void admm_iteration() {
        if constexpr (TWIN) admm_iteration_twin<FIRST, CAREFUL>();
        else admm_iteration_single<FIRST, CAREFUL>();
    }
    template <bool FIRST, bool CAREFUL>
    A1_DEV void admm_iteration_single() {
        // Loop-invariant scalars are laundered through row_opaque() once per iteration: otherwise LICM hoists every
        // per-step product that only depends on them out of the ADMM loop and the register file overflows.
        const double sigma_l = row_opaque(P.sigma);
        const double al = P.alpha, oma = 1.0 - P.alpha;
        const double muz = comp == 2 ? mu : 0.0, mux = comp < 2 ? mu : 0.0;  // selects folded into multipliers
        // One wave per SIMD: every instruction costs an issue slot, so the sweeps are written for instruction count --
        // accumulator chains are seeded with values that have to be added anyway (no zero-initialised partners, no final
        // adds), independent chains are interleaved by hand, subtraction rides on the NEG modifier of v_fmac_f64_dpp.
        double d[H];
        double pv = 0.0;  // costate p_{t+1}, state layout (dpp-ready at the top of every step)
        static_for<H>([&](auto TT) {
            constexpr int t = H - 1 - A1_CV(TT);
            const double* slot = lds + L::FAC + t * L::SLOT;
            // issue this step's LDS reads first: their latency overlaps the right-hand-side arithmetic below
            double Sr[12], Kc[12];
            static_for<12>([&](auto B) {
                constexpr int b = A1_CV(B);
                Sr[b] = slot[L::K_SZ + sinv_slot<b>(ci)];
                if constexpr (t > 0) Kc[b] = slot[b * L::KSTR + ci];
            });
            const double cgt = lds[L::CG + t * 12 + ci];
            [[maybe_unused]] double Btl[6];
            if constexpr (GEN) Bt_at(t, Btl);
            const double lbt = lbs<t>(t), ubt = ubs<t>(t);
            row_sched_fence();
            // rhs of update_xz_tilde premultiplied by D^-1:  b = sigma D^-2 xh - c g + A' [E (rho z_s - y_s)]
            double t0, t1;
            if constexpr (FIRST) {  // E (rho z_s - y_s) = rr (A x0) - c y0
                const double xz = quad_perm<2, 2, 2, 2>(xh[t]);
                const double yw0 = wh0[t], yw1 = wh1[t];  // y0 is parked in the (still unused) w registers, see park_warm_y()
                t0 = rr0[t] * (comp == 2 ? xh[t] : fma(mu, xz, xh[t])) - csc * yw0;
                t1 = rr1[t] * fma(-mu, xz, xh[t]) - csc * yw1;
            } else {                // E (rho z_s - y_s) = rr (2 Pi(wh) - wh)
                const double z0 = clamp_f64(wh0[t], lbt, ubt), z1 = min_f64(wh1[t], 0.0);
                t0 = rr0[t] * fma(2.0, z0, -wh0[t]);
                t1 = rr1[t] * fma(2.0, z1, -wh1[t]);
            }
            const double sm = t0 - t1;
            const double smx = quad_perm<0, 0, 0, 0>(sm), smy = quad_perm<1, 1, 1, 1>(sm);
            const double at = fma(muz, smx + smy, t0 + t1);  // fz lanes: t1 == 0 (rr1 == 0); fx/fy lanes: muz == 0
            // r = b - B~' p_{t+1};  d_t = S_t^-1 r  interleaved with  p_t = A' p_{t+1} + K_t' r   (instruction blocks, a1mpc_rowops.hpp)
            const double sd = sigma_l * dI2[t];
            double r, pa = 0.0, pb = 0.0;
            if constexpr (t < H - 1) {
                if constexpr (t > 0) pb = gV * row_ror<8>(pv);
                if constexpr (GEN) sweep_back_rhs(r, pa, pb, at, cgt, sd, xh[t], pv, Btl, gA, gB, gC);
                else sweep_back_rhs(r, pa, pb, at, cgt, sd, xh[t], pv, Bt, gA, gB, gC);
            } else {
                r = row_dpp_ready(fma(sd, xh[t], at - cgt));  // p_H = 0
            }
            row_lds_landed();
            if constexpr (t > 0) {
                sweep_back_chains(d[t], pa, pb, r, Sr, Kc);
                pv = pa;
            } else {
                d[t] = dot12_block(Sr, r);
            }
        });
        double s = 0.0;  // state x_t of the LQ roll-out (x_0 = 0); dpp-ready at the top of every step
        static_for<H>([&](auto T) {
            constexpr int t = A1_CV(T);
            const double* slot = lds + L::FAC + t * L::SLOT;
            // issue this step's LDS reads first (K_t row): they overlap the previous step's w update
            double Kr[12];
            static_for<12>([&](auto B) {
                constexpr int b = A1_CV(B);
                if constexpr (t > 0) Kr[b] = slot[krow + b];
            });
            [[maybe_unused]] double Brl[12];  // H > 10: my row of B~ is re-read per step (the 24 registers are worth more than 6 LDS reads there)
            if constexpr (!kBrowInRegs && t < H - 1) {
                const double* br = brow_at(t);
#pragma unroll
                for (int b = 0; b < 12; ++b) Brl[b] = br[b];
            }
            const double lbt = lbs<t>(t), ubt = ubs<t>(t);
            row_sched_fence();
            // v_t = d_t - K_t x_t,  xh <- alpha v + (1 - alpha) xh,  x_{t+1} = A x_t + B~ v_t   (instruction blocks)
            const double am = act ? 1.0 : 0.0;  // pad lanes carry no force
            double v = d[t], sa = 0.0, sb = 0.0, z0 = 0.0;
            [[maybe_unused]] double xz_first = 0.0;
            if constexpr (FIRST) xz_first = quad_perm<2, 2, 2, 2>(xh[t]);  // reads x0 before the block overwrites xh
            const double xh_old = xh[t];
            [[maybe_unused]] double gt0 = 0.0, gt1 = 0.0;  // CAREFUL: rr (2 Pi(w) - w) of the state BEFORE this iteration's update
            if constexpr (CAREFUL) {
                const double zp0 = clamp_f64(wh0[t], lbt, ubt), zp1 = min_f64(wh1[t], 0.0);
                gt0 = rr0[t] * fma(2.0, zp0, -wh0[t]);
                gt1 = rr1[t] * fma(2.0, zp1, -wh1[t]);
            }
            row_lds_landed();
            if constexpr (t == 0) {
                v = row_dpp_ready(am * v);  // x_0 = 0
                xh[t] = fma(al, v, oma * xh[t]);
            } else if constexpr (t < H - 1) {
                sb = fP * row_ror<8>(s);
                sweep_fwd_gain<true>(v, sa, sb, xh[t], s, Kr, fA, fB, fC, am, oma, al);
            } else {
                sweep_fwd_gain<false>(v, sa, sb, xh[t], s, Kr, fA, fB, fC, am, oma, al);
            }
            if constexpr (t < H - 1) {
                if constexpr (kBrowInRegs) sweep_fwd_input(sa, sb, z0, v, Brw, wh0[t], lbt, ubt);
                else sweep_fwd_input(sa, sb, z0, v, Brl, wh0[t], lbt, ubt);
                s = sa;  // lanes without a wrench state read the zero row of B~
            } else {
                z0 = clamp_f64(wh0[t], lbt, ubt);
            }
            // z~ = A v (unscaled), then update_x / update_z / update_y in the w form
            const double vz = quad_perm<2, 2, 2, 2>(v);
            const double av0 = fma(mux, vz, v);
            const double av1 = fma(-mux, vz, v);
            if constexpr (CAREFUL) {
                // the x-update identity  c P x~ + c g = sigma D^-2 (x_prev - x~) + A' [rr (2 z_prev - w_prev - z~)]  holds for an exact solve and has no
                // cancellation; G <- alpha (c P x~ + c g) + (1 - alpha) G follows x <- alpha x~ + (1 - alpha) x_prev.  Carried only while rho is small.
                const double d0 = fma(-rr0[t], av0, gt0), d1 = fma(-rr1[t], av1, gt1);
                const double sdm = d0 - d1;
                const double sdx = quad_perm<0, 0, 0, 0>(sdm), sdy = quad_perm<1, 1, 1, 1>(sdm);
                const double atd = fma(muz, sdx + sdy, d0 + d1);
                const double T = fma(sigma_l * dI2[t], xh_old - v, atd);
                double* G = lds + L::FAC + t * L::SLOT + ci * L::KSTR + L::GCOL;
                if (act) *G = fma(al, T, oma * *G);
            }
            if constexpr (FIRST) {  // w1 = alpha z~ + (1 - alpha) z0 + y0 / rho,  z0 = A x0
                const double xz = xz_first;
                const double yw0 = wh0[t], yw1 = wh1[t];
                const double z00 = comp == 2 ? xh_old : fma(mu, xz, xh_old), z01 = fma(-mu, xz, xh_old);
                wh0[t] = al * av0 + oma * z00 + (rr0[t] > 0.0 ? csc * yw0 / rr0[t] : 0.0);
                wh1[t] = comp < 2 ? al * av1 + oma * z01 + (rr1[t] > 0.0 ? csc * yw1 / rr1[t] : 0.0) : 0.0;
            } else {                // w+ = w + alpha (z~ - Pi(w))
                const double z1 = min_f64(wh1[t], 0.0);
                wh0[t] = fma(al, av0 - z0, wh0[t]);
                wh1[t] = fma(al, av1 - z1, wh1[t]);   // (fz lanes have no second row: see the twin-row code)
            }
        });
    }

    // ---------------------------------------------------------------- the same iteration for a main / twin pair of rows (TWIN)
    // The pair splits (a) the per-lane state: slot k of xh / wh / rr / dI2 is horizon step 2k on the main row and 2k + 1 on the twin, so all
    // element-wise work (projection, right-hand side, x / w update) is issued once per PAIR of steps and the register file holds half the
    // state; and (b) the two 12-term products of a backward step on the same right-hand side: one chain of v_fmac_f64_dpp and one LDS read
    // per term compute K_t' r on the main row and S_t^-1 r on the twin.  Everything that is sequential over the steps (the costate, the
    // roll-out) is computed redundantly and bit-identically by both rows; twin_exchange() moves the per-step values the other row needs
    // (the right-hand-side data e_t, then p_t / d_t).  Every value is formed by the same operations in the same order as in the
    // single-row code above: the two kernels agree bit for bit.
    // QUAD, a quad of rows on one QP: slot k is horizon step 4k + own, rows 1 / 3 repeat what rows 0 / 2 do in the sweeps and hold their own steps' state; the right-hand sides of a slot's four steps reach every row through one v_permlane32_swap and two v_permlane16_swap.
    // LDS reads are issued as soon as the block that consumed the previous step's has been issued (into the registers it frees): with one
    // wave per SIMD nothing else hides an LDS round trip.  (Measured and not kept: carrying the first reads of the next iteration across the
    // loop back-edge -- the loop-carried registers push loop invariants into scratch, 2.95 -> 3.56 us per iteration; double-buffering the
    // backward reads a whole step ahead -- 36 more AGPR moves per iteration, 2.77 -> 2.88 us; again in round 3 with both sweeps' reads two steps ahead in a second
    // register set, after the loop had lost its AGPR traffic: 44 AGPR moves per iteration come back, +6 ... 10 % per iteration.)
    template <int T_>
    A1_DEV void issue_back_reads(double (&M)[12]) const {
        const double* slot = lds + L::FAC + T_ * L::SLOT;
        static_for<12>([&](auto B) {  // one read serves both rows: the main row's K_t column entry, the twin's S_t^-1 row entry (t = 0: the main row's value is unused)
            constexpr int b = A1_CV(B);
            M[b] = slot[mofs[b]];
        });
    }
    template <bool FIRST, bool CAREFUL>
    A1_DEV void admm_iteration_twin() {
        static_assert(H % NS == 0, "twin rows split the horizon steps in pairs, quads in fours");
        const double sigma_l = row_opaque(P.sigma);
        const double al = P.alpha, oma = 1.0 - P.alpha;
        const double muz = comp == 2 ? mu : 0.0, mux = comp < 2 ? mu : 0.0;
        const double* cgp = lds + L::CG + own * 12 + ci;  // c g of my step of slot k: cgp[12 NS k]
        double d[H];
        double pv = 0.0;  // costate p_{t+1}, state layout
        double M[12], Kq[12];
        double pbn = 0.0;  // gV ror8(p_{t+1}) on the main row, 0 on the twin
        issue_back_reads<H - 1>(M);
        double cgv = cgp[12 * NS * (HS - 1)];
        row_sched_fence();
#ifdef A1X_CLK
        const long long c0_ = clock64();
#endif
        // one step of the backward sweep: r = e_t - B~' p_{t+1};  main row: p_t = A' p_{t+1} + K_t' r,  twin: d_t = S_t^-1 r;  then the rows swap
        auto back_step = [&](auto TQ, double e_t) {
            constexpr int t = A1_CV(TQ);
            double r = e_t, pa = 0.0, pb = 0.0;
            if constexpr (t < H - 1) {
                if constexpr (t > 0) pb = pbn;
                if constexpr (GEN) {
                    double Btl[6];
                    Bt_at(t, Btl);  // my column of this step's B~_t
                    sweep_back_rhs_twin(r, pa, pb, pv, Btl, gAm, gBm, gCm, hm);
                } else {
                    sweep_back_rhs_twin(r, pa, pb, pv, Bt, gAm, gBm, gCm, hm);
                }
            } else {
                r = row_dpp_ready(r);  // p_H = 0
            }
            sweep_back_chain_twin(pa, pb, r, M);
            row_sched_fence();
            if constexpr (t > 0) issue_back_reads<(t > 0 ? t - 1 : 0)>(M);  // the next step's reads, into the registers the chain has just freed
            else {
                const auto kr1 = row_lds_at<1 * L::SLOT>(lds + L::FAC + krow);  // ... or the forward sweep's first K row
                static_for<12>([&](auto B) { Kq[A1_CV(B)] = kr1[A1_CV(B)]; });
            }
            row_sched_fence();
            // the seed gV ror8(p_t) of the next step's costate, taken BEFORE the swap: the main row already holds p_t, and on the twin (which holds d_t
            // there) the multiplier is zero -- behind the swap the rotate would wait two states for it
            if constexpr (t > 1) pbn = gVm * row_ror<8>(pa);
            d[t] = twin_exchange_copied(pa, pb);  // (pb: the chain block's copy of pa)
            pv = pa;
        };
        static_for<HS>([&](auto KK) {
            constexpr int k = HS - 1 - A1_CV(KK);
            // my step's right-hand-side data  e = sigma D^-2 xh - c g + A' [E (rho z_s - y_s)]  (see the single-row code)
            double t0, t1;
            if constexpr (FIRST) {
                const double xz = quad_perm<2, 2, 2, 2>(xh[k]);
                const double yw0 = wh0[k], yw1 = wh1[k];
                t0 = rr0[k] * (comp == 2 ? xh[k] : fma(mu, xz, xh[k])) - csc * yw0;
                t1 = rr1[k] * fma(-mu, xz, xh[k]) - csc * yw1;
            } else {
                const double z0 = clamp_f64(wh0[k], lbs<k>(NS * k + own), ubs<k>(NS * k + own)), z1 = min_f64(wh1[k], 0.0);
                t0 = rr0[k] * fma(2.0, z0, -wh0[k]);
                t1 = rr1[k] * fma(2.0, z1, -wh1[k]);
            }
            const double sm = t0 - t1;
            const double smx = quad_perm<0, 0, 0, 0>(sm), smy = quad_perm<1, 1, 1, 1>(sm);
            const double at = fma(muz, smx + smy, t0 + t1);
            double e = fma(sigma_l * dI2[k], xh[k], at - cgv);
            if constexpr (k > 0) cgv = cgp[12 * NS * (k > 0 ? k - 1 : 0)];
            const double eo = twin_exchange(e);  // e: step 2k (the main row's) on both rows, eo: step 2k + 1 (the twin's)
            if constexpr (QUAD) {  // (rows 0, 2: steps 4k, 4k + 1; rows 1, 3: steps 4k + 2, 4k + 3 -- the halves' even and odd rows swap)
                double eb = eo;
                const double ec = quad_exchange(e), ed = quad_exchange(eb);
                back_step(std::integral_constant<int, 4 * k + 3>{}, ed);
                back_step(std::integral_constant<int, 4 * k + 2>{}, ec);
                back_step(std::integral_constant<int, 4 * k + 1>{}, eb);
                back_step(std::integral_constant<int, 4 * k>{}, e);
            } else {
                back_step(std::integral_constant<int, 2 * k + 1>{}, eo);
                back_step(std::integral_constant<int, 2 * k>{}, e);
            }
        });
#ifdef A1X_CLK
        const long long c1_ = clock64();
#endif
        double s = 0.0;  // state x_t of the LQ roll-out (x_0 = 0)
        // one step of the forward sweep: v_t = d_t - K_t x_t,  x_{t+1} = A x_t + B~ v_t
        auto fwd_step = [&](auto TQ) {
            constexpr int t = A1_CV(TQ);
            double v = d[t], sb = 0.0;
            // (one address register per step: the reads are base + immediate; formed ahead of the gain block -- hipcc pads an asm statement that follows another)
            [[maybe_unused]] lds_cptr krn = nullptr;
            if constexpr (t > 0 && t < H - 1) krn = row_lds_at<(t + 1) * L::SLOT>(lds + L::FAC + krow);
            // (no `am` multiply: a pad lane carries a copy of lane 0's v -- its matrix rows are lane 0's -- that no DPP read, no store and no norm ever looks at;
            // on the lanes that matter v * 1.0 is v)
            if constexpr (t == 0) {
                v = row_dpp_ready(v);  // x_0 = 0
            } else if constexpr (t < H - 1) {
                sb = fP * row_ror<8>(s);
                sweep_fwd_gain_twin<true>(v, s, Kq, fA);
            } else {
                sweep_fwd_gain_twin<false>(v, s, Kq, fA);
            }
            row_sched_fence();
            [[maybe_unused]] double Brl[12];
            if constexpr (GEN && t < H - 1) {  // my row of this step's B~_t
                const double* br = brow_at(t);
#pragma unroll
                for (int b = 0; b < 12; ++b) Brl[b] = br[b];
            }
            if constexpr (t > 0 && t < H - 1) {  // the next step's K row, into the registers the gain block has just freed
                static_for<12>([&](auto B) { Kq[A1_CV(B)] = krn[A1_CV(B)]; });
            }
            row_sched_fence();
            if constexpr (t < H - 1) {  // (s: zero at t = 0, x_t with the seeds of x_{t+1} on top after the gain block)
                if constexpr (GEN) sweep_fwd_input_twin<(t > 0)>(s, sb, v, Brl, fB, fC);
                else sweep_fwd_input_twin<(t > 0)>(s, sb, v, Brw, fB, fC);  // lanes without a wrench state read the zero row of B~
                // the next step's first DPP read of s: behind the row_ror<8> of its seed (hipcc pads that one itself), or -- last step -- right at the top of its block
                if constexpr (t == H - 2) s = row_dpp_ready(s);
            }
            return v;
        };
        static_for<HS>([&](auto K) {
            constexpr int k = A1_CV(K);
            const double va = fwd_step(std::integral_constant<int, NS * k>{});
            const double vb = fwd_step(std::integral_constant<int, NS * k + 1>{});
            double v = twin ? vb : va;  // my step's v
            if constexpr (QUAD) {
                const double vc = fwd_step(std::integral_constant<int, 4 * k + 2>{});
                const double vd = fwd_step(std::integral_constant<int, 4 * k + 3>{});
                if (own >= 2) v = twin ? vd : vc;
            }
            // update_x / update_z / update_y of my step in the w form (see the single-row code)
            [[maybe_unused]] double xz_first = 0.0;
            if constexpr (FIRST) xz_first = quad_perm<2, 2, 2, 2>(xh[k]);
            const double xh_old = xh[k];
            [[maybe_unused]] double gt0 = 0.0, gt1 = 0.0;
            if constexpr (CAREFUL) {
                const double zp0 = clamp_f64(wh0[k], lbs<k>(NS * k + own), ubs<k>(NS * k + own)), zp1 = min_f64(wh1[k], 0.0);
                gt0 = rr0[k] * fma(2.0, zp0, -wh0[k]);
                gt1 = rr1[k] * fma(2.0, zp1, -wh1[k]);
            }
            const double z0 = clamp_f64(wh0[k], lbs<k>(NS * k + own), ubs<k>(NS * k + own));
            xh[k] = fma(al, v, oma * xh[k]);
            const double vz = quad_perm<2, 2, 2, 2>(v);
            const double av0 = fma(mux, vz, v);
            const double av1 = fma(-mux, vz, v);
            if constexpr (CAREFUL) {
                const double d0 = fma(-rr0[k], av0, gt0), d1 = fma(-rr1[k], av1, gt1);
                const double sdm = d0 - d1;
                const double sdx = quad_perm<0, 0, 0, 0>(sdm), sdy = quad_perm<1, 1, 1, 1>(sdm);
                const double atd = fma(muz, sdx + sdy, d0 + d1);
                const double T = fma(sigma_l * dI2[k], xh_old - v, atd);
                double* G = lds + L::FAC + (NS * k + own) * L::SLOT + ci * L::KSTR + L::GCOL;
                if (act) *G = fma(al, T, oma * *G);
            }
            if constexpr (FIRST) {
                const double xz = xz_first;
                const double yw0 = wh0[k], yw1 = wh1[k];
                const double z00 = comp == 2 ? xh_old : fma(mu, xz, xh_old), z01 = fma(-mu, xz, xh_old);
                wh0[k] = al * av0 + oma * z00 + (rr0[k] > 0.0 ? csc * yw0 / rr0[k] : 0.0);
                wh1[k] = comp < 2 ? al * av1 + oma * z01 + (rr1[k] > 0.0 ? csc * yw1 / rr1[k] : 0.0) : 0.0;
            } else {
                const double z1 = min_f64(wh1[k], 0.0);
                wh0[k] = fma(al, av0 - z0, wh0[k]);
                // fz lanes own no second constraint row: their wh1 is never read where it matters (rr1 = E1^2 rho = 0 there, so every product with it vanishes; the residual
                // norms, the outputs and the carry test comp < 2), so it need not be held at zero by a per-lane multiplier -- at h = 20 that multiplier was reloaded
                // from scratch once per step pair (ten scratch loads per iteration)
                wh1[k] = fma(al, av1 - z1, wh1[k]);
            }
        });
#ifdef A1X_CLK
        const long long c2_ = clock64();
        clkB += c1_ - c0_; clkF += c2_ - c1_;
#endif
    }


    // ================================================================================ residuals (auxil.c compute_pri_res / compute_dua_res / ...)
    // maximum over the lanes of my row -- and, for a twin pair, of both rows (the pair's norms run over all horizon steps)
    A1_DEV double pair_allmax(double v) const {
        v = row_allmax(v);
        if constexpr (TWIN) {
            const double o = twin_exchange(v);
            v = max_f64(v, o);
            if constexpr (QUAD) {
                const double q = quad_exchange(v);
                v = max_f64(v, q);
            }
        }
        return v;
    }
    A1_DEV void update_info() {
        const double rho_c = row_opaque(rho), one_c = row_opaque(1.0);  // keep the cold path's invariants out of the hot loop's registers
        double Pu[HS];
        if constexpr (TWIN) {
            // a twin pair: both rows run the roll-out and the adjoint sweep over all steps (the rows swap their xh of every step pair);
            // each keeps the (P u) of its own steps
            double sv[H];
            double s = row_dpp_ready(0.0);
            auto Bx = [&](int t, double x_ready) {  // (B~_t x) on the wrench lanes
                if constexpr (GEN) {
                    double Br[12];
                    const double* br = brow_at(t);
#pragma unroll
                    for (int b = 0; b < 12; ++b) Br[b] = br[b];
                    return dot_bc<0>(Br, x_ready);
                } else {
                    return dot_bc<0>(Brw, x_ready);
                }
            };
            static_for<HS>([&](auto K) {
                constexpr int k = A1_CV(K);
                double xs[NS];
                xs[0] = xh[k];
                xs[1] = twin_exchange(xs[0]);  // xs[0]: step 2k (the main row's) on both rows, xs[1]: step 2k + 1; a quad: rows 0, 2 hold steps 4k, 4k + 1, rows 1, 3 the next two
                if constexpr (QUAD) { xs[2] = quad_exchange(xs[0]); xs[3] = quad_exchange(xs[1]); }
                static_for<NS>([&](auto O) {
                    constexpr int t = NS * k + A1_CV(O);
                    s = row_dpp_ready(opA(s) + Bx(t, row_dpp_ready(xs[O])));
                    sv[t] = q2s * s;
                });
            });
            double lam = row_dpp_ready(0.0);
            auto Btl_ = [&](int t, double lam_ready) {  // (B~_t' lambda) in force layout
                if constexpr (GEN) { double Bq[6]; Bt_at(t, Bq); return dot_bc<6>(Bq, lam_ready); }
                else return BtT(lam_ready);
            };
            static_for<HS>([&](auto KK) {
                constexpr int k = HS - 1 - A1_CV(KK);
                double bs[NS];
                static_for<NS>([&](auto OO) {
                    constexpr int o = NS - 1 - A1_CV(OO);
                    lam = row_dpp_ready(sv[NS * k + o] + opAT(lam));
                    bs[o] = Btl_(NS * k + o, lam);
                });
                double bo = twin ? bs[1] : bs[0];
                if constexpr (QUAD) { if (own >= 2) bo = twin ? bs[3] : bs[2]; }
                Pu[k] = fma(r2a, xh[k], bo);
            });
        } else {   // P u = B_qp' Q (B_qp u) + R u : roll-out, then adjoint
            double sv[H];
            double s = row_dpp_ready(0.0);
            static_for<H>([&](auto T) {
                // B~ u with my row of B~ from registers (loaded by factorize) where it is kept there
                if constexpr (kBrowInRegs) s = row_dpp_ready(opA(s) + dot_bc<0>(Brw, row_dpp_ready(xh[T])));
                else if constexpr (GEN) {
                    double Br[12];
                    const double* br = brow_at(A1_CV(T));
#pragma unroll
                    for (int b = 0; b < 12; ++b) Br[b] = br[b];
                    s = row_dpp_ready(opA(s) + dot_bc<0>(Br, row_dpp_ready(xh[T])));
                } else s = row_dpp_ready(opA(s) + Bu(row_dpp_ready(xh[T])));
                sv[T] = q2s * s;
            });
            double lam = row_dpp_ready(0.0);
            static_for<H>([&](auto TT) {
                constexpr int t = H - 1 - A1_CV(TT);
                lam = row_dpp_ready(sv[t] + opAT(lam));
                if constexpr (GEN) { double Bq[6]; Bt_at(t, Bq); Pu[t] = fma(r2a, xh[t], dot_bc<6>(Bq, lam)); }
                else Pu[t] = fma(r2a, xh[t], BtT(lam));
            });
        }
        // Norms of the scaled vectors (E r, D^-1 r: they only feed the rho estimate) are accumulated as squares -- E^2 = rr / rho_row and
        // D^-2 = 1 / dI2 are at hand, E and D^-1 would cost a square root and a division per row and step -- and rooted once per norm.
        double m_pri = 0, m_upri = 0, m_z = 0, m_uz = 0, m_Ax = 0, m_uAx = 0;
        double m_dua = 0, m_udua = 0, m_q = 0, m_uq = 0, m_Aty = 0, m_uAty = 0, m_Px = 0, m_uPx = 0;
        const double irho = one_c / rho_c, irho_eq = one_c / (kRhoEqOverIneq * rho_c);
        static_for<HS>([&](auto T) {
            constexpr int k = A1_CV(T);            // slot
            const int t = NS * k + own;   // its horizon step
            const double uz = quad_perm<2, 2, 2, 2>(xh[k]);
            const double ax0 = comp == 2 ? xh[k] : fma(mu, uz, xh[k]);  // E^-1 (A_s x)
            const double ax1 = comp < 2 ? fma(-mu, uz, xh[k]) : 0.0;
            const double z0 = clamp_f64(wh0[k], lbs<k>(t), ubs<k>(t)), z1 = comp < 2 ? min_f64(wh1[k], 0.0) : 0.0;  // E^-1 z (fz lanes: no second row, wh1 is a don't-care there)
            const double rp0 = ax0 - z0, rp1 = ax1 - z1;
            const bool eq = (eqmask >> t) & 1u;
            const double e0 = rr0[k] * (eq ? irho_eq : irho), e1 = rr1[k] * irho;  // E^2
            m_upri = max_f64(m_upri, max_f64(fabs(rp0), fabs(rp1)));
            m_pri = max_f64(m_pri, max_f64(e0 * (rp0 * rp0), e1 * (rp1 * rp1)));
            m_uz = max_f64(m_uz, max_f64(fabs(z0), fabs(z1)));
            m_z = max_f64(m_z, max_f64(e0 * (z0 * z0), e1 * (z1 * z1)));
            m_uAx = max_f64(m_uAx, max_f64(fabs(ax0), fabs(ax1)));
            m_Ax = max_f64(m_Ax, max_f64(e0 * (ax0 * ax0), e1 * (ax1 * ax1)));
            // D^-1 (A_s' y_s) = A' (E y_s) = A' [rr (wh - Pi(wh))]
            const double w0 = rr0[k] * (wh0[k] - z0), w1 = rr1[k] * (wh1[k] - z1);
            const double sm = w0 - w1;
            const double smx = quad_perm<0, 0, 0, 0>(sm), smy = quad_perm<1, 1, 1, 1>(sm);
            const double aty_u = comp == 2 ? fma(mu, smx + smy, w0) : w0 + w1;
            const double cgt = act ? lds[L::CG + t * 12 + ci] : 0.0;
            const double px_u = csc * Pu[k];        // = D^-1 (P_s x_s)
            // D^-1 (P_s x_s + q_s) : re-evaluated here, or -- while rho is small and the re-evaluation would mostly measure the backward error
            // of the Riccati solves -- the value carried through the x-update identity by the iterations (G, in the unused K_0 slot)
            double* G = lds + L::FAC + t * L::SLOT + ci * L::KSTR + L::GCOL;
            double pq_u = px_u + cgt;
            if (careful) pq_u = act ? *G : 0.0;
            else if (act) *G = pq_u;
            const double rd_u = pq_u + aty_u;        // = D^-1 (P_s x_s + q_s + A_s' y_s)
            const double D2 = one_c / dI2[k];        // D^2
            m_udua = max_f64(m_udua, fabs(rd_u));
            m_dua = max_f64(m_dua, D2 * (rd_u * rd_u));
            m_uq = max_f64(m_uq, fabs(cgt));
            m_q = max_f64(m_q, D2 * (cgt * cgt));
            m_uAty = max_f64(m_uAty, fabs(aty_u));
            m_Aty = max_f64(m_Aty, D2 * (aty_u * aty_u));
            m_uPx = max_f64(m_uPx, fabs(px_u));
            m_Px = max_f64(m_Px, D2 * (px_u * px_u));
        });
        info.pri_res = pair_allmax(act ? m_upri : 0.0);
        info.nEz = pair_allmax(act ? m_uz : 0.0);
        info.nEAx = pair_allmax(act ? m_uAx : 0.0);
        info.s_pri = sqrt(pair_allmax(act ? m_pri : 0.0));
        info.s_z = sqrt(pair_allmax(act ? m_z : 0.0));
        info.s_Ax = sqrt(pair_allmax(act ? m_Ax : 0.0));
        info.dua_res = cinv * pair_allmax(act ? m_udua : 0.0);
        info.nDq = pair_allmax(act ? m_uq : 0.0);
        info.nDAty = pair_allmax(act ? m_uAty : 0.0);
        info.nDPx = pair_allmax(act ? m_uPx : 0.0);
        info.s_dua = sqrt(pair_allmax(act ? m_dua : 0.0));
        info.s_q = sqrt(pair_allmax(act ? m_q : 0.0));
        info.s_Aty = sqrt(pair_allmax(act ? m_Aty : 0.0));
        info.s_Px = sqrt(pair_allmax(act ? m_Px : 0.0));
    }
    // auxil.c check_termination (feasibility certificates are not evaluated: u = 0 is always feasible and
    // P > 0, so this QP family is never primal or dual infeasible)
    A1_DEV bool check_termination(bool approximate) {
        double ea = P.eps_abs, er = P.eps_rel;
        if (!(info.pri_res <= kInfty) || !(info.dua_res <= kInfty)) { status = A1MPC_NON_CVX; return true; }
        if (approximate) { ea *= 10; er *= 10; }
        const bool prc = info.pri_res < ea + er * fmax(info.nEz, info.nEAx);
        const bool drc = info.dua_res < ea + er * cinv * fmax(fmax(info.nDq, info.nDAty), info.nDPx);
        if (prc && drc) { status = approximate ? A1MPC_SOLVED_INACCURATE : A1MPC_SOLVED; return true; }
        return false;
    }

    // ================================================================================ one segment of osqp_solve:
    // (re-)factorise if needed, iterate up to the next checkpoint (a multiple of check_termination / adaptive_rho_interval, or
    // max_iter), then the residual check and the rho update.  Everything that can differ between the rows of a wave
    // (termination, rho update) happens at segment boundaries, so rows that run advance() in lock-step stay aligned.
    template <bool UPD = false>
    A1_DEV void advance() {
#ifdef A1X_CLK
        const long long tf_ = clock64();
#endif
        [[maybe_unused]] long long pf0_ = 0;
        if constexpr (CLK) pf0_ = row_clock();
        if (need_factor) factorize();
        if constexpr (CLK) pfX += row_clock() - pf0_;
#ifdef A1X_CLK
        clkX += clock64() - tf_;
#endif
        if (!done) {
            int next = P.max_iter;
            if (P.check_every > 0) next = imin(next, (iter / P.check_every + 1) * P.check_every);
            if (P.adaptive_rho && P.adaptive_rho_every > 0) next = imin(next, (iter / P.adaptive_rho_every + 1) * P.adaptive_rho_every);
            if (iter == 0 && first_special) {
                admm_iteration<true>(); iter = 1;
                if constexpr (UPD && H > 1 && !SETUP_ONLY && MODE == kModeMpc) restore_cg();  // (unconditional in the UPD instantiations: a flag would have to live across the ADMM loop)
            }
#ifdef A1X_POISON  // test build (tests/test_emu_parity.py): the values nothing may depend on -- wh1 of the fz lanes (no second constraint row: any FINITE value), every
                   // per-step register of the pad lanes (anything) -- are overwritten at every segment start; not one output bit may change
            static_for<HS>([&](auto K) {
                if (act && comp == 2) wh1[K] = (A1_CV(K) & 1) ? 1e300 : -1e300;
                if (!act) { xh[K] = nan(""); wh0[K] = nan(""); wh1[K] = nan(""); }
            });
#endif
            // the rows of a wave share one instruction stream: if any of them carries G, all run that variant (a harmless extra for the others)
#ifdef A1X_CLK
            const long long t0_ = clock64();
#endif
            [[maybe_unused]] long long pf1_ = 0;
            if constexpr (CLK) pf1_ = row_clock();
            if (row_wave_any(careful)) {
                for (int k = iter; k < next; ++k) admm_iteration<false, true>();
            } else {
                for (int k = iter; k < next; ++k) admm_iteration<false, false>();
            }
#ifdef A1X_CLK
            clkT += clock64() - t0_;
#endif
            if constexpr (CLK) pfT += row_clock() - pf1_;
            iter = next;
            const bool can_check = P.check_every > 0 && (iter % P.check_every) == 0;
            const bool do_rho = P.adaptive_rho && P.adaptive_rho_every > 0 && (iter % P.adaptive_rho_every) == 0;
            const bool last = iter >= P.max_iter;
#ifdef A1X_CLK
            const long long t1_ = clock64();
#endif
            [[maybe_unused]] long long pf2_ = 0;
            if constexpr (CLK) pf2_ = row_clock();
            update_info();
            if constexpr (CLK) pfU += row_clock() - pf2_;
#ifdef A1X_CLK
            clkU += clock64() - t1_;
#endif
            if (can_check && check_termination(false)) {
                done = true;
            } else {
                if (do_rho) {  // auxil.c compute_rho_estimate / adapt_rho
                    const double pr = info.s_pri / (fmax(info.s_z, info.s_Ax) + 1e-10);
                    const double dr = info.s_dua / (fmax(fmax(info.s_q, info.s_Aty), info.s_Px) + 1e-10);
                    const double rn = fmin(fmax(rho * sqrt(pr / (dr + 1e-10)), kRhoMin), kRhoMax);
                    if (rn > rho * P.adaptive_rho_tol || rn < rho / P.adaptive_rho_tol) {
                        // y_s = rho E (wh - zh) is kept:  wh <- zh + (rho_old / rho_new) (wh - zh),  rr <- rr rho_new / rho_old
                        const double up = rn / rho, dn = rho / rn;
                        static_for<HS>([&](auto T) {
                            const int t_ = NS * A1_CV(T) + own;
                            const double z0 = fmin(fmax(wh0[T], lbs<A1_CV(T)>(t_)), ubs<A1_CV(T)>(t_)), z1 = fmin(wh1[T], 0.0);
                            wh0[T] = fma(dn, wh0[T] - z0, z0);
                            wh1[T] = fma(dn, wh1[T] - z1, z1);
                            rr0[T] *= up;
                            rr1[T] *= up;
                        });
                        rho = rn;
                        need_factor = true;
                        careful = rho <= kRhoCareful;  // G holds this checkpoint's evaluation (written by update_info above) or the carried value
                    }
                }
                if (last) {
                    if (!(!can_check && check_termination(false)) && !check_termination(true)) status = A1MPC_MAX_ITER_REACHED;
                    if (need_factor) ++nfact;  // the reference refactors before it notices the iteration limit
                    done = true;
                }
            }
        }
    }
    // fused driver: every loop leaves through its latch only (a divergent exit from the middle of a body makes the compiler
    // copy every live-out vector on every iteration)
    template <bool UPD = false>
    A1_DEV void solve() {
        do { advance<UPD>(); } while (!done);
    }

    // ================================================================================ store_solution + first-step GRFs in the body frame
    // carry (update path, warm_start = 2): this QP's Carry<H> record, or null -- the unscaled z = Pi(w) of the solve goes there
    A1_DEV void write_outputs(const ProblemIO& io, double* __restrict__ carry = nullptr) const {
        // a non-finite solution (NaN / Inf inputs) is reported as NON_CVX whatever the residual tests concluded: max-norms
        // skip NaNs, so OSQP's own termination test can "converge" on a NaN iterate
        double nf = 0.0;
#pragma unroll
        for (int t = 0; t < HS; ++t) nf = (xh[t] - xh[t] == 0.0) ? nf : 1.0;
        const int32_t status_out = pair_allmax(act ? nf : 0.0) > 0.0 ? A1MPC_NON_CVX : status;   // (pad lanes hold no force: what they carry is nobody's business)
        const bool nanout = status_out == A1MPC_NON_CVX;
        const double nanv = nan("");
        {
            const double f = nanout ? nanv : xh[0];
            const double f0 = quad_perm<0, 0, 0, 0>(f), f1 = quad_perm<1, 1, 1, 1>(f), f2 = quad_perm<2, 2, 2, 2>(f);
            const bool bad = (f0 != f0) || (f1 != f1) || (f2 != f2);
            double gout = 0.0;
            if (wr) {  // R' f (S/A1RobotControl.cpp:558-561); non-finite solution -> zeros + status
                const double gb = io.R[0 * 3 + comp] * f0 + io.R[1 * 3 + comp] * f1 + io.R[2 * 3 + comp] * f2;
                gout = bad ? 0.0 : gb;
                io.grf[3 * quad + comp] = gout;
            }
            if constexpr (MODE == kModeMpc && !GEN && H > 1) {
                // N3 in the output stage (a1mpc_control_tick_device): the joint torques of my leg from the GRF this lane has just written -- the three lanes
                // of a leg evaluate the leg's torque (each all of it, from the leg's three force components) and store their own component; a null tq_tau compiles to nothing
                // in the persistent rows (make_io leaves it null) and is one uniform branch in the fused / latency kernels
                if (io.tq_tau != nullptr) {
                    const double g0 = quad_perm<0, 0, 0, 0>(gout), g1 = quad_perm<1, 1, 1, 1>(gout), g2 = quad_perm<2, 2, 2, 2>(gout);
                    if (wr) {
                        double* out = io.tq_tau + 3 * quad + comp;
                        if (!*io.tq_active) {                                          // :294-295
                            *out = 0.0;
                        } else {
                            double tq[3];
                            const double* fk = io.tq_fkin + 3 * quad;
                            leg_joint_torque(io.tq_J + 9 * quad, io.contact[quad] != 0, g0, g1, g2, io.tq_km[0] * fk[0], io.tq_km[1] * fk[1], io.tq_km[2] * fk[2], tq);
                            const double v = tq[comp == 0 ? 0 : (comp == 1 ? 1 : 2)] + io.tq_tg[3 * quad + comp];   // :311
                            if (!(v != v)) *out = v;                                   // :314-317 (a NaN keeps the previous value)
                        }
                    }
                }
            }
        }
        static_for<HS>([&](auto T) {
            constexpr int k = A1_CV(T);
            const int t = NS * k + own;  // (a twin pair / a quad: each row writes its own steps)
            if (act) {
                const double xu = nanout ? nanv : xh[k];
                if (io.u_full) io.u_full[t * 12 + ci] = xu;
                // A failed solve must not poison the carried workspace (every later tick of this robot would start from NaN): the next
                // tick starts from cold iterates -- x = y = 0 (OSQP's store_solution() cold-starts its iterates after a failed solve) -- with the rho
                // the solver had reached, like OSQP (below)
                if (io.warm_x) io.warm_x[t * 12 + ci] = nanout ? 0.0 : xu;
                if (io.warm_y) {  // y = c^-1 E y_s = c^-1 rr (wh - Pi(wh))
                    const double z0 = fmin(fmax(wh0[k], lbs<k>(t)), ubs<k>(t)), z1 = fmin(wh1[k], 0.0);
                    io.warm_y[t * 20 + 5 * quad + r0] = nanout ? 0.0 : cinv * rr0[k] * (wh0[k] - z0);
                    if (comp < 2) io.warm_y[t * 20 + 5 * quad + r1] = nanout ? 0.0 : cinv * rr1[k] * (wh1[k] - z1);
                    if constexpr (H > 1 && MODE == kModeMpc) {
                        if (carry) {  // z_s / E = Pi(w / E): what OSQP leaves in work->z (zeros after a failed solve: cold_start)
                            carry[Carry<H>::Z0 + t * 12 + ci] = nanout ? 0.0 : z0;
                            if (comp < 2) carry[Carry<H>::Z1 + t * 12 + ci] = nanout ? 0.0 : z1;
                        }
                    }
                }
            }
        });
#ifdef A1X_CLK
        if (lead() && io.u_full) { io.u_full[0] = double(clkB); io.u_full[12] = double(clkF); io.u_full[24] = double(clkT); io.u_full[36] = double(clkU); io.u_full[48] = double(clkX); }
#endif
        if (lead()) {
            if (io.iters) *io.iters = iter;
            if (io.status) *io.status = status_out;
            if (io.nfact) *io.nfact = nfact;
            if (io.rho_io) *io.rho_io = rho;   // also after a failed solve: OSQP's cold_start() zeroes x, z, y and leaves the rho it had adapted in settings->rho (round 4; until then 0 = "settings.rho")
        }
    }
    // profiling instantiations of the fused / latency kernels: this QP's stage record (kTickStages cycles, layout kClk*) from the driver's stamps --
    // c0 entry | cF formation done | cR Ruiz passes done | c1 hot state + hand-off done | c2 solve() returned | c3 outputs written.  The iterations' share is what
    // solve() spent outside factor passes and residual checks (OSQP's first iteration and the loop glue included).
    A1_DEV void store_tick_stages(long long* __restrict__ clk, long long c0, long long cF, long long cR, long long c1, long long c2, long long c3) const {
        if (clk == nullptr || !lead()) return;
        clk[0] = cF - c0; clk[1] = cR - cF; clk[2] = c1 - cR; clk[3] = pfX; clk[4] = (c2 - c1) - pfX - pfU; clk[5] = pfU; clk[6] = c3 - c2; clk[7] = c3 - c0;
    }
};

// arguments of a batch launch (device pointers; per-QP records are contiguous, QP b at base + b * record size)
struct BatchArgs {
    DeviceParams P;
    const double* tab;
    int32_t n;
    const double *root_acc, *Rz;  // balance mode only
    const double* tick;           // n x 22 compact tick records, or null (N1)
    const double *x0, *xref, *R, *foot;
    const uint8_t* contact;
    const double* yaw_A;                  // general path only: n yaws for A_c, or null
    int32_t foot_stride, contact_stride;  // general path only (0 / 12 doubles and 0 / 4 bytes per horizon step): per-QP records are then 12H doubles / 4H bytes
    double *grf, *u_full, *warm_x, *warm_y, *rho;
    int32_t *iters, *status, *nfact;
    // work-queue order of the persistent ADMM rows (null = index order) and the per-QP cost record that the next solve's order is
    // built from (null = not recorded); scheduling only -- no result depends on either
    const int32_t* order;
    int32_t* cost;
    int32_t predict;  // the set-up kernel writes its cost guess to `cost` (first solve of a batch: no history to order the queue by)
    double* carry;    // warm_start = 2 (update path): n x Carry<H>::STRIDE, or null
    long long* clk;   // profiling instantiations (a1mpc_set_profiling): n x kTickStages shader-clock cycles per QP (layout below), or null
    // N3 in the output stage of the fused / latency kernels (a1mpc_control_tick_device), or tq_tau = null: per-robot records like the arrays above
    const uint8_t* tq_active;
    const double *tq_J, *tq_fkin, *tq_tg, *tq_km;
    double* tq_tau;
};
// per-QP stage record of the profiling instantiations: the split pipeline's persistent rows fill FACTOR / ITER / CHECK, the fused and the latency kernel all of it
enum : int { kClkForm = 0, kClkRuiz = 1, kClkHandoff = 2, kClkFactor = 3, kClkIter = 4, kClkCheck = 5, kClkOut = 6, kClkTotal = 7, kTickStages = 8 };
template <int H, int MODE>
A1_DEV ProblemIO make_io(const BatchArgs& a, int64_t b) {
    ProblemIO io;
    io.root_acc = MODE == kModeBalance ? a.root_acc + b * 6 : nullptr;
    io.Rz = MODE == kModeBalance ? a.Rz + b * 9 : nullptr;
    io.tick = (MODE == kModeMpc && a.tick) ? a.tick + b * 22 : nullptr;
    io.x0 = (MODE == kModeMpc && a.x0) ? a.x0 + b * 13 : nullptr;
    io.xref = (MODE == kModeMpc && a.xref) ? a.xref + b * 13 * H : nullptr;
    io.R = a.R + b * 9;
    io.foot = a.foot + b * 12;
    io.contact = a.contact + b * 4;
    // foot_stride / contact_stride / yaw_A are only read by RowSolver<..., GEN = true> and set by make_io_gen() below (deliberately not here)
    io.grf = a.grf + b * 12;
    io.u_full = a.u_full ? a.u_full + b * 12 * H : nullptr;
    io.warm_x = a.warm_x ? a.warm_x + b * 12 * H : nullptr;
    io.warm_y = a.warm_y ? a.warm_y + b * 20 * H : nullptr;
    io.rho_io = a.rho ? a.rho + b : nullptr;
    io.iters = a.iters ? a.iters + b : nullptr;
    io.status = a.status ? a.status + b : nullptr;
    io.nfact = a.nfact ? a.nfact + b : nullptr;
    io.carry = nullptr;  // (set by make_io_sched for the set-ups; the ADMM side gets the pointer where it writes, see carry_of)
    io.cost = nullptr; io.tq_tau = nullptr;   // (make_io_sched: the fused / latency kernels only -- the persistent rows keep the code they had)
    io.tq_active = nullptr; io.tq_J = nullptr; io.tq_fkin = nullptr; io.tq_tg = nullptr; io.tq_km = nullptr;
    return io;
}
template <int H>
A1_DEV double* carry_of(const BatchArgs& a, int64_t b) { return a.carry ? a.carry + b * Carry<H>::STRIDE : nullptr; }

// The fast path with a per-step contact schedule (contact_stride = 4; feet step-invariant): contacts only change the bounds and which rows are
// equalities, so every kernel of the fast path takes them -- the set-ups read the schedule, the hand-off record carries the contact bits.
template <int H, int MODE>
A1_DEV ProblemIO make_io_sched(const BatchArgs& a, int64_t b) {
    ProblemIO io = make_io<H, MODE>(a, b);
    io.contact = a.contact + b * (a.contact_stride ? 4 * H : 4);
    io.foot_stride = 0; io.contact_stride = a.contact_stride; io.yaw_A = nullptr;
    io.carry = carry_of<H>(a, b);
    io.cost = a.cost ? a.cost + b : nullptr;
    if (a.tq_tau != nullptr) {
        io.tq_tau = a.tq_tau + b * 12; io.tq_active = a.tq_active + b; io.tq_J = a.tq_J + b * 36; io.tq_fkin = a.tq_fkin + b * 12; io.tq_tg = a.tq_tg + b * 12; io.tq_km = a.tq_km;
    }
    return io;
}

// The general path's records (per-step feet / contacts, A_c yaw).  A separate function on purpose: make_io() is inlined into the persistent ADMM
// kernel, whose register allocation is sensitive to every instruction around the hot loop -- with the stride selects inside make_io() the
// same hot loop came out 7 % slower per iteration (tools/ab_kernels.sh: 10.3 -> 10.8 ms at 65 536 QPs).
template <int H>
A1_DEV ProblemIO make_io_gen(const BatchArgs& a, int64_t b) {
    ProblemIO io = make_io<H, kModeMpc>(a, b);
    io.foot = a.foot + b * (a.foot_stride ? 12 * H : 12);
    io.contact = a.contact + b * (a.contact_stride ? 4 * H : 4);
    io.foot_stride = a.foot_stride; io.contact_stride = a.contact_stride; io.yaw_A = a.yaw_A ? a.yaw_A + b : nullptr;
    io.carry = carry_of<H>(a, b);   // (warm_start = 2 on the general path: the fused general kernels, round 5)
    return io;
}

// split pipeline, kernel 1: formation + Ruiz for QP b, prepared state to global memory
// rows of a wavefront that share one QP in the general path's set-up kernel: 2 (two QPs per wavefront) at H = 10, 4 (one QP per wavefront) at H = 16 / 20 -- the per-lane state of
// the Ruiz passes then fits 256 registers and two wavefronts share a SIMD (a lone wavefront issues an FP64 instruction every 2.13 ns, two every 1.84: tools/ubench/f64_rate_ubench.hip)
constexpr int setup_gen_rows(int h) { return h >= 12 ? 4 : 2; }   // (12, 14: the extended horizons -- with two rows their set-up kernels spill, 84 / 196 B)
template <int H, bool GEN = false, bool UPD = false>
A1_DEV void setup_row(const BatchArgs& a, const double* __restrict__ tab, int64_t b, double* __restrict__ lds, double* __restrict__ prep) {
    RowSolver<H, kModeMpc, true, GEN> S(a.P, tab, lds);  // tab: the (alpha/beta, beta) table, staged in LDS by the kernel
    if constexpr (GEN) {
        // general path: rows r and r + 2 of the wavefront (H = 10; all four rows at H = 16 / 20: setup_gen_rows) share this QP's set-up -- each owns every N-th horizon step of the Ruiz passes, RowSolver::setup
        constexpr int N = setup_gen_rows(H);
        S.coop_id = N == 4 ? 2 * (row_is_twin() ? 1 : 0) + row_sub() : (row_is_twin() ? 1 : 0); S.coop_n = N;
        S.gen_rec = prep + b * Prep<H>::STRIDE_GEN;
        S.template setup<false, N>(make_io_gen<H>(a, b));
        if (S.coop_id == 0) S.template save_prepared<false>(prep + b * Prep<H>::STRIDE_GEN);
    } else {
        S.template setup<UPD>(make_io_sched<H, kModeMpc>(a, b));
        S.template save_prepared<UPD>(prep + b * Prep<H>::STRIDE);
    }
    if (a.predict && a.cost != nullptr && S.ln == 0) a.cost[b] = S.pred_cost;
}

// split pipeline, kernel 2: a persistent row.  It pulls prepared QPs from a shared counter and advances them one
// checkpoint-aligned segment per loop trip; a row whose QP has converged writes it out and pulls the next one at the next
// trip, so the rows of a wave never wait for each other's iteration counts -- only for each other's (rare) re-factorisations.
template <int H, bool TWIN = false, bool GEN = false, bool UPD = false, bool UNI = false, bool CLK = false, bool QUAD = false>
A1_DEV void admm_rows(const BatchArgs& a, const double* __restrict__ prep, int* __restrict__ counter, double* __restrict__ lds) {
    RowSolver<H, kModeMpc, false, GEN, TWIN, UNI, CLK, QUAD> S(a.P, a.tab, lds);
    bool alive = true, need_new = true, have = false;
    int64_t cur = 0;
    while (alive) {
        if (need_new) {
            if (have) {
                if constexpr (UPD) S.write_outputs(make_io<H, kModeMpc>(a, cur), carry_of<H>(a, cur));
                else S.write_outputs(make_io<H, kModeMpc>(a, cur));
                if (a.cost != nullptr && S.lead()) a.cost[cur] = S.iter + 10 * S.nfact;  // ~ ADMM-iteration equivalents (a factor pass ~ 10)
                if constexpr (CLK) { if (a.clk != nullptr && S.lead()) { a.clk[cur * kTickStages + kClkFactor] = S.pfX; a.clk[cur * kTickStages + kClkIter] = S.pfT; a.clk[cur * kTickStages + kClkCheck] = S.pfU; } }
            }
            double v = 0.0;
            if (S.lead()) {
                const int q = row_atomic_inc(counter);
                v = static_cast<double>((q < a.n && a.order != nullptr) ? a.order[q] : q);
            }
            v = row_bcast<0>(v);
            if constexpr (TWIN) v = twin_from_main(v);  // the twin row works on its main row's QP
            if constexpr (QUAD) (void)quad_exchange(v);  // ... and rows 1, 3 on row 0's
            cur = static_cast<int64_t>(v);
            if (cur >= a.n) {
                alive = false;
            } else {
                if constexpr (GEN) S.template load_prepared<false>(prep + cur * Prep<H>::STRIDE_GEN, make_io<H, kModeMpc>(a, cur), true);
                else S.template load_prepared<UPD>(prep + cur * Prep<H>::STRIDE, make_io<H, kModeMpc>(a, cur));
                have = true;
                need_new = false;
            }
        }
        if (alive) {
            S.template advance<UPD>();
            need_new = S.done;
        }
    }
}

// The fused / latency kernels' copy of the batch's (alpha / beta, beta) table inside the QP's LDS image (Layout::TAB, in the still empty factor region): the Ruiz sweeps read
// a column of it per visited block column, and from global memory that was one exposed round trip per column -- a lone wave per SIMD hides nothing (round 5: ~25 k of
// the ~100 k cycles the ten passes took in a warm-started tick).  The `parts` rows that share the image split the copy; all loads are issued before the first store.
template <int H, class LAY>
A1_DEV const double* stage_table(const double* __restrict__ tab, double* __restrict__ lds, int part, int parts) {
    constexpr int N2 = 2 * H * H, PER = (N2 + 15) / 16;
    double v[PER];
    const int l0 = row_lane() + 16 * part, step = 16 * parts;
#pragma unroll
    for (int k = 0; k < PER; ++k) { const int i = l0 + step * k; v[k] = (k * parts < PER && i < N2) ? tab[i] : 0.0; }
#pragma unroll
    for (int k = 0; k < PER; ++k) { const int i = l0 + step * k; if (k * parts < PER && i < N2) lds[LAY::TAB + i] = v[k]; }
    return lds + LAY::TAB;   // (the caller's first set_sync() orders the stores against the sweeps' reads)
}

// the fused path: one QP from inputs to outputs (small batches, the batch-1 latency path, the CPU test double).
// make_io() is called where the pointers are needed (set-up, hand-off, outputs) instead of once: a ProblemIO of per-row pointers that
// stays live across the ADMM loop costs that loop ~30 VGPRs it does not have.
// TWIN: the calling row is one of a main / twin pair (rows r and r + 2 of the wavefront, see RowSolver<.., TWIN>): the main row sets the QP up alone,
// both iterate.
// UPD: the instantiation that also serves warm_start = 2 (the reference's update path); UPD = false is the code of every other mode, as it was before that path existed
// CLK (fast path, H > 1): the profiling instantiation -- `clk` = this QP's kTickStages stage record (a1mpc_last_tick_stage_cycles); same arithmetic, same bits
template <int H, int MODE, bool GEN = false, bool TWIN = false, bool UPD = false, bool QUAD = false, bool CLK = false, class MakeIO>
A1_DEV void solve_row_with(const DeviceParams& P, const double* __restrict__ tab, MakeIO&& make_io_, double* __restrict__ lds, [[maybe_unused]] long long* __restrict__ clk = nullptr) {
    static_assert(!CLK || (!GEN && MODE == kModeMpc && Prep<H>::STRIDE <= H * Layout<H>::SLOT), "the profiling instantiation exists for the fast path's set-up | iteration hand-off");
    static_assert(!TWIN || (MODE == kModeMpc && Prep<H>::STRIDE <= H * Layout<H>::SLOT), "twin rows: the MPC solve with the set-up | iteration hand-off");
    static_assert(!QUAD || TWIN, "a quad of rows: a twin pair doubled");
    if constexpr (GEN) {
        // general path (per-step feet / contact schedules): the same set-up | iteration hand-off as below (its per-step tables live behind c*g
        // in the LDS image and survive it; the T*B~w table aliased into the factor region is dead once the Ruiz passes are done)
        static_assert(Prep<H>::STRIDE <= H * Layout<H, true>::SLOT, "the hand-off record fits the (still empty) factor region");
        {   // a main / twin pair shares the set-up like the rows of the latency kernel do: each takes every other column of the Ruiz sweeps and every
            // other horizon step of the D / E updates, everything else is computed redundantly (same values, same LDS image)
            int cid = 0, cn = 1;
            if constexpr (TWIN) { cid = row_is_twin() ? 1 : 0; cn = 2; }
            if constexpr (QUAD) { cid = 2 * cid + row_sub(); cn = 4; }   // (a quad: the four rows)
            RowSolver<H, MODE, false, true> S0(P, stage_table<H, Layout<H, true>>(tab, lds, cid, cn), lds);
            S0.coop_id = cid; S0.coop_n = cn;
            S0.template setup<UPD, QUAD ? 4 : (TWIN ? 2 : 1)>(make_io_());
            if constexpr (TWIN) pair_sync(); else row_sync();  // everybody is done with the set-up scratch aliased into the factor region
            if ((!TWIN || !row_is_twin()) && (!QUAD || row_sub() == 0)) S0.template save_prepared<UPD>(lds + Layout<H, true>::FAC);
        }
        if constexpr (TWIN) pair_sync();  // the twin reads the hand-off record and the per-step tables its main row wrote
        RowSolver<H, MODE, false, true, TWIN, false, false, QUAD> S(P, tab, lds);
        S.template load_prepared<UPD>(lds + Layout<H, true>::FAC, make_io_());
        S.template solve<UPD>();
        if constexpr (UPD) { const ProblemIO& io_ = make_io_(); S.write_outputs(io_, io_.carry); }
        else S.write_outputs(make_io_());
    } else if constexpr (MODE == kModeMpc && Prep<H>::STRIDE <= H * Layout<H>::SLOT) {
        // Set-up and iteration are two solver objects joined by the hand-off record of the split pipeline, staged in the (still
        // empty) factor region: the ADMM loop then gets the register allocation of the persistent kernel instead of one that
        // also carries the set-up's live values (scratch reloads inside the loop).  ~0.5 us per solve.
        [[maybe_unused]] long long c0 = 0, cF = 0, cR = 0, c1 = 0, c2 = 0;
        if constexpr (CLK) c0 = row_clock();
        {   // (a main / twin pair shares the set-up: see the general path above)
            int cid = 0, cn = 1;
            if constexpr (TWIN) { cid = row_is_twin() ? 1 : 0; cn = 2; }
            if constexpr (QUAD) { cid = 2 * cid + row_sub(); cn = 4; }   // (a quad: the four rows, like the latency kernel's)
            RowSolver<H, MODE, false, false, false, false, CLK> S0(P, stage_table<H, Layout<H>>(tab, lds, cid, cn), lds);
            S0.coop_id = cid; S0.coop_n = cn;
            S0.template setup<UPD>(make_io_());
            if constexpr (TWIN) pair_sync(); else row_sync();
            if ((!TWIN || !row_is_twin()) && (!QUAD || row_sub() == 0)) S0.template save_prepared<UPD>(lds + Layout<H>::FAC);
            if constexpr (CLK) { cF = S0.ckF; cR = S0.ckR; }
        }
        if constexpr (TWIN) pair_sync();  // the twin reads the hand-off record its main row wrote
        RowSolver<H, MODE, false, false, TWIN, false, CLK, QUAD> S(P, tab, lds);
        S.template load_prepared<UPD>(lds + Layout<H>::FAC, make_io_());
        if constexpr (CLK) c1 = row_clock();
        S.template solve<UPD>();
        if constexpr (CLK) c2 = row_clock();
        {
            const ProblemIO& io_ = make_io_();
            if constexpr (UPD) S.write_outputs(io_, io_.carry);
            else S.write_outputs(io_);
            if (io_.cost != nullptr && S.lead()) *io_.cost = S.iter + 10 * S.nfact;   // the next tick's launch order (longest first: a1mpc_solve_kernel)
        }
        if constexpr (CLK) S.store_tick_stages(clk, c0, cF, cR, c1, c2, row_clock());
    } else {
        RowSolver<H, MODE> S(P, tab, lds);
        S.setup(make_io_());
        S.solve();
        S.write_outputs(make_io_());
    }
}
// Latency variant of the general path's fused solve for a horizon at which a wavefront would hold two QPs (H = 10; round 5): a handful of QPs, one per wavefront, whose FOUR
// rows share the set-up -- each takes every fourth block column of the general path's Ruiz sweeps (every column is visited there: the sweeps are most of a general-path tick's
// set-up) and every fourth horizon step of the D / E updates; then rows 1 / 3 retire and rows 0 / 2 solve as a main / twin pair.  Same bits as the fused general kernel, whose
// pairs share the set-up two ways (the column maxima are exact and order-free).  At H = 16 / 20 the fused kernel's quad of rows already is this arrangement.
template <int H, bool UPD = false, class MakeIO>
A1_DEV void solve_latency_gen(const DeviceParams& P, const double* __restrict__ tab, MakeIO&& make_io_, double* __restrict__ lds) {
    static_assert(H % 2 == 0 && H % 4 != 0, "a main / twin pair per QP and no quad: H = 10");
    using LG = Layout<H, true>;
    static_assert(Prep<H>::STRIDE <= H * LG::SLOT, "the hand-off record fits the (still empty) factor region");
    const int cid = 2 * (row_is_twin() ? 1 : 0) + row_sub();
    {
        RowSolver<H, kModeMpc, false, true> S0(P, stage_table<H, LG>(tab, lds, cid, 4), lds);
        S0.coop_id = cid; S0.coop_n = 4;
        S0.template setup<UPD, 4>(make_io_());
        coop_sync();   // everybody is done with the set-up scratch aliased into the factor region
        if (cid == 0) S0.template save_prepared<UPD>(lds + LG::FAC);
    }
    if (row_sub()) return;   // rows 1 and 3 retire
    pair_sync();             // the twin reads the hand-off record and the per-step tables row 0 wrote
    RowSolver<H, kModeMpc, false, true, true> S(P, tab, lds);
    S.template load_prepared<UPD>(lds + LG::FAC, make_io_());
    S.template solve<UPD>();
    if constexpr (UPD) { const ProblemIO& io_ = make_io_(); S.write_outputs(io_, io_.carry); }
    else S.write_outputs(make_io_());
}
template <int H, int MODE = kModeMpc, bool GEN = false, bool UPD = false>
A1_DEV void solve_row(const DeviceParams& P, const double* __restrict__ tab, const ProblemIO& io, double* __restrict__ lds) {
    solve_row_with<H, MODE, GEN, false, UPD>(P, tab, [&]() -> const ProblemIO& { return io; }, lds);
}

}  // namespace a1mpc
