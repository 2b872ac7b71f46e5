// a1mpc_tables.hpp -- host-side tables shared by every QP of a given horizon.
//
// With the reference's step-invariant B_d (S/A1RobotControl.cpp:498-514) and A_d = I + dt*A_c
// (S/ConvexMpc.cpp:149-151; A_c^2 B_d = 0) block (s,t) of B_qp' Q B_qp (S/ConvexMpc.cpp:184-210) is
//     sum_{i=max(s,t)}^{H-1} (A^{i-s} B)' Q (A^{i-t} B) = alpha_st * U + beta_st * V,
//     alpha_st = sum_i (i-s)(i-t),   beta_st = H - max(s,t).
// The kernel wants (alpha/beta, beta) so that one fma gives the entry up to the positive factor beta.
#pragma once

namespace a1mpc {

inline void fill_gamma_beta_table(int H, double* tab /* [H][H][2] */) {
    for (int s = 0; s < H; ++s)
        for (int t = 0; t < H; ++t) {
            const int m = s > t ? s : t;
            long alpha = 0;
            for (int i = m; i < H; ++i) alpha += (long)(i - s) * (i - t);
            const double beta = (double)(H - m);
            tab[(s * H + t) * 2 + 0] = (double)alpha / beta;
            tab[(s * H + t) * 2 + 1] = beta;
        }
}

}  // namespace a1mpc
