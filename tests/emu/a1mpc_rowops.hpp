// tests/emu/a1mpc_rowops.hpp -- TEST DOUBLE of csrc/a1mpc_rowops.hpp (never shipped, never linked
// into the product library).  It lets the test-suite execute the solver source lane-for-lane on a
// CPU: the 16 lanes of a DPP row are 16 cooperative fibers (ucontext); every cross-lane primitive
// publishes the caller's value, yields until all 16 lanes have published, then reads its source
// lane -- the semantics of row_newbcast / row_ror / quad_perm on gfx950 for fully active rows.
// Between two cross-lane points the fibers run one after the other, so LDS hazards that the
// in-order wavefront would hide (a read that needs a row_sync()) show up as wrong results here.
#pragma once
#include <math.h>
#include <stdint.h>

#define A1_DEV inline

namespace a1mpc {

struct EmuRow;
extern thread_local int emu_lane;
const double* emu_publish(double v);  // returns the 16 published values of this exchange

inline int row_lane() { return emu_lane; }
template <int L>
inline double row_bcast(double v) { return emu_publish(v)[L]; }
template <int N>
inline double row_ror(double v) { return emu_publish(v)[(emu_lane - N) & 15]; }
template <int P0, int P1, int P2, int P3>
inline double quad_perm(double v) {
    const double* p = emu_publish(v);
    const int sel[4] = {P0, P1, P2, P3};
    return p[(emu_lane & ~3) + sel[emu_lane & 3]];
}
template <int L>
inline void fma_bcast(double& acc, double m, double x) { acc = fma(m, emu_publish(x)[L], acc); }
template <int L>
inline void fnma_bcast(double& acc, double m, double x) { acc = fma(-m, emu_publish(x)[L], acc); }
inline double max_f64(double a, double b) { return fmax(a, b); }
inline double min_f64(double a, double b) { return fmin(a, b); }
inline double row_dpp_ready(double x) { return x; }
inline void row_dpp_ready12(double (&)[12]) {}
inline void row_sync() { (void)emu_publish(0.0); }


inline void row_sched_fence() {}
inline double row_opaque(double v) { return v; }


inline int row_atomic_inc(int* p) { return (*p)++; }

}  // namespace a1mpc
