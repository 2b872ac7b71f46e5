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
extern thread_local int emu_lane;        // 0..15, or 0..31 when a main / twin pair of rows is emulated (16..31 = the twin), or 0..63 for a quad (the device's lane order)
extern thread_local int emu_lanes;       // 16 / 32 / 64
double emu_quad_exchange(double& a);     // a = [x | y] on the (even | odd) row of each half of a quad -> a = x on both, returns y
const double* emu_publish(double v);     // returns the 16 published values of MY row of this exchange
double emu_twin_exchange(double& a);     // a = [x | y] on (main | twin) -> a = [x | x], returns [y | y]

inline int row_lane() { return emu_lane & 15; }
template <int L>
inline double row_bcast(double v) { return emu_publish(v)[L]; }
template <int N>
inline double row_ror(double v) { return emu_publish(v)[(emu_lane - N) & 15]; }
template <int P0, int P1, int P2, int P3>
inline double quad_perm(double v) {
    const double* p = emu_publish(v);
    const int sel[4] = {P0, P1, P2, P3};
    return p[((emu_lane & 15) & ~3) + sel[emu_lane & 3]];
}
template <int L>
inline void fma_bcast(double& acc, double m, double x) { acc = fma(m, emu_publish(x)[L], acc); }
template <int L>
inline void fnma_bcast(double& acc, double m, double x) { acc = fma(-m, emu_publish(x)[L], acc); }
template <int L, int Q>
inline void fma_bcast_leg(double& acc, double m, double x) {
    const double v = emu_publish(x)[L];
    if (((emu_lane & 15) >> 2) == Q) acc = fma(m, v, acc);
}
template <int L, int Q>
inline void fnma_bcast_leg(double& acc, double m, double x) {
    const double v = emu_publish(x)[L];
    if (((emu_lane & 15) >> 2) == Q) acc = fma(-m, v, acc);
}
inline double row_rsqrt(double p) { return 1.0 / sqrt(p); }
inline double max_f64(double a, double b) { return fmax(a, b); }
inline double min_f64(double a, double b) { return fmin(a, b); }
inline double clamp_f64(double w, double lb, double ub) { return fmin(fmax(w, lb), ub); }
inline double max_abs_f64(double a, double x) { return fmax(a, fabs(x)); }
inline void max_abs3_f64(double& a, double& b, double x0, double x1, double x2) { a = fmax(fmax(a, fabs(x0)), fabs(x2)); b = fmax(b, fabs(x1)); }
inline double row_dpp_ready(double x) { return x; }
inline void row_dpp_ready12(double (&)[12]) {}
inline void row_sync() { (void)emu_publish(0.0); }



// ---- the Riccati sweep blocks (same operation order as the gfx950 instruction blocks) ----
namespace emu_detail {
constexpr int LANE[12] = {0, 1, 2, 4, 5, 6, 8, 9, 10, 12, 13, 14};
}
inline void sweep_back_rhs(double& r, double& pa, double& pb, double at, double cg, double sd, double xh, double p, const double (&Bt)[6],
                           double gA, double gB, double gC) {
    const double* P_ = emu_publish(p);
    double ra = fma(sd, xh, at - cg), rb = 0.0;  // e in the first accumulator, the second starts from zero (like the gfx950 block)
    pa = p;
    ra = fma(-P_[8], Bt[0], ra); rb = fma(-P_[9], Bt[1], rb); pa = fma(P_[0], gA, pa);
    ra = fma(-P_[10], Bt[2], ra); rb = fma(-P_[12], Bt[3], rb);
    ra = fma(-P_[13], Bt[4], ra); rb = fma(-P_[14], Bt[5], rb);
    r = ra + rb;
    pb = fma(P_[1], gB, pb); pa = fma(P_[2], gC, pa);
}
inline void sweep_back_chains(double& d, double& pa, double& pb, double r, const double (&Sr)[12], const double (&Kc)[12]) {
    const double* R_ = emu_publish(r);
    double da = 0.0, db = 0.0;  // even / odd terms, like the gfx950 block
    for (int b = 0; b < 12; ++b) {
        double& dacc = (b & 1) ? db : da;
        dacc = fma(R_[emu_detail::LANE[b]], Sr[b], dacc);
        double& pacc = (b & 1) ? pb : pa;
        pacc = fma(R_[emu_detail::LANE[b]], Kc[b], pacc);
    }
    pa = pa + pb;
    d = da + db;
}
// ---- a main / twin pair of rows (RowSolver<.., TWIN>): the same operation order as the gfx950 blocks
inline bool row_is_twin() { return emu_lanes == 64 ? (emu_lane & 32) != 0 : emu_lane >= 16; }
inline double twin_exchange(double& a) { return emu_twin_exchange(a); }
inline double twin_exchange_copied(double& a, double /*copy of a (the device block's hazard spacing)*/) { return emu_twin_exchange(a); }
inline double twin_from_main(double v) { (void)emu_twin_exchange(v); return v; }
inline void pair_sync() { double z = 0.0; (void)emu_twin_exchange(z); }  // both rows of the pair arrive before either goes on
// quads of rows (RowSolver<.., QUAD>): 64 fibers in the device's lane order
inline int row_sub() { return emu_lanes == 64 ? (emu_lane >> 4) & 1 : 0; }
inline double quad_exchange(double& a) { return emu_quad_exchange(a); }
inline void coop_sync() { pair_sync(); }  // a set-up shared by the rows of a pair or a quad (the exchange waits for every emulated lane)
inline void sweep_back_rhs_twin(double& r, double& pa, double& pb, double p, const double (&Bt)[6], double gA, double gB, double gC, double hm) {
    const double* P_ = emu_publish(p);
    double ra = r, rb = 0.0;
    pa = p * hm;
    ra = fma(-P_[8], Bt[0], ra); rb = fma(-P_[9], Bt[1], rb); pa = fma(P_[0], gA, pa);
    ra = fma(-P_[10], Bt[2], ra); rb = fma(-P_[12], Bt[3], rb);
    ra = fma(-P_[13], Bt[4], ra); rb = fma(-P_[14], Bt[5], rb);
    r = ra + rb;
    pb = fma(P_[1], gB, pb); pa = fma(P_[2], gC, pa);
}
inline void sweep_back_chain_twin(double& pa, double& pb, double r, const double (&M)[12]) {
    const double* R_ = emu_publish(r);
    for (int b = 0; b < 12; ++b) {
        double& acc = (b & 1) ? pb : pa;
        acc = fma(R_[emu_detail::LANE[b]], M[b], acc);
    }
    pa = pa + pb;
    pb = pa;  // (the device block leaves the copy that twin_exchange_copied() swaps with)
}
template <bool SEED>
inline void sweep_fwd_gain_twin(double& v, double& s, const double (&Kr)[12], double fA) {
    const double* S_ = emu_publish(s);
    double va = v, vb = 0.0;
    for (int b = 0; b < 12; b += 2) { va = fma(-S_[emu_detail::LANE[b]], Kr[b], va); vb = fma(-S_[emu_detail::LANE[b + 1]], Kr[b + 1], vb); }
    v = va + vb;   // (pad lanes keep a copy of lane 0's value that nothing reads)
    if (SEED) s = fma(S_[8], fA, s);  // (the device block accumulates the first seed onto s in place; the other two open the input block)
}
template <bool SEED>
inline void sweep_fwd_input_twin(double& sa, double& sb, double v, const double (&Br)[12], double fB, double fC) {
    if (SEED) { const double* S_ = emu_publish(sa); const double s9 = S_[9], s10 = S_[10]; sb = fma(s9, fB, sb); sa = fma(s10, fC, sa); }
    const double* V_ = emu_publish(v);
    for (int b = 0; b < 12; b += 2) { sa = fma(V_[emu_detail::LANE[b]], Br[b], sa); sb = fma(V_[emu_detail::LANE[b + 1]], Br[b + 1], sb); }
    sa = sa + sb;
}
inline double dot12_block(const double (&m)[12], double x) {
    const double* X_ = emu_publish(x);
    double a0 = 0.0, a1 = 0.0;
    for (int b = 0; b < 12; b += 2) { a0 = fma(X_[emu_detail::LANE[b]], m[b], a0); a1 = fma(X_[emu_detail::LANE[b + 1]], m[b + 1], a1); }
    return a0 + a1;
}
template <bool SEED>
inline void sweep_fwd_gain(double& v, double& sa, double& sb, double& xh, double s, const double (&Kr)[12], double fA, double fB, double fC,
                           double am, double oma, double al) {
    const double* S_ = emu_publish(s);
    double va = v, vb = 0.0;
    for (int b = 0; b < 12; b += 2) { va = fma(-S_[emu_detail::LANE[b]], Kr[b], va); vb = fma(-S_[emu_detail::LANE[b + 1]], Kr[b + 1], vb); }
    if (SEED) { sa = s; sa = fma(S_[8], fA, sa); sb = fma(S_[9], fB, sb); sa = fma(S_[10], fC, sa); }
    v = (va + vb) * am;
    xh = fma(al, v, oma * xh);
}
inline void sweep_fwd_input(double& sa, double& sb, double& z0, double v, const double (&Br)[12], double w0, double lb, double ub) {
    const double* V_ = emu_publish(v);
    for (int b = 0; b < 12; b += 2) { sa = fma(V_[emu_detail::LANE[b]], Br[b], sa); sb = fma(V_[emu_detail::LANE[b + 1]], Br[b + 1], sb); }
    sa = sa + sb;
    z0 = fmin(fmax(w0, lb), ub);
}

inline double row_recip(double p) { return 1.0 / p; }  // the device sequence is accurate to ~1 ulp, this is the correctly rounded value
template <int K>
inline void gj_pivot(double (&S)[12], double mlt, double& p, double& x) {
    constexpr int N = K + 1;
    auto elim = [&](int j) { S[j] = fma(emu_publish(S[j])[emu_detail::LANE[K]], mlt, S[j]); };
    if (N < 12) elim(N);
    for (int j = 0; j < 12; ++j)
        if (j != K && j != N) elim(j);
    if (N < 12) {
        p = emu_publish(S[N < 12 ? N : 0])[emu_detail::LANE[N < 12 ? N : 0]];
        x = 1.0 / p;
    }
}

inline void row_sched_fence() {}
inline void row_lds_landed() {}
inline double row_opaque(double v) { return v; }
inline int64_t row_opaque(int64_t v) { return v; }
inline int row_opaque(int v) { return v; }
using lds_cptr = const double*;
template <int OFF>
inline const double* row_lds_at(const double* p) { return p + OFF; }


inline bool row_wave_any(bool p) { return p; }  // one row per emulated wave
inline long long row_clock() { return 0; }   // (no clock on the host fibers: the profiling instantiation is never emulated)
inline int row_atomic_inc(int* p) { return (*p)++; }

}  // namespace a1mpc
