// a1mpc_rowops.hpp -- CDNA4 (gfx950) cross-lane primitives for the "one QP per DPP row" solver.
//
// A wavefront (64 lanes) is four DPP rows of 16 lanes; every QP lives in exactly one row, so all
// cross-lane traffic of the solver is row-local DPP (no LDS crossbar, no __shfl):
//   row_bcast<L>   v_mov_b64_dpp ... row_newbcast:L   (lane L of MY row -> every lane of the row)
//   row_rorN       v_mov_b32_dpp x2 ... row_ror:N     (rotation inside the row; only used where
//                                                      the direction does not matter: N=8 and the
//                                                      8/4/2/1 all-reduce)
//   quad_perm      v_mov_b32_dpp x2 ... quad_perm:[..] (a leg's fx,fy,fz live in one quad)
// Rows never talk to each other, so rows of one wave may diverge (different ADMM iteration counts):
// EXEC is then row-granular and every DPP source lane is still live.
//
// tests/emu/a1mpc_rowops.hpp provides the same names on top of host fibers so that the solver
// source can be executed lane-for-lane on a CPU by the test-suite (test double, never shipped).
#pragma once
#include <hip/hip_runtime.h>

#define A1_DEV __device__ __forceinline__

namespace a1mpc {

A1_DEV int row_lane() { return static_cast<int>(threadIdx.x) & 15; }

template <int L>
A1_DEV double row_bcast(double v) {
    static_assert(L >= 0 && L < 16, "lane");
    return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + L, 0xF, 0xF, true);  // row_newbcast:L (bound_ctrl: no `old` operand to initialise)
}
template <int N>
A1_DEV double row_ror(double v) {
    static_assert(N >= 1 && N < 16, "rot");
    return __builtin_amdgcn_update_dpp(0.0, v, 0x120 + N, 0xF, 0xF, true);  // row_ror:N
}
template <int P0, int P1, int P2, int P3>
A1_DEV double quad_perm(double v) {
    return __builtin_amdgcn_update_dpp(0.0, v, P0 | (P1 << 2) | (P2 << 4) | (P3 << 6), 0xF, 0xF, true);
}

// Orders LDS traffic between the lanes of a row.  All lanes of a row are in one wavefront and the
// LDS executes a wave's DS instructions in issue order, so this only has to stop the compiler from
// moving DS accesses across it.
A1_DEV void row_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


// acc += m * (lane L of my row's x) as ONE instruction: v_fmac_f64 with a DPP row_newbcast source (gfx90a+: DP-ALU
// DPP supports exactly this control).  hipcc emits v_mov_b64_dpp + v_fmac_f64 for the intrinsic form and does not
// combine them, which doubles the instruction count and the dependency depth of every 12x12 mat-vec.
// (v_fmac_f64 is the only FP64 arithmetic op with a VOP2 encoding, hence the only one with a DPP form: no v_mul_f64_dpp.)
// Hazard contract (hipcc pads nothing inside asm): a VGPR written by a VALU instruction needs 2 wait states before
// a DPP instruction reads it -- pass x through row_dpp_ready() once after computing it and before its first fma_bcast.
template <int L>
A1_DEV void fma_bcast(double& acc, double m, double x) {
    static_assert(L >= 0 && L < 16, "lane");
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(m), "n"(L));
}
// acc -= m * (lane L of my row's x): the same instruction with the NEG source modifier
template <int L>
A1_DEV void fnma_bcast(double& acc, double m, double x) {
    static_assert(L >= 0 && L < 16, "lane");
    asm("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(m), "n"(L));
}
// min / max as single instructions: fmin()/fmax() make hipcc canonicalise loop-carried operands first (v_max_f64 x, x, x),
// one extra FP64 issue slot per operand in the projection of every ADMM row.  Operands here are never signalling NaNs.
A1_DEV double max_f64(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
A1_DEV double min_f64(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
A1_DEV double row_dpp_ready(double x) {
    asm volatile("s_nop 1" : "+v"(x));
    return x;
}

// the same for twelve values at once (one wait instead of twelve)
A1_DEV void row_dpp_ready12(double (&v)[12]) {
    asm volatile("s_nop 1" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                 "+v"(v[9]), "+v"(v[10]), "+v"(v[11]));
}

// Scheduling fence: the machine scheduler moves nothing across it (keeps a step's LDS reads ahead of the arithmetic that hides them).
A1_DEV void row_sched_fence() { __builtin_amdgcn_sched_barrier(0); }

// Optimisation barrier: the value becomes opaque to the compiler (no code is emitted).
A1_DEV double row_opaque(double v) {
    asm volatile("" : "+v"(v));
    return v;
}


// returns the old value; called by one lane of a row (the work queue of the persistent ADMM rows)
A1_DEV int row_atomic_inc(int* p) { return atomicAdd(p, 1); }

}  // namespace a1mpc
