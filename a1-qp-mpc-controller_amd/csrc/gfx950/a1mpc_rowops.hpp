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
// the same restricted to the four lanes of leg Q (DPP bank_mask: bit q enables lanes 4q..4q+3 of every row; the other lanes keep acc)
template <int L, int Q>
A1_DEV void fma_bcast_leg(double& acc, double m, double x) {
    static_assert(L >= 0 && L < 16 && Q >= 0 && Q < 4, "lane / leg");
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:%4" : "+v"(acc) : "v"(x), "v"(m), "n"(L), "n"(1 << Q));
}
template <int L, int Q>
A1_DEV void fnma_bcast_leg(double& acc, double m, double x) {
    static_assert(L >= 0 && L < 16 && Q >= 0 && Q < 4, "lane / leg");
    asm("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:%4" : "+v"(acc) : "v"(x), "v"(m), "n"(L), "n"(1 << Q));
}
// min / max as single instructions: fmin()/fmax() make hipcc canonicalise loop-carried operands first (v_max_f64 x, x, x),
// one extra FP64 issue slot per operand in the projection of every ADMM row.  Operands here are never signalling NaNs.
A1_DEV double max_f64(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// max(a, |x|): the absolute value rides on the source modifier
A1_DEV double max_abs_f64(double a, double x) {
    double r;
    asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(x));
    return r;
}
// a = max(a, |x0|, |x2|), b = max(b, |x1|) as ONE statement (hipcc pads every dependent pair of asm statements with an s_nop; VALU -> VALU needs none)
A1_DEV void max_abs3_f64(double& a, double& b, double x0, double x1, double x2) {
    asm("v_max_f64 %0, %0, |%2|\n"
        "v_max_f64 %1, %1, |%3|\n"
        "v_max_f64 %0, %0, |%4|" : "+v"(a), "+v"(b) : "v"(x0), "v"(x1), "v"(x2));
}
A1_DEV double min_f64(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// min(max(w, lb), ub) as one statement: hipcc pads a dependent pair of asm statements with an s_nop
A1_DEV double clamp_f64(double w, double lb, double ub) {
    double r;
    asm("v_max_f64 %0, %1, %2\n"
        "v_min_f64 %0, %0, %3" : "=&v"(r) : "v"(w), "v"(lb), "v"(ub));
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

// ---- the Riccati sweeps of one ADMM iteration as monolithic instruction blocks -------------------------------------------------
// One wave per SIMD means every instruction costs an issue slot.  Written as separate asm statements the chains pay for it
// twice: hipcc counts an inline-asm statement as zero wait states, so it pads every dependent pair of them with s_nop, and it
// cannot see the DPP read hazard, so each vector needed an explicit s_nop 1 before its first broadcast.  Inside one block the
// order is ours: independent accumulators are interleaved, and the two instructions that must separate a VALU write from a
// DPP read of the same register are instructions that have to be issued anyway (the hazard contract of every block: its
// DPP-read inputs were written >= 2 instructions before the block ends / begins, as noted per block).
// Compact index j = 0..11 -> lane 4*(j/3) + j%3 = 0,1,2,4,5,6,8,9,10,12,13,14.
#define A1_FMAC(acc, x, m, L) "v_fmac_f64_dpp " acc ", " x ", " m " row_newbcast:" #L " row_mask:0xf bank_mask:0xf\n"
#define A1_FNMA(acc, x, m, L) "v_fmac_f64_dpp " acc ", -" x ", " m " row_newbcast:" #L " row_mask:0xf bank_mask:0xf\n"

// r = e - B~' p  with  e = sd*xh + (at - cg)  (even terms into e, odd terms into a second accumulator that starts from zero: the arithmetic
// of sweep_back_rhs_twin)  and the seeds of the costate accumulators  pa = p + gA p[0] + gC p[2],  pb += gB p[1]
// (pb comes in as gV * ror8(p)).  p: written >= 2 instructions ago (sweep_back ends with two d-chain instructions after it);
// r: followed by two instructions inside the block.
A1_DEV void sweep_back_rhs(double& r, double& pa, double& pb, double at, double cg, double sd, double xh, double p, const double (&Bt)[6],
                           double gA, double gB, double gC) {
    double rb;
    asm("v_add_f64 %0, %4, -%5\n"
        "v_mov_b64 %1, 0\n"
        "v_fmac_f64 %0, %6, %7\n"
        "v_mov_b64 %2, %8\n"
        A1_FNMA("%0", "%8", "%9", 8) A1_FNMA("%1", "%8", "%10", 9) A1_FMAC("%2", "%8", "%15", 0)
        A1_FNMA("%0", "%8", "%11", 10) A1_FNMA("%1", "%8", "%12", 12)
        A1_FNMA("%0", "%8", "%13", 13) A1_FNMA("%1", "%8", "%14", 14)
        "v_add_f64 %0, %0, %1\n"
        A1_FMAC("%3", "%8", "%16", 1) A1_FMAC("%2", "%8", "%17", 2)
        : "=&v"(r), "=&v"(rb), "=&v"(pa), "+v"(pb)
        : "v"(at), "v"(cg), "v"(sd), "v"(xh), "v"(p), "v"(Bt[0]), "v"(Bt[1]), "v"(Bt[2]), "v"(Bt[3]), "v"(Bt[4]), "v"(Bt[5]), "v"(gA), "v"(gB), "v"(gC));
}
// the same for a row of a twin pair (persistent ADMM kernel, see twin_exchange): r comes in as e (formed by the row that owns the step and
// swapped over), and the costate seed is  pa = hm * p  with hm = 1 on the main row and 0 on its twin (whose accumulators carry the d chain
// and must start from zero; gA, gB, gC and pb come in masked the same way).
A1_DEV void sweep_back_rhs_twin(double& r, double& pa, double& pb, double p, const double (&Bt)[6], double gA, double gB, double gC, double hm) {
    double rb;
    asm("v_mov_b64 %1, 0\n"
        "v_mul_f64 %2, %4, %14\n"
        A1_FNMA("%0", "%4", "%5", 8) A1_FNMA("%1", "%4", "%6", 9) A1_FMAC("%2", "%4", "%11", 0)
        A1_FNMA("%0", "%4", "%7", 10) A1_FNMA("%1", "%4", "%8", 12)
        A1_FNMA("%0", "%4", "%9", 13) A1_FNMA("%1", "%4", "%10", 14)
        "v_add_f64 %0, %0, %1\n"
        A1_FMAC("%3", "%4", "%12", 1) A1_FMAC("%2", "%4", "%13", 2)
        : "+v"(r), "=&v"(rb), "=&v"(pa), "+v"(pb)
        : "v"(p), "v"(Bt[0]), "v"(Bt[1]), "v"(Bt[2]), "v"(Bt[3]), "v"(Bt[4]), "v"(Bt[5]), "v"(gA), "v"(gB), "v"(gC), "v"(hm));
}
// One chain pair for a row and its twin:  (pa, pb) += sum_b M[b] r[b]  (even b into pa, odd b into pb),  pa <- pa + pb.  On the main row M is
// column `ci` of K_t and the pair arrives seeded with A' p_{t+1} (result: the costate p_t); on the twin M is row `ci` of S_t^-1 and the seeds are
// zero (result: d_t = S_t^-1 r).  ONE instruction stream and ONE LDS read per term serve both products.  r: written >= 2 instructions ago.
// The block ends with the copy of the result that twin_exchange_copied() swaps with (pb := pa): the swap reads it >= 2 instructions later -- the next
// step's LDS reads are issued in between -- where a copy made next to the swap costs a second v_mov and an s_nop.
A1_DEV void sweep_back_chain_twin(double& pa, double& pb, double r, const double (&M)[12]) {
    asm(A1_FMAC("%0", "%2", "%3", 0) A1_FMAC("%1", "%2", "%4", 1) A1_FMAC("%0", "%2", "%5", 2) A1_FMAC("%1", "%2", "%6", 4)
        A1_FMAC("%0", "%2", "%7", 5) A1_FMAC("%1", "%2", "%8", 6) A1_FMAC("%0", "%2", "%9", 8) A1_FMAC("%1", "%2", "%10", 9)
        A1_FMAC("%0", "%2", "%11", 10) A1_FMAC("%1", "%2", "%12", 12) A1_FMAC("%0", "%2", "%13", 13) A1_FMAC("%1", "%2", "%14", 14)
        "v_add_f64 %0, %0, %1\n"
        "v_mov_b64 %1, %0\n"
        : "+v"(pa), "+v"(pb)
        : "v"(r), "v"(M[0]), "v"(M[1]), "v"(M[2]), "v"(M[3]), "v"(M[4]), "v"(M[5]), "v"(M[6]), "v"(M[7]), "v"(M[8]), "v"(M[9]), "v"(M[10]), "v"(M[11]));
}
// d = S^-1 r (row Sr; even terms + odd terms, two accumulators: the arithmetic of sweep_back_chain_twin) interleaved with
// p_t = (pa + pb) + K' r (column Kc); the costate chain runs ahead so that its final add is followed by >= 2 instructions.  Returns p_t in pa.
A1_DEV void sweep_back_chains(double& d, double& pa, double& pb, double r, const double (&Sr)[12], const double (&Kc)[12]) {
    double db;
    asm("v_mov_b64 %0, 0\n"
        "v_mov_b64 %3, 0\n"
        A1_FMAC("%1", "%4", "%17", 0) A1_FMAC("%2", "%4", "%18", 1)
        A1_FMAC("%0", "%4", "%5", 0) A1_FMAC("%1", "%4", "%19", 2) A1_FMAC("%3", "%4", "%6", 1) A1_FMAC("%2", "%4", "%20", 4)
        A1_FMAC("%0", "%4", "%7", 2) A1_FMAC("%1", "%4", "%21", 5) A1_FMAC("%3", "%4", "%8", 4) A1_FMAC("%2", "%4", "%22", 6)
        A1_FMAC("%0", "%4", "%9", 5) A1_FMAC("%1", "%4", "%23", 8) A1_FMAC("%3", "%4", "%10", 6) A1_FMAC("%2", "%4", "%24", 9)
        A1_FMAC("%0", "%4", "%11", 8) A1_FMAC("%1", "%4", "%25", 10) A1_FMAC("%3", "%4", "%12", 9) A1_FMAC("%2", "%4", "%26", 12)
        A1_FMAC("%0", "%4", "%13", 10) A1_FMAC("%1", "%4", "%27", 13) A1_FMAC("%3", "%4", "%14", 12) A1_FMAC("%2", "%4", "%28", 14)
        "v_add_f64 %1, %1, %2\n"
        A1_FMAC("%0", "%4", "%15", 13) A1_FMAC("%3", "%4", "%16", 14)
        "v_add_f64 %0, %0, %3\n"
        : "=&v"(d), "+v"(pa), "+v"(pb), "=&v"(db)
        : "v"(r), "v"(Sr[0]), "v"(Sr[1]), "v"(Sr[2]), "v"(Sr[3]), "v"(Sr[4]), "v"(Sr[5]), "v"(Sr[6]), "v"(Sr[7]), "v"(Sr[8]), "v"(Sr[9]), "v"(Sr[10]),
          "v"(Sr[11]), "v"(Kc[0]), "v"(Kc[1]), "v"(Kc[2]), "v"(Kc[3]), "v"(Kc[4]), "v"(Kc[5]), "v"(Kc[6]), "v"(Kc[7]), "v"(Kc[8]), "v"(Kc[9]),
          "v"(Kc[10]), "v"(Kc[11]));
}
// init + sum_j m[j] x[j] with two accumulators in one block (x written >= 2 instructions ago)
A1_DEV double dot12_block(const double (&m)[12], double x) {
    double a0, a1;
    asm("v_mov_b64 %0, 0\n"
        "v_mov_b64 %1, 0\n"
        A1_FMAC("%0", "%2", "%3", 0) A1_FMAC("%1", "%2", "%4", 1) A1_FMAC("%0", "%2", "%5", 2) A1_FMAC("%1", "%2", "%6", 4)
        A1_FMAC("%0", "%2", "%7", 5) A1_FMAC("%1", "%2", "%8", 6) A1_FMAC("%0", "%2", "%9", 8) A1_FMAC("%1", "%2", "%10", 9)
        A1_FMAC("%0", "%2", "%11", 10) A1_FMAC("%1", "%2", "%12", 12) A1_FMAC("%0", "%2", "%13", 13) A1_FMAC("%1", "%2", "%14", 14)
        "v_add_f64 %0, %0, %1\n"
        : "=&v"(a0), "=&v"(a1)
        : "v"(x), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "v"(m[9]), "v"(m[10]), "v"(m[11]));
    return a0;
}
// v = am * (v - K s) (row Kr),  xh = oma*xh + al*v,  and (SEED) the seeds of x_{t+1}: sa = s + fA s[8] + fC s[10], sb += fB s[9]
// (sb comes in as fP * ror8(s)).  s: written >= 2 instructions ago (sweep_fwd_input ends with the two projection instructions);
// v: followed by the two xh instructions.
template <bool SEED>
A1_DEV void sweep_fwd_gain(double& v, double& sa, double& sb, double& xh, double s, const double (&Kr)[12], double fA, double fB, double fC,
                           double am, double oma, double al) {
    double vb;
    if constexpr (SEED) {
        asm("v_mov_b64 %2, %5\n"
            "v_mov_b64 %1, 0\n"
            A1_FNMA("%0", "%5", "%6", 0) A1_FNMA("%1", "%5", "%7", 1) A1_FMAC("%2", "%5", "%18", 8)
            A1_FNMA("%0", "%5", "%8", 2) A1_FNMA("%1", "%5", "%9", 4) A1_FMAC("%3", "%5", "%19", 9)
            A1_FNMA("%0", "%5", "%10", 5) A1_FNMA("%1", "%5", "%11", 6) A1_FMAC("%2", "%5", "%20", 10)
            A1_FNMA("%0", "%5", "%12", 8) A1_FNMA("%1", "%5", "%13", 9) A1_FNMA("%0", "%5", "%14", 10)
            A1_FNMA("%1", "%5", "%15", 12) A1_FNMA("%0", "%5", "%16", 13) A1_FNMA("%1", "%5", "%17", 14)
            "v_add_f64 %0, %0, %1\n"
            "v_mul_f64 %0, %0, %21\n"
            "v_mul_f64 %4, %22, %4\n"
            "v_fmac_f64 %4, %23, %0\n"
            : "+v"(v), "=&v"(vb), "=&v"(sa), "+v"(sb), "+v"(xh)
            : "v"(s), "v"(Kr[0]), "v"(Kr[1]), "v"(Kr[2]), "v"(Kr[3]), "v"(Kr[4]), "v"(Kr[5]), "v"(Kr[6]), "v"(Kr[7]), "v"(Kr[8]), "v"(Kr[9]), "v"(Kr[10]),
              "v"(Kr[11]), "v"(fA), "v"(fB), "v"(fC), "v"(am), "v"(oma), "v"(al));
    } else {
        asm("v_mov_b64 %1, 0\n"
            A1_FNMA("%0", "%3", "%4", 0) A1_FNMA("%1", "%3", "%5", 1) A1_FNMA("%0", "%3", "%6", 2) A1_FNMA("%1", "%3", "%7", 4)
            A1_FNMA("%0", "%3", "%8", 5) A1_FNMA("%1", "%3", "%9", 6) A1_FNMA("%0", "%3", "%10", 8) A1_FNMA("%1", "%3", "%11", 9)
            A1_FNMA("%0", "%3", "%12", 10) A1_FNMA("%1", "%3", "%13", 12) A1_FNMA("%0", "%3", "%14", 13) A1_FNMA("%1", "%3", "%15", 14)
            "v_add_f64 %0, %0, %1\n"
            "v_mul_f64 %0, %0, %16\n"
            "v_mul_f64 %2, %17, %2\n"
            "v_fmac_f64 %2, %18, %0\n"
            : "+v"(v), "=&v"(vb), "+v"(xh)
            : "v"(s), "v"(Kr[0]), "v"(Kr[1]), "v"(Kr[2]), "v"(Kr[3]), "v"(Kr[4]), "v"(Kr[5]), "v"(Kr[6]), "v"(Kr[7]), "v"(Kr[8]), "v"(Kr[9]), "v"(Kr[10]),
              "v"(Kr[11]), "v"(am), "v"(oma), "v"(al));
    }
}
// x_{t+1} = (sa + sb) + B~ v (row Br), returned in sa; z0 = min(max(w0, lb), ub) rides behind it (the two instructions that
// separate the new state from its first DPP read).  v: written >= 2 instructions ago (sweep_fwd_gain).
A1_DEV void sweep_fwd_input(double& sa, double& sb, double& z0, double v, const double (&Br)[12], double w0, double lb, double ub) {
    asm(A1_FMAC("%0", "%3", "%4", 0) A1_FMAC("%1", "%3", "%5", 1) A1_FMAC("%0", "%3", "%6", 2) A1_FMAC("%1", "%3", "%7", 4)
        A1_FMAC("%0", "%3", "%8", 5) A1_FMAC("%1", "%3", "%9", 6) A1_FMAC("%0", "%3", "%10", 8) A1_FMAC("%1", "%3", "%11", 9)
        A1_FMAC("%0", "%3", "%12", 10) A1_FMAC("%1", "%3", "%13", 12) A1_FMAC("%0", "%3", "%14", 13) A1_FMAC("%1", "%3", "%15", 14)
        "v_add_f64 %0, %0, %1\n"
        "v_max_f64 %2, %16, %17\n"
        "v_min_f64 %2, %2, %18\n"
        : "+v"(sa), "+v"(sb), "=&v"(z0)
        : "v"(v), "v"(Br[0]), "v"(Br[1]), "v"(Br[2]), "v"(Br[3]), "v"(Br[4]), "v"(Br[5]), "v"(Br[6]), "v"(Br[7]), "v"(Br[8]), "v"(Br[9]), "v"(Br[10]),
          "v"(Br[11]), "v"(w0), "v"(lb), "v"(ub));
}

// The forward blocks of a twin pair: the x / w updates of a step belong to ONE row of the pair (RowSolver::admm_iteration_twin), so the blocks
// only carry the roll-out.  v = v - K s (row Kr; the pad lanes' copy of lane 0's value is never read: no `am` multiply) and (SEED) the first seed of x_{t+1}: s += fA s[8]; the seed
// terms come last so that v is followed by >= 2 instructions before sweep_fwd_input_twin reads it through DPP.  s: written >= 2 instructions ago.
template <bool SEED>
A1_DEV void sweep_fwd_gain_twin(double& v, double& s, const double (&Kr)[12], double fA) {
    double vb;
    if constexpr (SEED) {
        // the first seed of x_{t+1} accumulates onto s IN PLACE (no copy): it comes after the last term that reads s as it was, changes lanes 0 and 1 only and
        // reads lane 8; the other two seeds open the input block (each write of s is two instructions away from the next DPP read of s)
        asm("v_mov_b64 %1, 0\n"
            A1_FNMA("%0", "%2", "%3", 0) A1_FNMA("%1", "%2", "%4", 1) A1_FNMA("%0", "%2", "%5", 2) A1_FNMA("%1", "%2", "%6", 4)
            A1_FNMA("%0", "%2", "%7", 5) A1_FNMA("%1", "%2", "%8", 6) A1_FNMA("%0", "%2", "%9", 8) A1_FNMA("%1", "%2", "%10", 9)
            A1_FNMA("%0", "%2", "%11", 10) A1_FNMA("%1", "%2", "%12", 12) A1_FNMA("%0", "%2", "%13", 13) A1_FNMA("%1", "%2", "%14", 14)
            A1_FMAC("%2", "%2", "%15", 8)
            "v_add_f64 %0, %0, %1\n"
            : "+v"(v), "=&v"(vb), "+v"(s)
            : "v"(Kr[0]), "v"(Kr[1]), "v"(Kr[2]), "v"(Kr[3]), "v"(Kr[4]), "v"(Kr[5]), "v"(Kr[6]), "v"(Kr[7]), "v"(Kr[8]), "v"(Kr[9]), "v"(Kr[10]),
              "v"(Kr[11]), "v"(fA));
    } else {
        asm("v_mov_b64 %1, 0\n"
            A1_FNMA("%0", "%2", "%3", 0) A1_FNMA("%1", "%2", "%4", 1) A1_FNMA("%0", "%2", "%5", 2) A1_FNMA("%1", "%2", "%6", 4)
            A1_FNMA("%0", "%2", "%7", 5) A1_FNMA("%1", "%2", "%8", 6) A1_FNMA("%0", "%2", "%9", 8) A1_FNMA("%1", "%2", "%10", 9)
            A1_FNMA("%0", "%2", "%11", 10) A1_FNMA("%1", "%2", "%12", 12) A1_FNMA("%0", "%2", "%13", 13) A1_FNMA("%1", "%2", "%14", 14)
            "v_add_f64 %0, %0, %1\n"
            : "+v"(v), "=&v"(vb)
            : "v"(s), "v"(Kr[0]), "v"(Kr[1]), "v"(Kr[2]), "v"(Kr[3]), "v"(Kr[4]), "v"(Kr[5]), "v"(Kr[6]), "v"(Kr[7]), "v"(Kr[8]), "v"(Kr[9]), "v"(Kr[10]),
              "v"(Kr[11]));
    }
}
// x_{t+1} = (sa + sb) + B~ v (row Br), returned in sa.  SEED: sa arrives as x_t + fA x_t[8] (sweep_fwd_gain_twin) and the block opens with the other two seeds,
// sb += fB sa[9], sa += fC sa[10] (they read the omega lanes before the B~ v terms change them).  v: written >= 2 instructions ago (the step's LDS reads and, with SEED,
// the two seed instructions lie in between).  The next DPP read of sa is two instructions away or more (row_dpp_ready / the row_ror of the next seed).
template <bool SEED>
A1_DEV void sweep_fwd_input_twin(double& sa, double& sb, double v, const double (&Br)[12], double fB, double fC) {
    if constexpr (SEED) {
        asm(A1_FMAC("%1", "%0", "%15", 9) A1_FMAC("%0", "%0", "%16", 10)
            A1_FMAC("%0", "%2", "%3", 0) A1_FMAC("%1", "%2", "%4", 1) A1_FMAC("%0", "%2", "%5", 2) A1_FMAC("%1", "%2", "%6", 4)
            A1_FMAC("%0", "%2", "%7", 5) A1_FMAC("%1", "%2", "%8", 6) A1_FMAC("%0", "%2", "%9", 8) A1_FMAC("%1", "%2", "%10", 9)
            A1_FMAC("%0", "%2", "%11", 10) A1_FMAC("%1", "%2", "%12", 12) A1_FMAC("%0", "%2", "%13", 13) A1_FMAC("%1", "%2", "%14", 14)
            "v_add_f64 %0, %0, %1\n"
            : "+v"(sa), "+v"(sb)
            : "v"(v), "v"(Br[0]), "v"(Br[1]), "v"(Br[2]), "v"(Br[3]), "v"(Br[4]), "v"(Br[5]), "v"(Br[6]), "v"(Br[7]), "v"(Br[8]), "v"(Br[9]), "v"(Br[10]),
              "v"(Br[11]), "v"(fB), "v"(fC));
    } else {
        asm(A1_FMAC("%0", "%2", "%3", 0) A1_FMAC("%1", "%2", "%4", 1) A1_FMAC("%0", "%2", "%5", 2) A1_FMAC("%1", "%2", "%6", 4)
            A1_FMAC("%0", "%2", "%7", 5) A1_FMAC("%1", "%2", "%8", 6) A1_FMAC("%0", "%2", "%9", 8) A1_FMAC("%1", "%2", "%10", 9)
            A1_FMAC("%0", "%2", "%11", 10) A1_FMAC("%1", "%2", "%12", 12) A1_FMAC("%0", "%2", "%13", 13) A1_FMAC("%1", "%2", "%14", 14)
            "v_add_f64 %0, %0, %1\n"
            : "+v"(sa), "+v"(sb)
            : "v"(v), "v"(Br[0]), "v"(Br[1]), "v"(Br[2]), "v"(Br[3]), "v"(Br[4]), "v"(Br[5]), "v"(Br[6]), "v"(Br[7]), "v"(Br[8]), "v"(Br[9]), "v"(Br[10]),
              "v"(Br[11]));
    }
}

// Scheduling fence: the machine scheduler moves nothing across it (keeps a step's LDS reads ahead of the arithmetic that hides them).
// 1 / p for a well-scaled positive p (the Gauss-Jordan pivots): v_rcp_f64 + two Newton steps, ~1 ulp, without the scaling / fix-up
// instructions of a correctly rounded division (12 dependent instructions shorter per pivot).
A1_DEV double row_recip(double p) {
    double x = __builtin_amdgcn_rcp(p);
    double e = __builtin_fma(-p, x, 1.0);
    x = __builtin_fma(x, e, x);
    e = __builtin_fma(-p, x, 1.0);
    return __builtin_fma(x, e, x);
}

// 1 / sqrt(p) for a positive, well-scaled p (the Ruiz factors: 1e-4 <= p <= 1e4 after limit_scaling): v_rsq_f64 + two Goldschmidt steps, ~1 ulp,
// ~9 instructions against ~35 for a correctly rounded sqrt followed by a correctly rounded division (thirty of them per Ruiz pass and QP).
A1_DEV double row_rsqrt(double p) {
    const double y = __builtin_amdgcn_rsq(p);
    double g = p * y, h = 0.5 * y;          // g -> sqrt(p), h -> 1 / (2 sqrt(p))
    double r = __builtin_fma(-g, h, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    r = __builtin_fma(-g, h, 0.5);
    h = __builtin_fma(h, r, h);
    return h + h;
}

// One Gauss-Jordan pivot of the row-distributed 12x12 (RowSolver::factorize): S[j] += mlt * (S[j] of lane K) for every j != K
// (S[K] itself is set by the caller).  For K < 11 the block also returns the next pivot p = S[K+1] of lane K+1 and x = 1 / p
// (row_recip's sequence): row K+1 is eliminated first, and the reciprocal's dependent chain is issued between the other ten
// eliminations instead of after them.  Hazards: S[K+1] is read through DPP three instructions after its write; v_rcp_f64's
// result is read two instructions later; every S[j] was written >= 2 instructions before the block (caller's glue code).
#define A1_GJ(s) "v_fmac_f64_dpp " s ", " s ", %14 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n"
#define A1_GJL(s) "v_fmac_f64_dpp " s ", " s ", %11 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n"
template <int K>
A1_DEV void gj_pivot(double (&S)[12], double mlt, double& p, double& x) {
    static_assert(K >= 0 && K < 12, "pivot");
    constexpr int LK = 4 * (K / 3) + K % 3;
    if constexpr (K < 11) {
        constexpr int N = K + 1, LN = 4 * (N / 3) + N % 3;
        // the ten rows other than K and K+1, ascending
        constexpr auto o = [](int i) { int j = i; if (j >= (K < N ? K : N)) ++j; if (j >= (K < N ? N : K)) ++j; return j; };
        double e;
        asm(A1_GJ("%0") A1_GJ("%1") A1_GJ("%2")
            "v_mov_b64_dpp %11, %0 row_newbcast:%16 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            A1_GJ("%3")
            "v_rcp_f64 %12, %11\n"
            A1_GJ("%4") A1_GJ("%5")
            "v_fma_f64 %13, -%11, %12, 1.0\n"
            A1_GJ("%6")
            "v_fmac_f64 %12, %12, %13\n"
            A1_GJ("%7")
            "v_fma_f64 %13, -%11, %12, 1.0\n"
            A1_GJ("%8")
            "v_fmac_f64 %12, %12, %13\n"
            A1_GJ("%9") A1_GJ("%10")
            : "+v"(S[N]), "+v"(S[o(0)]), "+v"(S[o(1)]), "+v"(S[o(2)]), "+v"(S[o(3)]), "+v"(S[o(4)]), "+v"(S[o(5)]), "+v"(S[o(6)]), "+v"(S[o(7)]),
              "+v"(S[o(8)]), "+v"(S[o(9)]), "=&v"(p), "=&v"(x), "=&v"(e)
            : "v"(mlt), "n"(LK), "n"(LN));
    } else {
        asm(A1_GJL("%0") A1_GJL("%1") A1_GJL("%2") A1_GJL("%3") A1_GJL("%4") A1_GJL("%5") A1_GJL("%6") A1_GJL("%7") A1_GJL("%8") A1_GJL("%9") A1_GJL("%10")
            : "+v"(S[0]), "+v"(S[1]), "+v"(S[2]), "+v"(S[3]), "+v"(S[4]), "+v"(S[5]), "+v"(S[6]), "+v"(S[7]), "+v"(S[8]), "+v"(S[9]), "+v"(S[10])
            : "v"(mlt), "n"(LK));
    }
}
#undef A1_GJL
#undef A1_GJ

A1_DEV void row_sched_fence() { __builtin_amdgcn_sched_barrier(0); }

// All LDS reads issued so far have landed (s_waitcnt lgkmcnt(0)): one wait in front of a chain instead of one per operand.
A1_DEV void row_lds_landed() { __builtin_amdgcn_s_waitcnt(0xc07f); }

// Optimisation barrier: the value becomes opaque to the compiler (no code is emitted).
A1_DEV double row_opaque(double v) {
    asm volatile("" : "+v"(v));
    return v;
}
A1_DEV int64_t row_opaque(int64_t v) {
    asm volatile("" : "+v"(v));
    return v;
}
A1_DEV int row_opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}
// p + OFF doubles as an LDS address in a register of its own, formed by ONE v_add_u32 where it is used: the reads that follow become base + immediate.
// Left to itself hipcc folds a step's constant into every read of the step, and ds_read2_b64's 8-bit offsets do not reach it -- one v_add_u32 per read
// instruction instead of one per step; an opaque copy of the sum gets hoisted out of the ADMM loop instead (one more loop-carried register per step).
using lds_cptr = const __attribute__((address_space(3))) double*;
template <int OFF>
A1_DEV lds_cptr row_lds_at(const double* p) {
    const unsigned b = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_cptr)p));
    unsigned a;
    asm volatile("v_add_u32 %0, %1, %2" : "=v"(a) : "n"(OFF * 8), "v"(b));
    return reinterpret_cast<lds_cptr>(static_cast<uintptr_t>(a));
}


// ---- twin rows (persistent ADMM kernel) ------------------------------------------------------------------------------------------
// A wavefront costs the same issue slots with 32 or 64 live lanes, and the LDS image bounds residency at two QPs per wavefront: rows 2 and 3
// would idle.  In the persistent ADMM kernel they run as TWINS of rows 0 and 1 -- the same QP, the same instruction stream, bit-identical
// state and control flow -- except in the backward Riccati sweep, where the two 12-term products of a step on the same right-hand side
// (K_t' r on the main row, S_t^-1 r on the twin) share one chain of v_fmac_f64_dpp and one LDS read per term, and the halves then swap results.
A1_DEV bool row_is_twin() { return (static_cast<int>(threadIdx.x) & 32) != 0; }
// a = [x | y] on (main | twin)  ->  a = [x | x], returns [y | y]: v_permlane32_swap (lanes 32-63 of vdst <-> lanes 0-31 of src) on both dwords.
// Builtin, not asm: hipcc places the 2 wait states the swap needs after the copies itself.
A1_DEV double twin_exchange_copied(double& a, double c) {  // c: a copy of a, made >= 2 instructions ago (VALU write -> v_permlane32_swap read: 2 wait states)
    const unsigned long long bits = __builtin_bit_cast(unsigned long long, a), cbits = __builtin_bit_cast(unsigned long long, c);
    const unsigned lo = static_cast<unsigned>(bits), hi = static_cast<unsigned>(bits >> 32);
    const unsigned clo = static_cast<unsigned>(cbits), chi = static_cast<unsigned>(cbits >> 32);
    const auto r0 = __builtin_amdgcn_permlane32_swap(lo, clo, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(hi, chi, false, false);
    a = __builtin_bit_cast(double, static_cast<unsigned long long>(r0[0]) | (static_cast<unsigned long long>(r1[0]) << 32));
    return __builtin_bit_cast(double, static_cast<unsigned long long>(r0[1]) | (static_cast<unsigned long long>(r1[1]) << 32));
}
A1_DEV double twin_exchange(double& a) {
    // the swap rewrites both of its operands, so it needs a copy of `a`: ONE v_mov_b64 (hipcc makes two v_mov_b32 of it) and the two wait states
    // the swap needs after a VALU write, which hipcc cannot see behind an asm statement
    double c;
    asm("v_mov_b64 %0, %1\n"
        "s_nop 1\n" : "=v"(c) : "v"(a));
    return twin_exchange_copied(a, c);
}
// LDS ordering between the two rows of a pair (one wavefront: the same as row_sync(); the CPU test double needs the distinction)
A1_DEV void pair_sync() { row_sync(); }
// rows of one wavefront that share a QP's set-up (RowSolver::coop_n): the same wave-level ordering
A1_DEV void coop_sync() { row_sync(); }
// the twin takes its main row's value
A1_DEV double twin_from_main(double v) {
    (void)twin_exchange(v);
    return v;
}

// ---- quads of rows (kernels with one QP per wavefront, horizon a multiple of 4) ----------------------------------------------------------------
// With one LDS image per wavefront all four rows work on the same QP: rows 0 / 1 in the main role, rows 2 / 3 as their twins, rows 1 and 3 bit-identical
// copies of rows 0 and 2 through the sweeps -- and the per-lane ADMM state (x^, w, rho rows, D^-2 of a step) split four ways instead of two, so that the
// element-wise third of an iteration is issued for four steps at once (RowSolver<.., QUAD>).
A1_DEV int row_sub() { return (static_cast<int>(threadIdx.x) >> 4) & 1; }
// a = [x | y] on the (even | odd) row of each half  ->  a = x on both, returns y: v_permlane16_swap (odd rows of vdst <-> even rows of src) on both dwords
A1_DEV double quad_exchange(double& a) {
    double c;
    asm("v_mov_b64 %0, %1\n"
        "s_nop 1\n" : "=v"(c) : "v"(a));
    const unsigned long long bits = __builtin_bit_cast(unsigned long long, a), cbits = __builtin_bit_cast(unsigned long long, c);
    const unsigned lo = static_cast<unsigned>(bits), hi = static_cast<unsigned>(bits >> 32);
    const unsigned clo = static_cast<unsigned>(cbits), chi = static_cast<unsigned>(cbits >> 32);
    const auto r0 = __builtin_amdgcn_permlane16_swap(lo, clo, false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(hi, chi, false, false);
    a = __builtin_bit_cast(double, static_cast<unsigned long long>(r0[0]) | (static_cast<unsigned long long>(r1[0]) << 32));
    return __builtin_bit_cast(double, static_cast<unsigned long long>(r0[1]) | (static_cast<unsigned long long>(r1[1]) << 32));
}

// true if the predicate holds on any live lane of the wavefront (= any row that is still running)
A1_DEV bool row_wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0; }

// shader clock (s_memtime); the profiling instantiation of the persistent ADMM kernel stamps its stages with it (RowSolver<.., CLK>)
A1_DEV long long row_clock() { return clock64(); }

// returns the old value; called by one lane of a row (the work queue of the persistent ADMM rows)
A1_DEV int row_atomic_inc(int* p) { return atomicAdd(p, 1); }

}  // namespace a1mpc
