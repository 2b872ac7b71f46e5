// a1mpc_hip.hip -- the C ABI of include/a1mpc.h (liba1mpc.so) and the caller-side kernels (gait plan, contacts / terrain, swing legs, leg kinematics, EKF, joint torques,
// the dense-QP formation entry, the queue-order kernel).
//
// The solver's kernels live in csrc/a1mpc_kernels.hpp and are compiled as one translation unit per (horizon, pipeline): csrc/a1mpc_k_*.hip (round 6; build.py compiles the
// units in parallel).  This unit sees only the DECLARATIONS of their launch entry points (csrc/a1mpc_common.hpp) and decides which way a batch goes (launch_mpc,
// solve_device_impl): the latency kernel (<= 256 QPs: the rows of a wave share one QP's set-up), the fused kernel (up to the resident rows: one row pair = one QP from
// inputs to outputs; also warm-started ticks of a known batch) or the split pipeline (set-up kernel -> queue-order kernel -> persistent ADMM rows that drain the queue
// longest-first); the general path (per-step feet / contact schedules) has the same three.  The QPs of a batch are independent; nothing is shared between workgroups except the
// read-only (alpha/beta) table and the queue counter, so the blockIdx -> XCD mapping is irrelevant here (no L2 reuse to localise).
//
// There is no CPU path in this file: without a HIP device every entry point fails.
#include <rccl/rccl.h>  // types and prototypes only: librccl.so is dlopen()ed by a1mpc_sharded_create(transport = 1), never linked

#include <atomic>
#include <cmath>
#include <cstdio>
#include <new>

#include "a1mpc_common.hpp"

extern "C" const char a1mpc_build_id_[];   // a1mpc_build_id.cpp: "sources <hash> arch gfx950"

namespace a1mpc {

// the set-up kernel's cost guess (RowSolver::predict_cost) from the inputs alone, one thread per QP: the queue order of a first solve through a1mpc_solve_queue_kernel
__global__ __launch_bounds__(256) void a1mpc_predict_kernel(const KernelArgs a, int H) {
    const int64_t b = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (b >= a.n) return;
    double ev[3];
    if (a.tick) {
        const double* k = a.tick + b * 22; const double* R = a.R + b * 9;
        const double vwx = R[0] * k[15] + R[1] * k[16] + R[2] * k[17], vwy = R[3] * k[15] + R[4] * k[16] + R[5] * k[17];
        ev[0] = vwx - k[9]; ev[1] = vwy - k[10]; ev[2] = 0.0 - k[11];
    } else {
        const double* x0 = a.x0 + b * 13; const double* xr = a.xref + b * 13 * H;
        for (int i = 0; i < 3; ++i) ev[i] = xr[9 + i] - x0[9 + i];
    }
    const uint8_t* c = a.contact + b * (a.contact_stride ? 4 * H : 4);
    const bool all_stance = c[0] && c[1] && c[2] && c[3];
    const double hard = 30.0 * ev[2] + 27.0 * sqrt(ev[0] * ev[0] + ev[1] * ev[1]) + (all_stance ? 10.0 : 0.0);
    const double cc = 8.0 * hard + 400.0;
    a.cost[b] = cc > 0.0 ? (cc < 2047.0 ? static_cast<int>(cc) : 2047) : 0;
}

// K3: the next solve's queue order = this solve's QPs by decreasing cost (counting sort, one workgroup).  A batch of 1-4x the resident
// rows is otherwise finished by whichever long QP happened to start last; longest-first makes the makespan max(longest, total / rows).
// Round 6: 8 registers per wavefront and 516 B of LDS, so that the workgroup's four wavefronts per SIMD fit BESIDE a resident persistent wavefront (424 of 512 registers,
// 1 KB of a CU's LDS left at H = 10).  At 24 registers (a serial loop over the buckets on one thread) it had to wait for a CU with no persistent wavefront at all: 6 us
// behind its own batch's set-up with two batches in flight, but 0.4-0.5 ms per launch as soon as a third slot lets a batch's set-up run ahead of its turn (kernel
// trace, profiles/r06_setup_ahead.md); the build's resource gate keeps it at <= 16 (isa_check.MAX_VGPR_BESIDE_PERSISTENT).  The costs (0 .. 2047) are bucketed by 16: scheduling only
constexpr int kOrderThreads = 1024, kOrderBins = 128, kOrderShift = 4;
__global__ __launch_bounds__(kOrderThreads) void a1mpc_order_kernel(int n, const int32_t* __restrict__ cost, int32_t* __restrict__ order) {
    __shared__ int hist[kOrderBins];
    const int tid = static_cast<int>(threadIdx.x);
    if (tid < kOrderBins) hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += kOrderThreads) {
        const int c = cost[i] >> kOrderShift;
        atomicAdd(&hist[c < 0 ? 0 : (c > kOrderBins - 1 ? kOrderBins - 1 : c)], 1);
    }
    __syncthreads();
    // first queue position of every bucket, most expensive bucket first: an exclusive scan over the buckets in descending order -- thread t < 128 takes bucket 127 - t,
    // wave-level shuffles inside each of the two wavefronts, one word of LDS across them (a serial loop over the buckets on one thread was half of the kernel's 10 us)
    static_assert(kOrderBins == 128 && kOrderThreads >= 128, "two wavefronts scan the buckets");
    __shared__ int carry;
    int mine = 0, incl = 0;
    if (tid < kOrderBins) {
        mine = hist[kOrderBins - 1 - tid]; incl = mine;
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off, 64);
            if ((tid & 63) >= off) incl += up;
        }
        if (tid == 63) carry = incl;   // buckets 127 .. 64 in total
    }
    __syncthreads();
    if (tid < kOrderBins) hist[kOrderBins - 1 - tid] = incl - mine + (tid >= 64 ? carry : 0);
    __syncthreads();
    for (int i = tid; i < n; i += kOrderThreads) {
        const int c = cost[i] >> kOrderShift;
        order[atomicAdd(&hist[c < 0 ? 0 : (c > kOrderBins - 1 ? kOrderBins - 1 : c)], 1)] = i;
    }
}
void launch_order_kernel(int n, const int32_t* cost, int32_t* order, hipStream_t stream) { hipLaunchKernelGGL(a1mpc_order_kernel, dim3(1), dim3(kOrderThreads), 0, stream, n, cost, order); }
void launch_predict_kernel(const KernelArgs& a, int H, hipStream_t stream) { hipLaunchKernelGGL(a1mpc_predict_kernel, dim3(static_cast<unsigned>((a.n + 255) / 256)), dim3(256), 0, stream, a, H); }


// ---- debug / verification: the dense QP data the reference's ConvexMpc holds in its public members (hessian, gradient, lb, ub --
// S/ConvexMpc.h:84-93, filled by calculate_qp_mats, S/ConvexMpc.cpp:158-245) materialised from the same inputs, per-step feet and
// contacts included.  The solver never forms these (a1mpc_solver.hpp); this kernel exists so that callers which poke the members
// (S/A1RobotControl.cpp:527-537, S/test/test_mpc.cpp:136-140) keep working and so that the implicit Hessian can be compared entry by
// entry with the reference's dense product.  One workgroup per QP; not on the hot path.
struct FormArgs {
    DeviceParams P;
    int32_t n, H, foot_stride, contact_stride;
    const double *x0, *xref, *R, *foot;
    const uint8_t* contact;
    const double* yaw_A;
    double *Pout, *gout, *lout, *uout;   // n x (12H)^2 row-major, n x 12H, n x 20H, n x 20H
};
__global__ __launch_bounds__(256) void a1mpc_form_kernel(const FormArgs a) {
    __shared__ double Bw[20 * 36], TB[20 * 36], w[20 * 12], Ii[9];
    const int64_t b = blockIdx.x;
    const int tid = static_cast<int>(threadIdx.x), H = a.H, n = 12 * H;
    const DeviceParams& P = a.P;
    const double* R = a.R + b * 9;
    const double* x0 = a.x0 + b * 13;
    const double yaw = a.yaw_A ? a.yaw_A[b] : x0[2];
    const double dt = P.dt, cy = cos(yaw), sy = sin(yaw), bv = dt / P.mass;
    if (tid == 0) {  // I_world = R I_b R', inverse by cofactors (S/ConvexMpc.cpp:136-141)
        double t9[9], Iw[9];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += R[i * 3 + k] * P.inertia[k * 3 + j]; t9[i * 3 + j] = s; }
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += t9[i * 3 + k] * R[j * 3 + k]; Iw[i * 3 + j] = s; }
        const double c00 = Iw[4] * Iw[8] - Iw[5] * Iw[7], c01 = Iw[5] * Iw[6] - Iw[3] * Iw[8], c02 = Iw[3] * Iw[7] - Iw[4] * Iw[6];
        const double id = 1.0 / (Iw[0] * c00 + Iw[1] * c01 + Iw[2] * c02);
        Ii[0] = c00 * id; Ii[1] = (Iw[2] * Iw[7] - Iw[1] * Iw[8]) * id; Ii[2] = (Iw[1] * Iw[5] - Iw[2] * Iw[4]) * id;
        Ii[3] = c01 * id; Ii[4] = (Iw[0] * Iw[8] - Iw[2] * Iw[6]) * id; Ii[5] = (Iw[2] * Iw[3] - Iw[0] * Iw[5]) * id;
        Ii[6] = c02 * id; Ii[7] = (Iw[1] * Iw[6] - Iw[0] * Iw[7]) * id; Ii[8] = (Iw[0] * Iw[4] - Iw[1] * Iw[3]) * id;
        // w_i = Q (A_d^{i+1} x0 - x_ref_i): roll-out with A_d = I + dt A_c (S/ConvexMpc.cpp:123-129,150)
        double xs[12];
        for (int k = 0; k < 12; ++k) xs[k] = x0[k];
        const double* xr = a.xref + b * 13 * H;
        for (int i = 0; i < H; ++i) {
            const double o0 = xs[6], o1 = xs[7], o2 = xs[8];
            xs[0] += dt * (cy * o0 + sy * o1); xs[1] += dt * (-sy * o0 + cy * o1); xs[2] += dt * o2;
            xs[3] += dt * xs[9]; xs[4] += dt * xs[10]; xs[5] += dt * xs[11];
            xs[11] += dt * x0[12];
            for (int k = 0; k < 12; ++k) w[i * 12 + k] = P.q2[k] * (xs[k] - xr[i * 13 + k]);
        }
    }
    __syncthreads();
    for (int e = tid; e < H * 12; e += 256) {  // B~w_t = dt Iw^-1 skew(r_t), T B~w_t
        const int t = e / 12, c = e % 12, leg = c / 3, comp = c % 3;
        const double* fp = a.foot + b * (a.foot_stride ? 12 * H : 12) + static_cast<int64_t>(t) * a.foot_stride + 3 * leg;
        const double rx = fp[0], ry = fp[1], rz = fp[2];
        const double k0 = comp == 0 ? 0.0 : (comp == 1 ? -rz : ry), k1 = comp == 0 ? rz : (comp == 1 ? 0.0 : -rx), k2 = comp == 0 ? -ry : (comp == 1 ? rx : 0.0);
        double bw[3];
        for (int k = 0; k < 3; ++k) bw[k] = (Ii[k * 3 + 0] * k0 + Ii[k * 3 + 1] * k1 + Ii[k * 3 + 2] * k2) * dt;
        for (int k = 0; k < 3; ++k) Bw[(t * 3 + k) * 12 + c] = bw[k];
        TB[(t * 3 + 0) * 12 + c] = cy * bw[0] + sy * bw[1]; TB[(t * 3 + 1) * 12 + c] = -sy * bw[0] + cy * bw[1]; TB[(t * 3 + 2) * 12 + c] = bw[2];
    }
    __syncthreads();
    double* Po = a.Pout + b * static_cast<int64_t>(n) * n;
    for (int e = tid; e < n * n; e += 256) {
        const int i = e / n, j = e % n, s = i / 12, ca = i % 12, t = j / 12, cb = j % 12, m = s > t ? s : t;
        double al = 0.0;
        for (int k = m; k < H; ++k) al += static_cast<double>((k - s) * (k - t));
        const double be = H - m;
        double u = 0.0, v = 0.0;
        for (int c = 0; c < 3; ++c) { u += P.q2[c] * TB[(s * 3 + c) * 12 + ca] * TB[(t * 3 + c) * 12 + cb]; v += P.q2[6 + c] * Bw[(s * 3 + c) * 12 + ca] * Bw[(t * 3 + c) * 12 + cb]; }
        if (ca % 3 == cb % 3) { u += P.q2[3 + ca % 3] * bv * bv; v += P.q2[9 + ca % 3] * bv * bv; }
        Po[e] = al * (u * dt * dt) + be * v + (i == j ? P.r2[ca] : 0.0);
    }
    for (int j = tid; j < n; j += 256) {
        const int t = j / 12, cb = j % 12;
        double g = 0.0;
        for (int i = t; i < H; ++i) {
            const double k = (i - t) * dt;
            double lo = bv * w[i * 12 + 3 + cb % 3], hi = bv * w[i * 12 + 9 + cb % 3];
            for (int c = 0; c < 3; ++c) { lo += TB[(t * 3 + c) * 12 + cb] * w[i * 12 + c]; hi += Bw[(t * 3 + c) * 12 + cb] * w[i * 12 + 6 + c]; }
            g += k * lo + hi;
        }
        a.gout[b * n + j] = g;
    }
    for (int r = tid; r < 20 * H; r += 256) {  // S/ConvexMpc.cpp:223-245, per step
        const int t = r / 20, leg = (r % 20) / 5, row = r % 5;
        const uint8_t* cs = a.contact + b * (a.contact_stride ? 4 * H : 4) + static_cast<int64_t>(t) * a.contact_stride;
        const double cf = cs[leg] ? 1.0 : 0.0;
        double lo, hi;
        if (row == 4) { lo = P.fz_min * cf; hi = P.fz_max * cf; }
        else if (row == 0 || row == 2) { lo = 0.0; hi = kInfty; }
        else { lo = -kInfty; hi = 0.0; }
        a.lout[b * 20 * H + r] = lo; a.uout[b * 20 * H + r] = hi;
    }
}

__global__ void a1mpc_noop_kernel() {}


// ---- N2a: update_plan (S/A1RobotControl.cpp:148-202), one lane per (robot, leg) ------------------------------------------------
// Element-wise and HBM-bound: ~0.5 KB in + 0.3 KB out per robot.  The four lanes of a robot read the same robot-level words
// (one transaction) and write consecutive 24-byte segments.  No FMA contraction: the results are bit-identical to the reference's
// C++ arithmetic (and to the oracle, which is compiled the same way).
// Wave-private LDS stage for the kernels that run one lane per (robot, leg) (round 4).  A lane's results are 3-vectors at a stride of 24 bytes between lanes: stored
// straight from the registers, every store instruction touches 12 cache lines for 512 bytes of data.  The 16 robots of a wavefront own one contiguous run of
// every [robot][12] array (192 doubles), so the wave parks three such arrays at a time in 576 doubles of LDS and stores them back out with 64 consecutive doubles
// per instruction.  Pure data movement: no arithmetic, no bit changes (leg kernel 2.9 -> 5.8 TB/s at 524 288 robots).
struct WaveStage {
    double* sg;          // 576 doubles of LDS owned by this wavefront
    int lane;            // 0..63 = (robot of the wave) * 4 + leg
    int robots;          // live robots of this wavefront (16, fewer in the last one)
    int64_t first;       // first robot of this wavefront
    __device__ static void sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
    // null output pointers are skipped
    __device__ void flush3(const double (&x0)[3], const double (&x1)[3], const double (&x2)[3], double* o0, double* o1, double* o2) const {
#pragma unroll
        for (int r = 0; r < 3; ++r) { sg[3 * lane + r] = x0[r]; sg[192 + 3 * lane + r] = x1[r]; sg[384 + 3 * lane + r] = x2[r]; }
        sync();
        const int cnt = robots * 12;
        double* outs[3] = {o0, o1, o2};
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            if (outs[t] == nullptr) continue;
            double* out = outs[t] + first * 12;
#pragma unroll
            for (int m = 0; m < 3; ++m) { const int e = lane + 64 * m; if (e < cnt) out[e] = sg[192 * t + e]; }
        }
        sync();
    }
};
struct PlanArgs {
    a1mpc_gait_config g;
    int32_t n;
    const uint8_t* movement_mode;
    double* gait_counter;
    const double *gait_counter_speed, *root_lin_vel, *Rz, *Rw, *root_pos, *root_lin_vel_d;
    uint8_t* plan_contacts;
    double *rel, *abs_, *world;
};
// one (robot, leg) of update_plan: the new gait counter (also stored), the planned contact and the three foothold vectors
__device__ __forceinline__ void plan_lane(const PlanArgs& a, const int64_t b, const int leg, double& gc, double (&rel)[3], double (&ab)[3], double (&wo)[3]) {
#pragma clang fp contract(off)
    gc = a.gait_counter[b * 4 + leg];
    const double spd = a.gait_counter_speed[b * 4 + leg];
    uint8_t pc;
    if (!a.movement_mode[b]) {                                  // :150-153
        pc = 1; gc = a.g.gait_counter_reset[leg];
    } else {                                                    // :155-165
        gc = gc + spd;
        gc = fmod(gc, a.g.counter_per_gait);
        pc = gc <= a.g.counter_per_swing ? 1 : 0;
    }
    a.gait_counter[b * 4 + leg] = gc;
    a.plan_contacts[b * 4 + leg] = pc;
    const double* Rz = a.Rz + b * 9; const double* Rw = a.Rw + b * 9;
    const double* v = a.root_lin_vel + b * 3; const double* vd = a.root_lin_vel_d + b * 3; const double* pos = a.root_pos + b * 3;
    const double vrx = Rz[0] * v[0] + Rz[3] * v[1] + Rz[6] * v[2];   // :168-169  Rz' v
    const double vry = Rz[1] * v[0] + Rz[4] * v[1] + Rz[7] * v[2];
    const double k = sqrt(fabs(a.g.default_foot_pos[2]) / 9.8);       // default_foot_pos(2): linear index 2 = z of leg 0
    const double half_swing = ((a.g.counter_per_swing / spd) * a.g.control_dt) / 2.0;
    double dx = k * (vrx - vd[0]) + half_swing * vd[0];              // :175-182
    double dy = k * (vry - vd[1]) + half_swing * vd[1];
    if (dx < -a.g.foot_delta_x_limit) dx = -a.g.foot_delta_x_limit;
    if (dx > a.g.foot_delta_x_limit) dx = a.g.foot_delta_x_limit;
    if (dy < -a.g.foot_delta_y_limit) dy = -a.g.foot_delta_y_limit;
    if (dy > a.g.foot_delta_y_limit) dy = a.g.foot_delta_y_limit;
    rel[0] = a.g.default_foot_pos[3 * leg + 0] + dx; rel[1] = a.g.default_foot_pos[3 * leg + 1] + dy; rel[2] = a.g.default_foot_pos[3 * leg + 2];
#pragma unroll
    for (int r = 0; r < 3; ++r) {                                    // :198-199
        ab[r] = Rw[r * 3 + 0] * rel[0] + Rw[r * 3 + 1] * rel[1] + Rw[r * 3 + 2] * rel[2];
        wo[r] = ab[r] + pos[r];
    }
}
__global__ __launch_bounds__(256) void a1mpc_plan_kernel(const PlanArgs a) {
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(16))) double stage[4][576];
    const int lane = static_cast<int>(threadIdx.x) & 63, wv = static_cast<int>(threadIdx.x) >> 6;
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t b = gid >> 2;
    const int leg = static_cast<int>(gid & 3);
    const int64_t wave_first = (static_cast<int64_t>(blockIdx.x) * 256 + wv * 64) >> 2;
    if (wave_first >= a.n) return;
    const WaveStage ws{stage[wv], lane, static_cast<int>(a.n - wave_first < 16 ? a.n - wave_first : 16), wave_first};
    double rel[3] = {0, 0, 0}, ab[3] = {0, 0, 0}, wo[3] = {0, 0, 0}, gc = 0.0;
    if (b < a.n) plan_lane(a, b, leg, gc, rel, ab, wo);
    ws.flush3(rel, ab, wo, a.rel, a.abs_, a.world);   // (the three [robot][12] outputs leave through the wave's LDS stage: WaveStage)
}

thread_local std::string g_last_error;
std::mutex g_cache_mu;
thread_local bool g_clk_ran = false;
thread_local bool g_gen_prefer_one_wave = false;
// dynamic-LDS limit of a kernel, once per device and kernel
a1mpc_status set_lds_attr(const void* fn, size_t bytes) {
    static std::vector<std::pair<const void*, int>> done;
    static std::mutex mu;   // handles of different threads may launch at the same time
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    A1_HIP(hipGetDevice(&dev));
    for (const auto& d : done) if (d.first == fn && d.second == dev) return A1MPC_OK;
    A1_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes)));
    done.emplace_back(fn, dev);
    return A1MPC_OK;
}

static size_t carry_stride(int horizon) {
    switch (horizon) {
#define A1_CASE(H) case H: return Carry<H>::STRIDE;
        A1MPC_MULTI_STEP_HORIZONS(A1_CASE)
#undef A1_CASE
    }
    return 0;  // (horizon 1: the update path does not apply -- warm_start = 2 behaves like 1)
}
static size_t prep_stride(int horizon) {
    switch (horizon) {
#define A1_CASE(H) case H: return Prep<H>::STRIDE;
        A1MPC_FAST_HORIZONS(A1_CASE)
#undef A1_CASE
    }
    return 0;
}
static size_t prep_stride_gen(int horizon) {
    switch (horizon) {
#define A1_CASE(H, R) case H: return Prep<H>::STRIDE_GEN;
        A1MPC_GEN_HORIZONS(A1_CASE)
#undef A1_CASE
    }
    return 0;
}
// rows of the general path's ADMM kernel that are resident at once (0: horizon without a general path)
static a1mpc_status resident_rows_gen(int horizon, int* rows) {
    int wg = 0;
    a1mpc_status st = A1MPC_OK;
    *rows = 0;
    switch (horizon) {
#define A1_CASE(H, R) case H: st = resident_workgroups_gen<H, R>(&wg); *rows = R * wg; break;
        A1MPC_GEN_HORIZONS(A1_CASE)
#undef A1_CASE
    }
    return st;
}
static a1mpc_status launch_gen_split(int horizon, const KernelArgs& a, double* prep, int* counter, hipStream_t s, hipEvent_t mid) {
    int wg = 0;
    switch (horizon) {
#define A1_CASE(H, R) case H: if (a1mpc_status st = resident_workgroups_gen<H, R>(&wg); st != A1MPC_OK) return st; return launch_gen_split_rows<H, R>(a, prep, counter, s, mid, wg);
        A1MPC_GEN_HORIZONS(A1_CASE)
#undef A1_CASE
    }
    return fail(A1MPC_ERR_UNSUPPORTED_HORIZON, "per-step feet (foot_stride = 12) and a separate A_c yaw need a horizon > 1");
}
// LDS per QP (round 6: the per-step bounds left the image): 22.7 KB (H = 10: six QPs per CU in one-wave workgroups, seven in the CU-wide persistent kernel), 35.9 KB (H = 16: four,
// one per wavefront like the fast path's), 44.7 KB (H = 20: three, one row per workgroup) -- which also leaves 13 / 26 KB of a CU's LDS free at H = 16 / 20: the set-up kernel of a
// second batch in flight (11.7 / 15.7 KB per workgroup) now runs BESIDE the persistent kernel instead of behind it (profiles/r06_general_pipeline.md)
static a1mpc_status launch_gen(int horizon, const KernelArgs& a, hipStream_t s) {
    switch (horizon) {
#define A1_CASE(H, R) case H: return launch_gen_rows<H, R>(a, s);
        A1MPC_GEN_HORIZONS(A1_CASE)
#undef A1_CASE
    }
    return fail(A1MPC_ERR_UNSUPPORTED_HORIZON, "per-step feet (foot_stride = 12) and a separate A_c yaw need a horizon > 1");
}

static bool warm_fused_enabled() {
    static const bool on = [] { const char* e = getenv("A1MPC_WARM_FUSED"); return !(e && !strcmp(e, "0")); }();
    return on;
}
// largest warm-started batch that runs the fused kernel (see solve_device_impl): measured crossovers, in QPs
static int warm_fused_max(int horizon) {
    static const int over = [] { const char* e = getenv("A1MPC_WARM_FUSED_MAX"); return e ? atoi(e) : 0; }();
    if (over > 0) return over;
    return horizon == 10 ? 8192 : (horizon == 20 ? 2048 : 0);   // (h = 16 / 20 with the fused kernel's quads of rows: profiles/r04_warm_ticks_h16_h20_fused_vs_split.txt -- 2048 x h20
                                                                 // 0.72 -> 0.63 ms per tick; from 4096 on, and at h = 16 everywhere, the split pipeline's denser residency wins)
}
static constexpr int kScheduleMinBatch = 1024;  // below this every QP is resident at once and the order cannot matter

// does a batch of n QPs go through the split pipeline?
static a1mpc_status use_split_pipeline(int horizon, int n, bool have_prep, bool* split) {
    *split = false;
    if (!have_prep || pipeline_mode() == 2) return A1MPC_OK;
    if (pipeline_mode() == 1) { *split = true; return A1MPC_OK; }
    int rows = 0;
    a1mpc_status st = A1MPC_OK;
    switch (horizon) {
#define A1_CASE(H) case H: st = resident_rows<H>(&rows); break;
        A1MPC_FAST_HORIZONS(A1_CASE)
#undef A1_CASE
        default: return A1MPC_OK;
    }
    *split = n > rows;
    return st;
}

static bool fused_queue_enabled() {
    static const bool on = [] { const char* e = getenv("A1MPC_FUSED_QUEUE"); return e && !strcmp(e, "1"); }();
    return on;
}
static a1mpc_status launch_mpc(int horizon, const KernelArgs& a, double* prep, int* counter, hipStream_t s, bool split, hipEvent_t mid) {
    RoctxRange range(split ? "a1mpc solve (split pipeline)" : "a1mpc solve (fused)");
    if (split && counter && horizon == 10 && fused_queue_enabled() && a.carry == nullptr && a.clk == nullptr) {
        if (mid) A1_HIP(hipEventRecord(mid, s));
        return launch_fused_queue(a, counter, s);
    }
    if (split && prep && counter) {
        switch (horizon) {
#define A1_CASE(H) case H: return launch_split<H>(a, prep, counter, s, mid);
            A1MPC_FAST_HORIZONS(A1_CASE)
#undef A1_CASE
        }
    }
    switch (horizon) {
#define A1_CASE(H) case H: return launch<H, kModeMpc>(a, s);
        A1MPC_FAST_HORIZONS(A1_CASE)
#undef A1_CASE
    }
    return fail(A1MPC_ERR_UNSUPPORTED_HORIZON, "horizon must be " A1MPC_HORIZON_LIST);
}
static size_t lds_bytes_of(int horizon) {
    const int r = rows_per_wg(horizon);
    switch (horizon) {
#define A1_CASE(H) case H: return lds_bytes<H>(r);
        A1MPC_FAST_HORIZONS(A1_CASE)
#undef A1_CASE
    }
    return 0;
}

static void to_device_params(const a1mpc_config& c, DeviceParams* p) {
    std::memset(p, 0, sizeof *p);
    p->dt = c.dt; p->mu = c.mu; p->fz_min = c.fz_min; p->fz_max = c.fz_max;
    for (int i = 0; i < 12; ++i) { p->q2[i] = 2.0 * c.q[i]; p->r2[i] = 2.0 * c.r[i]; }  // S/ConvexMpc.cpp:20,41
    p->mass = c.mass;
    for (int i = 0; i < 9; ++i) p->inertia[i] = c.inertia_body[i];
    p->rho0 = c.rho; p->sigma = c.sigma; p->alpha = c.alpha; p->eps_abs = c.eps_abs; p->eps_rel = c.eps_rel;
    p->adaptive_rho_tol = c.adaptive_rho_tolerance;
    p->max_iter = c.max_iter; p->check_every = c.check_termination; p->adaptive_rho = c.adaptive_rho;
    // adaptive_rho_interval = 0 is OSQP's default "automatic" (what the reference actually runs, S/A1RobotControl.cpp:523-524): OSQP fixes the
    // interval in its first solve to c_max(c_roundmultiple(iter, check_termination), check_termination), iter = the iteration at which
    // 0.4 x setup_time of WALL CLOCK has passed (osqp.c, PROFILING).  That rule is evaluated here with the outcome it has whenever 0.4 x setup_time
    // is worth fewer than 1.5 x check_termination iterations (true for these QP sizes on the CPUs the reference runs on): check_termination.
    // Its other possible outcomes (50, 75, ...) are selected by passing that number.
    p->adaptive_rho_every = c.adaptive_rho_interval > 0 ? c.adaptive_rho_interval : (c.check_termination > 0 ? c.check_termination : 25);
    p->scaling_iters = c.scaling; p->warm_start = c.warm_start;
}

// Which configurations the engine accepts (a13).  OSQP validates its data and settings in osqp_setup (auxil.c validate_data: every l_i <= u_i;
// validate_settings: rho, sigma > 0, 0 < alpha < 2, eps >= 0, max_iter > 0, scaling >= 0, adaptive_rho_tolerance >= 1, ...) and refuses to set up otherwise; the
// reference never looks at the outcome (S/A1RobotControl.cpp:532,540) and then "solves" on an uninitialised workspace.  Here the same conditions are a loud
// A1MPC_ERR_INVALID_ARGUMENT at a1mpc_create / a1mpc_update_config / a1mpc_balance_solve_batch.  On everything that passes, the QP family is feasible and bounded:
//   * 0 <= mu, fz_min <= fz_max, fz_max >= 0: per leg and step (0, 0, max(fz_min, 0) c) satisfies the pyramid and the box (c = the contact flag) -- never primal infeasible;
//   * the box and the pyramid bound every force by fz_max (1 + mu) -- the feasible set is compact, so the cost is bounded below whatever the weights: never dual infeasible;
//   * q >= 0, r >= 0 (finite): P = B'QB + R is positive semi-definite -- never OSQP_NON_CVX from the data (the status remains for non-finite INPUTS).
// Hence OSQP's statuses -3 / 3 (primal infeasible) and -4 / 4 (dual infeasible) have no exact certificate on any accepted configuration and the kernels do not evaluate
// the approximate certificate tests (auxil.c is_primal_infeasible / is_dual_infeasible); the oracle does evaluate them, and over every QP the suites compare (> 2 M) it never reported one.
static bool all_finite(const double* v, int n) {
    for (int i = 0; i < n; ++i) if (!std::isfinite(v[i])) return false;
    return true;
}
static const char* invalid_config(const a1mpc_config& c) {
    if (!std::isfinite(c.dt) || !(c.dt > 0)) return "dt must be positive and finite";
    if (!std::isfinite(c.mass) || !(c.mass > 0)) return "mass must be positive and finite";
    if (!std::isfinite(c.mu) || c.mu < 0) return "mu must be >= 0 and finite (a negative friction coefficient collapses the pyramid to the zero force, and to the empty set -- primal infeasible -- when fz_min > 0)";
    if (!std::isfinite(c.fz_min) || !std::isfinite(c.fz_max)) return "fz_min / fz_max must be finite";
    if (c.fz_min > c.fz_max) return "fz_min > fz_max: OSQP's validate_data refuses l > u (the problem would be primal infeasible)";
    if (c.fz_max < 0) return "fz_max < 0: a stance leg would need fz <= fz_max < 0 inside a pyramid that asks fz >= 0 (primal infeasible)";
    if (!all_finite(c.q, A1MPC_STATE_DIM) || !all_finite(c.r, A1MPC_NUM_DOF)) return "q / r weights must be finite";
    for (int i = 0; i < A1MPC_STATE_DIM; ++i) if (c.q[i] < 0) return "q weights must be >= 0 (a negative weight makes the Hessian indefinite: OSQP_NON_CVX)";
    for (int i = 0; i < A1MPC_NUM_DOF; ++i) if (c.r[i] < 0) return "r weights must be >= 0 (a negative weight makes the Hessian indefinite: OSQP_NON_CVX)";
    if (!all_finite(c.inertia_body, 9)) return "inertia_body must be finite";
    {   // the body inertia is inverted (S/ConvexMpc.cpp:136-141): it must be invertible
        const double* I = c.inertia_body;
        const double det = I[0] * (I[4] * I[8] - I[5] * I[7]) - I[1] * (I[3] * I[8] - I[5] * I[6]) + I[2] * (I[3] * I[7] - I[4] * I[6]);
        if (!(std::fabs(det) > 0)) return "inertia_body must be invertible";
    }
    // osqp auxil.c validate_settings
    if (!std::isfinite(c.rho) || !(c.rho > 0)) return "rho must be positive";
    if (!std::isfinite(c.sigma) || !(c.sigma > 0)) return "sigma must be positive";
    if (!std::isfinite(c.alpha) || !(c.alpha > 0) || !(c.alpha < 2)) return "alpha must lie in (0, 2)";
    if (!std::isfinite(c.eps_abs) || !std::isfinite(c.eps_rel) || c.eps_abs < 0 || c.eps_rel < 0) return "eps_abs / eps_rel must be >= 0 and finite";
    if (c.eps_abs == 0 && c.eps_rel == 0) return "eps_abs and eps_rel must not both be zero";
    if (c.max_iter <= 0) return "max_iter must be positive";
    if (c.check_termination < 0) return "check_termination must be >= 0";
    if (c.scaling < 0) return "scaling must be >= 0";
    if (c.adaptive_rho != 0 && c.adaptive_rho != 1) return "adaptive_rho must be 0 or 1";
    if (c.adaptive_rho_interval < 0) return "adaptive_rho_interval must be >= 0";
    if (c.adaptive_rho && (!std::isfinite(c.adaptive_rho_tolerance) || c.adaptive_rho_tolerance < 1.0)) return "adaptive_rho_tolerance must be >= 1";
    if (c.warm_start < 0 || c.warm_start > 2) return "warm_start must be 0, 1 or 2";
    return nullptr;
}
static const char* invalid_balance_config(const a1mpc_balance_config& q) {
    if (!all_finite(q.Q, 6) || !std::isfinite(q.R) || !std::isfinite(q.mu) || !std::isfinite(q.F_min) || !std::isfinite(q.F_max)) return "balance-QP constants must be finite";
    for (int i = 0; i < 6; ++i) if (q.Q[i] < 0) return "balance-QP weights Q must be >= 0";
    if (q.R < 0) return "balance-QP weight R must be >= 0";
    if (q.mu < 0) return "balance-QP mu must be >= 0";
    if (q.F_min > q.F_max) return "balance-QP F_min > F_max (primal infeasible; OSQP's validate_data refuses l > u)";
    if (q.F_max < 0) return "balance-QP F_max < 0 (primal infeasible)";
    return nullptr;
}

}  // namespace a1mpc

using namespace a1mpc;

struct a1mpc_handle_s {
    a1mpc_config cfg;
    DeviceParams dp;
    int device = 0;
    int max_batch = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    hipEvent_t ev_mark = nullptr;   // marker_at(): a no-timing event recorded where a timing event would be (A1MPC_TICK_MARKERS, profiles/r06_control_tick_timeline.md)
    bool timing = true;   // a1mpc_set_timing: HIP events around the launches (a1mpc_last_kernel_ms / _stage_ms / _control_tick_ms); off = three to five event records less per tick
    double* d_tab = nullptr;      // (alpha/beta, beta) table of cfg.horizon
    double* d_tab1 = nullptr;     // the H = 1 table (balance QP)
    // device staging for the host-pointer entry points
    double *d_x0 = nullptr, *d_xref = nullptr, *d_R = nullptr, *d_foot = nullptr, *d_aux = nullptr, *d_Rz = nullptr;
    uint8_t* d_contact = nullptr;
    double *d_grf = nullptr, *d_u = nullptr;
    int32_t *d_iters = nullptr, *d_status = nullptr, *d_nfact = nullptr;
    hipStream_t last_stream = nullptr;
    // Every launch touches handle-owned scratch (prepared-state records, queue counter / order / cost, nfact, the carried OSQP workspace,
    // the filter states): a call on another stream than the previous call's first waits for that call's work (ev_order, recorded after
    // every launch); the reset functions wait for it on the host.
    double* d_carry = nullptr;     // warm_start = 2 (the reference's update path): n x Carry<H>::STRIDE, allocated on first use
    hipEvent_t ev_order = nullptr;
    hipEvent_t ev_mid = nullptr;   // between the set-up kernel and the persistent ADMM kernel of the split pipeline (a1mpc_last_stage_ms)
    hipEvent_t ev_tick0 = nullptr, ev_tick1 = nullptr;   // around a1mpc_control_tick_device
    bool tick_timed = false, tick_fused = false, tick_km_set = false;
    double tick_km[3] = {0, 0, 0};
    bool staged = false;
    bool busy = false;
    size_t zc_poll_bytes = 0;   // > 0: host_submit launched a kernel that writes its outputs into the pinned block itself and filled that many bytes of it with the in-flight pattern (a1mpc_solve_batch polls them)
    int pipeline_depth = 0;      // > 0: this handle is a slot of an a1mpc_pipeline with that many slots (g_gen_prefer_one_wave)
    // carried OSQP workspace (warm start)
    double *d_wx = nullptr, *d_wy = nullptr, *d_rho = nullptr;
    // split pipeline: prepared state of every QP (set-up kernel -> ADMM kernel) and the work-queue counter
    double* d_prep = nullptr;
    double* d_prep_gen = nullptr;  // the general path's records (B~w_t of every step included), allocated on first use
    int* d_counter = nullptr;
    // queue order of the next solve (longest-first by the previous solve's per-QP cost) -- see a1mpc_set_schedule
    int32_t *d_order = nullptr, *d_cost = nullptr;
    double* d_foot_steps = nullptr;      // general path: n x 12H per-step feet (allocated on first use)
    uint8_t* d_contact_steps = nullptr;  // general path: n x 4H per-step contacts
    double* d_ct_state = nullptr;  // N2b filter state of every robot (allocated on first use)
    double* d_ekf_state = nullptr;  // N4c Kalman filter state of every robot (allocated on first use)
    double* d_tickrec = nullptr;    // a1mpc_control_tick_device: n x 22 tick records + 3 doubles (km_foot), allocated on first use
    int32_t ekf_ready_n = 0;        // robots 0 .. ekf_ready_n - 1 have had their filter initialised (the init kernel is not launched for them again)
    // staging of the element-wise entry points (N2a, N2b, N3), allocated on first use: 64 / 96 doubles and 16 bytes per robot
    double *d_aux_in = nullptr, *d_aux_out = nullptr;
    uint8_t* d_aux_u8 = nullptr;
    int32_t hint_n = 0;  // batch size the order was built for (0 = none)
    bool profiling = false;        // a1mpc_set_profiling: split-pipeline solves run the clock-stamped instantiation of the ADMM kernel
    long long* d_clk = nullptr;    // n x kTickStages cycles per QP (allocated on first use)
    int32_t clk_n = 0;             // QPs of the last profiled solve (0: the last solve was not profiled)
    bool clk_tick = false;         // ... and it ran the fused / latency kernel: every stage of the record is filled (a1mpc_last_tick_stage_cycles)
    int32_t last_ws_mode = -1;  // warm-start semantics the last MPC solve ran (a1mpc_last_warm_start_mode): the configured mode, or 1 where mode 2 does not exist
    int schedule = 1;    // 1 = history (default), 0 = index order
    // packed device blocks + pinned host mirrors of the host-pointer MPC entry (one copy each way per call)
    char *d_in = nullptr, *d_out = nullptr;
    char* h_pin = nullptr;
    char* d_pin = nullptr;   // the pinned block as the device sees it (hipHostGetDevicePointer), or null: small batches read / write it directly (host_submit)
    size_t h_pin_bytes = 0, h_pin_in_bytes = 0;
    char *h_pin_gen = nullptr, *d_pin_gen = nullptr;   // a small pinned block for the general path's inputs of <= 8 QPs (per-step feet / contacts do not fit the block above), allocated on first use
    size_t h_pin_gen_bytes = 0;
};

static a1mpc_status order_streams(a1mpc_handle h, hipStream_t s) {
    if (h->busy && h->last_stream != s) A1_HIP(hipStreamWaitEvent(s, h->ev_order, 0));
    return A1MPC_OK;
}
static a1mpc_status mark_launched(a1mpc_handle h, hipStream_t s) {
    A1_HIP(hipEventRecord(h->ev_order, s));
    h->busy = true; h->last_stream = s;
    return A1MPC_OK;
}
#define A1_ORDER(h, s) do { if (a1mpc_status so_ = order_streams(h, s); so_ != A1MPC_OK) return so_; } while (0)
#define A1_MARK(h, s) do { if (a1mpc_status sm_ = mark_launched(h, s); sm_ != A1MPC_OK) return sm_; } while (0)

extern "C" {

void a1mpc_default_config(a1mpc_config* c) {
    if (!c) return;
    std::memset(c, 0, sizeof *c);
    c->horizon = 10;         // PLAN_HORIZON, S/A1Params.h:26
    c->dt = 0.0025;          // S/A1RobotControl.cpp:462
    c->mu = 0.3;             // S/ConvexMpc.cpp:8
    c->fz_min = 0.0;         // S/ConvexMpc.cpp:223
    c->fz_max = 180.0;       // S/ConvexMpc.cpp:224
    c->rho = 0.1; c->sigma = 1e-6; c->alpha = 1.6; c->eps_abs = 1e-3; c->eps_rel = 1e-3;  // OSQP 0.6 defaults
    c->adaptive_rho_tolerance = 5.0;
    c->max_iter = 4000; c->check_termination = 25; c->adaptive_rho = 1; c->adaptive_rho_interval = 25; c->scaling = 10;
    c->warm_start = 1;       // S/A1RobotControl.cpp:524
}

void a1mpc_default_balance_config(a1mpc_balance_config* q) {
    if (!q) return;
    const double Q[6] = {1.0, 1.0, 1.0, 400.0, 400.0, 100.0};  // S/A1RobotControl.cpp:11
    std::memcpy(q->Q, Q, sizeof Q);
    q->R = 1e-3; q->mu = 0.7; q->F_min = 0.0; q->F_max = 180.0;  // S/A1RobotControl.cpp:12-15
}

// ---- N2b: contact logic + recent-contact filters, walking-surface fit, terrain pitch (S/A1RobotControl.cpp:256-282, 566-582, 335-376) --------
// Device-resident state per robot, laid out so that one tick touches whole cache lines whatever phase the robots' filters are in (round 3; until then
// every field was its own [robot] array, which coalesces only while all robots' ring cursors agree -- they part with the first early contact):
//   record, 384 B, 128-aligned: 4 x CtLeg (64 B: the leg's window count and cursor -- its three position filters are always updated together, so they
//           share them --, the early-contact flag, 3 x Neumaier sum / correction) + CtRobot (128 B: foot_pos_recent_contact, the terrain-angle filter's header);
//   leg rings  [robot][leg][60 slots][4 doubles] (x, y, z of one tick in one 32-byte sector; the 4th is padding);
//   terrain ring [100 slots][robot]: this filter advances on every tick of a robot that stands (root height > 0.1 m), so the robots' cursors agree unless one of them
//           has been lying down -- neighbouring robots then touch neighbouring words (the leg rings cannot have that: early contacts part their cursors for good).
// One lane per robot: 3 lines of record + one sector per leg in contact + one word of the terrain ring, read and written once.  No FMA contraction.
constexpr int kLegWindow = 60, kTerrainWindow = 100;
struct CtLeg { int32_t count, head; double early, sum[3], corr[3]; };
struct CtRobot { double recent[12]; int32_t count, head; double sum, corr, pad_; };
struct CtRecord { CtLeg leg[4]; CtRobot rb; };   // (no alignas: the records start a hipMalloc block, which is 256-byte aligned, and their LDS image below is not 128-aligned)
static_assert(sizeof(CtLeg) == 64 && sizeof(CtRobot) == 128 && sizeof(CtRecord) == 384, "contact-state record: three 128-byte lines");
constexpr size_t kCtLegRing = 4 * kLegWindow * 4;  // doubles per robot
constexpr size_t kCtBytesPerRobot = sizeof(CtRecord) + (kCtLegRing + kTerrainWindow) * sizeof(double);
struct ContactArgs {
    int32_t n;
    double counter_per_swing, foot_force_low;
    int32_t use_terrain_adapt;
    CtRecord* rec;
    double *leg_ring, *terrain_ring;
    int64_t stride;  // robots per terrain-ring slot (= max_batch)
    const double *gait_counter, *foot_force, *foot_pos_abs, *root_pos_z;
    const uint8_t* plan_contacts;
    double* pitch_d;
    int32_t z_stride, pitch_stride;   // doubles between two robots' root_pos_z / root_euler_d pitch (1: arrays of their own; 3: element 2 of root_pos / element 1 of root_euler_d, a1mpc_control_tick_device)
    // a1mpc_control_tick_device: the lane that has just written a robot's terrain pitch also assembles its 22-number MPC tick record (a1mpc_tick_pack_kernel's job, one
    // launch less per tick), or pk_tick = null
    const double *pk_euler, *pk_pos, *pk_ang_vel, *pk_lin_vel, *pk_euler_d, *pk_lin_vel_d, *pk_ang_vel_d, *pk_pos_d_z;
    double* pk_tick;
    uint8_t* contacts;
    double *recent_out, *terrain_out;
    const double* recent_in;  // terrain-only entry: foot_pos_recent_contact comes from the caller instead of the handle's contact state
};
struct Neumaier {  // S/utils/filter.hpp:26-39,53-66: the moving-window sum with its running compensation
    double sum, corr;
    __device__ inline void add(double val) {
#pragma clang fp contract(off)
        const double ns = sum + val;
        if (fabs(sum) >= fabs(val)) corr += (sum - ns) + val; else corr += (val - ns) + sum;
        sum = ns;
    }
};
__device__ inline void sym3_pinv(const double* m, double* out) {  // pseudo-inverse of a symmetric PSD 3x3 by cyclic Jacobi (S/utils/Utils.cpp:44-52)
#pragma clang fp contract(off)
    double a[9], v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int k = 0; k < 9; ++k) a[k] = m[k];
    for (int sweep = 0; sweep < 8; ++sweep) {
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            const double apq = a[3 * p + q];
            if (apq == 0.0) continue;
            const double th = (a[3 * q + q] - a[3 * p + p]) / (2.0 * apq);
            const double t = (th >= 0.0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
            for (int k = 0; k < 3; ++k) { const double akp = a[3 * k + p], akq = a[3 * k + q]; a[3 * k + p] = c * akp - sn * akq; a[3 * k + q] = sn * akp + c * akq; }
            for (int k = 0; k < 3; ++k) { const double apk = a[3 * p + k], aqk = a[3 * q + k]; a[3 * p + k] = c * apk - sn * aqk; a[3 * q + k] = sn * apk + c * aqk; }
            for (int k = 0; k < 3; ++k) { const double vkp = v[3 * k + p], vkq = v[3 * k + q]; v[3 * k + p] = c * vkp - sn * vkq; v[3 * k + q] = sn * vkp + c * vkq; }
        }
    }
    double lmax = fabs(a[0]);
    if (fabs(a[4]) > lmax) lmax = fabs(a[4]);
    if (fabs(a[8]) > lmax) lmax = fabs(a[8]);
    const double tol = 2.220446049250313e-16 * 3.0 * lmax;
    double inv[3];
    for (int k = 0; k < 3; ++k) inv[k] = fabs(a[4 * k]) > tol ? 1.0 / a[4 * k] : 0.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            out[3 * i + j] = v[3 * i + 0] * inv[0] * v[3 * j + 0] + v[3 * i + 1] * inv[1] * v[3 * j + 1] + v[3 * i + 2] * inv[2] * v[3 * j + 2];
}
// a robot's inputs of one tick: loaded BEFORE its record is staged, so that they travel with the record instead of after it (round 6: until then they were loaded behind the
// staging barrier, the ring sectors behind them, root_pos_z behind the plane fit and the terrain ring's word behind that -- five exposures to the memory latency per robot, now two)
struct CtInputs { double gc[4], ff[4], fp[12], rz; uint8_t plan[4]; };
__device__ __forceinline__ CtInputs contact_terrain_inputs(const ContactArgs& a, const int64_t b) {
    CtInputs in;
    in.rz = a.root_pos_z[b * a.z_stride];
    if (!a.recent_in) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { in.gc[i] = a.gait_counter[b * 4 + i]; in.ff[i] = a.foot_force[b * 4 + i]; in.plan[i] = a.plan_contacts[b * 4 + i]; }
#pragma unroll
        for (int k = 0; k < 12; ++k) in.fp[k] = a.foot_pos_abs[b * 12 + k];
    }
    return in;
}
// one robot's tick on its record `st` (the kernel hands in the record's LDS image; the terrain-only entry and the host-compiled test double the record itself)
__device__ __forceinline__ void contact_terrain_robot_in(const ContactArgs& a, const int64_t b, CtRecord* st, const CtInputs& in) {
#pragma clang fp contract(off)
    double rc[12];
    // the terrain-angle filter's word under its cursor is read here, with the leg rings' sectors: everything a tick reads in place is in flight before anything is computed
    const bool standing = in.rz > 0.1;
    const int tcount = st->rb.count, thead = st->rb.head;
    double* tr = a.terrain_ring + thead * a.stride + b;
    const double tr_old = (standing && tcount >= kTerrainWindow) ? *tr : 0.0;
    if (a.recent_in) {
#pragma unroll
        for (int k = 0; k < 12; ++k) rc[k] = a.recent_in[b * 12 + k];
    } else {
        CtLeg L[4];
        double slot[4][3];
        const double (&gc)[4] = in.gc, (&ff)[4] = in.ff, (&fp)[12] = in.fp;
        const uint8_t (&plan)[4] = in.plan;
        bool c[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) L[i] = st->leg[i];
#pragma unroll
        for (int k = 0; k < 12; ++k) rc[k] = st->rb.recent[k];
        double* ring = a.leg_ring + b * kCtLegRing;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double e = L[i].early;
            if (gc[i] <= a.counter_per_swing * 1.5) e = 0.0;                                                        // :260-262
            if (!plan[i] && gc[i] > a.counter_per_swing * 1.5 && ff[i] > a.foot_force_low) e = 1.0;                 // :263-267
            c[i] = plan[i] != 0 || e != 0.0;                                                                        // :271
            if (e != L[i].early) st->leg[i].early = e;
            a.contacts[b * 4 + i] = c[i] ? 1 : 0;
            const double* sl = ring + (i * kLegWindow + L[i].head) * 4;
            const bool full = c[i] && L[i].count >= kLegWindow;   // only then does the slot under the cursor hold a sample to retire
#pragma unroll
            for (int k = 0; k < 3; ++k) slot[i][k] = full ? sl[k] : 0.0;
        }
        bool any = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (!c[i]) continue;                                                                                    // :274-281
            any = true;
            const bool full = L[i].count >= kLegWindow;
            double* sl = ring + (i * kLegWindow + L[i].head) * 4;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                Neumaier f{L[i].sum[k], L[i].corr[k]};
                if (full) f.add(-slot[i][k]);
                f.add(fp[3 * i + k]);
                sl[k] = fp[3 * i + k];
                st->leg[i].sum[k] = f.sum; st->leg[i].corr[k] = f.corr;
                rc[3 * i + k] = (f.sum + f.corr) / static_cast<double>(kLegWindow);
            }
            st->leg[i].count = full ? L[i].count : L[i].count + 1;
            st->leg[i].head = (L[i].head + 1) % kLegWindow;
        }
        if (any) {
#pragma unroll
            for (int k = 0; k < 12; ++k) st->rb.recent[k] = rc[k];
        }
    }
    if (a.recent_out) {
#pragma unroll
        for (int k = 0; k < 12; ++k) a.recent_out[b * 12 + k] = rc[k];
    }
    double M[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, rhs[3] = {0, 0, 0}, P3[9], co[3];   // :566-582  a = pinv(W'W) W' z
    for (int i = 0; i < 4; ++i) {
        const double w[3] = {1.0, rc[3 * i + 0], rc[3 * i + 1]};
        for (int r = 0; r < 3; ++r) { for (int cc = 0; cc < 3; ++cc) M[3 * r + cc] += w[r] * w[cc]; rhs[r] += w[r] * rc[3 * i + 2]; }
    }
    sym3_pinv(M, P3);
    for (int r = 0; r < 3; ++r) co[r] = P3[3 * r + 0] * rhs[0] + P3[3 * r + 1] * rhs[1] + P3[3 * r + 2] * rhs[2];
    const double s0 = co[1], s1 = co[2], s2 = -1.0;
    double terrain_angle = 0.0;                                                             // :339-352
    if (standing) {
        const double angle_cos = fabs(0.0 * s0 + 0.0 * s1 + 1.0 * s2) / (sqrt(0.0 * 0.0 + 0.0 * 0.0 + 1.0 * 1.0) * sqrt(s0 * s0 + s1 * s1 + s2 * s2));
        const double v = acos(angle_cos);
        Neumaier f{st->rb.sum, st->rb.corr};
        if (tcount >= kTerrainWindow) f.add(-tr_old); else st->rb.count = tcount + 1;
        f.add(v);
        *tr = v;
        st->rb.head = (thead + 1) % kTerrainWindow; st->rb.sum = f.sum; st->rb.corr = f.corr;
        terrain_angle = (f.sum + f.corr) / static_cast<double>(kTerrainWindow);
    }
    if (terrain_angle > 0.5) terrain_angle = 0.5;
    if (terrain_angle < -0.5) terrain_angle = -0.5;
    const double F_R_diff = rc[2] + rc[5] - rc[8] - rc[11];                                // :355
    if (a.use_terrain_adapt) a.pitch_d[b * a.pitch_stride] = F_R_diff > 0.05 ? -terrain_angle : terrain_angle;  // :358-364
    a.terrain_out[b] = terrain_angle;
}
// (a name of its own, not an overload: this section sits inside the ABI's extern "C" block, where two functions of one name are ONE symbol)
__device__ __forceinline__ void contact_terrain_robot(const ContactArgs& a, const int64_t b, CtRecord* st) { contact_terrain_robot_in(a, b, st, contact_terrain_inputs(a, b)); }
// One wavefront = 64 robots.  A lane that walks its own 384-byte record in HBM makes every load instruction touch 64 different lines (measured: 2.5 TB/s of algorithmic
// bytes at 524 288 robots whether the plane fit runs or not).  The wavefront therefore moves its 24 KB of records as 24 fully coalesced 16-byte-per-lane loads into LDS,
// every lane works on the LDS image of its record (stride 400 B = 25 x 16 B: the 16 lanes of a ds_read_b128 phase hit 16 different quad-banks), and the image goes back
// the same way -- whole lines in, whole lines out.  Only the ring sectors are touched in place (one 32-byte sector per leg in contact, wherever that robot's cursor is).
constexpr int kCtLdsStride = 400, kCtUnits = sizeof(CtRecord) / 16;   // bytes per record image; 16-byte units per record
__global__ __launch_bounds__(64) void a1mpc_contact_terrain_kernel(const ContactArgs a) {
    const int64_t base = static_cast<int64_t>(blockIdx.x) * 64;
    const int lane = threadIdx.x;
    const int64_t b = base + lane;
    if (a.recent_in) {   // terrain-only entry: 24 bytes of the record matter
        if (b < a.n) contact_terrain_robot(a, b, a.rec + b);
        return;
    }
    __shared__ __attribute__((aligned(16))) unsigned char img[64 * kCtLdsStride];
    const int cnt = a.n - base < 64 ? static_cast<int>(a.n - base) : 64;
    const CtInputs in = contact_terrain_inputs(a, b < a.n ? b : a.n - 1);   // (in flight together with the record)
    const uint4* src = reinterpret_cast<const uint4*>(a.rec + base);
    uint4 v[kCtUnits];
#pragma unroll
    for (int i = 0; i < kCtUnits; ++i) { const int o = i * 64 + lane; v[i] = o < cnt * kCtUnits ? src[o] : uint4{0, 0, 0, 0}; }
#pragma unroll
    for (int i = 0; i < kCtUnits; ++i) { const int o = i * 64 + lane, r = o / kCtUnits, w = o - r * kCtUnits; *reinterpret_cast<uint4*>(img + r * kCtLdsStride + w * 16) = v[i]; }
    __syncthreads();
    if (b < a.n) contact_terrain_robot_in(a, b, reinterpret_cast<CtRecord*>(img + lane * kCtLdsStride), in);
    if (b < a.n && a.pk_tick != nullptr) {   // (S/A1RobotControl.cpp:452-456, :470-488 read exactly these fields; root_euler_d[1] is the pitch this lane has just written)
        double* t = a.pk_tick + b * 22;
        double tv[22];   // (every load before the first store: written field by field each load waited for the store before it -- the compiler cannot know that t[] and the sources are apart)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            tv[k] = a.pk_euler[b * 3 + k]; tv[3 + k] = a.pk_pos[b * 3 + k]; tv[6 + k] = a.pk_ang_vel[b * 3 + k]; tv[9 + k] = a.pk_lin_vel[b * 3 + k];
            tv[12 + k] = a.pk_euler_d[b * 3 + k]; tv[15 + k] = a.pk_lin_vel_d[b * 3 + k]; tv[18 + k] = a.pk_ang_vel_d[b * 3 + k];
        }
        tv[21] = a.pk_pos_d_z[b];
#pragma unroll
        for (int k = 0; k < 22; ++k) t[k] = tv[k];
    }
    __syncthreads();
    uint4* dst = reinterpret_cast<uint4*>(a.rec + base);
#pragma unroll
    for (int i = 0; i < kCtUnits; ++i) { const int o = i * 64 + lane, r = o / kCtUnits, w = o - r * kCtUnits; if (o < cnt * kCtUnits) dst[o] = *reinterpret_cast<const uint4*>(img + r * kCtLdsStride + w * 16); }
}
// the handle's contact state: records, then the leg rings, then the terrain rings (zero = every filter empty)
static a1mpc_status ensure_contact_state(a1mpc_handle h, hipStream_t s) {
    if (h->d_ct_state) return A1MPC_OK;
    A1_HIP(hipMalloc(&h->d_ct_state, static_cast<size_t>(h->max_batch) * kCtBytesPerRobot));
    A1_HIP(hipMemsetAsync(h->d_ct_state, 0, static_cast<size_t>(h->max_batch) * kCtBytesPerRobot, s));
    return A1MPC_OK;
}
static void contact_state_pointers(a1mpc_handle h, ContactArgs& a) {
    a.rec = reinterpret_cast<CtRecord*>(h->d_ct_state);
    a.leg_ring = reinterpret_cast<double*>(a.rec + h->max_batch);
    a.terrain_ring = a.leg_ring + static_cast<size_t>(h->max_batch) * kCtLegRing;
    a.stride = h->max_batch;
}
static void launch_contact_terrain(const ContactArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(a1mpc_contact_terrain_kernel, dim3(static_cast<unsigned>((a.n + 63) / 64)), dim3(64), 0, s, a);
}

static a1mpc_status ensure_aux(a1mpc_handle h) {
    if (h->d_aux_in) return A1MPC_OK;
    const size_t n = static_cast<size_t>(h->max_batch);
    A1_HIP(hipMalloc(&h->d_aux_in, n * 64 * sizeof(double)));
    A1_HIP(hipMalloc(&h->d_aux_out, n * 96 * sizeof(double)));
    A1_HIP(hipMalloc(&h->d_aux_u8, n * 16));
    return A1MPC_OK;
}

void a1mpc_default_contact_config(a1mpc_contact_config* c) {
    if (!c) return;
    c->counter_per_swing = 120.0; c->foot_force_low = 30.0; c->use_terrain_adapt = 1;
}
a1mpc_status a1mpc_reset_contact_state(a1mpc_handle h) {
    if (!h) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle");
    if (!h->d_ct_state) return A1MPC_OK;
    A1_HIP(hipSetDevice(h->device));
    A1_ORDER(h, h->stream);  // the memsets below must not overtake a launch still running on a caller's stream
    A1_HIP(hipMemsetAsync(h->d_ct_state, 0, static_cast<size_t>(h->max_batch) * kCtBytesPerRobot, h->stream));
    A1_HIP(hipStreamSynchronize(h->stream));
    return A1MPC_OK;
}
a1mpc_status a1mpc_contact_terrain_batch(a1mpc_handle h, const a1mpc_contact_config* cfg, int32_t n, const double* gait_counter,
                                         const uint8_t* plan_contacts, const double* foot_force, const double* foot_pos_abs,
                                         const double* root_pos_z, double* root_euler_d_pitch, uint8_t* contacts_out,
                                         double* foot_pos_recent_contact_out, double* terrain_angle_out) {
    if (!h || !cfg) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle/config");
    if (n < 0 || !gait_counter || !plan_contacts || !foot_force || !foot_pos_abs || !root_pos_z || !root_euler_d_pitch || !contacts_out ||
        !foot_pos_recent_contact_out || !terrain_angle_out)
        return fail(A1MPC_ERR_INVALID_ARGUMENT, "null input/output pointer");
    if (n > h->max_batch) return fail(A1MPC_ERR_BATCH_TOO_LARGE, "n > max_batch given to a1mpc_create");
    if (n == 0) return A1MPC_OK;
    A1_HIP(hipSetDevice(h->device));
    if (a1mpc_status st = ensure_aux(h); st != A1MPC_OK) return st;
    const size_t N = n;
    hipStream_t s = h->stream;
    A1_ORDER(h, s);
    if (a1mpc_status st = ensure_contact_state(h, s); st != A1MPC_OK) return st;
    // staging: in [gc 4 | ff 4 | foot 12 | z 1 | pitch 1], out [recent 12 | terrain 1]
    double *d_gc = h->d_aux_in, *d_ff = d_gc + 4 * N, *d_fp = d_ff + 4 * N, *d_z = d_fp + 12 * N, *d_pd = d_z + N;
    double *d_rec = h->d_aux_out, *d_ta = d_rec + 12 * N;
    uint8_t *d_pc = h->d_aux_u8, *d_ct = h->d_aux_u8 + 8 * N;
    A1_HIP(hipMemcpyAsync(d_gc, gait_counter, N * 4 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_ff, foot_force, N * 4 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_fp, foot_pos_abs, N * 12 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_z, root_pos_z, N * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_pd, root_euler_d_pitch, N * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_pc, plan_contacts, N * 4, hipMemcpyHostToDevice, s));
    ContactArgs a;
    a.recent_in = nullptr; a.z_stride = 1; a.pitch_stride = 1; a.pk_tick = nullptr;
    a.n = n; a.counter_per_swing = cfg->counter_per_swing; a.foot_force_low = cfg->foot_force_low; a.use_terrain_adapt = cfg->use_terrain_adapt;
    contact_state_pointers(h, a); a.gait_counter = d_gc; a.foot_force = d_ff; a.foot_pos_abs = d_fp; a.root_pos_z = d_z; a.plan_contacts = d_pc;
    a.pitch_d = d_pd; a.contacts = d_ct; a.recent_out = d_rec; a.terrain_out = d_ta;
    if (h->timing) A1_HIP(hipEventRecord(h->ev0, s));
    launch_contact_terrain(a, s);
    A1_HIP(hipGetLastError());
    if (h->timing) A1_HIP(hipEventRecord(h->ev1, s));
    h->timed = h->timing; A1_MARK(h, s);
    A1_HIP(hipMemcpyAsync(contacts_out, d_ct, N * 4, hipMemcpyDeviceToHost, s));
    A1_HIP(hipMemcpyAsync(foot_pos_recent_contact_out, d_rec, N * 12 * sizeof(double), hipMemcpyDeviceToHost, s));
    A1_HIP(hipMemcpyAsync(terrain_angle_out, d_ta, N * sizeof(double), hipMemcpyDeviceToHost, s));
    A1_HIP(hipMemcpyAsync(root_euler_d_pitch, d_pd, N * sizeof(double), hipMemcpyDeviceToHost, s));
    A1_HIP(hipStreamSynchronize(s));
    return A1MPC_OK;
}

// ---- N4a: swing-leg targets + foot PD force (S/A1RobotControl.cpp:204-254, Bezier: S/utils/Utils.cpp:64-104), one lane per (robot, leg) ------
struct SwingArgs {
    int32_t n;
    double counter_per_swing, dt, kp[3], kd[3];
    const double *Rz, *foot_pos_abs, *gait_counter, *target_rel;
    double *start, *rel_last, *target_last, *cur_out, *kin_out;
};
// one (robot, leg) of generate_swing_legs_ctrl's first block: gc = the leg's gait counter, target_rel = its foothold (both may come straight from plan_lane's registers)
__device__ __forceinline__ void swing_lane(const SwingArgs& a, const int64_t b, const int i, const double gc, const double (&target_rel)[3], double (&cur)[3], double (&st)[3],
                                           double (&tl)[3], double (&kin)[3]) {
#pragma clang fp contract(off)
    const double* Rz = a.Rz + b * 9;
    const double* fa = a.foot_pos_abs + b * 12 + 3 * i;
    const int64_t o = b * 12 + 3 * i;
#pragma unroll
    for (int r = 0; r < 3; ++r) cur[r] = Rz[0 * 3 + r] * fa[0] + Rz[1 * 3 + r] * fa[1] + Rz[2 * 3 + r] * fa[2];   // :224
    float spline_time = 0.0f;
    if (gc <= a.counter_per_swing) {                                                                                  // :227-232  (foot_pos_start <- the current position)
#pragma unroll
        for (int r = 0; r < 3; ++r) st[r] = cur[r];
    } else {                                                                                                          // :233-236  (foot_pos_start stays: written back as read)
        spline_time = static_cast<float>(gc - a.counter_per_swing) / static_cast<float>(a.counter_per_swing);
#pragma unroll
        for (int r = 0; r < 3; ++r) st[r] = a.start[o + r];
    }
    const double t = spline_time, u = 1 - t;
    const double t2 = t * t, t3 = t2 * t, t4 = t2 * t2, u2 = u * u, u3 = u2 * u, u4 = u2 * u2;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double fin = target_rel[r];
        double P1 = st[r], P2 = fin;
        if (r == 2) { P1 += 0.0f; P2 += 0.4f + 0.5 * 0.0; }                                                           // FOOT_SWING_CLEARANCE1 / 2
        double y = 0;                                                                                                 // Utils.cpp:97-104
        y += 1.0 * 1.0 * u4 * st[r];
        y += 4.0 * t * u3 * P1;
        y += 6.0 * t2 * u2 * P2;
        y += 4.0 * t3 * u * fin;
        y += 1.0 * t4 * 1.0 * fin;
        const double vel_cur = (cur[r] - a.rel_last[o + r]) / a.dt;                                                   // :243-252
        const double vel_tgt = (y - a.target_last[o + r]) / a.dt;
        tl[r] = y;
        kin[r] = (y - cur[r]) * a.kp[r] + (vel_tgt - vel_cur) * a.kd[r];
    }
}
__global__ __launch_bounds__(256) void a1mpc_swing_kernel(const SwingArgs a) {
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(16))) double stage[4][576];
    const int lane = static_cast<int>(threadIdx.x) & 63, wv = static_cast<int>(threadIdx.x) >> 6;
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t b = gid >> 2;
    const int i = static_cast<int>(gid & 3);
    const int64_t wave_first = (static_cast<int64_t>(blockIdx.x) * 256 + wv * 64) >> 2;
    if (wave_first >= a.n) return;
    const WaveStage ws{stage[wv], lane, static_cast<int>(a.n - wave_first < 16 ? a.n - wave_first : 16), wave_first};
    double cur[3] = {0, 0, 0}, st[3] = {0, 0, 0}, tl[3] = {0, 0, 0}, kin[3] = {0, 0, 0};
    if (b < a.n) {
        const double tr[3] = {a.target_rel[b * 12 + 3 * i], a.target_rel[b * 12 + 3 * i + 1], a.target_rel[b * 12 + 3 * i + 2]};
        swing_lane(a, b, i, a.gait_counter[b * 4 + i], tr, cur, st, tl, kin);
    }
    // start / rel_last / target_last are updated in place, and a staged store writes OTHER lanes' words of this wavefront's 16 robots (no other wavefront touches
    // these robots).  Every load above must have returned before the first such store: an explicit s_waitcnt vmcnt(0) -- rel_last / target_last only feed `kin`, which
    // is parked in the SECOND flush, so the first flush's data dependences alone would not cover them
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
    ws.flush3(st, cur, tl, a.start, a.rel_last, a.target_last);   // foot_pos_start, foot_pos_rel_last_time <- cur, foot_pos_target_last_time <- y
    ws.flush3(kin, cur, cur, a.kin_out, a.cur_out, nullptr);
}
// a1mpc_control_tick_device: update_plan and the swing-leg block in ONE launch -- both run one lane per (robot, leg), and the swing block's inputs from the plan (the
// leg's new gait counter and foothold) are the lane's own registers.  Same lane functions as the two kernels: same bits, one launch less per tick.
__global__ __launch_bounds__(256) void a1mpc_plan_swing_kernel(const PlanArgs pa, const SwingArgs sa) {
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(16))) double stage[4][576];
    const int lane = static_cast<int>(threadIdx.x) & 63, wv = static_cast<int>(threadIdx.x) >> 6;
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t b = gid >> 2;
    const int leg = static_cast<int>(gid & 3);
    const int64_t wave_first = (static_cast<int64_t>(blockIdx.x) * 256 + wv * 64) >> 2;
    if (wave_first >= pa.n) return;
    const WaveStage ws{stage[wv], lane, static_cast<int>(pa.n - wave_first < 16 ? pa.n - wave_first : 16), wave_first};
    double rel[3] = {0, 0, 0}, ab[3] = {0, 0, 0}, wo[3] = {0, 0, 0}, gc = 0.0;
    double cur[3] = {0, 0, 0}, st[3] = {0, 0, 0}, tl[3] = {0, 0, 0}, kin[3] = {0, 0, 0};
    if (b < pa.n) {
        plan_lane(pa, b, leg, gc, rel, ab, wo);
        swing_lane(sa, b, leg, gc, rel, cur, st, tl, kin);
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): every load of the in-place arrays has returned before the staged stores write other lanes' words of them (see a1mpc_swing_kernel)
    ws.flush3(rel, ab, wo, pa.rel, pa.abs_, pa.world);
    ws.flush3(st, cur, tl, sa.start, sa.rel_last, sa.target_last);
    ws.flush3(kin, cur, cur, sa.kin_out, sa.cur_out, nullptr);
}

a1mpc_status a1mpc_swing_legs_batch(a1mpc_handle h, int32_t n, double counter_per_swing, double dt, const double* R_z,
                                    const double* foot_pos_abs, const double* gait_counter, const double* foot_pos_target_rel,
                                    const double* kp_foot, const double* kd_foot, double* foot_pos_start, double* foot_pos_rel_last_time,
                                    double* foot_pos_target_last_time, double* foot_pos_cur_out, double* foot_forces_kin_out) {
    if (!h) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle");
    if (n < 0 || !R_z || !foot_pos_abs || !gait_counter || !foot_pos_target_rel || !kp_foot || !kd_foot || !foot_pos_start || !foot_pos_rel_last_time ||
        !foot_pos_target_last_time || !foot_pos_cur_out || !foot_forces_kin_out || !(dt > 0))
        return fail(A1MPC_ERR_INVALID_ARGUMENT, "null input/output pointer or dt <= 0");
    if (n > h->max_batch) return fail(A1MPC_ERR_BATCH_TOO_LARGE, "n > max_batch given to a1mpc_create");
    if (n == 0) return A1MPC_OK;
    A1_HIP(hipSetDevice(h->device));
    if (a1mpc_status st = ensure_aux(h); st != A1MPC_OK) return st;
    const size_t N = n;
    hipStream_t s = h->stream;
    A1_ORDER(h, s);
    // staging: in [Rz 9 | foot 12 | gc 4 | target_rel 12] = 37, in/out + out [start 12 | rel_last 12 | target_last 12] (aux_in tail, 36) and [cur 12 | kin 12]
    double *d_Rz = h->d_aux_in, *d_fa = d_Rz + 9 * N, *d_gc = d_fa + 12 * N, *d_tr = d_gc + 4 * N;
    double *d_st = h->d_aux_out, *d_rl = d_st + 12 * N, *d_tl = d_rl + 12 * N;
    double *d_cur = d_tr + 12 * N, *d_kin = d_cur + 12 * N;   // aux_in has 64 doubles per robot: 37 + 24 = 61
    A1_HIP(hipMemcpyAsync(d_Rz, R_z, N * 9 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_fa, foot_pos_abs, N * 12 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_gc, gait_counter, N * 4 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_tr, foot_pos_target_rel, N * 12 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_st, foot_pos_start, N * 12 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_rl, foot_pos_rel_last_time, N * 12 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_tl, foot_pos_target_last_time, N * 12 * sizeof(double), hipMemcpyHostToDevice, s));
    SwingArgs a;
    a.n = n; a.counter_per_swing = counter_per_swing; a.dt = dt;
    for (int k = 0; k < 3; ++k) { a.kp[k] = kp_foot[k]; a.kd[k] = kd_foot[k]; }
    a.Rz = d_Rz; a.foot_pos_abs = d_fa; a.gait_counter = d_gc; a.target_rel = d_tr; a.start = d_st; a.rel_last = d_rl; a.target_last = d_tl;
    a.cur_out = d_cur; a.kin_out = d_kin;
    if (h->timing) A1_HIP(hipEventRecord(h->ev0, s));
    hipLaunchKernelGGL(a1mpc_swing_kernel, dim3(static_cast<unsigned>((N * 4 + 255) / 256)), dim3(256), 0, s, a);
    A1_HIP(hipGetLastError());
    if (h->timing) A1_HIP(hipEventRecord(h->ev1, s));
    h->timed = h->timing; A1_MARK(h, s);
    A1_HIP(hipMemcpyAsync(foot_pos_start, d_st, N * 12 * sizeof(double), hipMemcpyDeviceToHost, s));
    A1_HIP(hipMemcpyAsync(foot_pos_rel_last_time, d_rl, N * 12 * sizeof(double), hipMemcpyDeviceToHost, s));
    A1_HIP(hipMemcpyAsync(foot_pos_target_last_time, d_tl, N * 12 * sizeof(double), hipMemcpyDeviceToHost, s));
    A1_HIP(hipMemcpyAsync(foot_pos_cur_out, d_cur, N * 12 * sizeof(double), hipMemcpyDeviceToHost, s));
    A1_HIP(hipMemcpyAsync(foot_forces_kin_out, d_kin, N * 12 * sizeof(double), hipMemcpyDeviceToHost, s));
    A1_HIP(hipStreamSynchronize(s));
    return A1MPC_OK;
}

// ---- N4b: leg kinematics (S/GazeboA1ROS.cpp:264-279, A1Kinematics fk / jac restated from the leg model), one lane per (robot, leg) -----------
struct LegArgs {
    int32_t n;
    double rho_fix[20], rho_opt[12];
    const double *q, *qd, *R, *pos, *vel;
    double *rel, *Jb, *vrel, *pabs, *vabs, *pworld, *vworld;
};
// Round 4: the outputs (9 + 6 x 3 doubles per lane at strides of 72 / 24 bytes between lanes) leave through the wave's LDS stage (WaveStage): 64 consecutive doubles per
// store instruction instead of 36 / 12 cache lines touched for 512 bytes; 0.213 -> 0.106 ms at 524 288 robots, bit-identical.
__global__ __launch_bounds__(256) void a1mpc_leg_kernel(const LegArgs a) {
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(16))) double stage[4][576];
    const int lane = static_cast<int>(threadIdx.x) & 63, wv = static_cast<int>(threadIdx.x) >> 6;
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t b = gid >> 2;
    const int i = static_cast<int>(gid & 3);
    const int64_t wave_first = (static_cast<int64_t>(blockIdx.x) * 256 + wv * 64) >> 2;     // first robot of this wavefront
    if (wave_first >= a.n) return;
    const int wave_robots = static_cast<int>(a.n - wave_first < 16 ? a.n - wave_first : 16);
    const bool live = b < a.n;
    double p[3] = {0, 0, 0}, v[3] = {0, 0, 0}, J[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, pa[3] = {0, 0, 0}, va[3] = {0, 0, 0}, pw[3] = {0, 0, 0}, vw[3] = {0, 0, 0};
    if (live) {
        const double *q = a.q + b * 12 + 3 * i, *qd = a.qd + b * 12 + 3 * i, *f = a.rho_fix + 5 * i, *o = a.rho_opt + 3 * i;
        const double ox = f[0], oy = f[1], L = f[2] + o[1], lt = f[3], al = f[4] - o[2], r0 = o[0];
        const double s0 = sin(q[0]), c0 = cos(q[0]), s1 = sin(q[1]), c1 = cos(q[1]), s12 = sin(q[1] + q[2]), c12 = cos(q[1] + q[2]);
        const double Xq = r0 * c12 - al * s12, Zq = -(al * c12) - r0 * s12;
        const double Xr = Xq - lt * s1, Zp = Zq - lt * c1;
        p[0] = ox + Xr; p[1] = oy + (L * c0 - Zp * s0); p[2] = L * s0 + Zp * c0;
        J[0] = 0.0; J[1] = -p[2]; J[2] = p[1] - oy; J[3] = Zp; J[4] = s0 * Xr; J[5] = -(c0 * Xr); J[6] = Zq; J[7] = s0 * Xq; J[8] = -(c0 * Xq);
#pragma unroll
        for (int r = 0; r < 3; ++r) v[r] = J[r] * qd[0] + J[3 + r] * qd[1] + J[6 + r] * qd[2];
        const double* R = a.R + b * 9;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            pa[r] = R[3 * r] * p[0] + R[3 * r + 1] * p[1] + R[3 * r + 2] * p[2];
            va[r] = R[3 * r] * v[0] + R[3 * r + 1] * v[1] + R[3 * r + 2] * v[2];
            pw[r] = pa[r] + a.pos[b * 3 + r];
            vw[r] = va[r] + a.vel[b * 3 + r];
        }
    }
    const WaveStage ws{stage[wv], lane, wave_robots, wave_first};
    double* sg = stage[wv];
    // J blocks: lane (robot, leg) holds doubles [9 lane, 9 lane + 9) of the wave's 576
#pragma unroll
    for (int k = 0; k < 9; ++k) sg[9 * lane + k] = J[k];
    WaveStage::sync();
    {
        double* out = a.Jb + wave_first * 36;
        const int cnt = wave_robots * 36;
#pragma unroll
        for (int m = 0; m < 9; ++m) { const int e = lane + 64 * m; if (e < cnt) out[e] = sg[e]; }
    }
    WaveStage::sync();
    ws.flush3(p, v, pa, a.rel, a.vrel, a.pabs);
    ws.flush3(va, pw, vw, a.vabs, a.pworld, a.vworld);
}

a1mpc_status a1mpc_leg_state_batch(a1mpc_handle h, int32_t n, const double* joint_pos, const double* joint_vel, const double* R_world,
                                   const double* root_pos, const double* root_lin_vel, const double* rho_fix, const double* rho_opt,
                                   double* foot_pos_rel_out, double* j_foot_blocks_out, double* foot_vel_rel_out, double* foot_pos_abs_out,
                                   double* foot_vel_abs_out, double* foot_pos_world_out, double* foot_vel_world_out) {
    if (!h) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle");
    if (n < 0 || !joint_pos || !joint_vel || !R_world || !root_pos || !root_lin_vel || !rho_fix || !rho_opt || !foot_pos_rel_out || !j_foot_blocks_out)
        return fail(A1MPC_ERR_INVALID_ARGUMENT, "null input/output pointer");
    if (n > h->max_batch) return fail(A1MPC_ERR_BATCH_TOO_LARGE, "n > max_batch given to a1mpc_create");
    if (n == 0) return A1MPC_OK;
    A1_HIP(hipSetDevice(h->device));
    if (a1mpc_status st = ensure_aux(h); st != A1MPC_OK) return st;
    const size_t N = n;
    hipStream_t s = h->stream;
    A1_ORDER(h, s);
    // staging: aux_in [q 12 | qd 12 | R 9 | pos 3 | vel 3 | rel 12] = 51 of 64, aux_out [Jb 36 | vrel 12 | pabs 12 | vabs 12 | pworld 12 | vworld 12] = 96
    double *d_q = h->d_aux_in, *d_qd = d_q + 12 * N, *d_R = d_qd + 12 * N, *d_pos = d_R + 9 * N, *d_vel = d_pos + 3 * N, *d_rel = d_vel + 3 * N;
    double *d_Jb = h->d_aux_out, *d_vrel = d_Jb + 36 * N, *d_pabs = d_vrel + 12 * N, *d_vabs = d_pabs + 12 * N, *d_pw = d_vabs + 12 * N, *d_vw = d_pw + 12 * N;
    A1_HIP(hipMemcpyAsync(d_q, joint_pos, N * 12 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_qd, joint_vel, N * 12 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_R, R_world, N * 9 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_pos, root_pos, N * 3 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_vel, root_lin_vel, N * 3 * sizeof(double), hipMemcpyHostToDevice, s));
    LegArgs a;
    a.n = n;
    std::memcpy(a.rho_fix, rho_fix, sizeof a.rho_fix); std::memcpy(a.rho_opt, rho_opt, sizeof a.rho_opt);
    a.q = d_q; a.qd = d_qd; a.R = d_R; a.pos = d_pos; a.vel = d_vel; a.rel = d_rel; a.Jb = d_Jb;
    a.vrel = foot_vel_rel_out ? d_vrel : nullptr; a.pabs = foot_pos_abs_out ? d_pabs : nullptr; a.vabs = foot_vel_abs_out ? d_vabs : nullptr;
    a.pworld = foot_pos_world_out ? d_pw : nullptr; a.vworld = foot_vel_world_out ? d_vw : nullptr;
    if (h->timing) A1_HIP(hipEventRecord(h->ev0, s));
    hipLaunchKernelGGL(a1mpc_leg_kernel, dim3(static_cast<unsigned>((N * 4 + 255) / 256)), dim3(256), 0, s, a);
    A1_HIP(hipGetLastError());
    if (h->timing) A1_HIP(hipEventRecord(h->ev1, s));
    h->timed = h->timing; A1_MARK(h, s);
    A1_HIP(hipMemcpyAsync(foot_pos_rel_out, d_rel, N * 12 * sizeof(double), hipMemcpyDeviceToHost, s));
    A1_HIP(hipMemcpyAsync(j_foot_blocks_out, d_Jb, N * 36 * sizeof(double), hipMemcpyDeviceToHost, s));
    if (foot_vel_rel_out) A1_HIP(hipMemcpyAsync(foot_vel_rel_out, d_vrel, N * 12 * sizeof(double), hipMemcpyDeviceToHost, s));
    if (foot_pos_abs_out) A1_HIP(hipMemcpyAsync(foot_pos_abs_out, d_pabs, N * 12 * sizeof(double), hipMemcpyDeviceToHost, s));
    if (foot_vel_abs_out) A1_HIP(hipMemcpyAsync(foot_vel_abs_out, d_vabs, N * 12 * sizeof(double), hipMemcpyDeviceToHost, s));
    if (foot_pos_world_out) A1_HIP(hipMemcpyAsync(foot_pos_world_out, d_pw, N * 12 * sizeof(double), hipMemcpyDeviceToHost, s));
    if (foot_vel_world_out) A1_HIP(hipMemcpyAsync(foot_vel_world_out, d_vw, N * 12 * sizeof(double), hipMemcpyDeviceToHost, s));
    A1_HIP(hipStreamSynchronize(s));
    return A1MPC_OK;
}

// ---- N4c: A1BasicEKF (S/A1BasicEKF.cpp), 32 lanes per robot, two robots per wavefront -----------------------------------------------
// Lane r owns row r of every matrix (18 state rows, 28 measurement rows); rows meet through the robot's LDS image (P, Pbar, S / S^-1 C, the
// pivot row of the elimination).  The arithmetic follows the dense products of the reference term for term (inner index ascending, the
// exact zeros of A, B, C dropped), no FMA contraction, so the result is bit-identical to the oracle's dense restatement.
constexpr int kEkfState = 18 + 18 * 18 + 1;
struct EkfArgs {
    int32_t n;
    double dt;
    int32_t flat;
    double* state;
    const uint8_t* mode;
    const double *ff, *R, *acc, *w, *fk, *fv;
    double *pos_out, *vel_out;
    uint8_t* ec_out;
};
__global__ __launch_bounds__(256) void a1mpc_ekf_init_kernel(const EkfArgs a) {  // init_state :54-68 for robots whose filter is new
#pragma clang fp contract(off)
    const int64_t b = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (b >= a.n) return;
    double* st = a.state + b * kEkfState;
    if (st[18 + 324] != 0.0) return;
    double *x = st, *P = st + 18;
    for (int i = 0; i < 324; ++i) P[i] = 0.0;
    for (int i = 0; i < 18; ++i) { P[i * 18 + i] = 1.0 * 3; x[i] = 0.0; }
    x[2] = 0.09;
    const double* R = a.R + b * 9; const double* fk = a.fk + b * 12;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 3; ++r) x[6 + i * 3 + r] = (R[3 * r] * fk[3 * i] + R[3 * r + 1] * fk[3 * i + 1] + R[3 * r + 2] * fk[3 * i + 2]) + x[r];
    st[18 + 324] = 2.0;  // initialised in THIS call: the update kernel turns it into 1 and leaves the robot alone
    for (int r = 0; r < 3; ++r) { a.pos_out[b * 3 + r] = x[r]; a.vel_out[b * 3 + r] = x[3 + r]; }
    for (int i = 0; i < 4; ++i) a.ec_out[b * 4 + i] = 0;
}
__device__ inline void half_sync() {  // orders LDS traffic between the lanes of a wavefront (see row_sync)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// column list of C's row r: up to two (column, sign) pairs in ascending column order
__device__ inline void ekf_c_row(int r, int& c0, double& s0, int& c1, double& s1) {
    c1 = -1; s1 = 0.0;
    if (r < 12) { c0 = r % 3; s0 = -1.0; c1 = 6 + r; s1 = 1.0; }            // -pos + foot
    else if (r < 24) { c0 = 3 + (r - 12) % 3; s0 = 1.0; }                    // vel
    else { c0 = 6 + (r - 24) * 3 + 2; s0 = 1.0; }                            // foot height
}
// ONE arithmetic, round 6: the measurement update never forms S^-1.  S = L D L' by forward elimination without pivoting (S is symmetric positive definite), the
// right-hand sides [C Pbar | error_y] ride along, and with Y = L^-1 [C Pbar | error_y]
//     x = xbar + Y_P' D^-1 y_e,      P = Pbar - Y_P' D^-1 Y_P                                                                  (:134-139)
// -- nothing is solved backwards, and the four dense products behind the explicit inverse (S^-1 error_y, S^-1 C, Pbar C' (S^-1 C), (..) Pbar) are one 28-term
// rank update.  A third of the inverse's arithmetic, and closer to an 80-bit evaluation of the filter than the explicit inverse is (state 9.8e-15 against 7.3e-14,
// covariance 1.0e-15 against 1.0e-12 per tick: the CPU suite measures it).  Until round 6: in-place Gauss-Jordan sweeps to -S^-1 (2 x 28^3 multiply-adds per robot) with the
// pivot column exchanged through LDS, two residencies of it (profiles/r05_ekf_residency.txt).
// No LDS in the elimination.  A robot is two DPP rows of 16 lanes; lane i holds ROW i of S (M[28]) and COLUMN i of the right-hand sides (B[28]: column c < 18 of C Pbar,
// column 18 = error_y).  Step k, p = a_kk:  f_i = a_ik / p,  a_ij -= f_i a_jk (j > k);  t_c = b_kc / p,  b_ic -= a_ik t_c (i > k).  a_jk and a_ik live in lanes j and i: the column reaches
// the lane that needs it as the DPP source of a v_fmac_f64_dpp (row_newbcast), after one v_permlane16_swap pair has put each 16-lane half of it on both rows of the
// robot (ekf_rows) -- one exchange per step serves both updates.  The pivot row is taken from COLUMN k as the other lanes hold it (a_jk for a_kj): only entries of the lower triangle ever feed another entry, so this
// is the standard right-looking L D L' and the upper triangle a lane drags along is never read.  Rows / columns that are done keep executing the updates on dead
// registers (no branch inside a step).  The rank update reads y_rb from lane b the same way.  (First cut of this round: rows of Y published to LDS by the pivot lane,
// 10 masked 16-byte stores per step -- 40 % fewer VALU instructions than the inverse and SLOWER, 0.59 ms per 65 536 robots against 0.49: every multiply-add took a fresh
// broadcast operand from LDS and the CU's one LDS served four SIMDs, profiles/r06_ekf_ldl.md.)  oracle/a1mpc_oracle.c ekf_step_impl(device = 1) is the same
// arithmetic in the same order.
extern "C++" {   // (this section sits inside the ABI's extern "C" block)
template <int I, int N, class F>
__device__ __forceinline__ void ekf_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); ekf_for<I + 1, N>(f); }
}
// (ua, ub) <- x:  ua = the lanes of the robot's EVEN row of x on both of its rows, ub = the odd row's (v_permlane16_swap: odd rows of vdst <-> even rows of src)
__device__ __forceinline__ void ekf_rows(double x, double& ua, double& ub) {
    double c0, c1;
    asm("v_mov_b64 %0, %2\n"
        "v_mov_b64 %1, %2\n"
        "s_nop 1" : "=&v"(c0), "=&v"(c1) : "v"(x));   // (VALU write -> v_permlane16_swap read: 2 wait states, which hipcc cannot see behind an asm statement)
    const unsigned long long b0 = __builtin_bit_cast(unsigned long long, c0), b1 = __builtin_bit_cast(unsigned long long, c1);
    const auto r0 = __builtin_amdgcn_permlane16_swap(static_cast<unsigned>(b0), static_cast<unsigned>(b1), false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(static_cast<unsigned>(b0 >> 32), static_cast<unsigned>(b1 >> 32), false, false);
    ua = __builtin_bit_cast(double, static_cast<unsigned long long>(r0[0]) | (static_cast<unsigned long long>(r1[0]) << 32));
    ub = __builtin_bit_cast(double, static_cast<unsigned long long>(r0[1]) | (static_cast<unsigned long long>(r1[1]) << 32));
    asm volatile("s_nop 1" : "+v"(ua), "+v"(ub));     // (VALU write -> DPP read: 2 wait states; the DPP reads below sit inside asm statements)
}
template <int L>
__device__ __forceinline__ double ekf_bcast(double v) { return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + L, 0xF, 0xF, true); }   // row_newbcast:L
// acc += m * (lane L of my row's x)  /  acc -= ..  as ONE instruction (v_fmac_f64 is the only FP64 arithmetic with a DPP form)
template <int L>
__device__ __forceinline__ void ekf_fma(double& acc, double m, double x) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(m), "n"(L));
}
template <int L>
__device__ __forceinline__ void ekf_fnma(double& acc, double m, double x) {
    asm("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(m), "n"(L));
}
constexpr int kEkfLds = 324 + 16 + 18 + 18 + 28;   // doubles per robot: Pbar (+ 16 words that idle lanes read past it), x, xbar, error_y: 3.2 KB
__device__ __forceinline__ void ekf_update_robot(const EkfArgs& a, double* __restrict__ lds_g, const int g, const int l) {
#pragma clang fp contract(off)
    const int64_t b = static_cast<int64_t>(blockIdx.x) * 2 + g;
    if (b >= a.n) return;
    double* st = a.state + b * kEkfState;
    double *Pb = lds_g, *xs = Pb + 324 + 16, *xb = xs + 18, *zs = xb + 18;
    const double dt = a.dt;
    // Every global load of the tick is issued here, before the first use of any of them: ONE exposure to the memory latency per robot.  (Until round 6 the loads sat where
    // they were used, behind the lane-dependent branches of the process update and of the measurement vector -- 18 pairs of row loads each waited for on its own, the
    // contact inputs behind the flag: half of a wavefront's life was spent parked at s_waitcnt, SQ_WAIT_ANY / SQ_WAVE_CYCLES = 0.49, profiles/r06_ekf_ldl.md.)
    const double* Pg = st + 18;  // P of the previous tick: lane i < 18 reads row i (and row 3 + i for i < 3: A P) straight from global memory
    const int rowA = l < 18 ? l : 0, rowB = l < 3 ? 3 + l : rowA;
    const int r = l < 28 ? l : 27;                                       // my row of S / C (lanes 28 - 31 idle along on row 27's inputs)
    const int leg = r < 24 ? (r % 12) / 3 : r - 24, cr = l % 3;         // the leg my measurement row belongs to; my component (process rows 3 - 5 and measurement rows alike)
    const double flag = st[18 + 324];
    double Pa[18], Pc[18];
#pragma unroll
    for (int j = 0; j < 18; ++j) { Pa[j] = Pg[rowA * 18 + j]; Pc[j] = Pg[rowB * 18 + j]; }
    const double xl = st[rowA];
    const int mode = a.mode[b];
    double ffv[4], accv[3], wv[3], Rc[3], f[3], v[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) ffv[i] = a.ff[b * 4 + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) { accv[i] = a.acc[b * 3 + i]; wv[i] = a.w[b * 3 + i]; Rc[i] = a.R[b * 9 + 3 * cr + i]; f[i] = a.fk[b * 12 + 3 * leg + i]; v[i] = a.fv[b * 12 + 3 * leg + i]; }
    // (a robot whose filter was initialised in THIS call -- flag 2 -- or never -- flag 0 -- runs through the arithmetic on whatever its state holds and stores nothing: the
    // flag is looked at where the results leave, so its load is not a round trip of its own ahead of the others)
    if (l < 18) xs[l] = xl;
    double ec[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ec[i] = mode == 0 ? 1.0 : fmin(fmax(ffv[i] / (100.0 - 0.0), 0.0), 1.0);   // :79-86
    const double PIMU = 0.01, VIMU = 0.01, PFOOT = 0.01, S_PIMU_REL = 0.001, S_VIMU_REL = 0.1, S_ZFOOT = 0.001;       // A1BasicEKF.h:15-20
    half_sync();
    // ---- process update (:72-112): xbar = A x + B u, Pbar = A P A' + Q; lane i < 18 owns row i
    double xbv = 0.0;
    if (l < 18) {
        if (l < 3) xbv = (xl + dt * xs[3 + l]) + 0.0;
        else if (l < 6) { const double u = (Rc[0] * accv[0] + Rc[1] * accv[1] + Rc[2] * accv[2]) + (cr == 2 ? -9.81 : 0.0); xbv = xl + dt * u; }
        else xbv = xl + 0.0;
        xb[l] = xbv;
        double T[18];
#pragma unroll
        for (int j = 0; j < 18; ++j) T[j] = l < 3 ? Pa[j] + dt * Pc[j] : Pa[j];
        double q;
        if (l < 3) q = PIMU * dt / 20.0; else if (l < 6) q = VIMU * dt * 9.8 / 20.0; else q = (1 + (1 - ec[(l - 6) / 3]) * 1e3) * dt * PFOOT;
#pragma unroll
        for (int j = 0; j < 18; ++j) Pb[l * 18 + j] = (j < 3 ? T[j] + T[3 + j] * dt : T[j]) + (j == l ? q : 0.0);
    }
    half_sync();
    // ---- innovation (:115-131): lane r < 28 owns row r of S and its error_y entry
    double M[28];
#pragma unroll
    for (int c = 0; c < 28; ++c) M[c] = 0.0;
    if (l < 28) {
        int c0, c1; double s0, s1;
        ekf_c_row(r, c0, s0, c1, s1);
        const double yhat = c1 >= 0 ? s0 * xb[c0] + s1 * xb[c1] : s0 * xb[c0];
        // the three kinds of measurement rows, each evaluated by every lane on its own leg / component and selected (the same operations as a branch per kind would do)
        const double ecl = leg == 0 ? ec[0] : (leg == 1 ? ec[1] : (leg == 2 ? ec[2] : ec[3]));
        const double wgt = 1 + (1 - ecl) * 1e3;
        const double y_pos = Rc[0] * f[0] + Rc[1] * f[1] + Rc[2] * f[2];
        const double wx = wv[0], wy = wv[1], wz = wv[2];
        const double sk0 = 0.0 * f[0] + -wz * f[1] + wy * f[2], sk1 = wz * f[0] + 0.0 * f[1] + -wx * f[2], sk2 = -wy * f[0] + wx * f[1] + 0.0 * f[2];
        const double lv0 = -v[0] - sk0, lv1 = -v[1] - sk1, lv2 = -v[2] - sk2;
        const double rl = Rc[0] * lv0 + Rc[1] * lv1 + Rc[2] * lv2;
        const double y_vel = (1.0 - ecl) * xs[3 + cr] + ecl * rl;
        const double y_z = (1.0 - ecl) * (xs[2] + f[2]) + ecl * 0;
        const double y = r < 12 ? y_pos : (r < 24 ? y_vel : y_z);
        const double rd = r < 12 ? wgt * S_PIMU_REL : (r < 24 ? wgt * S_VIMU_REL : (a.flat ? (1 + (1 - ecl) * 1e3) * S_ZFOOT : 1e5));
        double CP[18];   // my row of C Pbar
#pragma unroll
        for (int j = 0; j < 18; ++j) CP[j] = c1 >= 0 ? s0 * Pb[c0 * 18 + j] + s1 * Pb[c1 * 18 + j] : s0 * Pb[c0 * 18 + j];
#pragma unroll
        for (int c = 0; c < 28; ++c) {
            int d0, d1; double t0, t1;
            ekf_c_row(c, d0, t0, d1, t1);
            const double vv = d1 >= 0 ? CP[d0] * t0 + CP[d1] * t1 : CP[d0] * t0;
            M[c] = vv + (c == r ? rd : 0.0);
        }
        zs[r] = y - yhat;   // error_y: column 18 of the right-hand sides, which lane 18 collects below
        // :131  S <- (S + S') / 2.  S[c][l], the entry lane c holds, is recomputed here from Pbar by lane c's own operations (C row of c times Pbar, then my C row)
#pragma unroll
        for (int c = 0; c < 28; ++c) {
            int e0, e1; double u0, u1;
            ekf_c_row(c, e0, u0, e1, u1);
            const double cp0 = e1 >= 0 ? u0 * Pb[e0 * 18 + c0] + u1 * Pb[e1 * 18 + c0] : u0 * Pb[e0 * 18 + c0];
            const double cp1 = c1 >= 0 ? (e1 >= 0 ? u0 * Pb[e0 * 18 + c1] + u1 * Pb[e1 * 18 + c1] : u0 * Pb[e0 * 18 + c1]) : 0.0;
            const double tr = (c1 >= 0 ? cp0 * s0 + cp1 * s1 : cp0 * s0) + 0.0;   // (+ 0.0: lane c added its "not the diagonal" zero)
            M[c] = c == l ? 0.5 * (M[c] + M[c]) : (c > l ? 0.5 * (M[c] + tr) : 0.5 * (tr + M[c]));
        }
    }
    half_sync();
    // the right-hand sides, a COLUMN per lane: B[r] = (C Pbar)[r][l] for l < 18 (row r of C has one or two entries: known at compile time), error_y[r] on lane 18
    double B[28];
    ekf_for<0, 28>([&](auto Rr) {
        constexpr int r = decltype(Rr)::value;
        constexpr int c0 = r < 12 ? r % 3 : (r < 24 ? 3 + (r - 12) % 3 : 6 + (r - 24) * 3 + 2);   // (ekf_c_row, as constants)
        double v;
        if constexpr (r < 12) v = -1.0 * Pb[c0 * 18 + l] + 1.0 * Pb[(6 + r) * 18 + l];
        else v = 1.0 * Pb[c0 * 18 + l];
        B[r] = l < 18 ? v : (l == 18 ? zs[r] : 0.0);
    });
    // ---- L D L' with the right-hand sides riding along (see the header of this section); every lane of the robot executes every instruction
    double D = 0.0;   // 1 / d_r on lane r
    ekf_for<0, 28>([&](auto K) {
        constexpr int k = decltype(K)::value;
        double ua, ub;
        ekf_rows(M[k], ua, ub);                                    // column k of the trailing matrix: a_jk on lane j
        const double pinv = 1.0 / ekf_bcast<k % 16>(k < 16 ? ua : ub);
        D = l == k ? pinv : D;
        if constexpr (k < 27) {
            const double f = M[k] * pinv, t = B[k] * pinv;   // f_i = a_ik / p on lane i (my row's multiplier);  t_c = b_kc / p on lane c (my column of the scaled pivot row)
            ekf_for<k + 1, 28>([&](auto J) { constexpr int j = decltype(J)::value; ekf_fnma<j % 16>(M[j], f, j < 16 ? ua : ub); });   // a_ij -= f_i a_jk
            ekf_for<k + 1, 28>([&](auto I) { constexpr int i = decltype(I)::value; ekf_fnma<i % 16>(B[i], t, i < 16 ? ua : ub); });   // b_ic -= a_ik t_c: the column serves both updates
        }
    });
    // ---- measurement update (:134-140): lane a < 18 owns row a of  Y_P' D^-1 [Y_P | y_e]  (the sums run over r = 0 .. 27 in ascending order); y_rb comes from lane b
    double G[19];
#pragma unroll
    for (int j = 0; j < 19; ++j) G[j] = 0.0;
    {
        double da, db;
        ekf_rows(D, da, db);
        ekf_for<0, 28>([&](auto Rr) {
            constexpr int r = decltype(Rr)::value;
            const double wr = B[r] * ekf_bcast<r % 16>(r < 16 ? da : db);
            double va, vb;
            ekf_rows(B[r], va, vb);
            ekf_for<0, 19>([&](auto Cc) { constexpr int c = decltype(Cc)::value; ekf_fma<c % 16>(G[c], wr, c < 16 ? va : vb); });
        });
    }
    double Tn[18], xnew = 0.0;
    if (l < 18) {
        xnew = xbv + G[18];
#pragma unroll
        for (int j = 0; j < 18; ++j) Tn[j] = Pb[l * 18 + j] - G[j];
    }
    half_sync();  // every lane is done reading Pbar: its region now stages the unsymmetrised update for the transposed read
    double* Pm = Pb;
    if (l < 18)
        for (int j = 0; j < 18; ++j) Pm[l * 18 + j] = Tn[j];
    half_sync();
    const bool live = flag == 1.0;
    if (l == 0 && flag == 2.0) st[18 + 324] = 1.0;   // initialised in this call: an ordinary robot from the next tick on
    if (l < 18 && live) {
        double Pn[18];
        for (int j = 0; j < 18; ++j) Pn[j] = 0.5 * (Tn[j] + Pm[j * 18 + l]);                                       // :140
        const double p00 = 0.5 * (Pm[0] + Pm[0]), p01 = 0.5 * (Pm[1] + Pm[18]), p10 = 0.5 * (Pm[18] + Pm[1]), p11 = 0.5 * (Pm[19] + Pm[19]);
        if (p00 * p11 - p01 * p10 > 1e-6) {                                                                         // :143-147
            for (int j = 0; j < 18; ++j) {
                if (l < 2 && j >= 2) Pn[j] = 0.0;
                if (l >= 2 && j < 2) Pn[j] = 0.0;
                if (l < 2 && j < 2) Pn[j] /= 10.0;
            }
        }
        for (int j = 0; j < 18; ++j) st[18 + l * 18 + j] = Pn[j];
        st[l] = xnew;
        if (l < 3) a.pos_out[b * 3 + l] = xnew; else if (l < 6) a.vel_out[b * 3 + l - 3] = xnew;
    }
    if (l < 4 && live) a.ec_out[b * 4 + l] = ec[l] < 0.5 ? 0 : 1;                                                   // :151-157
}
}  // extern "C++"
__global__ __launch_bounds__(64, 3) void a1mpc_ekf_kernel(const EkfArgs a) {
    __shared__ __attribute__((aligned(16))) double lds[2][kEkfLds];
    const int g = static_cast<int>(threadIdx.x) >> 5;
    ekf_update_robot(a, lds[g], g, static_cast<int>(threadIdx.x) & 31);
}
static void launch_ekf_update(const EkfArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(a1mpc_ekf_kernel, dim3(static_cast<unsigned>((a.n + 1) / 2)), dim3(64), 0, s, a);
}
#define EKF_LAUNCH(a) launch_ekf_update(a, 0)   // (tools/ubench/ekf_bench.py cuts this section into a stand-alone timing harness)

a1mpc_status a1mpc_reset_ekf_state(a1mpc_handle h) {
    if (!h) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle");
    if (!h->d_ekf_state) return A1MPC_OK;
    A1_HIP(hipSetDevice(h->device));
    A1_ORDER(h, h->stream);  // the memsets below must not overtake a launch still running on a caller's stream
    A1_HIP(hipMemsetAsync(h->d_ekf_state, 0, static_cast<size_t>(h->max_batch) * kEkfState * sizeof(double), h->stream));
    A1_HIP(hipStreamSynchronize(h->stream));
    h->ekf_ready_n = 0;
    return A1MPC_OK;
}
a1mpc_status a1mpc_ekf_update_batch(a1mpc_handle h, int32_t n, double dt, int32_t assume_flat_ground, const uint8_t* movement_mode,
                                    const double* foot_force, const double* R_world, const double* imu_acc, const double* imu_ang_vel,
                                    const double* foot_pos_rel, const double* foot_vel_rel, double* root_pos_out, double* root_lin_vel_out,
                                    uint8_t* estimated_contacts_out) {
    if (!h) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle");
    if (n < 0 || !movement_mode || !foot_force || !R_world || !imu_acc || !imu_ang_vel || !foot_pos_rel || !foot_vel_rel || !root_pos_out ||
        !root_lin_vel_out || !estimated_contacts_out || !(dt > 0))
        return fail(A1MPC_ERR_INVALID_ARGUMENT, "null input/output pointer or dt <= 0");
    if (n > h->max_batch) return fail(A1MPC_ERR_BATCH_TOO_LARGE, "n > max_batch given to a1mpc_create");
    if (n == 0) return A1MPC_OK;
    A1_HIP(hipSetDevice(h->device));
    if (a1mpc_status st = ensure_aux(h); st != A1MPC_OK) return st;
    const size_t N = n;
    hipStream_t s = h->stream;
    A1_ORDER(h, s);
    if (!h->d_ekf_state) {
        A1_HIP(hipMalloc(&h->d_ekf_state, static_cast<size_t>(h->max_batch) * kEkfState * sizeof(double)));
        A1_HIP(hipMemsetAsync(h->d_ekf_state, 0, static_cast<size_t>(h->max_batch) * kEkfState * sizeof(double), s));
    }
    // staging: aux_in [ff 4 | R 9 | acc 3 | w 3 | fk 12 | fv 12] = 43, aux_out [pos 3 | vel 3]
    double *d_ff = h->d_aux_in, *d_R = d_ff + 4 * N, *d_acc = d_R + 9 * N, *d_w = d_acc + 3 * N, *d_fk = d_w + 3 * N, *d_fv = d_fk + 12 * N;
    double *d_pos = h->d_aux_out, *d_vel = d_pos + 3 * N;
    uint8_t *d_mm = h->d_aux_u8, *d_ec = h->d_aux_u8 + 8 * N;
    A1_HIP(hipMemcpyAsync(d_ff, foot_force, N * 4 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_R, R_world, N * 9 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_acc, imu_acc, N * 3 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_w, imu_ang_vel, N * 3 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_fk, foot_pos_rel, N * 12 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_fv, foot_vel_rel, N * 12 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_mm, movement_mode, N, hipMemcpyHostToDevice, s));
    EkfArgs a;
    a.n = n; a.dt = dt; a.flat = assume_flat_ground; a.state = h->d_ekf_state; a.mode = d_mm; a.ff = d_ff; a.R = d_R; a.acc = d_acc; a.w = d_w;
    a.fk = d_fk; a.fv = d_fv; a.pos_out = d_pos; a.vel_out = d_vel; a.ec_out = d_ec;
    if (h->timing) A1_HIP(hipEventRecord(h->ev0, s));
    if (n > h->ekf_ready_n) {   // (robots 0 .. ekf_ready_n - 1 have been through init_state: nothing for the init kernel to do -- one launch less per tick)
        hipLaunchKernelGGL(a1mpc_ekf_init_kernel, dim3(static_cast<unsigned>((N + 255) / 256)), dim3(256), 0, s, a);
        if (hipError_t ei = hipGetLastError(); ei != hipSuccess) { h->ekf_ready_n = 0; return fail(A1MPC_ERR_HIP, std::string("a1mpc_ekf_init_kernel: ") + hipGetErrorString(ei)); }
        h->ekf_ready_n = n;   // (only once the launch has been accepted -- ADVICE r5: a failed init launch must not leave the robots marked as initialised)
    }
    launch_ekf_update(a, s);
    A1_HIP(hipGetLastError());
    if (h->timing) A1_HIP(hipEventRecord(h->ev1, s));
    h->timed = h->timing; A1_MARK(h, s);
    A1_HIP(hipMemcpyAsync(root_pos_out, d_pos, N * 3 * sizeof(double), hipMemcpyDeviceToHost, s));
    A1_HIP(hipMemcpyAsync(root_lin_vel_out, d_vel, N * 3 * sizeof(double), hipMemcpyDeviceToHost, s));
    A1_HIP(hipMemcpyAsync(estimated_contacts_out, d_ec, N * 4, hipMemcpyDeviceToHost, s));
    A1_HIP(hipStreamSynchronize(s));
    return A1MPC_OK;
}

// ---- N3: compute_joint_torques (S/A1RobotControl.cpp:289-319), one lane per (robot, leg); no FMA contraction ---------------------
struct TorqueArgs {
    int32_t n;
    const uint8_t *active, *contacts;
    const double *Jb, *grf, *fkin, *tg;
    double km[3];
    double* tau;
};
__global__ __launch_bounds__(256) void a1mpc_torque_kernel(const TorqueArgs a) {
#pragma clang fp contract(off)
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t b = gid >> 2;
    const int leg = static_cast<int>(gid & 3);
    if (b >= a.n) return;
    double* out = a.tau + b * 12 + 3 * leg;
    if (!a.active[b]) { out[0] = 0.0; out[1] = 0.0; out[2] = 0.0; return; }   // :294-295
    const double* gr = a.grf + b * 12 + 3 * leg;
    const double* fk = a.fkin + b * 12 + 3 * leg;
    double t[3];
    leg_joint_torque(a.Jb + b * 36 + 9 * leg, a.contacts[b * 4 + leg] != 0, gr[0], gr[1], gr[2], a.km[0] * fk[0], a.km[1] * fk[1], a.km[2] * fk[2], t);   // (a1mpc_solver.hpp: shared with the MPC kernels' output stage)
    const double t0 = t[0], t1 = t[1], t2 = t[2];
    const double* g = a.tg + b * 12 + 3 * leg;
    const double v0 = t0 + g[0], v1 = t1 + g[1], v2 = t2 + g[2];                // :311
    if (!isnan(v0)) out[0] = v0;                                                // :314-317
    if (!isnan(v1)) out[1] = v1;
    if (!isnan(v2)) out[2] = v2;
}

a1mpc_status a1mpc_joint_torques_batch(a1mpc_handle h, int32_t n, const uint8_t* active, const uint8_t* contacts, const double* j_foot_blocks,
                                       const double* grf, const double* f_kin, const double* km_foot, const double* torques_gravity,
                                       double* joint_torques) {
    if (!h) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle");
    if (n < 0 || !active || !contacts || !j_foot_blocks || !grf || !f_kin || !km_foot || !torques_gravity || !joint_torques)
        return fail(A1MPC_ERR_INVALID_ARGUMENT, "null input/output pointer");
    if (n > h->max_batch) return fail(A1MPC_ERR_BATCH_TOO_LARGE, "n > max_batch given to a1mpc_create");
    if (n == 0) return A1MPC_OK;
    A1_HIP(hipSetDevice(h->device));
    if (a1mpc_status st = ensure_aux(h); st != A1MPC_OK) return st;
    const size_t N = n;
    hipStream_t s = h->stream;
    A1_ORDER(h, s);
    // staging: in [J 36 | grf 12 | f_kin 12], out [tg 12 | tau 12]
    double *d_J = h->d_aux_in, *d_grf = d_J + 36 * N, *d_fk = d_grf + 12 * N, *d_tg = h->d_aux_out, *d_tau = d_tg + 12 * N;
    uint8_t *d_c = h->d_aux_u8, *d_act = h->d_aux_u8 + 8 * N;
    A1_HIP(hipMemcpyAsync(d_J, j_foot_blocks, N * 36 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_grf, grf, N * 12 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_fk, f_kin, N * 12 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_tg, torques_gravity, N * 12 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_tau, joint_torques, N * 12 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_c, contacts, N * 4, hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_act, active, N, hipMemcpyHostToDevice, s));
    TorqueArgs a;
    a.n = n; a.active = d_act; a.contacts = d_c; a.Jb = d_J; a.grf = d_grf; a.fkin = d_fk; a.tg = d_tg; a.tau = d_tau;
    a.km[0] = km_foot[0]; a.km[1] = km_foot[1]; a.km[2] = km_foot[2];
    if (h->timing) A1_HIP(hipEventRecord(h->ev0, s));
    hipLaunchKernelGGL(a1mpc_torque_kernel, dim3(static_cast<unsigned>((N * 4 + 255) / 256)), dim3(256), 0, s, a);
    A1_HIP(hipGetLastError());
    if (h->timing) A1_HIP(hipEventRecord(h->ev1, s));
    h->timed = h->timing; A1_MARK(h, s);
    A1_HIP(hipMemcpyAsync(joint_torques, d_tau, N * 12 * sizeof(double), hipMemcpyDeviceToHost, s));
    A1_HIP(hipStreamSynchronize(s));
    return A1MPC_OK;
}

void a1mpc_default_gait_config(a1mpc_gait_config* g) {
    if (!g) return;
    std::memset(g, 0, sizeof *g);
    g->counter_per_gait = 240.0; g->counter_per_swing = 120.0;        // S/A1CtrlStates.h:24-25
    g->control_dt = 0.0025;                                            // MAIN_UPDATE_FREQUENCY 2.5 ms, S/A1CtrlStates.h:332
    g->foot_delta_x_limit = 0.1; g->foot_delta_y_limit = 0.1;          // S/A1Params.h:44-45
    const double dfp[12] = {0.17, 0.15, -0.35, 0.17, -0.15, -0.35, -0.17, 0.15, -0.35, -0.17, -0.15, -0.35};  // S/A1CtrlStates.h:45-47
    std::memcpy(g->default_foot_pos, dfp, sizeof dfp);
    const double rs[4] = {0.0, 120.0, 120.0, 0.0};                     // S/A1CtrlStates.h:324
    std::memcpy(g->gait_counter_reset, rs, sizeof rs);
}

a1mpc_status a1mpc_update_plan_batch(a1mpc_handle h, const a1mpc_gait_config* gait, int32_t n, const uint8_t* movement_mode,
                                     double* gait_counter, const double* gait_counter_speed, const double* root_lin_vel,
                                     const double* R_z, const double* R_world, const double* root_pos, const double* root_lin_vel_d,
                                     uint8_t* plan_contacts_out, double* rel_out, double* abs_out, double* world_out) {
    if (!h || !gait) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle/config");
    if (n < 0 || !movement_mode || !gait_counter || !gait_counter_speed || !root_lin_vel || !R_z || !R_world || !root_pos || !root_lin_vel_d ||
        !plan_contacts_out)
        return fail(A1MPC_ERR_INVALID_ARGUMENT, "null input/output pointer");
    if (n > h->max_batch) return fail(A1MPC_ERR_BATCH_TOO_LARGE, "n > max_batch given to a1mpc_create");
    if (n == 0) return A1MPC_OK;
    A1_HIP(hipSetDevice(h->device));
    if (a1mpc_status st = ensure_aux(h); st != A1MPC_OK) return st;
    const size_t N = n;
    hipStream_t s = h->stream;
    A1_ORDER(h, s);
    // staging: in [gc 4 | spd 4 | v 3 | vd 3 | pos 3 | Rz 9 | Rw 9], out [rel 12 | abs 12 | world 12]
    double* din = h->d_aux_in;
    double *d_gc = din, *d_spd = d_gc + 4 * N, *d_v = d_spd + 4 * N, *d_vd = d_v + 3 * N, *d_pos = d_vd + 3 * N, *d_Rz = d_pos + 3 * N, *d_Rw = d_Rz + 9 * N;
    double *d_rel = h->d_aux_out, *d_abs = d_rel + 12 * N, *d_world = d_abs + 12 * N;
    uint8_t *d_mm = h->d_aux_u8, *d_pc = h->d_aux_u8 + 8 * N;
    A1_HIP(hipMemcpyAsync(d_gc, gait_counter, N * 4 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_spd, gait_counter_speed, N * 4 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_v, root_lin_vel, N * 3 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_vd, root_lin_vel_d, N * 3 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_pos, root_pos, N * 3 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_Rz, R_z, N * 9 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_Rw, R_world, N * 9 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_mm, movement_mode, N, hipMemcpyHostToDevice, s));
    PlanArgs a;
    a.g = *gait; a.n = n; a.movement_mode = d_mm; a.gait_counter = d_gc; a.gait_counter_speed = d_spd; a.root_lin_vel = d_v; a.Rz = d_Rz; a.Rw = d_Rw;
    a.root_pos = d_pos; a.root_lin_vel_d = d_vd; a.plan_contacts = d_pc; a.rel = d_rel; a.abs_ = d_abs; a.world = d_world;
    if (h->timing) A1_HIP(hipEventRecord(h->ev0, s));
    hipLaunchKernelGGL(a1mpc_plan_kernel, dim3(static_cast<unsigned>((N * 4 + 255) / 256)), dim3(256), 0, s, a);
    A1_HIP(hipGetLastError());
    if (h->timing) A1_HIP(hipEventRecord(h->ev1, s));
    h->timed = h->timing; A1_MARK(h, s);
    A1_HIP(hipMemcpyAsync(gait_counter, d_gc, N * 4 * sizeof(double), hipMemcpyDeviceToHost, s));
    A1_HIP(hipMemcpyAsync(plan_contacts_out, d_pc, N * 4, hipMemcpyDeviceToHost, s));
    if (rel_out) A1_HIP(hipMemcpyAsync(rel_out, d_rel, N * 12 * sizeof(double), hipMemcpyDeviceToHost, s));
    if (abs_out) A1_HIP(hipMemcpyAsync(abs_out, d_abs, N * 12 * sizeof(double), hipMemcpyDeviceToHost, s));
    if (world_out) A1_HIP(hipMemcpyAsync(world_out, d_world, N * 12 * sizeof(double), hipMemcpyDeviceToHost, s));
    A1_HIP(hipStreamSynchronize(s));
    return A1MPC_OK;
}

// ---- device-pointer variants of the caller-side entry points: every array is device-resident, the call is asynchronous on `hip_stream`
// (NULL = the handle's stream), so a whole control tick -- leg state, EKF, plan, swing legs, contacts / terrain, MPC, joint torques --
// chains on the GPU without a PCIe hop.  Same kernels as the host-pointer entries.
#define A1_DEV_PROLOGUE(cond)                                                                               \
    if (!h) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle");                                          \
    if (n < 0 || !(cond)) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null input/output pointer");              \
    if (n > h->max_batch) return fail(A1MPC_ERR_BATCH_TOO_LARGE, "n > max_batch given to a1mpc_create");      \
    if (n == 0) return A1MPC_OK;                                                                             \
    A1_HIP(hipSetDevice(h->device));                                                                         \
    hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->stream;                           \
    A1_ORDER(h, s);                                                                                          \
    const size_t N = n;                                                                                      \
    (void)N
#define A1_DEV_EPILOGUE()              \
    A1_HIP(hipGetLastError());         \
    if (h->timing) A1_HIP(hipEventRecord(h->ev1, s)); \
    h->timed = h->timing; A1_MARK(h, s); \
    return A1MPC_OK

a1mpc_status a1mpc_update_plan_batch_device(a1mpc_handle h, const a1mpc_gait_config* gait, int32_t n, const uint8_t* movement_mode,
                                            double* gait_counter, const double* gait_counter_speed, const double* root_lin_vel, const double* R_z,
                                            const double* R_world, const double* root_pos, const double* root_lin_vel_d, uint8_t* plan_contacts_out,
                                            double* rel_out, double* abs_out, double* world_out, void* hip_stream) {
    A1_DEV_PROLOGUE(gait && movement_mode && gait_counter && gait_counter_speed && root_lin_vel && R_z && R_world && root_pos && root_lin_vel_d && plan_contacts_out);
    PlanArgs a;
    a.g = *gait; a.n = n; a.movement_mode = movement_mode; a.gait_counter = gait_counter; a.gait_counter_speed = gait_counter_speed; a.root_lin_vel = root_lin_vel;
    a.Rz = R_z; a.Rw = R_world; a.root_pos = root_pos; a.root_lin_vel_d = root_lin_vel_d; a.plan_contacts = plan_contacts_out; a.rel = rel_out; a.abs_ = abs_out;
    a.world = world_out;
    if (h->timing) A1_HIP(hipEventRecord(h->ev0, s));
    hipLaunchKernelGGL(a1mpc_plan_kernel, dim3(static_cast<unsigned>((N * 4 + 255) / 256)), dim3(256), 0, s, a);
    A1_DEV_EPILOGUE();
}
a1mpc_status a1mpc_swing_legs_batch_device(a1mpc_handle h, int32_t n, double counter_per_swing, double dt, const double* R_z, const double* foot_pos_abs,
                                           const double* gait_counter, const double* foot_pos_target_rel, const double* kp_foot_host, const double* kd_foot_host,
                                           double* foot_pos_start, double* foot_pos_rel_last_time, double* foot_pos_target_last_time, double* foot_pos_cur_out,
                                           double* foot_forces_kin_out, void* hip_stream) {
    A1_DEV_PROLOGUE(R_z && foot_pos_abs && gait_counter && foot_pos_target_rel && kp_foot_host && kd_foot_host && foot_pos_start && foot_pos_rel_last_time &&
                    foot_pos_target_last_time && foot_pos_cur_out && foot_forces_kin_out && dt > 0);
    SwingArgs a;
    a.n = n; a.counter_per_swing = counter_per_swing; a.dt = dt;
    for (int k = 0; k < 3; ++k) { a.kp[k] = kp_foot_host[k]; a.kd[k] = kd_foot_host[k]; }
    a.Rz = R_z; a.foot_pos_abs = foot_pos_abs; a.gait_counter = gait_counter; a.target_rel = foot_pos_target_rel; a.start = foot_pos_start;
    a.rel_last = foot_pos_rel_last_time; a.target_last = foot_pos_target_last_time; a.cur_out = foot_pos_cur_out; a.kin_out = foot_forces_kin_out;
    if (h->timing) A1_HIP(hipEventRecord(h->ev0, s));
    hipLaunchKernelGGL(a1mpc_swing_kernel, dim3(static_cast<unsigned>((N * 4 + 255) / 256)), dim3(256), 0, s, a);
    A1_DEV_EPILOGUE();
}
a1mpc_status a1mpc_contact_terrain_batch_device(a1mpc_handle h, const a1mpc_contact_config* cfg, int32_t n, const double* gait_counter,
                                                const uint8_t* plan_contacts, const double* foot_force, const double* foot_pos_abs, const double* root_pos_z,
                                                double* root_euler_d_pitch, uint8_t* contacts_out, double* foot_pos_recent_contact_out, double* terrain_angle_out,
                                                void* hip_stream) {
    A1_DEV_PROLOGUE(cfg && gait_counter && plan_contacts && foot_force && foot_pos_abs && root_pos_z && root_euler_d_pitch && contacts_out &&
                    foot_pos_recent_contact_out && terrain_angle_out);
    if (a1mpc_status st = ensure_contact_state(h, s); st != A1MPC_OK) return st;
    ContactArgs a;
    a.recent_in = nullptr; a.z_stride = 1; a.pitch_stride = 1; a.pk_tick = nullptr;
    a.n = n; a.counter_per_swing = cfg->counter_per_swing; a.foot_force_low = cfg->foot_force_low; a.use_terrain_adapt = cfg->use_terrain_adapt;
    contact_state_pointers(h, a); a.gait_counter = gait_counter; a.foot_force = foot_force; a.foot_pos_abs = foot_pos_abs; a.root_pos_z = root_pos_z;
    a.plan_contacts = plan_contacts; a.pitch_d = root_euler_d_pitch; a.contacts = contacts_out; a.recent_out = foot_pos_recent_contact_out; a.terrain_out = terrain_angle_out;
    if (h->timing) A1_HIP(hipEventRecord(h->ev0, s));
    launch_contact_terrain(a, s);
    A1_DEV_EPILOGUE();
}
a1mpc_status a1mpc_leg_state_batch_device(a1mpc_handle h, int32_t n, const double* joint_pos, const double* joint_vel, const double* R_world,
                                          const double* root_pos, const double* root_lin_vel, const double* rho_fix_host, const double* rho_opt_host,
                                          double* foot_pos_rel_out, double* j_foot_blocks_out, double* foot_vel_rel_out, double* foot_pos_abs_out,
                                          double* foot_vel_abs_out, double* foot_pos_world_out, double* foot_vel_world_out, void* hip_stream) {
    A1_DEV_PROLOGUE(joint_pos && joint_vel && R_world && root_pos && root_lin_vel && rho_fix_host && rho_opt_host && foot_pos_rel_out && j_foot_blocks_out);
    LegArgs a;
    a.n = n;
    std::memcpy(a.rho_fix, rho_fix_host, sizeof a.rho_fix); std::memcpy(a.rho_opt, rho_opt_host, sizeof a.rho_opt);
    a.q = joint_pos; a.qd = joint_vel; a.R = R_world; a.pos = root_pos; a.vel = root_lin_vel; a.rel = foot_pos_rel_out; a.Jb = j_foot_blocks_out;
    a.vrel = foot_vel_rel_out; a.pabs = foot_pos_abs_out; a.vabs = foot_vel_abs_out; a.pworld = foot_pos_world_out; a.vworld = foot_vel_world_out;
    if (h->timing) A1_HIP(hipEventRecord(h->ev0, s));
    hipLaunchKernelGGL(a1mpc_leg_kernel, dim3(static_cast<unsigned>((N * 4 + 255) / 256)), dim3(256), 0, s, a);
    A1_DEV_EPILOGUE();
}
a1mpc_status a1mpc_ekf_update_batch_device(a1mpc_handle h, int32_t n, double dt, int32_t assume_flat_ground, const uint8_t* movement_mode,
                                           const double* foot_force, const double* R_world, const double* imu_acc, const double* imu_ang_vel,
                                           const double* foot_pos_rel, const double* foot_vel_rel, double* root_pos_out, double* root_lin_vel_out,
                                           uint8_t* estimated_contacts_out, void* hip_stream) {
    A1_DEV_PROLOGUE(movement_mode && foot_force && R_world && imu_acc && imu_ang_vel && foot_pos_rel && foot_vel_rel && root_pos_out && root_lin_vel_out &&
                    estimated_contacts_out && dt > 0);
    if (!h->d_ekf_state) {
        A1_HIP(hipMalloc(&h->d_ekf_state, static_cast<size_t>(h->max_batch) * kEkfState * sizeof(double)));
        A1_HIP(hipMemsetAsync(h->d_ekf_state, 0, static_cast<size_t>(h->max_batch) * kEkfState * sizeof(double), s));
    }
    EkfArgs a;
    a.n = n; a.dt = dt; a.flat = assume_flat_ground; a.state = h->d_ekf_state; a.mode = movement_mode; a.ff = foot_force; a.R = R_world; a.acc = imu_acc;
    a.w = imu_ang_vel; a.fk = foot_pos_rel; a.fv = foot_vel_rel; a.pos_out = root_pos_out; a.vel_out = root_lin_vel_out; a.ec_out = estimated_contacts_out;
    if (h->timing) A1_HIP(hipEventRecord(h->ev0, s));
    if (n > h->ekf_ready_n) {   // (robots 0 .. ekf_ready_n - 1 have been through init_state: nothing for the init kernel to do -- one launch less per tick)
        hipLaunchKernelGGL(a1mpc_ekf_init_kernel, dim3(static_cast<unsigned>((N + 255) / 256)), dim3(256), 0, s, a);
        if (hipError_t ei = hipGetLastError(); ei != hipSuccess) { h->ekf_ready_n = 0; return fail(A1MPC_ERR_HIP, std::string("a1mpc_ekf_init_kernel: ") + hipGetErrorString(ei)); }
        h->ekf_ready_n = n;   // (only once the launch has been accepted -- ADVICE r5: a failed init launch must not leave the robots marked as initialised)
    }
    launch_ekf_update(a, s);
    A1_DEV_EPILOGUE();
}
a1mpc_status a1mpc_joint_torques_batch_device(a1mpc_handle h, int32_t n, const uint8_t* active, const uint8_t* contacts, const double* j_foot_blocks,
                                              const double* grf, const double* f_kin, const double* km_foot_host, const double* torques_gravity,
                                              double* joint_torques, void* hip_stream) {
    A1_DEV_PROLOGUE(active && contacts && j_foot_blocks && grf && f_kin && km_foot_host && torques_gravity && joint_torques);
    TorqueArgs a;
    a.n = n; a.active = active; a.contacts = contacts; a.Jb = j_foot_blocks; a.grf = grf; a.fkin = f_kin; a.tg = torques_gravity; a.tau = joint_torques;
    a.km[0] = km_foot_host[0]; a.km[1] = km_foot_host[1]; a.km[2] = km_foot_host[2];
    if (h->timing) A1_HIP(hipEventRecord(h->ev0, s));
    hipLaunchKernelGGL(a1mpc_torque_kernel, dim3(static_cast<unsigned>((N * 4 + 255) / 256)), dim3(256), 0, s, a);
    A1_DEV_EPILOGUE();
}
#undef A1_DEV_PROLOGUE
#undef A1_DEV_EPILOGUE

const char* a1mpc_status_string(a1mpc_status s) {
    switch (s) {
        case A1MPC_OK: return "ok";
        case A1MPC_ERR_INVALID_ARGUMENT: return "invalid argument";
        case A1MPC_ERR_UNSUPPORTED_HORIZON: return "unsupported horizon";
        case A1MPC_ERR_NO_DEVICE: return "no HIP device";
        case A1MPC_ERR_HIP: return "HIP runtime error";
        case A1MPC_ERR_BATCH_TOO_LARGE: return "batch larger than max_batch";
    }
    return "unknown";
}
const char* a1mpc_last_error(void) { return g_last_error.c_str(); }

// which sources this library is the compilation of (build.py: sha256 over csrc/ + include/a1mpc.h) -- a shipped .so is matched against the sources beside it.
// The string lives in a unit of its own (a1mpc_build_id.cpp, compiled in a second): a change of any source re-stamps the library without recompiling this file.
const char* a1mpc_build_info(void) { return a1mpc_build_id_; }

void a1mpc_destroy(a1mpc_handle h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    void* ptrs[] = {h->d_tab, h->d_tab1, h->d_x0, h->d_xref, h->d_R, h->d_foot, h->d_aux, h->d_Rz, h->d_contact, h->d_grf,
                    h->d_u, h->d_iters, h->d_status, h->d_nfact, h->d_wx, h->d_wy, h->d_rho, h->d_prep, h->d_counter, h->d_in, h->d_out, h->d_order, h->d_cost, h->d_ct_state, h->d_ekf_state, h->d_aux_in, h->d_aux_out, h->d_aux_u8, h->d_foot_steps, h->d_contact_steps, h->d_prep_gen, h->d_carry, h->d_clk, h->d_tickrec};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (h->h_pin) (void)hipHostFree(h->h_pin);
    if (h->h_pin_gen) (void)hipHostFree(h->h_pin_gen);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->ev_order) (void)hipEventDestroy(h->ev_order);
    if (h->ev_mark) (void)hipEventDestroy(h->ev_mark);
    if (h->ev_mid) (void)hipEventDestroy(h->ev_mid);
    if (h->ev_tick0) (void)hipEventDestroy(h->ev_tick0);
    if (h->ev_tick1) (void)hipEventDestroy(h->ev_tick1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

a1mpc_status a1mpc_create(const a1mpc_config* cfg, int32_t max_batch, int32_t device, a1mpc_handle* out) {
    if (!cfg || !out || max_batch <= 0 || device < 0) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null config/out or bad batch/device");
    *out = nullptr;
    if (lds_bytes_of(cfg->horizon) == 0) return fail(A1MPC_ERR_UNSUPPORTED_HORIZON, "horizon must be " A1MPC_HORIZON_LIST);
    if (const char* why = invalid_config(*cfg)) return fail(A1MPC_ERR_INVALID_ARGUMENT, why);   // (before any device query: a bad configuration is reported as such on every machine)
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(A1MPC_ERR_NO_DEVICE, "hipGetDeviceCount found no device");
    if (device >= ndev) return fail(A1MPC_ERR_NO_DEVICE, "device ordinal out of range");
    A1_HIP(hipSetDevice(device));
    a1mpc_handle h = new (std::nothrow) a1mpc_handle_s();
    if (!h) return fail(A1MPC_ERR_HIP, "out of host memory");
    h->cfg = *cfg;
    to_device_params(*cfg, &h->dp);
    h->device = device;
    h->max_batch = max_batch;
    const int H = cfg->horizon;
    const size_t n = static_cast<size_t>(max_batch);
#define A1_TRY(call)                                  \
    do {                                              \
        hipError_t e_ = (call);                       \
        if (e_ != hipSuccess) {                       \
            a1mpc_destroy(h);                         \
            return fail(A1MPC_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
        }                                             \
    } while (0)
    A1_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    A1_TRY(hipEventCreate(&h->ev0));
    A1_TRY(hipEventCreate(&h->ev1));
    A1_TRY(hipEventCreateWithFlags(&h->ev_order, hipEventDisableTiming));
    A1_TRY(hipEventCreateWithFlags(&h->ev_mark, hipEventDisableTiming));
    A1_TRY(hipEventCreate(&h->ev_mid));
    A1_TRY(hipEventCreate(&h->ev_tick0));
    A1_TRY(hipEventCreate(&h->ev_tick1));
    std::vector<double> tab(2 * H * H), tab1(2);
    fill_gamma_beta_table(H, tab.data());
    fill_gamma_beta_table(1, tab1.data());
    A1_TRY(hipMalloc(&h->d_tab, tab.size() * sizeof(double)));
    A1_TRY(hipMalloc(&h->d_tab1, tab1.size() * sizeof(double)));
    A1_TRY(hipMemcpy(h->d_tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
    A1_TRY(hipMemcpy(h->d_tab1, tab1.data(), tab1.size() * sizeof(double), hipMemcpyHostToDevice));
    A1_TRY(hipMalloc(&h->d_x0, n * 13 * sizeof(double)));
    A1_TRY(hipMalloc(&h->d_xref, n * 13 * H * sizeof(double)));
    A1_TRY(hipMalloc(&h->d_R, n * 9 * sizeof(double)));
    A1_TRY(hipMalloc(&h->d_Rz, n * 9 * sizeof(double)));
    A1_TRY(hipMalloc(&h->d_foot, n * 12 * sizeof(double)));
    A1_TRY(hipMalloc(&h->d_aux, n * 6 * sizeof(double)));
    A1_TRY(hipMalloc(&h->d_contact, n * 4));
    A1_TRY(hipMalloc(&h->d_grf, n * 12 * sizeof(double)));
    A1_TRY(hipMalloc(&h->d_u, n * 12 * H * sizeof(double)));
    A1_TRY(hipMalloc(&h->d_iters, n * sizeof(int32_t)));
    A1_TRY(hipMalloc(&h->d_status, n * sizeof(int32_t)));
    A1_TRY(hipMalloc(&h->d_nfact, n * sizeof(int32_t)));
    A1_TRY(hipMalloc(&h->d_wx, n * 12 * H * sizeof(double)));
    A1_TRY(hipMalloc(&h->d_wy, n * 20 * H * sizeof(double)));
    A1_TRY(hipMalloc(&h->d_rho, n * sizeof(double)));
    A1_TRY(hipMemset(h->d_wx, 0, n * 12 * H * sizeof(double)));
    A1_TRY(hipMemset(h->d_wy, 0, n * 20 * H * sizeof(double)));
    A1_TRY(hipMemset(h->d_rho, 0, n * sizeof(double)));
    if (pipeline_mode() != 2) A1_TRY(hipMalloc(&h->d_prep, n * prep_stride(H) * sizeof(double)));
    A1_TRY(hipMalloc(&h->d_counter, sizeof(int)));
    A1_TRY(hipMalloc(&h->d_order, n * sizeof(int32_t)));
    A1_TRY(hipMalloc(&h->d_cost, n * sizeof(int32_t)));
    if (const char* e = std::getenv("A1MPC_SCHEDULE")) h->schedule = std::strcmp(e, "index") != 0;
    // pinned mirror: inputs (x0, xref, R, Rz, foot, aux, contact) then outputs (grf, u, iters, status)
    h->h_pin_in_bytes = n * ((13 + 13 * H + 9 + 12) * sizeof(double) + 8);
    const size_t out_max = n * ((12 + 12 * H) * sizeof(double) + 2 * sizeof(int32_t));
    h->h_pin_bytes = h->h_pin_in_bytes + out_max;
    A1_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_in), h->h_pin_in_bytes));
    A1_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_out), out_max));
    A1_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->h_pin), h->h_pin_bytes, hipHostMallocDefault));
    { void* dp = nullptr; if (hipHostGetDevicePointer(&dp, h->h_pin, 0) == hipSuccess) h->d_pin = static_cast<char*>(dp); else (void)hipGetLastError(); }
    // One-time, per process: the first launches / copies through a fresh HIP runtime cost milliseconds (code-object load, pool set-up);
    // a 400 Hz loop should not pay that on its first tick (measured 5 ms -> 0.4 ms), so the tick's operation mix is exercised here.
    static bool runtime_warmed_dev[64] = {};   // per device: a sharded handle (a1mpc_sharded_*) creates one engine handle on every GPU of the node
    std::unique_lock<std::mutex> warm_lock(g_cache_mu);   // (handles may be created from several threads)
    bool& runtime_warmed = runtime_warmed_dev[device < 64 ? device : 63];
    if (!runtime_warmed) {
        for (int i = 0; i < 256; ++i) {  // the operation mix of a tick: copies both ways from pinned memory, memset, launches, events
            A1_TRY(hipMemcpyAsync(h->d_x0, h->h_pin, 8, hipMemcpyHostToDevice, h->stream));
            A1_TRY(hipMemsetAsync(h->d_counter, 0, sizeof(int), h->stream));
            A1_TRY(hipEventRecord(h->ev0, h->stream));
            hipLaunchKernelGGL(a1mpc_noop_kernel, dim3(1), dim3(64), 0, h->stream);
            hipLaunchKernelGGL(a1mpc_noop_kernel, dim3(1), dim3(64), 0, h->stream);
            A1_TRY(hipEventRecord(h->ev1, h->stream));
            A1_TRY(hipMemcpyAsync(h->h_pin + 64, h->d_x0, 8, hipMemcpyDeviceToHost, h->stream));
            if ((i & 63) == 63) A1_TRY(hipStreamSynchronize(h->stream));
        }
        A1_TRY(hipStreamSynchronize(h->stream));
        runtime_warmed = true;
    }
    warm_lock.unlock();
#undef A1_TRY
    *out = h;
    return A1MPC_OK;
}

a1mpc_status a1mpc_update_config(a1mpc_handle h, const a1mpc_config* cfg) {
    if (!h || !cfg) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle/config");
    if (cfg->horizon != h->cfg.horizon) return fail(A1MPC_ERR_UNSUPPORTED_HORIZON, "the horizon of a handle is fixed at a1mpc_create");
    if (const char* why = invalid_config(*cfg)) return fail(A1MPC_ERR_INVALID_ARGUMENT, why);   // (the handle keeps the configuration it had)
    // every constant travels to the kernels by value with each launch: nothing on the device has to change, launches already
    // queued keep the values they were issued with, the carried warm start stays
    if (cfg->warm_start != h->cfg.warm_start && h->d_carry) {
        // leaving or entering the update path: ticks solved in another mode do not refresh its carry (previous scalings / gradient / z), so what is there is stale
        A1_HIP(hipSetDevice(h->device));
        A1_ORDER(h, h->stream);
        A1_HIP(hipMemsetAsync(h->d_carry, 0, static_cast<size_t>(h->max_batch) * carry_stride(h->cfg.horizon) * sizeof(double), h->stream));
        A1_MARK(h, h->stream);
    }
    h->cfg = *cfg;
    to_device_params(*cfg, &h->dp);
    return A1MPC_OK;
}

a1mpc_status a1mpc_warm_start(a1mpc_handle h, int32_t n, const double* x, const double* y, const double* rho) {
    if (!h) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle");
    if (n < 0 || n > h->max_batch) return fail(A1MPC_ERR_BATCH_TOO_LARGE, "n > max_batch given to a1mpc_create");
    if (n == 0 || (!x && !y && !rho)) return A1MPC_OK;
    A1_HIP(hipSetDevice(h->device));
    A1_ORDER(h, h->stream);
    const size_t N = n, H = h->cfg.horizon;
    if (x) A1_HIP(hipMemcpyAsync(h->d_wx, x, N * 12 * H * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (y) A1_HIP(hipMemcpyAsync(h->d_wy, y, N * 20 * H * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (rho) A1_HIP(hipMemcpyAsync(h->d_rho, rho, N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    // warm_start = 2: the injected state is re-expressed on the update path's workspace the way osqp_warm_start_x / _y do it on the reference's persistent solver:
    // x and y replace the carried (unscaled) iterates above, z becomes A x (osqp_warm_start_x: z = A x), the previous tick's scalings, gradient and pattern stay --
    // the next tick follows the update path from there.  (Until round 4 the carry was cleared: that tick was a fresh set-up.)
    std::vector<double> zbuf;
    if (h->d_carry && x) {
        const size_t cs = carry_stride(h->cfg.horizon);
        const double mu = h->cfg.mu;
        zbuf.resize(N * 24 * H);
        for (size_t b = 0; b < N; ++b)
            for (size_t t = 0; t < H; ++t)
                for (int leg = 0; leg < 4; ++leg) {
                    const double* f = x + (b * H + t) * 12 + 3 * leg;
                    double* z0 = zbuf.data() + b * 24 * H + t * 12 + 3 * leg;   // [Z0: 12H | Z1: 12H] of Carry<H>, per lane (leg, comp)
                    double* z1 = z0 + 12 * H;
                    z0[0] = std::fma(mu, f[2], f[0]); z0[1] = std::fma(mu, f[2], f[1]); z0[2] = f[2];
                    z1[0] = std::fma(-mu, f[2], f[0]); z1[1] = std::fma(-mu, f[2], f[1]); z1[2] = 0.0;
                }
        A1_HIP(hipMemcpy2DAsync(h->d_carry + (1 + 48 * H), cs * sizeof(double), zbuf.data(), 24 * H * sizeof(double), 24 * H * sizeof(double), N, hipMemcpyHostToDevice, h->stream));
    }
    A1_HIP(hipStreamSynchronize(h->stream));
    return A1MPC_OK;
}

a1mpc_status a1mpc_get_warm_start(a1mpc_handle h, int32_t n, double* x_out, double* y_out, double* rho_out) {
    if (!h) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle");
    if (n < 0 || n > h->max_batch) return fail(A1MPC_ERR_BATCH_TOO_LARGE, "n > max_batch given to a1mpc_create");
    if (n == 0) return A1MPC_OK;
    A1_HIP(hipSetDevice(h->device));
    A1_ORDER(h, h->stream);
    const size_t N = n, H = h->cfg.horizon;
    if (x_out) A1_HIP(hipMemcpyAsync(x_out, h->d_wx, N * 12 * H * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (y_out) A1_HIP(hipMemcpyAsync(y_out, h->d_wy, N * 20 * H * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (rho_out) A1_HIP(hipMemcpyAsync(rho_out, h->d_rho, N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    A1_HIP(hipStreamSynchronize(h->stream));
    return A1MPC_OK;
}

a1mpc_status a1mpc_set_timing(a1mpc_handle h, int32_t on) {
    if (!h) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle");
    h->timing = on != 0;
    if (!h->timing) { h->timed = false; h->tick_timed = false; h->staged = false; }
    return A1MPC_OK;
}
a1mpc_status a1mpc_set_profiling(a1mpc_handle h, int32_t on) {
    if (!h) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle");
    h->profiling = on != 0;
    return A1MPC_OK;
}
a1mpc_status a1mpc_last_stage_cycles(a1mpc_handle h, double* cycles3_out, int32_t* qps_out) {
    if (!h || !cycles3_out) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle / output");
    cycles3_out[0] = cycles3_out[1] = cycles3_out[2] = 0.0;
    if (qps_out) *qps_out = h->clk_n;
    if (h->clk_n == 0 || !h->d_clk) return A1MPC_OK;
    A1_HIP(hipSetDevice(h->device));
    A1_ORDER(h, h->stream);
    std::vector<long long> c(static_cast<size_t>(h->clk_n) * kTickStages);
    A1_HIP(hipMemcpyAsync(c.data(), h->d_clk, c.size() * sizeof(long long), hipMemcpyDeviceToHost, h->stream));
    A1_HIP(hipStreamSynchronize(h->stream));
    for (int b = 0; b < h->clk_n; ++b) for (int k = 0; k < 3; ++k) cycles3_out[k] += static_cast<double>(c[static_cast<size_t>(b) * kTickStages + kClkFactor + k]);
    return A1MPC_OK;
}
a1mpc_status a1mpc_last_tick_stage_cycles(a1mpc_handle h, double* cycles8_out, int32_t* qps_out) {
    if (!h || !cycles8_out) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle / output");
    for (int k = 0; k < kTickStages; ++k) cycles8_out[k] = 0.0;
    const int32_t nq = h->clk_tick ? h->clk_n : 0;
    if (qps_out) *qps_out = nq;
    if (nq == 0 || !h->d_clk) return A1MPC_OK;
    A1_HIP(hipSetDevice(h->device));
    A1_ORDER(h, h->stream);
    std::vector<long long> c(static_cast<size_t>(nq) * kTickStages);
    A1_HIP(hipMemcpyAsync(c.data(), h->d_clk, c.size() * sizeof(long long), hipMemcpyDeviceToHost, h->stream));
    A1_HIP(hipStreamSynchronize(h->stream));
    for (int b = 0; b < nq; ++b) for (int k = 0; k < kTickStages; ++k) cycles8_out[k] += static_cast<double>(c[static_cast<size_t>(b) * kTickStages + k]);
    return A1MPC_OK;
}

a1mpc_status a1mpc_last_warm_start_mode(a1mpc_handle h, int32_t* mode_out) {
    if (!h || !mode_out) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle / output");
    *mode_out = h->last_ws_mode;
    return A1MPC_OK;
}

// one Carry<H> record -> 20H rows: lane (leg, comp) holds rows r0 (Z0) and, for fx / fy, r1 (Z1) of its leg's block
static void unpack_carry_z(int H, const double* rec, double* z) {
    const int Z0 = 1 + 48 * H, Z1 = 1 + 60 * H;
    static_assert(Carry<10>::Z0 == 1 + 48 * 10 && Carry<10>::Z1 == 1 + 60 * 10 && Carry<20>::Z0 == 1 + 48 * 20 && Carry<20>::Z1 == 1 + 60 * 20, "Carry<H> layout");
    for (int t = 0; t < H; ++t)
        for (int ci = 0; ci < 12; ++ci) {
            const int leg = ci / 3, comp = ci % 3;
            const int r0 = comp == 0 ? 0 : (comp == 1 ? 2 : 4), r1 = comp == 0 ? 1 : 3;
            z[t * 20 + 5 * leg + r0] = rec[Z0 + t * 12 + ci];
            if (comp < 2) z[t * 20 + 5 * leg + r1] = rec[Z1 + t * 12 + ci];
        }
}
a1mpc_status a1mpc_get_workspace_z(a1mpc_handle h, int32_t n, double* z_out) {
    if (!h || !z_out) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle / output");
    if (n < 0 || n > h->max_batch) return fail(A1MPC_ERR_BATCH_TOO_LARGE, "n > max_batch given to a1mpc_create");
    const size_t cs = carry_stride(h->cfg.horizon);
    if (h->cfg.warm_start != 2 || cs == 0 || !h->d_carry) return fail(A1MPC_ERR_INVALID_ARGUMENT, "no update-path carry (warm_start = 2, horizon > 1, after the first tick)");
    if (n == 0) return A1MPC_OK;
    A1_HIP(hipSetDevice(h->device));
    A1_ORDER(h, h->stream);
    std::vector<double> rec(static_cast<size_t>(n) * cs);
    A1_HIP(hipMemcpyAsync(rec.data(), h->d_carry, rec.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    A1_HIP(hipStreamSynchronize(h->stream));
    const size_t H = h->cfg.horizon;
    for (int b = 0; b < n; ++b) {
        const double* r = rec.data() + static_cast<size_t>(b) * cs;
        double* z = z_out + static_cast<size_t>(b) * 20 * H;
        unpack_carry_z(h->cfg.horizon, r, z);
    }
    return A1MPC_OK;
}

a1mpc_status a1mpc_get_workspace_scaling(a1mpc_handle h, int32_t n, double* D_out, double* E_out, double* c_out) {
    if (!h) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle");
    if (n < 0 || n > h->max_batch) return fail(A1MPC_ERR_BATCH_TOO_LARGE, "n > max_batch given to a1mpc_create");
    const size_t cs = carry_stride(h->cfg.horizon);
    if (h->cfg.warm_start != 2 || cs == 0 || !h->d_carry) return fail(A1MPC_ERR_INVALID_ARGUMENT, "no update-path carry (warm_start = 2, horizon > 1, after the first tick)");
    if (n == 0) return A1MPC_OK;
    A1_HIP(hipSetDevice(h->device));
    A1_ORDER(h, h->stream);
    std::vector<double> rec(static_cast<size_t>(n) * cs);
    A1_HIP(hipMemcpyAsync(rec.data(), h->d_carry, rec.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    A1_HIP(hipStreamSynchronize(h->stream));
    const int H = h->cfg.horizon;
    const int oD = 1, oE0 = 1 + 12 * H, oE1 = 1 + 24 * H;   // Carry<H>::D, E0, E1
    static_assert(Carry<10>::D == 1 && Carry<10>::E0 == 1 + 120 && Carry<10>::E1 == 1 + 240 && Carry<16>::E1 == 1 + 24 * 16, "Carry<H> layout");
    for (int b = 0; b < n; ++b) {
        const double* r = rec.data() + static_cast<size_t>(b) * cs;
        if (c_out) c_out[b] = r[0];
        for (int t = 0; t < H; ++t)
            for (int ci = 0; ci < 12; ++ci) {
                const int leg = ci / 3, comp = ci % 3;
                const int r0 = comp == 0 ? 0 : (comp == 1 ? 2 : 4), r1 = comp == 0 ? 1 : 3;
                if (D_out) D_out[(static_cast<size_t>(b) * H + t) * 12 + ci] = r[oD + t * 12 + ci];
                if (E_out) {
                    double* e = E_out + (static_cast<size_t>(b) * H + t) * 20 + 5 * leg;
                    e[r0] = r[oE0 + t * 12 + ci];
                    if (comp < 2) e[r1] = r[oE1 + t * 12 + ci];
                }
            }
    }
    return A1MPC_OK;
}

a1mpc_status a1mpc_reset_warm_start(a1mpc_handle h) {
    if (!h) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle");
    A1_HIP(hipSetDevice(h->device));
    A1_ORDER(h, h->stream);  // the memsets below must not overtake a launch still running on a caller's stream
    const size_t n = h->max_batch, H = h->cfg.horizon;
    A1_HIP(hipMemsetAsync(h->d_wx, 0, n * 12 * H * sizeof(double), h->stream));
    A1_HIP(hipMemsetAsync(h->d_wy, 0, n * 20 * H * sizeof(double), h->stream));
    A1_HIP(hipMemsetAsync(h->d_rho, 0, n * sizeof(double), h->stream));
    if (h->d_carry) A1_HIP(hipMemsetAsync(h->d_carry, 0, n * carry_stride(h->cfg.horizon) * sizeof(double), h->stream));
    A1_HIP(hipStreamSynchronize(h->stream));
    h->hint_n = 0;
    return A1MPC_OK;
}

a1mpc_status a1mpc_set_schedule(a1mpc_handle h, int32_t history) {
    if (!h) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle");
    h->schedule = history != 0;
    h->hint_n = 0;
    return A1MPC_OK;
}

// N3 in the MPC kernels' output stage (a1mpc_control_tick_device): the device arrays compute_joint_torques reads and writes; `fused` reports whether the launch that
// solve_device_impl chose carries the stage (the fused / latency kernels do; the split pipeline's persistent rows do not -- the caller then launches the torque kernel)
struct TorqueFuse {
    const uint8_t* active;
    const double *J, *fkin, *tg, *km;
    double* tau;
    bool fused;
};
// A1MPC_WARM_ORDER=1: launch the fused kernel of a warm-started tick in the order of the previous tick's per-QP cost (longest first).  OFF by default: measured and
// lost (profiles/r05_warm_tick_order.txt: 4096 x h10 ticks 0.335 -> 0.348 ms, update path 0.362 -> 0.373) -- a warm tick has no tail worth ordering for (3 of 4096
// QPs need a second 25-iteration segment and the two rounds of the resident rows absorb them), and the order kernel in front of the tick costs its ~12 us.
// Round 6 experiment switch (profiles/r06_control_tick_timeline.md): with the handle's timing events OFF, record a no-timing marker event where a timing event would have been
// -- bit 0: in front of the MPC launch (ev0), bit 1: at the start of a control tick (ev_tick0), bit 2: behind the MPC launch (ev1), bit 3: at the end of a tick (ev_tick1)
static int tick_markers() {
    static const int m = [] { const char* e = getenv("A1MPC_TICK_MARKERS"); return e ? atoi(e) : 0; }();
    return m;
}
static bool warm_order_enabled() {
    static const bool on = [] { const char* e = getenv("A1MPC_WARM_ORDER"); return e && !strcmp(e, "1"); }();
    return on;
}
static a1mpc_status solve_device_impl(a1mpc_handle h, int32_t n, const double* d_tick, const double* d_x0, const double* d_x_ref,
                                      const double* d_R_world, const double* d_foot_abs, const uint8_t* d_contact,
                                      double* d_grf_body_out, double* d_u_full_out, int32_t* d_iters_out, int32_t* d_status_out,
                                      void* hip_stream, int32_t foot_stride = 0, int32_t contact_stride = 0, const double* d_yaw_A = nullptr, TorqueFuse* tq = nullptr) {
    if (tq) tq->fused = false;
    if (!h) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle");
    if (n < 0 || (!d_tick && (!d_x0 || !d_x_ref)) || !d_R_world || !d_foot_abs || !d_contact || !d_grf_body_out)
        return fail(A1MPC_ERR_INVALID_ARGUMENT, "null input/output pointer");
    if (n > h->max_batch) return fail(A1MPC_ERR_BATCH_TOO_LARGE, "n > max_batch given to a1mpc_create");
    if (n == 0) return A1MPC_OK;
    A1_HIP(hipSetDevice(h->device));
    hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->stream;
    A1_ORDER(h, s);
    h->clk_n = 0; h->clk_tick = false;   // (set again below when this solve runs a profiling instantiation)
    KernelArgs a;
    std::memset(&a, 0, sizeof a);
    a.P = h->dp; a.tab = h->d_tab; a.n = n;
    a.tick = d_tick; a.x0 = d_x0; a.xref = d_x_ref; a.R = d_R_world; a.foot = d_foot_abs; a.contact = d_contact;
    a.grf = d_grf_body_out; a.u_full = d_u_full_out; a.iters = d_iters_out; a.status = d_status_out; a.nfact = h->d_nfact;
    if (h->cfg.warm_start) { a.warm_x = h->d_wx; a.warm_y = h->d_wy; a.rho = h->d_rho; }
    if (h->cfg.warm_start == 2 && carry_stride(h->cfg.horizon) != 0) {   // the reference's update path on ticks >= 2 (a1mpc.h): what its persistent OSQP workspace carries
        if (!h->d_carry) {
            const size_t bytes = static_cast<size_t>(h->max_batch) * carry_stride(h->cfg.horizon) * sizeof(double);
            A1_HIP(hipMalloc(&h->d_carry, bytes));
            A1_HIP(hipMemsetAsync(h->d_carry, 0, bytes, s));
        }
        a.carry = h->d_carry;
    }
    h->last_ws_mode = (h->cfg.warm_start == 2 && a.carry == nullptr) ? 1 : h->cfg.warm_start;   // (the general path's split pipeline revises this below)
    a.contact_stride = contact_stride;  // a per-step contact schedule alone (feet step-invariant) stays on the fast path: contacts only change bounds and equality rows
    if (foot_stride != 0 || d_yaw_A != nullptr) {  // general path: per-step B_d (and / or its own A_c yaw), with or without a contact schedule
        if (d_tick) return fail(A1MPC_ERR_INVALID_ARGUMENT, "per-step feet / contacts are not combined with tick records");
        a.foot_stride = foot_stride; a.contact_stride = contact_stride; a.yaw_A = d_yaw_A;
        // a batch beyond the resident rows of the general path's ADMM kernel runs its split pipeline (set-up kernel + persistent rows on a queue, like the
        // fast path); its hand-off records (B~w_t of every step included) live in a buffer of their own, allocated on first use
        int rows_gen = 0;
        if (a1mpc_status st0 = resident_rows_gen(h->cfg.horizon, &rows_gen); st0 != A1MPC_OK) return st0;
        bool split_gen = pipeline_mode() != 2 && rows_gen > 0 && (pipeline_mode() == 1 || n > rows_gen) && h->d_counter != nullptr;
        // warm_start = 2: the FUSED general kernels follow the update path like the fast path's (round 5: batches within the resident rows).  Round 6: every batch size --
        // an update-path batch beyond the resident rows runs the fused kernel as well, in as many rounds as it takes (the hardware dispatches the grid's workgroups as
        // LDS frees up).  Ticks on the update path are warm-started and all alike (25 iterations once the loop is closed): there is nothing for a queue to balance, which
        // is why the fast path's warm ticks of a known batch take the fused kernel too (solve_device_impl below); only the very first tick of a fleet (cold, no
        // history) pays a tail for it.  The general path's split pipeline itself has no update-path hand-off and keeps serving modes 0 / 1.
        if (a.carry != nullptr && split_gen && pipeline_mode() != 1) split_gen = false;
        if (a.carry != nullptr && split_gen) {   // (A1MPC_PIPELINE=split forced: warm_start = 1 semantics, visibly -- the carry is marked "no previous tick" so that a later
                                                 // update-path tick does not pair stale scalings with fresh iterates)
            A1_HIP(hipMemset2DAsync(h->d_carry, carry_stride(h->cfg.horizon) * sizeof(double), 0, sizeof(double), static_cast<size_t>(n), s));
            a.carry = nullptr;
            h->last_ws_mode = 1;
        }
        if (split_gen && !h->d_prep_gen) {
            A1_HIP(hipMalloc(&h->d_prep_gen, static_cast<size_t>(h->max_batch) * prep_stride_gen(h->cfg.horizon) * sizeof(double)));
        }
        const bool hints_gen = h->schedule && split_gen && n >= kScheduleMinBatch && h->d_order != nullptr && h->d_cost != nullptr;
        a.order = hints_gen ? h->d_order : nullptr;
        a.cost = hints_gen ? h->d_cost : nullptr;
        a.predict = (hints_gen && h->hint_n != -n) ? 1 : 0;  // (the history of a general-path batch is remembered as -n: never mixed up with a fast-path batch of the same size)
        h->staged = split_gen && h->timing;
        if (h->timing) A1_HIP(hipEventRecord(h->ev0, s));
        if (split_gen) {
            static const bool pipe_hint = [] { const char* e = getenv("A1MPC_GEN_PIPE_ONE_WAVE"); return !(e && !strcmp(e, "0")); }();
            g_gen_prefer_one_wave = pipe_hint && h->pipeline_depth == 2;
            const a1mpc_status stg = launch_gen_split(h->cfg.horizon, a, h->d_prep_gen, h->d_counter, s, h->timing ? h->ev_mid : nullptr);
            g_gen_prefer_one_wave = false;
            if (stg != A1MPC_OK) return stg;
            if (hints_gen) h->hint_n = -n;   // the cost buffer now holds this batch's costs: the next general-path solve of this size is ordered by them
        } else {
            if (a1mpc_status stg = launch_gen(h->cfg.horizon, a, s); stg != A1MPC_OK) return stg;
        }
        if (h->timing) A1_HIP(hipEventRecord(h->ev1, s));
        h->timed = h->timing;
        A1_MARK(h, s);
        return A1MPC_OK;
    }
    // Straggler-aware queue order: batches beyond the resident rows are issued longest-first by the cost each QP had in the previous solve of
    // this handle (the same robots tick after tick); the first solve of a batch size is ordered by the set-up kernel's cost guess
    // (RowSolver::predict_cost).  Scheduling only.
    bool split = false;
    if (a1mpc_status st0 = use_split_pipeline(h->cfg.horizon, n, h->d_prep != nullptr, &split); st0 != A1MPC_OK) return st0;
    // Warm-started ticks of the same robots (the closed-loop regime: second and later ticks of a batch size) take ~25 iterations each, all alike: nothing is left
    // for the queue to balance, and the fused kernel -- set-up and solve in one launch, no hand-off through memory, no kernel boundary for the rows to idle at --
    // wins up to a few rounds of the resident rows (4096 x h10: 0.397 -> 0.350 ms per tick, 8192: 0.665 -> 0.652; profiles/r04_warm_ticks_fused_vs_split.txt).
    // Same results bit for bit (the pipelines are tested against each other).  A1MPC_WARM_FUSED=0 keeps the split pipeline (A/B runs).
    bool warm_fused = false;
    if (split && h->cfg.warm_start != 0 && h->hint_n == n && pipeline_mode() == 0 && warm_fused_enabled() && n <= warm_fused_max(h->cfg.horizon)) { split = false; warm_fused = true; }
    const bool hints = h->schedule && split && n >= kScheduleMinBatch;
    a.order = hints ? h->d_order : nullptr;
    a.cost = hints ? h->d_cost : nullptr;
    a.predict = (hints && h->hint_n != n) ? 1 : 0;  // first solve of this batch size: order by the set-up kernel's guess instead
    if (h->timing) A1_HIP(hipEventRecord(h->ev0, s)); else if (tick_markers() & 1) A1_HIP(hipEventRecord(h->ev_mark, s));
    // Round 5 trial (opt-in, see warm_order_enabled): the fused kernel of a warm-started tick launches its workgroups in the order of the previous tick's per-QP
    // cost, longest first (the cost buffer holds it: hint_n == n; the fused kernel records this tick's).  Scheduling only: every result is bit-identical in any order.
    if (warm_fused && h->schedule && n >= kScheduleMinBatch && n > coop_max_batch() && warm_order_enabled()) {
        RoctxRange range("a1mpc order");
        launch_order_kernel(static_cast<int>(n), static_cast<const int32_t*>(h->d_cost), h->d_order, s);
        A1_HIP(hipGetLastError());
        a.order = h->d_order; a.cost = h->d_cost;
    }
    if (tq != nullptr && !split && h->cfg.horizon > 1) {   // the fused / latency kernels write the joint torques in their output stage
        a.tq_active = tq->active; a.tq_J = tq->J; a.tq_fkin = tq->fkin; a.tq_tg = tq->tg; a.tq_km = tq->km; a.tq_tau = tq->tau;
        tq->fused = true;
    }
    h->staged = split && h->timing;
    h->clk_n = 0; h->clk_tick = false;
    if (h->profiling && h->cfg.horizon > 1 && a.contact_stride == 0) {
        if (!h->d_clk) A1_HIP(hipMalloc(&h->d_clk, static_cast<size_t>(h->max_batch) * kTickStages * sizeof(long long)));
        A1_HIP(hipMemsetAsync(h->d_clk, 0, static_cast<size_t>(n) * kTickStages * sizeof(long long), s));
        a.clk = h->d_clk;
    }
    g_clk_ran = false;
    a1mpc_status st = launch_mpc(h->cfg.horizon, a, h->d_prep, h->d_counter, s, split, h->timing ? h->ev_mid : nullptr);
    if (st != A1MPC_OK) return st;
    if (a.clk != nullptr && g_clk_ran) { h->clk_n = n; h->clk_tick = !split; }   // (a kernel without stamps ran: the record stays "not profiled")
    if (hints) h->hint_n = n;   // the cost buffer now holds this batch's costs: the next solve of this size is ordered by them (sorted in front of its ADMM kernel)
    if (h->timing) A1_HIP(hipEventRecord(h->ev1, s)); else if (tick_markers() & 4) A1_HIP(hipEventRecord(h->ev_mark, s));
    h->timed = h->timing;
    A1_MARK(h, s);
    return A1MPC_OK;
}

a1mpc_status a1mpc_solve_batch_device(a1mpc_handle h, int32_t n, const double* d_x0, const double* d_x_ref, const double* d_R_world,
                                      const double* d_foot_abs, const uint8_t* d_contact, double* d_grf_body_out,
                                      double* d_u_full_out, int32_t* d_iters_out, int32_t* d_status_out, void* hip_stream) {
    if (!d_x0 || !d_x_ref) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null input/output pointer");
    return solve_device_impl(h, n, nullptr, d_x0, d_x_ref, d_R_world, d_foot_abs, d_contact, d_grf_body_out, d_u_full_out, d_iters_out,
                             d_status_out, hip_stream);
}
a1mpc_status a1mpc_solve_batch_ticks_device(a1mpc_handle h, int32_t n, const double* d_tick, const double* d_R_world,
                                            const double* d_foot_abs, const uint8_t* d_contact, double* d_grf_body_out,
                                            double* d_u_full_out, int32_t* d_iters_out, int32_t* d_status_out, void* hip_stream) {
    if (!d_tick) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null input/output pointer");
    return solve_device_impl(h, n, d_tick, nullptr, nullptr, d_R_world, d_foot_abs, d_contact, d_grf_body_out, d_u_full_out, d_iters_out,
                             d_status_out, hip_stream);
}

void a1mpc_default_tick_params(a1mpc_tick_params* p) {
    if (!p) return;
    std::memset(p, 0, sizeof *p);
    a1mpc_default_gait_config(&p->gait);
    a1mpc_default_contact_config(&p->contact);
    p->control_dt = 0.0025; p->assume_flat_ground = 1;
    const double kp[3] = {300.0, 400.0, 400.0}, kd[3] = {8.0, 8.0, 8.0}, km[3] = {0.1, 0.1, 0.04};   // kp_foot / kd_foot / km_foot of the Gazebo parameter set (config/gazebo_a1_mpc.yaml)
    std::memcpy(p->kp_foot, kp, sizeof kp); std::memcpy(p->kd_foot, kd, sizeof kd); std::memcpy(p->km_foot, km, sizeof km);
    const double ox = 0.1805, oy = 0.047, d = 0.0838, lt = 0.21, lc = 0.21;   // A1 leg geometry, S/GazeboA1ROS.cpp:20-50 (leg_offset_x/y, motor_offset, upper / lower leg length)
    const double sx[4] = {1, 1, -1, -1}, sy[4] = {1, -1, 1, -1};
    for (int i = 0; i < 4; ++i) { double* f = p->rho_fix + 5 * i; f[0] = sx[i] * ox; f[1] = sy[i] * oy; f[2] = sy[i] * d; f[3] = lt; f[4] = lc; }
}

// One control tick of n robots in ONE call, device-resident: the reference's chain  joint-state callback (leg FK / Jacobians, S/GazeboA1ROS.cpp:264-279)
// -> A1BasicEKF::update_estimation (S/A1BasicEKF.cpp:70-163) -> update_plan (S/A1RobotControl.cpp:148-202) -> generate_swing_legs_ctrl (:204-287, with the contact logic and
// the terrain fit of compute_grf :335-376) -> compute_grf (:446-562) -> compute_joint_torques (:289-319), as driven by S/MainGazebo.cpp:47-119 -- six kernels, the tick-record
// pack and the MPC launch back to back on one stream, no host round trip, and N3 inside the MPC kernel's output stage whenever the tick runs the fused / latency kernel
// (every warm-started tick of a known batch).  Bit-identical to chaining the seven *_device entry points.
a1mpc_status a1mpc_control_tick_device(a1mpc_handle h, const a1mpc_tick_params* p, const a1mpc_tick_buffers* bf, int32_t n, void* hip_stream) {
    if (!h || !p || !bf) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle / params / buffers");
    if (n < 0) return fail(A1MPC_ERR_INVALID_ARGUMENT, "negative n");
    if (n > h->max_batch) return fail(A1MPC_ERR_BATCH_TOO_LARGE, "n > max_batch given to a1mpc_create");
    if (h->cfg.horizon < 2) return fail(A1MPC_ERR_UNSUPPORTED_HORIZON, "tick records need horizon >= 2");
    const void* need[] = {bf->joint_pos, bf->joint_vel, bf->R_world, bf->R_z, bf->root_euler, bf->root_ang_vel, bf->imu_acc, bf->imu_ang_vel, bf->foot_force, bf->movement_mode,
                          bf->mpc_active, bf->root_lin_vel_d, bf->root_ang_vel_d, bf->root_pos_d_z, bf->gait_counter_speed, bf->torques_gravity, bf->gait_counter, bf->foot_pos_start,
                          bf->foot_pos_rel_last_time, bf->foot_pos_target_last_time, bf->root_euler_d, bf->joint_torques, bf->root_pos, bf->root_lin_vel, bf->estimated_contacts,
                          bf->plan_contacts, bf->contacts, bf->foot_pos_rel, bf->j_foot_blocks, bf->foot_vel_rel, bf->foot_pos_abs, bf->foot_pos_target_rel, bf->foot_pos_cur,
                          bf->foot_forces_kin, bf->foot_pos_recent_contact, bf->terrain_angle, bf->grf};
    for (const void* q : need) if (!q) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null device pointer in a1mpc_tick_buffers (only the optional outputs may be null)");
    if (!(p->control_dt > 0)) return fail(A1MPC_ERR_INVALID_ARGUMENT, "control_dt <= 0");
    if (n == 0) return A1MPC_OK;
    A1_HIP(hipSetDevice(h->device));
    hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->stream;
    A1_ORDER(h, s);
    const size_t N = n;
    if (!h->d_tickrec) A1_HIP(hipMalloc(&h->d_tickrec, (static_cast<size_t>(h->max_batch) * 22 + 3) * sizeof(double)));
    if (!h->d_ekf_state) {
        A1_HIP(hipMalloc(&h->d_ekf_state, static_cast<size_t>(h->max_batch) * kEkfState * sizeof(double)));
        A1_HIP(hipMemsetAsync(h->d_ekf_state, 0, static_cast<size_t>(h->max_batch) * kEkfState * sizeof(double), s));
    }
    if (a1mpc_status st = ensure_contact_state(h, s); st != A1MPC_OK) return st;
    if (!h->timing && (tick_markers() & 2)) A1_HIP(hipEventRecord(h->ev_mark, s));
    if (h->timing) A1_HIP(hipEventRecord(h->ev_tick0, s));   // (ev_tick0 .. ev_tick1 = the whole tick; ev0 .. ev1 = the MPC launch, as after every solve)
    {   // 1. leg state (uses the previous estimate of root_pos / root_lin_vel for the world-frame outputs, like the reference's callback)
        LegArgs a;
        a.n = n;
        std::memcpy(a.rho_fix, p->rho_fix, sizeof a.rho_fix); std::memcpy(a.rho_opt, p->rho_opt, sizeof a.rho_opt);
        a.q = bf->joint_pos; a.qd = bf->joint_vel; a.R = bf->R_world; a.pos = bf->root_pos; a.vel = bf->root_lin_vel; a.rel = bf->foot_pos_rel; a.Jb = bf->j_foot_blocks;
        a.vrel = bf->foot_vel_rel; a.pabs = bf->foot_pos_abs; a.vabs = bf->foot_vel_abs; a.pworld = bf->foot_pos_world; a.vworld = bf->foot_vel_world;
        hipLaunchKernelGGL(a1mpc_leg_kernel, dim3(static_cast<unsigned>((N * 4 + 255) / 256)), dim3(256), 0, s, a);
    }
    {   // 2. EKF
        EkfArgs a;
        a.n = n; a.dt = p->control_dt; a.flat = p->assume_flat_ground; a.state = h->d_ekf_state; a.mode = bf->movement_mode; a.ff = bf->foot_force; a.R = bf->R_world;
        a.acc = bf->imu_acc; a.w = bf->imu_ang_vel; a.fk = bf->foot_pos_rel; a.fv = bf->foot_vel_rel; a.pos_out = bf->root_pos; a.vel_out = bf->root_lin_vel;
        a.ec_out = bf->estimated_contacts;
        if (n > h->ekf_ready_n) {   // (robots 0 .. ekf_ready_n - 1 have been through init_state: nothing for the init kernel to do -- one launch less per tick)
            hipLaunchKernelGGL(a1mpc_ekf_init_kernel, dim3(static_cast<unsigned>((N + 255) / 256)), dim3(256), 0, s, a);
            if (hipError_t ei = hipGetLastError(); ei != hipSuccess) { h->ekf_ready_n = 0; return fail(A1MPC_ERR_HIP, std::string("a1mpc_ekf_init_kernel: ") + hipGetErrorString(ei)); }
            h->ekf_ready_n = n;
        }
        launch_ekf_update(a, s);
    }
    {   // 3. update_plan + 4. swing legs, one launch (a1mpc_plan_swing_kernel).  (Round 5 trial: the swing block as its own launch on a second stream beside stage 5 --
        //    both only need the plan's outputs -- joined in front of the MPC launch: the two cross-stream event waits cost more than the 5 us kernel they hide,
        //    0.490 -> 0.503 ms per tick at 4096 robots; profiles/r05_control_tick.txt)
        PlanArgs a;
        a.g = p->gait; a.n = n; a.movement_mode = bf->movement_mode; a.gait_counter = bf->gait_counter; a.gait_counter_speed = bf->gait_counter_speed;
        a.root_lin_vel = bf->root_lin_vel; a.Rz = bf->R_z; a.Rw = bf->R_world; a.root_pos = bf->root_pos; a.root_lin_vel_d = bf->root_lin_vel_d;
        a.plan_contacts = bf->plan_contacts; a.rel = bf->foot_pos_target_rel; a.abs_ = bf->foot_pos_target_abs; a.world = bf->foot_pos_target_world;
        SwingArgs w;
        w.n = n; w.counter_per_swing = p->gait.counter_per_swing; w.dt = p->control_dt;
        for (int k = 0; k < 3; ++k) { w.kp[k] = p->kp_foot[k]; w.kd[k] = p->kd_foot[k]; }
        w.Rz = bf->R_z; w.foot_pos_abs = bf->foot_pos_abs; w.gait_counter = bf->gait_counter; w.target_rel = bf->foot_pos_target_rel; w.start = bf->foot_pos_start;
        w.rel_last = bf->foot_pos_rel_last_time; w.target_last = bf->foot_pos_target_last_time; w.cur_out = bf->foot_pos_cur; w.kin_out = bf->foot_forces_kin;
        hipLaunchKernelGGL(a1mpc_plan_swing_kernel, dim3(static_cast<unsigned>((N * 4 + 255) / 256)), dim3(256), 0, s, a, w);
    }
    {   // 5. contacts / terrain: root_pos[2] and root_euler_d[1] are read / written in place (strides of 3); 6. the tick record, by the same lanes
        ContactArgs a;
        a.recent_in = nullptr; a.z_stride = 3; a.pitch_stride = 3;
        a.pk_euler = bf->root_euler; a.pk_pos = bf->root_pos; a.pk_ang_vel = bf->root_ang_vel; a.pk_lin_vel = bf->root_lin_vel; a.pk_euler_d = bf->root_euler_d;
        a.pk_lin_vel_d = bf->root_lin_vel_d; a.pk_ang_vel_d = bf->root_ang_vel_d; a.pk_pos_d_z = bf->root_pos_d_z; a.pk_tick = h->d_tickrec;
        a.n = n; a.counter_per_swing = p->contact.counter_per_swing; a.foot_force_low = p->contact.foot_force_low; a.use_terrain_adapt = p->contact.use_terrain_adapt;
        contact_state_pointers(h, a); a.gait_counter = bf->gait_counter; a.foot_force = bf->foot_force; a.foot_pos_abs = bf->foot_pos_abs; a.root_pos_z = bf->root_pos + 2;
        a.plan_contacts = bf->plan_contacts; a.pitch_d = bf->root_euler_d + 1; a.contacts = bf->contacts; a.recent_out = bf->foot_pos_recent_contact; a.terrain_out = bf->terrain_angle;
        launch_contact_terrain(a, s);
    }
    A1_HIP(hipGetLastError());
    double* d_km = h->d_tickrec + static_cast<size_t>(h->max_batch) * 22;
    if (std::memcmp(h->tick_km, p->km_foot, sizeof h->tick_km) != 0 || !h->tick_km_set) {   // km_foot travels once (and again when it changes): the output stage reads it from device memory
        std::memcpy(h->tick_km, p->km_foot, sizeof h->tick_km); h->tick_km_set = true;
        std::memcpy(h->h_pin, p->km_foot, 3 * sizeof(double));   // (a pinned source; the handle's pinned block is free: this entry point takes no host arrays)
        A1_HIP(hipMemcpyAsync(d_km, h->h_pin, 3 * sizeof(double), hipMemcpyHostToDevice, s));
        A1_HIP(hipStreamSynchronize(s));   // (once: the pinned words may be overwritten by a later host-pointer call)
    }
    // 7. MPC from the tick records + N3 in its output stage
    TorqueFuse tq{bf->mpc_active, bf->j_foot_blocks, bf->foot_forces_kin, bf->torques_gravity, d_km, bf->joint_torques, false};
    if (a1mpc_status st = solve_device_impl(h, n, h->d_tickrec, nullptr, nullptr, bf->R_world, bf->foot_pos_abs, bf->contacts, bf->grf, nullptr, bf->iters, bf->status, s, 0, 0, nullptr, &tq);
        st != A1MPC_OK) return st;
    if (!tq.fused) {   // the split pipeline solved this tick (a first tick, a batch beyond the fused kernel's range): N3 as its own launch
        TorqueArgs a;
        a.n = n; a.active = bf->mpc_active; a.contacts = bf->contacts; a.Jb = bf->j_foot_blocks; a.grf = bf->grf; a.fkin = bf->foot_forces_kin; a.tg = bf->torques_gravity; a.tau = bf->joint_torques;
        a.km[0] = p->km_foot[0]; a.km[1] = p->km_foot[1]; a.km[2] = p->km_foot[2];
        hipLaunchKernelGGL(a1mpc_torque_kernel, dim3(static_cast<unsigned>((N * 4 + 255) / 256)), dim3(256), 0, s, a);
        A1_HIP(hipGetLastError());
    }
    h->tick_fused = tq.fused;
    if (h->timing) A1_HIP(hipEventRecord(h->ev_tick1, s)); else if (tick_markers() & 8) A1_HIP(hipEventRecord(h->ev_mark, s));
    h->tick_timed = h->timing;
    A1_MARK(h, s);
    return A1MPC_OK;
}
a1mpc_status a1mpc_last_control_tick_ms(a1mpc_handle h, float* ms_out, int32_t* torques_fused_out) {
    if (!h || !ms_out) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle/out");
    if (!h->tick_timed) return fail(A1MPC_ERR_INVALID_ARGUMENT, "no control tick has been launched through this handle");
    A1_HIP(hipSetDevice(h->device));
    A1_HIP(hipEventSynchronize(h->ev_tick1));
    A1_HIP(hipEventElapsedTime(ms_out, h->ev_tick0, h->ev_tick1));
    if (torques_fused_out) *torques_fused_out = h->tick_fused ? 1 : 0;
    return A1MPC_OK;
}

static a1mpc_status ticks_small_batch(a1mpc_handle h, int32_t n, const double* tick, const double* R_world, const double* foot_abs, const uint8_t* contact, double* grf_body_out,
                                      double* u_full_out, int32_t* iters_out, int32_t* status_out, bool* taken);
a1mpc_status a1mpc_solve_batch_ticks(a1mpc_handle h, int32_t n, const double* tick, const double* R_world, const double* foot_abs,
                                     const uint8_t* contact, double* grf_body_out, double* u_full_out, int32_t* iters_out,
                                     int32_t* status_out) {
    if (!h) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle");
    if (n < 0 || !tick || !R_world || !foot_abs || !contact || !grf_body_out) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null input/output pointer");
    if (n > h->max_batch) return fail(A1MPC_ERR_BATCH_TOO_LARGE, "n > max_batch given to a1mpc_create");
    if (n == 0) return A1MPC_OK;
    A1_HIP(hipSetDevice(h->device));
    const size_t N = n, H = h->cfg.horizon;
    if (H < 2) return fail(A1MPC_ERR_UNSUPPORTED_HORIZON, "tick records need horizon >= 2");
    {   // a handful of ticks (the drop-in's compute_grf is n = 1): the kernel reads the handle's pinned block itself, the host polls the outputs (ticks_small_batch)
        bool taken = false;
        const a1mpc_status stz = ticks_small_batch(h, n, tick, R_world, foot_abs, contact, grf_body_out, u_full_out, iters_out, status_out, &taken);
        if (taken) return stz;
    }
    hipStream_t s = h->stream;
    A1_ORDER(h, s);
    // 22 + 9 + 12 doubles + 4 bytes per QP: the tick record rides in the x_ref staging buffer (13H >= 22 for H >= 2; H = 1 has its own room: 13 + 13)
    double* d_tick = (H >= 2) ? h->d_xref : h->d_x0;
    A1_HIP(hipMemcpyAsync(d_tick, tick, N * 22 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(h->d_R, R_world, N * 9 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(h->d_foot, foot_abs, N * 12 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(h->d_contact, contact, N * 4, hipMemcpyHostToDevice, s));
    a1mpc_status st = solve_device_impl(h, n, d_tick, nullptr, nullptr, h->d_R, h->d_foot, h->d_contact, h->d_grf,
                                        u_full_out ? h->d_u : nullptr, h->d_iters, h->d_status, s);
    if (st != A1MPC_OK) return st;
    A1_HIP(hipMemcpyAsync(grf_body_out, h->d_grf, N * 12 * sizeof(double), hipMemcpyDeviceToHost, s));
    if (u_full_out) A1_HIP(hipMemcpyAsync(u_full_out, h->d_u, N * 12 * H * sizeof(double), hipMemcpyDeviceToHost, s));
    if (iters_out) A1_HIP(hipMemcpyAsync(iters_out, h->d_iters, N * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (status_out) A1_HIP(hipMemcpyAsync(status_out, h->d_status, N * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    A1_HIP(hipStreamSynchronize(s));
    return A1MPC_OK;
}

// The host-pointer MPC entry in two halves, so that a pipeline slot can leave a batch in flight (a1mpc_pipeline_submit):
//   host_submit   snapshot the caller's arrays into ONE pinned block (the caller's program mutates them concurrently) laid out exactly like the device
//                 block, one H2D copy, the launches, one D2H copy into the pinned mirror -- all queued on the handle's stream, nothing waited for
//                 (a tick is one H2D copy, one memset, two launches and one D2H copy -- API calls, not bytes, set batch-1 latency);
//   host_collect  (after the stream has drained) the pinned mirror into the caller's output arrays.
constexpr unsigned long long kInFlightWord = 0x7ff85a5a7ff85a5aull;   // what a1mpc_solve_batch leaves in every output word of the pinned block until the kernel has written it (host_submit)
struct HostOut { size_t q_grf, q_it, q_st, q_u; };
static HostOut host_out_layout(size_t N) {
    HostOut o;
    o.q_grf = 0; o.q_it = o.q_grf + N * 12 * sizeof(double); o.q_st = o.q_it + N * sizeof(int32_t); o.q_u = o.q_st + N * sizeof(int32_t);
    return o;
}
static a1mpc_status host_submit(a1mpc_handle h, int32_t n, const double* x0, const double* x_ref, const double* R_world, const double* foot_abs,
                                const uint8_t* contact, bool want_u) {
    A1_HIP(hipSetDevice(h->device));
    const size_t N = n, H = h->cfg.horizon;
    const size_t o_x0 = 0, o_xr = o_x0 + N * 13 * sizeof(double), o_R = o_xr + N * 13 * H * sizeof(double),
                 o_f = o_R + N * 9 * sizeof(double), o_c = o_f + N * 12 * sizeof(double), in_bytes = o_c + N * 4;
    const HostOut q = host_out_layout(N);
    const size_t out_bytes = want_u ? q.q_u + N * 12 * H * sizeof(double) : q.q_u;
    char* hin = h->h_pin;
    char* hout = h->h_pin + h->h_pin_in_bytes;
    std::memcpy(hin + o_x0, x0, N * 13 * sizeof(double));
    std::memcpy(hin + o_xr, x_ref, N * 13 * H * sizeof(double));
    std::memcpy(hin + o_R, R_world, N * 9 * sizeof(double));
    std::memcpy(hin + o_f, foot_abs, N * 12 * sizeof(double));
    std::memcpy(hin + o_c, contact, N * 4);
    hipStream_t s = h->stream;
    A1_ORDER(h, s);
    // Round 5, the batch-1 control tick (BASELINE's latency metric): for a handful of QPs the kernel reads its inputs from, and writes its results to, the handle's
    // PINNED block itself (device-mapped host memory: 1.3 KB in, ~100 B out per QP over PCIe, a few transactions) -- the two staging copies each cost more on the
    // GPU's timeline than the bytes they move.  Same kernel, same bits.  A1MPC_ZERO_COPY_MAX (default 8 QPs; 0 = always stage through device memory).
    static const int zero_copy_max = [] { const char* e = getenv("A1MPC_ZERO_COPY_MAX"); return e ? atoi(e) : 8; }();
    h->zc_poll_bytes = 0;
    if (n <= zero_copy_max && h->d_pin != nullptr) {
        char* din = h->d_pin; char* dout = h->d_pin + h->h_pin_in_bytes;
        // Round 6: every output word doubles as its own completion flag.  The block the kernel is about to write is filled with a bit pattern no result has (a quiet NaN
        // with a payload as a double -- forces are finite or exactly zero, a failed solve's u is the payload-free NaN --, 0x7ff85a5a as an int32: no iteration count, no
        // status), and a1mpc_solve_batch polls until no word holds it any more instead of hipStreamSynchronize(): the host is awake 5 us earlier (p99: 9 us;
        // tools/ubench/completion_wake_ubench.hip).  No ordering between the kernel's stores is assumed: a first version that polled `status` alone, stored last behind a
        // system-scope release, saw the forces arrive AFTER it on 2 of 10 000 ticks (posted writes over PCIe with relaxed ordering).
        static const bool poll = [] { const char* e = getenv("A1MPC_POLL_COMPLETION"); return e ? atoi(e) != 0 : true; }();
        if (poll) {
            static_assert(sizeof(unsigned long long) == 8, "");
            unsigned long long* w = reinterpret_cast<unsigned long long*>(hout);
            const size_t words = (out_bytes + 7) / 8;
            for (size_t i = 0; i < words; ++i) w[i] = kInFlightWord;
            std::atomic_thread_fence(std::memory_order_release);
            h->zc_poll_bytes = out_bytes;
        }
        return a1mpc_solve_batch_device(
            h, n, reinterpret_cast<const double*>(din + o_x0), reinterpret_cast<const double*>(din + o_xr), reinterpret_cast<const double*>(din + o_R),
            reinterpret_cast<const double*>(din + o_f), reinterpret_cast<const uint8_t*>(din + o_c), reinterpret_cast<double*>(dout + q.q_grf),
            want_u ? reinterpret_cast<double*>(dout + q.q_u) : nullptr, reinterpret_cast<int32_t*>(dout + q.q_it), reinterpret_cast<int32_t*>(dout + q.q_st), s);
    }
    A1_HIP(hipMemcpyAsync(h->d_in, hin, in_bytes, hipMemcpyHostToDevice, s));
    a1mpc_status st = a1mpc_solve_batch_device(
        h, n, reinterpret_cast<const double*>(h->d_in + o_x0), reinterpret_cast<const double*>(h->d_in + o_xr),
        reinterpret_cast<const double*>(h->d_in + o_R), reinterpret_cast<const double*>(h->d_in + o_f),
        reinterpret_cast<const uint8_t*>(h->d_in + o_c), reinterpret_cast<double*>(h->d_out + q.q_grf),
        want_u ? reinterpret_cast<double*>(h->d_out + q.q_u) : nullptr, reinterpret_cast<int32_t*>(h->d_out + q.q_it),
        reinterpret_cast<int32_t*>(h->d_out + q.q_st), s);
    if (st != A1MPC_OK) return st;
    A1_HIP(hipMemcpyAsync(hout, h->d_out, out_bytes, hipMemcpyDeviceToHost, s));
    return A1MPC_OK;
}
static void host_collect(a1mpc_handle h, int32_t n, double* grf_body_out, double* u_full_out, int32_t* iters_out, int32_t* status_out) {
    const size_t N = n, H = h->cfg.horizon;
    const HostOut q = host_out_layout(N);
    const char* hout = h->h_pin + h->h_pin_in_bytes;
    std::memcpy(grf_body_out, hout + q.q_grf, N * 12 * sizeof(double));
    if (u_full_out) std::memcpy(u_full_out, hout + q.q_u, N * 12 * H * sizeof(double));
    if (iters_out) std::memcpy(iters_out, hout + q.q_it, N * sizeof(int32_t));
    if (status_out) std::memcpy(status_out, hout + q.q_st, N * sizeof(int32_t));
}

// Waits for the outputs of the launch host_submit / ticks_small_batch has just queued: polls the pinned block's output words when the kernel writes them itself
// (zc_poll_bytes, see host_submit), synchronises the stream otherwise.
static a1mpc_status host_wait_outputs(a1mpc_handle h, int32_t n) {
    bool done = false;
    if (h->zc_poll_bytes) {   // the kernel writes into the pinned block itself: poll its output words (and the stream now and then: a launch that failed never writes them)
        const HostOut q = host_out_layout(static_cast<size_t>(n));
        const char* hout = h->h_pin + h->h_pin_in_bytes;
        const volatile unsigned long long* w64 = reinterpret_cast<const volatile unsigned long long*>(hout);
        const volatile uint32_t* w32 = reinterpret_cast<const volatile uint32_t*>(hout);
        const uint32_t lo = static_cast<uint32_t>(kInFlightWord);
        const size_t i32_first = q.q_it / 4, i32_end = q.q_u / 4, d_end = h->zc_poll_bytes / 8;   // [grf doubles | iters, status int32 | u doubles]
        for (unsigned spin = 1; !done; ++spin) {
            done = true;
            for (size_t i = i32_first; i < i32_end && done; ++i) done = w32[i] != lo;                  // (written last by the kernel: the cheap test first)
            for (size_t i = 0; i < q.q_it / 8 && done; ++i) done = w64[i] != kInFlightWord;
            for (size_t i = (q.q_u + 7) / 8; i < d_end && done; ++i) done = w64[i] != kInFlightWord;
            if (done) break;
            if ((spin & 0xffff) == 0 && hipStreamQuery(h->stream) != hipErrorNotReady) break;   // (about once a millisecond) finished -- the words arrive with it -- or failed: the synchronisation below reports which
            __builtin_ia32_pause();
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        h->zc_poll_bytes = 0;
    }
    if (!done) A1_HIP(hipStreamSynchronize(h->stream));
    return A1MPC_OK;
}
// a1mpc_solve_batch_ticks for a handful of robots -- what the drop-in's compute_grf calls with n = 1 (include/a1mpc_dropin.hpp) -- the way host_submit serves a1mpc_solve_batch: the
// tick records, R, feet and contacts are snapshotted into the pinned block, the kernel reads and writes that block itself, the host polls the output words.  (Until round 6's
// last session this entry always took four pageable host-to-device copies, the launch and up to four copies back.)
static a1mpc_status ticks_small_batch(a1mpc_handle h, int32_t n, const double* tick, const double* R_world, const double* foot_abs, const uint8_t* contact, double* grf_body_out,
                                      double* u_full_out, int32_t* iters_out, int32_t* status_out, bool* taken) {
    static const int zero_copy_max = [] { const char* e = getenv("A1MPC_ZERO_COPY_MAX"); return e ? atoi(e) : 8; }();
    *taken = false;
    if (n > zero_copy_max || h->d_pin == nullptr) return A1MPC_OK;
    *taken = true;
    const size_t N = n, H = h->cfg.horizon;
    const size_t o_t = 0, o_R = o_t + N * 22 * sizeof(double), o_f = o_R + N * 9 * sizeof(double), o_c = o_f + N * 12 * sizeof(double);   // (< h_pin_in_bytes: 43 doubles + 4 bytes per QP)
    const HostOut q = host_out_layout(N);
    const bool want_u = u_full_out != nullptr;
    const size_t out_bytes = want_u ? q.q_u + N * 12 * H * sizeof(double) : q.q_u;
    char* hin = h->h_pin; char* hout = h->h_pin + h->h_pin_in_bytes;
    std::memcpy(hin + o_t, tick, N * 22 * sizeof(double));
    std::memcpy(hin + o_R, R_world, N * 9 * sizeof(double));
    std::memcpy(hin + o_f, foot_abs, N * 12 * sizeof(double));
    std::memcpy(hin + o_c, contact, N * 4);
    h->zc_poll_bytes = 0;
    static const bool poll = [] { const char* e = getenv("A1MPC_POLL_COMPLETION"); return e ? atoi(e) != 0 : true; }();
    if (poll) {
        unsigned long long* w = reinterpret_cast<unsigned long long*>(hout);
        for (size_t i = 0; i < (out_bytes + 7) / 8; ++i) w[i] = kInFlightWord;
        std::atomic_thread_fence(std::memory_order_release);
        h->zc_poll_bytes = out_bytes;
    }
    char* din = h->d_pin; char* dout = h->d_pin + h->h_pin_in_bytes;
    const a1mpc_status st = a1mpc_solve_batch_ticks_device(
        h, n, reinterpret_cast<const double*>(din + o_t), reinterpret_cast<const double*>(din + o_R), reinterpret_cast<const double*>(din + o_f),
        reinterpret_cast<const uint8_t*>(din + o_c), reinterpret_cast<double*>(dout + q.q_grf), want_u ? reinterpret_cast<double*>(dout + q.q_u) : nullptr,
        reinterpret_cast<int32_t*>(dout + q.q_it), reinterpret_cast<int32_t*>(dout + q.q_st), h->stream);
    if (st != A1MPC_OK) { h->zc_poll_bytes = 0; return st; }
    if (a1mpc_status sw = host_wait_outputs(h, n); sw != A1MPC_OK) return sw;
    host_collect(h, n, grf_body_out, u_full_out, iters_out, status_out);
    return A1MPC_OK;
}

a1mpc_status a1mpc_solve_batch(a1mpc_handle h, int32_t n, const double* x0, const double* x_ref, const double* R_world,
                               const double* foot_abs, const uint8_t* contact, double* grf_body_out, double* u_full_out,
                               int32_t* iters_out, int32_t* status_out) {
    if (!h) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle");
    if (n < 0 || !x0 || !x_ref || !R_world || !foot_abs || !contact || !grf_body_out)
        return fail(A1MPC_ERR_INVALID_ARGUMENT, "null input/output pointer");
    if (n > h->max_batch) return fail(A1MPC_ERR_BATCH_TOO_LARGE, "n > max_batch given to a1mpc_create");
    if (n == 0) return A1MPC_OK;
    if (a1mpc_status st = host_submit(h, n, x0, x_ref, R_world, foot_abs, contact, u_full_out != nullptr); st != A1MPC_OK) { h->zc_poll_bytes = 0; return st; }
    if (a1mpc_status st = host_wait_outputs(h, n); st != A1MPC_OK) return st;
    host_collect(h, n, grf_body_out, u_full_out, iters_out, status_out);
    return A1MPC_OK;
}

// ---- terrain block of compute_grf alone (S/A1RobotControl.cpp:335-376 + compute_walking_surface :566-582): foot_pos_recent_contact comes from
// the caller's A1CtrlStates (the reference fills it in generate_swing_legs_ctrl); the terrain-angle filter of every robot lives in the handle
a1mpc_status a1mpc_terrain_batch(a1mpc_handle h, int32_t use_terrain_adapt, int32_t n, const double* foot_pos_recent_contact, const double* root_pos_z,
                                 double* root_euler_d_pitch, double* terrain_angle_out) {
    if (!h) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle");
    if (n < 0 || !foot_pos_recent_contact || !root_pos_z || !root_euler_d_pitch || !terrain_angle_out) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null input/output pointer");
    if (n > h->max_batch) return fail(A1MPC_ERR_BATCH_TOO_LARGE, "n > max_batch given to a1mpc_create");
    if (n == 0) return A1MPC_OK;
    A1_HIP(hipSetDevice(h->device));
    if (a1mpc_status st = ensure_aux(h); st != A1MPC_OK) return st;
    const size_t N = n;
    hipStream_t s = h->stream;
    A1_ORDER(h, s);
    if (a1mpc_status st = ensure_contact_state(h, s); st != A1MPC_OK) return st;
    double *d_rec = h->d_aux_in, *d_z = d_rec + 12 * N, *d_pd = d_z + N, *d_ta = h->d_aux_out;
    A1_HIP(hipMemcpyAsync(d_rec, foot_pos_recent_contact, N * 12 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_z, root_pos_z, N * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(d_pd, root_euler_d_pitch, N * sizeof(double), hipMemcpyHostToDevice, s));
    ContactArgs a;
    std::memset(&a, 0, sizeof a);
    a.n = n; a.use_terrain_adapt = use_terrain_adapt; contact_state_pointers(h, a); a.root_pos_z = d_z; a.pitch_d = d_pd; a.z_stride = 1; a.pitch_stride = 1; a.pk_tick = nullptr;
    a.recent_in = d_rec; a.recent_out = nullptr; a.terrain_out = d_ta;
    launch_contact_terrain(a, s);
    A1_HIP(hipGetLastError());
    A1_MARK(h, s);
    A1_HIP(hipMemcpyAsync(terrain_angle_out, d_ta, N * sizeof(double), hipMemcpyDeviceToHost, s));
    A1_HIP(hipMemcpyAsync(root_euler_d_pitch, d_pd, N * sizeof(double), hipMemcpyDeviceToHost, s));
    A1_HIP(hipStreamSynchronize(s));
    return A1MPC_OK;
}

// ---- the dense QP the reference's ConvexMpc members hold (debug / verification; see a1mpc_form_kernel)
a1mpc_status a1mpc_form_qp_batch(a1mpc_handle h, int32_t n, const double* x0, const double* x_ref, const double* R_world, const double* foot_abs,
                                 int32_t foot_stride, const uint8_t* contact, int32_t contact_stride, const double* yaw_A, double* P_out, double* g_out,
                                 double* l_out, double* u_out) {
    if (!h) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle");
    if (n < 0 || !x0 || !x_ref || !R_world || !foot_abs || !contact || !P_out || !g_out || !l_out || !u_out)
        return fail(A1MPC_ERR_INVALID_ARGUMENT, "null input/output pointer");
    if (!((foot_stride == 0 || foot_stride == 12) && (contact_stride == 0 || contact_stride == 4)))
        return fail(A1MPC_ERR_INVALID_ARGUMENT, "foot_stride must be 0 or 12, contact_stride 0 or 4");
    if (n > h->max_batch) return fail(A1MPC_ERR_BATCH_TOO_LARGE, "n > max_batch given to a1mpc_create");
    if (n == 0) return A1MPC_OK;
    A1_HIP(hipSetDevice(h->device));
    const size_t N = n, H = h->cfg.horizon, nv = 12 * H, nc = 20 * H, nfoot = foot_stride ? 12 * H : 12, ncont = contact_stride ? 4 * H : 4;
    hipStream_t s = h->stream;
    A1_ORDER(h, s);
    // own, call-scoped device buffers: this is a verification path, not a tick
    double *d_in = nullptr, *d_out = nullptr;
    uint8_t* d_c = nullptr;
    const size_t in_d = N * (13 + 13 * H + 9 + nfoot + 1), out_d = N * (nv * nv + nv + 2 * nc);
    A1_HIP(hipMalloc(&d_in, in_d * sizeof(double)));
    hipError_t e1 = hipMalloc(&d_out, out_d * sizeof(double)), e2 = hipMalloc(&d_c, N * ncont);
    if (e1 != hipSuccess || e2 != hipSuccess) { (void)hipFree(d_in); (void)hipFree(d_out); (void)hipFree(d_c); return fail(A1MPC_ERR_HIP, "hipMalloc (a1mpc_form_qp_batch)"); }
    double *dx0 = d_in, *dxr = dx0 + N * 13, *dR = dxr + N * 13 * H, *df = dR + N * 9, *dyaw = df + N * nfoot;
    double *dP = d_out, *dg = dP + N * nv * nv, *dl = dg + N * nv, *du = dl + N * nc;
    hipError_t e = hipMemcpyAsync(dx0, x0, N * 13 * sizeof(double), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(dxr, x_ref, N * 13 * H * sizeof(double), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(dR, R_world, N * 9 * sizeof(double), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(df, foot_abs, N * nfoot * sizeof(double), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_c, contact, N * ncont, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && yaw_A) e = hipMemcpyAsync(dyaw, yaw_A, N * sizeof(double), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        FormArgs a;
        a.P = h->dp; a.n = n; a.H = static_cast<int32_t>(H); a.foot_stride = foot_stride; a.contact_stride = contact_stride;
        a.x0 = dx0; a.xref = dxr; a.R = dR; a.foot = df; a.contact = d_c; a.yaw_A = yaw_A ? dyaw : nullptr; a.Pout = dP; a.gout = dg; a.lout = dl; a.uout = du;
        hipLaunchKernelGGL(a1mpc_form_kernel, dim3(static_cast<unsigned>(N)), dim3(256), 0, s, a);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(P_out, dP, N * nv * nv * sizeof(double), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(g_out, dg, N * nv * sizeof(double), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(l_out, dl, N * nc * sizeof(double), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(u_out, du, N * nc * sizeof(double), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_in); (void)hipFree(d_out); (void)hipFree(d_c);
    if (e != hipSuccess) return fail(A1MPC_ERR_HIP, std::string("a1mpc_form_qp_batch: ") + hipGetErrorString(e));
    return A1MPC_OK;
}

static bool strides_ok(int32_t foot_stride, int32_t contact_stride) {
    return (foot_stride == 0 || foot_stride == 12) && (contact_stride == 0 || contact_stride == 4);
}
a1mpc_status a1mpc_solve_batch_strided_device(a1mpc_handle h, int32_t n, const double* d_x0, const double* d_x_ref, const double* d_R_world,
                                              const double* d_foot_abs, int32_t foot_stride, const uint8_t* d_contact, int32_t contact_stride,
                                              const double* d_yaw_A, double* d_grf_body_out, double* d_u_full_out, int32_t* d_iters_out,
                                              int32_t* d_status_out, void* hip_stream) {
    if (!d_x0 || !d_x_ref) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null input/output pointer");
    if (!strides_ok(foot_stride, contact_stride)) return fail(A1MPC_ERR_INVALID_ARGUMENT, "foot_stride must be 0 or 12, contact_stride 0 or 4");
    return solve_device_impl(h, n, nullptr, d_x0, d_x_ref, d_R_world, d_foot_abs, d_contact, d_grf_body_out, d_u_full_out, d_iters_out,
                             d_status_out, hip_stream, foot_stride, contact_stride, d_yaw_A);
}
// The host-pointer strided entry in two halves, like host_submit / host_collect (a pipeline slot leaves the batch in flight: a1mpc_pipeline_submit_strided).
// strided_host_submit: the caller's arrays are snapshotted by pageable H2D copies on the handle's stream (synchronous w.r.t. the host buffers), the launches and ONE
// D2H copy into the handle's pinned mirror are queued; host_collect hands the mirror to the caller's arrays once the stream has drained.
static a1mpc_status strided_host_submit(a1mpc_handle h, int32_t n, const double* x0, const double* x_ref, const double* R_world, const double* foot_abs, int32_t foot_stride,
                                        const uint8_t* contact, int32_t contact_stride, const double* yaw_A, bool want_u) {
    A1_HIP(hipSetDevice(h->device));
    const size_t N = n, H = h->cfg.horizon, nfoot = foot_stride ? 12 * H : 12, ncont = contact_stride ? 4 * H : 4;
    hipStream_t s = h->stream;
    A1_ORDER(h, s);
    if (!h->d_foot_steps) {
        A1_HIP(hipMalloc(&h->d_foot_steps, static_cast<size_t>(h->max_batch) * 12 * H * sizeof(double)));
        A1_HIP(hipMalloc(&h->d_contact_steps, static_cast<size_t>(h->max_batch) * 4 * H));
    }
    A1_HIP(hipMemcpyAsync(h->d_x0, x0, N * 13 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(h->d_xref, x_ref, N * 13 * H * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(h->d_R, R_world, N * 9 * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(h->d_foot_steps, foot_abs, N * nfoot * sizeof(double), hipMemcpyHostToDevice, s));
    A1_HIP(hipMemcpyAsync(h->d_contact_steps, contact, N * ncont, hipMemcpyHostToDevice, s));
    if (yaw_A) A1_HIP(hipMemcpyAsync(h->d_aux, yaw_A, N * sizeof(double), hipMemcpyHostToDevice, s));  // d_aux: n x 6 doubles of balance-QP staging, free here
    const HostOut q = host_out_layout(N);
    const size_t out_bytes = want_u ? q.q_u + N * 12 * H * sizeof(double) : q.q_u;
    a1mpc_status st = solve_device_impl(h, n, nullptr, h->d_x0, h->d_xref, h->d_R, h->d_foot_steps, h->d_contact_steps, reinterpret_cast<double*>(h->d_out + q.q_grf),
                                        want_u ? reinterpret_cast<double*>(h->d_out + q.q_u) : nullptr, reinterpret_cast<int32_t*>(h->d_out + q.q_it),
                                        reinterpret_cast<int32_t*>(h->d_out + q.q_st), s, foot_stride, contact_stride, yaw_A ? h->d_aux : nullptr);
    if (st != A1MPC_OK) return st;
    A1_HIP(hipMemcpyAsync(h->h_pin + h->h_pin_in_bytes, h->d_out, out_bytes, hipMemcpyDeviceToHost, s));
    return A1MPC_OK;
}
a1mpc_status a1mpc_solve_batch_strided(a1mpc_handle h, int32_t n, const double* x0, const double* x_ref, const double* R_world,
                                       const double* foot_abs, int32_t foot_stride, const uint8_t* contact, int32_t contact_stride,
                                       const double* yaw_A, double* grf_body_out, double* u_full_out, int32_t* iters_out, int32_t* status_out) {
    if (!strides_ok(foot_stride, contact_stride)) return fail(A1MPC_ERR_INVALID_ARGUMENT, "foot_stride must be 0 or 12, contact_stride 0 or 4");
    if (foot_stride == 0 && contact_stride == 0 && !yaw_A)  // the reference controller's case: the fast path, bit for bit
        return a1mpc_solve_batch(h, n, x0, x_ref, R_world, foot_abs, contact, grf_body_out, u_full_out, iters_out, status_out);
    if (!h) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle");
    if (n < 0 || !x0 || !x_ref || !R_world || !foot_abs || !contact || !grf_body_out)
        return fail(A1MPC_ERR_INVALID_ARGUMENT, "null input/output pointer");
    if (n > h->max_batch) return fail(A1MPC_ERR_BATCH_TOO_LARGE, "n > max_batch given to a1mpc_create");
    if (n == 0) return A1MPC_OK;
    {   // A handful of QPs on the general path (the drop-in's ConvexMpc-level call is n = 1): the small-batch path of a1mpc_solve_batch -- inputs in a pinned block the kernel
        // reads itself (one of its own: per-step feet / contacts do not fit the handle's), outputs in the handle's pinned block, the output words polled.  Until round 6's last
        // session: five or six pageable copies in, one back, a synchronisation.
        static const int zero_copy_max = [] { const char* e = getenv("A1MPC_ZERO_COPY_MAX"); return e ? atoi(e) : 8; }();
        static const bool poll = [] { const char* e = getenv("A1MPC_POLL_COMPLETION"); return e ? atoi(e) != 0 : true; }();
        if (n <= zero_copy_max && zero_copy_max <= 64 && h->d_pin != nullptr) {
            A1_HIP(hipSetDevice(h->device));
            const size_t N = n, H = h->cfg.horizon, M = static_cast<size_t>(zero_copy_max);
            if (!h->h_pin_gen) {
                h->h_pin_gen_bytes = M * ((13 + 13 * H + 9 + 12 * H + 1) * sizeof(double) + 4 * H + 8);
                A1_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->h_pin_gen), h->h_pin_gen_bytes, hipHostMallocDefault));
                void* dp = nullptr;
                if (hipHostGetDevicePointer(&dp, h->h_pin_gen, 0) == hipSuccess) h->d_pin_gen = static_cast<char*>(dp); else (void)hipGetLastError();
            }
            if (h->d_pin_gen != nullptr) {
                const size_t nfoot = foot_stride ? 12 * H : 12, ncont = contact_stride ? 4 * H : 4;
                const size_t o_x0 = 0, o_xr = o_x0 + N * 13 * sizeof(double), o_R = o_xr + N * 13 * H * sizeof(double), o_f = o_R + N * 9 * sizeof(double),
                             o_y = o_f + N * nfoot * sizeof(double), o_c = o_y + N * sizeof(double);
                const HostOut q = host_out_layout(N);
                const bool want_u = u_full_out != nullptr;
                const size_t out_bytes = want_u ? q.q_u + N * 12 * H * sizeof(double) : q.q_u;
                char* hin = h->h_pin_gen; char* hout = h->h_pin + h->h_pin_in_bytes;
                std::memcpy(hin + o_x0, x0, N * 13 * sizeof(double)); std::memcpy(hin + o_xr, x_ref, N * 13 * H * sizeof(double)); std::memcpy(hin + o_R, R_world, N * 9 * sizeof(double));
                std::memcpy(hin + o_f, foot_abs, N * nfoot * sizeof(double)); if (yaw_A) std::memcpy(hin + o_y, yaw_A, N * sizeof(double)); std::memcpy(hin + o_c, contact, N * ncont);
                h->zc_poll_bytes = 0;
                if (poll) {
                    unsigned long long* w = reinterpret_cast<unsigned long long*>(hout);
                    for (size_t i = 0; i < (out_bytes + 7) / 8; ++i) w[i] = kInFlightWord;
                    std::atomic_thread_fence(std::memory_order_release);
                    h->zc_poll_bytes = out_bytes;
                }
                const char* din = h->d_pin_gen; char* dout = h->d_pin + h->h_pin_in_bytes;
                const a1mpc_status st = solve_device_impl(
                    h, n, nullptr, reinterpret_cast<const double*>(din + o_x0), reinterpret_cast<const double*>(din + o_xr), reinterpret_cast<const double*>(din + o_R),
                    reinterpret_cast<const double*>(din + o_f), reinterpret_cast<const uint8_t*>(din + o_c), reinterpret_cast<double*>(dout + q.q_grf),
                    want_u ? reinterpret_cast<double*>(dout + q.q_u) : nullptr, reinterpret_cast<int32_t*>(dout + q.q_it), reinterpret_cast<int32_t*>(dout + q.q_st), h->stream,
                    foot_stride, contact_stride, yaw_A ? reinterpret_cast<const double*>(din + o_y) : nullptr);
                if (st != A1MPC_OK) { h->zc_poll_bytes = 0; return st; }
                if (a1mpc_status sw = host_wait_outputs(h, n); sw != A1MPC_OK) return sw;
                host_collect(h, n, grf_body_out, u_full_out, iters_out, status_out);
                return A1MPC_OK;
            }
        }
    }
    if (a1mpc_status st = strided_host_submit(h, n, x0, x_ref, R_world, foot_abs, foot_stride, contact, contact_stride, yaw_A, u_full_out != nullptr); st != A1MPC_OK) return st;
    A1_HIP(hipStreamSynchronize(h->stream));
    host_collect(h, n, grf_body_out, u_full_out, iters_out, status_out);
    return A1MPC_OK;
}
a1mpc_status a1mpc_balance_solve_batch(a1mpc_handle h, const a1mpc_balance_config* qp, int32_t n, const double* root_acc,
                                       const double* R_world, const double* R_z, const double* foot_abs, const uint8_t* contact,
                                       double* grf_body_out, double* f_world_out, int32_t* iters_out, int32_t* status_out) {
    if (!h || !qp) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle/config");
    if (const char* why = invalid_balance_config(*qp)) return fail(A1MPC_ERR_INVALID_ARGUMENT, why);
    if (n < 0 || !root_acc || !R_world || !R_z || !foot_abs || !contact || !grf_body_out)
        return fail(A1MPC_ERR_INVALID_ARGUMENT, "null input/output pointer");
    if (n > h->max_batch) return fail(A1MPC_ERR_BATCH_TOO_LARGE, "n > max_batch given to a1mpc_create");
    if (n == 0) return A1MPC_OK;
    A1_HIP(hipSetDevice(h->device));
    const size_t N = n;
    hipStream_t s = h->stream;
    A1_ORDER(h, s);
    // A handful of QPs (the drop-in's compute_grf with stance_leg_control_type = 0 is n = 1): inputs and outputs in the handle's pinned block, read and written by the kernel
    // itself, the output words polled -- the small-batch path of a1mpc_solve_batch (host_submit).  Larger batches: the transfers are small (344 B per QP); pageable copies on
    // the stream are synchronous w.r.t. the host buffer
    static const int zero_copy_max = [] { const char* e = getenv("A1MPC_ZERO_COPY_MAX"); return e ? atoi(e) : 8; }();
    static const bool poll = [] { const char* e = getenv("A1MPC_POLL_COMPLETION"); return e ? atoi(e) != 0 : true; }();
    const bool small = n <= zero_copy_max && h->d_pin != nullptr;
    const size_t o_a = 0, o_R = o_a + N * 6 * sizeof(double), o_Rz = o_R + N * 9 * sizeof(double), o_f = o_Rz + N * 9 * sizeof(double), o_c = o_f + N * 12 * sizeof(double);
    const HostOut q = host_out_layout(N);
    const size_t out_bytes = f_world_out ? q.q_u + N * 12 * sizeof(double) : q.q_u;
    h->zc_poll_bytes = 0;
    if (small) {
        char* hin = h->h_pin; char* hout = h->h_pin + h->h_pin_in_bytes;
        std::memcpy(hin + o_a, root_acc, N * 6 * sizeof(double)); std::memcpy(hin + o_R, R_world, N * 9 * sizeof(double)); std::memcpy(hin + o_Rz, R_z, N * 9 * sizeof(double));
        std::memcpy(hin + o_f, foot_abs, N * 12 * sizeof(double)); std::memcpy(hin + o_c, contact, N * 4);
        if (poll) {
            unsigned long long* w = reinterpret_cast<unsigned long long*>(hout);
            for (size_t i = 0; i < (out_bytes + 7) / 8; ++i) w[i] = kInFlightWord;
            std::atomic_thread_fence(std::memory_order_release);
            h->zc_poll_bytes = out_bytes;
        }
    } else {
        A1_HIP(hipMemcpyAsync(h->d_aux, root_acc, N * 6 * sizeof(double), hipMemcpyHostToDevice, s));
        A1_HIP(hipMemcpyAsync(h->d_R, R_world, N * 9 * sizeof(double), hipMemcpyHostToDevice, s));
        A1_HIP(hipMemcpyAsync(h->d_Rz, R_z, N * 9 * sizeof(double), hipMemcpyHostToDevice, s));
        A1_HIP(hipMemcpyAsync(h->d_foot, foot_abs, N * 12 * sizeof(double), hipMemcpyHostToDevice, s));
        A1_HIP(hipMemcpyAsync(h->d_contact, contact, N * 4, hipMemcpyHostToDevice, s));
    }
    KernelArgs a;
    std::memset(&a, 0, sizeof a);
    a.P = h->dp;
    // the H = 1 member of the family: dt = 0, wrench weights (torque first) in q2[6:12], R on the diagonal
    a.P.dt = 0.0; a.P.mu = qp->mu; a.P.fz_min = qp->F_min; a.P.fz_max = qp->F_max; a.P.warm_start = 0;
    for (int i = 0; i < 12; ++i) { a.P.q2[i] = 0.0; a.P.r2[i] = qp->R; }
    for (int k = 0; k < 3; ++k) { a.P.q2[6 + k] = qp->Q[3 + k]; a.P.q2[9 + k] = qp->Q[k]; }
    a.tab = h->d_tab1; a.n = n;
    a.root_acc = h->d_aux; a.Rz = h->d_Rz; a.R = h->d_R; a.foot = h->d_foot; a.contact = h->d_contact;
    a.grf = h->d_grf; a.u_full = h->d_u; a.iters = h->d_iters; a.status = h->d_status; a.nfact = h->d_nfact;
    if (small) {
        const char* din = h->d_pin; char* dout = h->d_pin + h->h_pin_in_bytes;
        a.root_acc = reinterpret_cast<const double*>(din + o_a); a.R = reinterpret_cast<const double*>(din + o_R); a.Rz = reinterpret_cast<const double*>(din + o_Rz);
        a.foot = reinterpret_cast<const double*>(din + o_f); a.contact = reinterpret_cast<const uint8_t*>(din + o_c);
        a.grf = reinterpret_cast<double*>(dout + q.q_grf); a.u_full = f_world_out ? reinterpret_cast<double*>(dout + q.q_u) : nullptr;
        a.iters = reinterpret_cast<int32_t*>(dout + q.q_it); a.status = reinterpret_cast<int32_t*>(dout + q.q_st);
    }
    if (h->timing) A1_HIP(hipEventRecord(h->ev0, s));
    h->staged = false;
    a1mpc_status st = launch<1, kModeBalance>(a, s);
    if (st != A1MPC_OK) { h->zc_poll_bytes = 0; return st; }
    if (h->timing) A1_HIP(hipEventRecord(h->ev1, s));
    h->timed = h->timing;
    A1_MARK(h, s);
    if (small) {
        if (a1mpc_status sw = host_wait_outputs(h, n); sw != A1MPC_OK) return sw;
        const char* hout = h->h_pin + h->h_pin_in_bytes;
        std::memcpy(grf_body_out, hout + q.q_grf, N * 12 * sizeof(double));
        if (f_world_out) std::memcpy(f_world_out, hout + q.q_u, N * 12 * sizeof(double));
        if (iters_out) std::memcpy(iters_out, hout + q.q_it, N * sizeof(int32_t));
        if (status_out) std::memcpy(status_out, hout + q.q_st, N * sizeof(int32_t));
        return A1MPC_OK;
    }
    A1_HIP(hipMemcpyAsync(grf_body_out, h->d_grf, N * 12 * sizeof(double), hipMemcpyDeviceToHost, s));
    if (f_world_out) A1_HIP(hipMemcpyAsync(f_world_out, h->d_u, N * 12 * sizeof(double), hipMemcpyDeviceToHost, s));
    if (iters_out) A1_HIP(hipMemcpyAsync(iters_out, h->d_iters, N * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (status_out) A1_HIP(hipMemcpyAsync(status_out, h->d_status, N * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    A1_HIP(hipStreamSynchronize(s));
    return A1MPC_OK;
}

a1mpc_status a1mpc_last_kernel_ms(a1mpc_handle h, float* ms_out) {
    if (!h || !ms_out) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle/out");
    if (!h->timed) return fail(A1MPC_ERR_INVALID_ARGUMENT, "no kernel has been launched through this handle");
    A1_HIP(hipSetDevice(h->device));
    A1_HIP(hipEventSynchronize(h->ev1));
    A1_HIP(hipEventElapsedTime(ms_out, h->ev0, h->ev1));
    return A1MPC_OK;
}

a1mpc_status a1mpc_last_stage_ms(a1mpc_handle h, float* form_ms_out, float* solve_ms_out) {
    if (!h || !form_ms_out || !solve_ms_out) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle/out");
    if (!h->timed) return fail(A1MPC_ERR_INVALID_ARGUMENT, "no kernel has been launched through this handle");
    A1_HIP(hipSetDevice(h->device));
    A1_HIP(hipEventSynchronize(h->ev1));
    if (h->staged) {
        A1_HIP(hipEventElapsedTime(form_ms_out, h->ev0, h->ev_mid));
        A1_HIP(hipEventElapsedTime(solve_ms_out, h->ev_mid, h->ev1));
    } else {  // fused / latency kernel: one launch from inputs to outputs, the stages are not separable from outside
        *form_ms_out = 0.0f;
        A1_HIP(hipEventElapsedTime(solve_ms_out, h->ev0, h->ev1));
    }
    return A1MPC_OK;
}

a1mpc_status a1mpc_last_nfact(a1mpc_handle h, int32_t n, int32_t* nfact_out) {
    if (!h || !nfact_out || n < 0 || n > h->max_batch) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle/out or bad n");
    if (!h->timed) return fail(A1MPC_ERR_INVALID_ARGUMENT, "no kernel has been launched through this handle");
    A1_HIP(hipSetDevice(h->device));
    A1_HIP(hipStreamSynchronize(h->last_stream));
    A1_HIP(hipMemcpy(nfact_out, h->d_nfact, static_cast<size_t>(n) * sizeof(int32_t), hipMemcpyDeviceToHost));
    return A1MPC_OK;
}

a1mpc_status a1mpc_kernel_info(a1mpc_handle h, int32_t* lds_bytes_per_workgroup, int32_t* qps_per_workgroup,
                               int32_t* threads_per_workgroup) {
    if (!h) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle");
    // the geometry of the persistent ADMM kernel large batches run (the fused / latency kernels of small batches: one wavefront of rows_per_wg QPs)
    if (cu_wide_qps(h->cfg.horizon) > 0 && cu_wide_enabled() && rows_per_wg(h->cfg.horizon) == default_rows_per_wg(h->cfg.horizon)) {   // h = 16: one CU-wide workgroup, five QPs
        const int q = cu_wide_qps(h->cfg.horizon);
        if (lds_bytes_per_workgroup) *lds_bytes_per_workgroup = static_cast<int32_t>(lds_bytes_of(h->cfg.horizon) / rows_per_wg(h->cfg.horizon) * q);
        if (qps_per_workgroup) *qps_per_workgroup = q;
        if (threads_per_workgroup) *threads_per_workgroup = 256;
        return A1MPC_OK;
    }
    if (lds_bytes_per_workgroup) *lds_bytes_per_workgroup = static_cast<int32_t>(lds_bytes_of(h->cfg.horizon));
    if (qps_per_workgroup) *qps_per_workgroup = rows_per_wg(h->cfg.horizon);
    if (threads_per_workgroup) *threads_per_workgroup = h->cfg.horizon > 1 && rows_per_wg(h->cfg.horizon) <= 2 ? 64 : 16 * rows_per_wg(h->cfg.horizon);  // (twin rows: a full wavefront)
    return A1MPC_OK;
}

// ======================================================================================================================
// Batch sharding across the GPUs of one node inside the C ABI (SURVEY 8b "device = -1 = all", 8e): one host thread, one handle and one
// stream per device, the batch cut into contiguous shards (sizes differ by at most one, remainder to the low shards -- the QPs are
// independent, there is no data-path collective).  Two transports for scatter-inputs / gather-GRFs:
//   0  pinned host memory, one hipMemcpyAsync fan-out per device each way (no root hop; every GPU is fed over its own PCIe link)
//   1  RCCL over xGMI: the whole batch goes to shard 0's GPU in one copy, grouped ncclSend / ncclRecv move the other shards' inputs
//      out and their results back (the north_star's named path); librccl.so is dlopen()ed on first use
// The same device may be listed twice (two shards on one GPU: how the single-GPU test box exercises transport 0).
struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool load() {
        if (lib) return true;
        lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!lib) return false;
#define A1_SYM(n) n = reinterpret_cast<decltype(n)>(dlsym(lib, "nccl" #n))
        A1_SYM(CommInitAll); A1_SYM(CommDestroy); A1_SYM(CommAbort); A1_SYM(GroupStart); A1_SYM(GroupEnd); A1_SYM(Send); A1_SYM(Recv); A1_SYM(GetErrorString);
#undef A1_SYM
        return CommInitAll && CommDestroy && CommAbort && GroupStart && GroupEnd && Send && Recv && GetErrorString;
    }
};
static RcclApi g_rccl;

struct a1mpc_sharded_s {
    int ndev = 0, transport = 0, max_batch = 0, horizon = 0;
    std::vector<int> dev;
    std::vector<a1mpc_handle> h;            // one engine handle per shard (its own stream, staging, warm start)
    char* pin = nullptr;                    // pinned mirror of inputs + outputs (transport 0; transport 1 uses it for the two big copies)
    size_t pin_bytes = 0;
    // transport 1: the whole batch on shard 0's device + communicators
    double *r_x0 = nullptr, *r_xref = nullptr, *r_R = nullptr, *r_foot = nullptr, *r_grf = nullptr;
    uint8_t* r_contact = nullptr;
    int32_t *r_iters = nullptr, *r_status = nullptr;
    std::vector<ncclComm_t> comm;
    std::vector<hipEvent_t> ev;             // per shard: "my part of this call is finished"
    bool broken = false;                    // an error inside an open RCCL group: the communicators were aborted, the handle only accepts a1mpc_sharded_destroy
    long long last_scatter_bytes = 0, last_gather_bytes = 0;   // what the last solve moved from the root to the other shards and back (a1mpc_sharded_last_transfer)
};
static void shard_range(int n, int g, int G, int* start, int* count) {
    const int base = n / G, rem = n % G;
    *count = base + (g < rem ? 1 : 0);
    *start = g * base + (g < rem ? g : rem);
}

void a1mpc_sharded_destroy(a1mpc_sharded S) {
    if (!S) return;
    for (size_t g = 0; g < S->comm.size(); ++g) if (S->comm[g] && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(S->comm[g]);
    if (!S->dev.empty()) (void)hipSetDevice(S->dev[0]);
    void* ptrs[] = {S->r_x0, S->r_xref, S->r_R, S->r_foot, S->r_grf, S->r_contact, S->r_iters, S->r_status};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (size_t g = 0; g < S->ev.size(); ++g) if (S->ev[g]) { (void)hipSetDevice(S->dev[g]); (void)hipEventDestroy(S->ev[g]); }
    for (a1mpc_handle h : S->h) a1mpc_destroy(h);
    if (S->pin) (void)hipHostFree(S->pin);
    delete S;
}

a1mpc_status a1mpc_sharded_create(const a1mpc_config* cfg, int32_t max_batch, const int32_t* devices, int32_t n_devices, int32_t transport,
                                  a1mpc_sharded* out) {
    if (!cfg || !out || max_batch <= 0 || (transport != 0 && transport != 1)) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null config/out, bad batch or transport");
    *out = nullptr;
    if (lds_bytes_of(cfg->horizon) == 0) return fail(A1MPC_ERR_UNSUPPORTED_HORIZON, "horizon must be " A1MPC_HORIZON_LIST);
    if (const char* why = invalid_config(*cfg)) return fail(A1MPC_ERR_INVALID_ARGUMENT, why);
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) return fail(A1MPC_ERR_NO_DEVICE, "hipGetDeviceCount found no device");
    a1mpc_sharded S = new (std::nothrow) a1mpc_sharded_s();
    if (!S) return fail(A1MPC_ERR_HIP, "out of host memory");
    if (n_devices <= 0 || !devices) { for (int d = 0; d < visible; ++d) S->dev.push_back(d); }   // -1 = all visible devices
    else for (int g = 0; g < n_devices; ++g) {
        if (devices[g] < 0 || devices[g] >= visible) { delete S; return fail(A1MPC_ERR_NO_DEVICE, "device ordinal out of range"); }
        S->dev.push_back(devices[g]);
    }
    S->ndev = static_cast<int>(S->dev.size()); S->transport = transport; S->max_batch = max_batch; S->horizon = cfg->horizon;
    const int per = (max_batch + S->ndev - 1) / S->ndev;
    const size_t N = max_batch, H = cfg->horizon;
    for (int g = 0; g < S->ndev; ++g) {
        a1mpc_handle hg = nullptr;
        const a1mpc_status st = a1mpc_create(cfg, per, S->dev[g], &hg);
        if (st != A1MPC_OK) { a1mpc_sharded_destroy(S); return st; }
        S->h.push_back(hg);
        hipEvent_t e = nullptr;
        if (hipSetDevice(S->dev[g]) != hipSuccess || hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { a1mpc_sharded_destroy(S); return fail(A1MPC_ERR_HIP, "hipEventCreate"); }
        S->ev.push_back(e);
    }
    S->pin_bytes = N * ((13 + 13 * H + 9 + 12 + 12) * sizeof(double) + 4 + 2 * sizeof(int32_t));
    if (hipHostMalloc(reinterpret_cast<void**>(&S->pin), S->pin_bytes, hipHostMallocPortable) != hipSuccess) { a1mpc_sharded_destroy(S); return fail(A1MPC_ERR_HIP, "hipHostMalloc"); }
    if (transport == 1) {
        for (int a = 0; a < S->ndev; ++a) for (int b = a + 1; b < S->ndev; ++b)
            if (S->dev[a] == S->dev[b]) { a1mpc_sharded_destroy(S); return fail(A1MPC_ERR_INVALID_ARGUMENT, "RCCL transport needs distinct devices"); }
        if (!g_rccl.load()) { a1mpc_sharded_destroy(S); return fail(A1MPC_ERR_HIP, std::string("cannot load librccl.so: ") + (dlerror() ? dlerror() : "missing symbols")); }
        S->comm.assign(S->ndev, nullptr);
        const ncclResult_t rc = g_rccl.CommInitAll(S->comm.data(), S->ndev, S->dev.data());
        if (rc != ncclSuccess) { S->comm.clear(); a1mpc_sharded_destroy(S); return fail(A1MPC_ERR_HIP, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(rc)); }
        bool ok = hipSetDevice(S->dev[0]) == hipSuccess;
        ok = ok && hipMalloc(&S->r_x0, N * 13 * sizeof(double)) == hipSuccess && hipMalloc(&S->r_xref, N * 13 * H * sizeof(double)) == hipSuccess;
        ok = ok && hipMalloc(&S->r_R, N * 9 * sizeof(double)) == hipSuccess && hipMalloc(&S->r_foot, N * 12 * sizeof(double)) == hipSuccess;
        ok = ok && hipMalloc(&S->r_contact, N * 4) == hipSuccess && hipMalloc(&S->r_grf, N * 12 * sizeof(double)) == hipSuccess;
        ok = ok && hipMalloc(&S->r_iters, N * sizeof(int32_t)) == hipSuccess && hipMalloc(&S->r_status, N * sizeof(int32_t)) == hipSuccess;
        if (!ok) { a1mpc_sharded_destroy(S); return fail(A1MPC_ERR_HIP, "hipMalloc (root staging of the RCCL transport)"); }
    }
    *out = S;
    return A1MPC_OK;
}

a1mpc_status a1mpc_sharded_info(a1mpc_sharded S, int32_t* n_shards, int32_t* devices_out, int32_t* transport) {
    if (!S) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle");
    if (n_shards) *n_shards = S->ndev;
    if (devices_out) for (int g = 0; g < S->ndev; ++g) devices_out[g] = S->dev[g];
    if (transport) *transport = S->transport;
    return A1MPC_OK;
}

// One implementation behind the four entry points (round 6): inputs as (x0, x_ref) or as the compact tick records (S/A1RobotControl.cpp:452-488), in HOST
// arrays (snapshotted into the pinned mirror first) or in DEVICE arrays resident on shard 0's GPU (the root: the fleet's state lives there, shards are fed over xGMI --
// peer copies with transport 0, grouped ncclSend / ncclRecv with transport 1 -- and their results come back into the caller's arrays on the root).  Every shard's engine
// handle carries its own warm start / update-path workspace: with cfg.warm_start = 1 | 2 and a constant n the closed loop of a1mpc_solve_batch* runs unchanged, shard by shard.
struct ShardedIo {
    bool device = false;                      // the arrays below live on shard 0's device (else: host)
    const double *x0 = nullptr, *xref = nullptr, *tick = nullptr, *R = nullptr, *foot = nullptr;
    const uint8_t* contact = nullptr;
    double* grf = nullptr; int32_t *iters = nullptr, *status = nullptr;
    void* stream = nullptr;                   // device form: the root-device stream the inputs were produced on (NULL: they are ready)
};
static a1mpc_status sharded_solve_impl(a1mpc_sharded S, int32_t n, const ShardedIo& io) {
    if (!S) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle");
    const bool ticks = io.tick != nullptr;
    if (n < 0 || (!ticks && (!io.x0 || !io.xref)) || !io.R || !io.foot || !io.contact || !io.grf) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null input/output pointer");
    if (n > S->max_batch) return fail(A1MPC_ERR_BATCH_TOO_LARGE, "n > max_batch given to a1mpc_sharded_create");
    if (S->broken) return fail(A1MPC_ERR_HIP, "an earlier call failed inside an RCCL group and the communicators were aborted: destroy this handle and create a new one");
    if (ticks && S->horizon < 2) return fail(A1MPC_ERR_UNSUPPORTED_HORIZON, "tick records need horizon >= 2");
    if (n == 0) return A1MPC_OK;
    const size_t N = n, H = S->horizon, G = S->ndev;
    // the input fields of a QP: bytes per QP, where the whole batch lies (host forms: the pinned snapshot `pin` and, transport 1, the root staging `root`; device forms: the
    // caller's array on the root device) and which staging buffer of a shard's engine handle receives a slice (tick records ride in the x_ref buffer: 13 H >= 22)
    struct Field { size_t bytes; const char* pin; const char* root; int which; };
    Field in[5]; int nin = 0;
    char* p = S->pin;
    auto take = [&](size_t bytes) { char* q = p; p += N * bytes; return q; };
    char* p_a = take(ticks ? 22 * sizeof(double) : 13 * sizeof(double));
    char* p_b = ticks ? nullptr : take(13 * H * sizeof(double));
    char* p_R = take(9 * sizeof(double)); char* p_f = take(12 * sizeof(double));
    double* p_grf = reinterpret_cast<double*>(take(12 * sizeof(double)));
    int32_t* p_it = reinterpret_cast<int32_t*>(take(sizeof(int32_t))); int32_t* p_st = reinterpret_cast<int32_t*>(take(sizeof(int32_t)));
    char* p_c = take(4);
    if (io.device) {
        if (ticks) in[nin++] = {22 * sizeof(double), nullptr, reinterpret_cast<const char*>(io.tick), 1};
        else { in[nin++] = {13 * sizeof(double), nullptr, reinterpret_cast<const char*>(io.x0), 0}; in[nin++] = {13 * H * sizeof(double), nullptr, reinterpret_cast<const char*>(io.xref), 1}; }
        in[nin++] = {9 * sizeof(double), nullptr, reinterpret_cast<const char*>(io.R), 2}; in[nin++] = {12 * sizeof(double), nullptr, reinterpret_cast<const char*>(io.foot), 3};
        in[nin++] = {4, nullptr, reinterpret_cast<const char*>(io.contact), 4};
    } else {
        // pinned snapshot (the caller's program mutates its arrays concurrently, see a1mpc.h), field after field like the device layout
        if (ticks) { std::memcpy(p_a, io.tick, N * 22 * sizeof(double)); in[nin++] = {22 * sizeof(double), p_a, reinterpret_cast<const char*>(S->r_xref), 1}; }
        else {
            std::memcpy(p_a, io.x0, N * 13 * sizeof(double)); std::memcpy(p_b, io.xref, N * 13 * H * sizeof(double));
            in[nin++] = {13 * sizeof(double), p_a, reinterpret_cast<const char*>(S->r_x0), 0}; in[nin++] = {13 * H * sizeof(double), p_b, reinterpret_cast<const char*>(S->r_xref), 1};
        }
        std::memcpy(p_R, io.R, N * 9 * sizeof(double)); std::memcpy(p_f, io.foot, N * 12 * sizeof(double)); std::memcpy(p_c, io.contact, N * 4);
        in[nin++] = {9 * sizeof(double), p_R, reinterpret_cast<const char*>(S->r_R), 2}; in[nin++] = {12 * sizeof(double), p_f, reinterpret_cast<const char*>(S->r_foot), 3};
        in[nin++] = {4, p_c, reinterpret_cast<const char*>(S->r_contact), 4};
    }
    auto shard_buf = [](a1mpc_handle h, int which) -> char* {
        switch (which) {
            case 0: return reinterpret_cast<char*>(h->d_x0);
            case 1: return reinterpret_cast<char*>(h->d_xref);
            case 2: return reinterpret_cast<char*>(h->d_R);
            case 3: return reinterpret_cast<char*>(h->d_foot);
        }
        return reinterpret_cast<char*>(h->d_contact);
    };
    // where the results of the whole batch are collected on the root device (device forms: the caller's arrays; host forms with transport 1: the root staging)
    double* root_grf = io.device ? io.grf : S->r_grf;
    int32_t* root_it = io.device ? io.iters : S->r_iters;
    int32_t* root_st = io.device ? io.status : S->r_status;
    // Error handling of a multi-device call: nothing may be left behind -- an open RCCL group is closed and every shard's stream is drained (their async copies
    // target the shared pinned mirror, which the next call overwrites) before the status goes back to the caller.
    bool group_open = false;
    auto drain = [&]() {
        if (group_open && g_rccl.GroupEnd) {
            // A failure between paired ncclSend / ncclRecv calls leaves point-to-point operations without their partner: GroupEnd submits them and a stream
            // synchronise would then wait for ever.  Close the group, ABORT every communicator -- that terminates the operations in flight -- and only
            // then drain the streams; the handle is unusable afterwards (the solve entries refuse, a1mpc_sharded_destroy frees it).
            (void)g_rccl.GroupEnd(); group_open = false;
            for (size_t g = 0; g < S->comm.size(); ++g)
                if (S->comm[g]) { if (hipSetDevice(S->dev[g]) == hipSuccess) (void)g_rccl.CommAbort(S->comm[g]); S->comm[g] = nullptr; }
            S->broken = true;
        }
        for (size_t g = 0; g < G; ++g) { if (hipSetDevice(S->h[g]->device) == hipSuccess) (void)hipStreamSynchronize(S->h[g]->stream); }
    };
#define A1_SH_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { drain(); return fail(A1MPC_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } } while (0)
#define A1_SH_ST(call) do { a1mpc_status s_ = (call); if (s_ != A1MPC_OK) { drain(); return s_; } } while (0)
    a1mpc_handle h0 = S->h[0];
    const int dev0 = h0->device;
    S->last_scatter_bytes = S->last_gather_bytes = 0;
    // device forms: every shard's stream starts behind what the caller has queued on `stream` (an event of the root device; cross-device waits are stream-ordered)
    if (io.device && io.stream != nullptr) {
        A1_SH_HIP(hipSetDevice(dev0));
        A1_SH_HIP(hipEventRecord(S->ev[0], static_cast<hipStream_t>(io.stream)));
        for (size_t g = 0; g < G; ++g) { A1_SH_HIP(hipSetDevice(S->h[g]->device)); A1_SH_HIP(hipStreamWaitEvent(S->h[g]->stream, S->ev[0], 0)); }
    }
    // the solve of shard g: on its handle's staging buffers, or -- shard 0 of the device forms and of transport 1 -- in place on the root arrays
    auto solve_shard = [&](size_t g, int c, bool in_place, size_t o) -> a1mpc_status {
        a1mpc_handle h = S->h[g];
        if (in_place) {
            const char* base[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
            for (int f = 0; f < nin; ++f) base[in[f].which] = in[f].root + o * in[f].bytes;
            return solve_device_impl(h, c, ticks ? reinterpret_cast<const double*>(base[1]) : nullptr, ticks ? nullptr : reinterpret_cast<const double*>(base[0]),
                                     ticks ? nullptr : reinterpret_cast<const double*>(base[1]), reinterpret_cast<const double*>(base[2]), reinterpret_cast<const double*>(base[3]),
                                     reinterpret_cast<const uint8_t*>(base[4]), root_grf + o * 12, nullptr, root_it ? root_it + o : h->d_iters, root_st ? root_st + o : h->d_status, h->stream);
        }
        return solve_device_impl(h, c, ticks ? h->d_xref : nullptr, ticks ? nullptr : h->d_x0, ticks ? nullptr : h->d_xref, h->d_R, h->d_foot, h->d_contact, h->d_grf, nullptr,
                                 h->d_iters, h->d_status, h->stream);
    };
    if (S->transport == 0) {
        for (size_t g = 0; g < G; ++g) {   // everything asynchronous: the G devices copy and solve concurrently
            int s0 = 0, c = 0;
            shard_range(n, static_cast<int>(g), static_cast<int>(G), &s0, &c);
            if (c == 0) continue;
            a1mpc_handle h = S->h[g];
            A1_SH_HIP(hipSetDevice(h->device));
            hipStream_t st = h->stream;
            A1_SH_ST(order_streams(h, st));
            const size_t o = s0, C = c;
            const bool in_place = io.device && g == 0;   // shard 0 of a device-resident batch reads and writes the caller's arrays
            if (!in_place) {
                for (int f = 0; f < nin; ++f) {
                    if (io.device && h->device != dev0) A1_SH_HIP(hipMemcpyPeerAsync(shard_buf(h, in[f].which), h->device, in[f].root + o * in[f].bytes, dev0, C * in[f].bytes, st));   // over xGMI
                    else if (io.device) A1_SH_HIP(hipMemcpyAsync(shard_buf(h, in[f].which), in[f].root + o * in[f].bytes, C * in[f].bytes, hipMemcpyDeviceToDevice, st));   // (a second shard on the root's own GPU)
                    else A1_SH_HIP(hipMemcpyAsync(shard_buf(h, in[f].which), in[f].pin + o * in[f].bytes, C * in[f].bytes, hipMemcpyHostToDevice, st));
                    S->last_scatter_bytes += static_cast<long long>(C * in[f].bytes);
                }
            }
            A1_SH_ST(solve_shard(g, c, in_place, o));
            if (in_place) continue;
            if (io.device && h->device != dev0) {
                A1_SH_HIP(hipMemcpyPeerAsync(root_grf + o * 12, dev0, h->d_grf, h->device, C * 12 * sizeof(double), st));
                if (root_it) A1_SH_HIP(hipMemcpyPeerAsync(root_it + o, dev0, h->d_iters, h->device, C * sizeof(int32_t), st));
                if (root_st) A1_SH_HIP(hipMemcpyPeerAsync(root_st + o, dev0, h->d_status, h->device, C * sizeof(int32_t), st));
            } else if (io.device) {
                A1_SH_HIP(hipMemcpyAsync(root_grf + o * 12, h->d_grf, C * 12 * sizeof(double), hipMemcpyDeviceToDevice, st));
                if (root_it) A1_SH_HIP(hipMemcpyAsync(root_it + o, h->d_iters, C * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
                if (root_st) A1_SH_HIP(hipMemcpyAsync(root_st + o, h->d_status, C * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
            } else {
                A1_SH_HIP(hipMemcpyAsync(p_grf + o * 12, h->d_grf, C * 12 * sizeof(double), hipMemcpyDeviceToHost, st));
                A1_SH_HIP(hipMemcpyAsync(p_it + o, h->d_iters, C * sizeof(int32_t), hipMemcpyDeviceToHost, st));
                A1_SH_HIP(hipMemcpyAsync(p_st + o, h->d_status, C * sizeof(int32_t), hipMemcpyDeviceToHost, st));
            }
            S->last_gather_bytes += static_cast<long long>(C * (12 * sizeof(double) + 2 * sizeof(int32_t)));
        }
        for (size_t g = 0; g < G; ++g) { A1_SH_HIP(hipSetDevice(S->h[g]->device)); A1_SH_HIP(hipStreamSynchronize(S->h[g]->stream)); }
    } else {
        // every RCCL call is issued with the device of ITS communicator current (one thread drives all the communicators of ncclCommInitAll: the stream
        // handed to a call belongs to that device)
#define A1_NCCL(dev, call) do { A1_SH_HIP(hipSetDevice(dev)); ncclResult_t r_ = (call); if (r_ != ncclSuccess) { drain(); return fail(A1MPC_ERR_HIP, std::string(#call) + ": " + g_rccl.GetErrorString(r_)); } } while (0)
        A1_SH_HIP(hipSetDevice(dev0));
        hipStream_t s0 = h0->stream;
        A1_SH_ST(order_streams(h0, s0));
        if (!io.device)   // host form: the whole batch to the root staging in one copy per field
            for (int f = 0; f < nin; ++f) A1_SH_HIP(hipMemcpyAsync(const_cast<char*>(in[f].root), in[f].pin, N * in[f].bytes, hipMemcpyHostToDevice, s0));
        for (size_t g = 1; g < G; ++g) { A1_SH_HIP(hipSetDevice(S->h[g]->device)); A1_SH_ST(order_streams(S->h[g], S->h[g]->stream)); }
        // scatter: shard g > 0 receives its slice of every field from shard 0's device (root drives all its xGMI links concurrently)
        A1_NCCL(dev0, g_rccl.GroupStart()); group_open = true;
        for (size_t g = 1; g < G; ++g) {
            int st0 = 0, c = 0;
            shard_range(n, static_cast<int>(g), static_cast<int>(G), &st0, &c);
            if (c == 0) continue;
            a1mpc_handle h = S->h[g];
            const size_t o = st0, C = c;
            const int gi = static_cast<int>(g);
            for (int f = 0; f < nin; ++f) {
                A1_NCCL(dev0, g_rccl.Send(in[f].root + o * in[f].bytes, C * in[f].bytes, ncclUint8, gi, S->comm[0], s0));
                A1_NCCL(h->device, g_rccl.Recv(shard_buf(h, in[f].which), C * in[f].bytes, ncclUint8, 0, S->comm[g], h->stream));
                S->last_scatter_bytes += static_cast<long long>(C * in[f].bytes);
            }
        }
        group_open = false;
        A1_NCCL(dev0, g_rccl.GroupEnd());
        // every shard solves on its own stream (stream order: after its receives); shard 0 works in place on the root buffers
        for (size_t g = 0; g < G; ++g) {
            int st0 = 0, c = 0;
            shard_range(n, static_cast<int>(g), static_cast<int>(G), &st0, &c);
            if (c == 0) continue;
            A1_SH_HIP(hipSetDevice(S->h[g]->device));
            A1_SH_ST(solve_shard(g, c, g == 0, static_cast<size_t>(st0)));
        }
        // gather: results of shard g > 0 back to their slice of the root buffers
        A1_NCCL(dev0, g_rccl.GroupStart()); group_open = true;
        for (size_t g = 1; g < G; ++g) {
            int st0 = 0, c = 0;
            shard_range(n, static_cast<int>(g), static_cast<int>(G), &st0, &c);
            if (c == 0) continue;
            a1mpc_handle h = S->h[g];
            const size_t o = st0, C = c;
            const int gi = static_cast<int>(g);
            A1_NCCL(h->device, g_rccl.Send(h->d_grf, C * 12, ncclFloat64, 0, S->comm[g], h->stream));
            A1_NCCL(dev0, g_rccl.Recv(root_grf + o * 12, C * 12, ncclFloat64, gi, S->comm[0], s0));
            if (root_it) { A1_NCCL(h->device, g_rccl.Send(h->d_iters, C, ncclInt32, 0, S->comm[g], h->stream)); A1_NCCL(dev0, g_rccl.Recv(root_it + o, C, ncclInt32, gi, S->comm[0], s0)); }
            if (root_st) { A1_NCCL(h->device, g_rccl.Send(h->d_status, C, ncclInt32, 0, S->comm[g], h->stream)); A1_NCCL(dev0, g_rccl.Recv(root_st + o, C, ncclInt32, gi, S->comm[0], s0)); }
            S->last_gather_bytes += static_cast<long long>(C * (12 * sizeof(double) + 2 * sizeof(int32_t)));
        }
        group_open = false;
        A1_NCCL(dev0, g_rccl.GroupEnd());
        // the RCCL operations ran on the handles' streams behind their solves: the handles' "last launch" events move with them
        for (size_t g = 0; g < G; ++g) { A1_SH_HIP(hipSetDevice(S->h[g]->device)); A1_SH_ST(mark_launched(S->h[g], S->h[g]->stream)); }
        A1_SH_HIP(hipSetDevice(dev0));
        if (!io.device) {
            A1_SH_HIP(hipMemcpyAsync(p_grf, S->r_grf, N * 12 * sizeof(double), hipMemcpyDeviceToHost, s0));
            A1_SH_HIP(hipMemcpyAsync(p_it, S->r_iters, N * sizeof(int32_t), hipMemcpyDeviceToHost, s0));
            A1_SH_HIP(hipMemcpyAsync(p_st, S->r_status, N * sizeof(int32_t), hipMemcpyDeviceToHost, s0));
        }
        for (size_t g = 0; g < G; ++g) { A1_SH_HIP(hipSetDevice(S->h[g]->device)); A1_SH_HIP(hipStreamSynchronize(S->h[g]->stream)); }
#undef A1_NCCL
    }
#undef A1_SH_HIP
#undef A1_SH_ST
    if (!io.device) {
        std::memcpy(io.grf, p_grf, N * 12 * sizeof(double));
        if (io.iters) std::memcpy(io.iters, p_it, N * sizeof(int32_t));
        if (io.status) std::memcpy(io.status, p_st, N * sizeof(int32_t));
    }
    return A1MPC_OK;
}

a1mpc_status a1mpc_sharded_solve_batch(a1mpc_sharded S, int32_t n, const double* x0, const double* x_ref, const double* R_world, const double* foot_abs,
                                       const uint8_t* contact, double* grf_body_out, int32_t* iters_out, int32_t* status_out) {
    ShardedIo io; io.x0 = x0; io.xref = x_ref; io.R = R_world; io.foot = foot_abs; io.contact = contact; io.grf = grf_body_out; io.iters = iters_out; io.status = status_out;
    if (!x0 || !x_ref) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null input/output pointer");
    return sharded_solve_impl(S, n, io);
}
a1mpc_status a1mpc_sharded_solve_batch_ticks(a1mpc_sharded S, int32_t n, const double* tick, const double* R_world, const double* foot_abs, const uint8_t* contact,
                                             double* grf_body_out, int32_t* iters_out, int32_t* status_out) {
    ShardedIo io; io.tick = tick; io.R = R_world; io.foot = foot_abs; io.contact = contact; io.grf = grf_body_out; io.iters = iters_out; io.status = status_out;
    if (!tick) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null input/output pointer");
    return sharded_solve_impl(S, n, io);
}
a1mpc_status a1mpc_sharded_solve_batch_device(a1mpc_sharded S, int32_t n, const double* d_x0, const double* d_x_ref, const double* d_R_world, const double* d_foot_abs,
                                              const uint8_t* d_contact, double* d_grf_body_out, int32_t* d_iters_out, int32_t* d_status_out, void* hip_stream) {
    ShardedIo io; io.device = true; io.x0 = d_x0; io.xref = d_x_ref; io.R = d_R_world; io.foot = d_foot_abs; io.contact = d_contact; io.grf = d_grf_body_out;
    io.iters = d_iters_out; io.status = d_status_out; io.stream = hip_stream;
    if (!d_x0 || !d_x_ref) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null input/output pointer");
    return sharded_solve_impl(S, n, io);
}
a1mpc_status a1mpc_sharded_solve_batch_ticks_device(a1mpc_sharded S, int32_t n, const double* d_tick, const double* d_R_world, const double* d_foot_abs,
                                                    const uint8_t* d_contact, double* d_grf_body_out, int32_t* d_iters_out, int32_t* d_status_out, void* hip_stream) {
    ShardedIo io; io.device = true; io.tick = d_tick; io.R = d_R_world; io.foot = d_foot_abs; io.contact = d_contact; io.grf = d_grf_body_out;
    io.iters = d_iters_out; io.status = d_status_out; io.stream = hip_stream;
    if (!d_tick) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null input/output pointer");
    return sharded_solve_impl(S, n, io);
}
// shard g's engine handle (warm-start I/O, a1mpc_reset_warm_start, the instrumentation calls) and the bytes the last solve moved between the root and the other shards
a1mpc_status a1mpc_sharded_handle(a1mpc_sharded S, int32_t shard, a1mpc_handle* out) {
    if (!S || !out || shard < 0 || shard >= S->ndev) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle/out or shard out of range");
    *out = S->h[shard];
    return A1MPC_OK;
}
a1mpc_status a1mpc_sharded_last_transfer(a1mpc_sharded S, int64_t* scatter_bytes, int64_t* gather_bytes) {
    if (!S) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null handle");
    if (scatter_bytes) *scatter_bytes = S->last_scatter_bytes;
    if (gather_bytes) *gather_bytes = S->last_gather_bytes;
    return A1MPC_OK;
}


// ======================================================================================================================
// Batch pipeline: `depth` engine handles on `depth` HIP streams of ONE device, batches submitted round-robin.  A launch of a few thousand QPs
// ends in a tail (its few 150-225-iteration QPs keep a handful of wavefronts busy, the rest of the chip idles); with a second batch in flight
// on another stream the hardware dispatches that batch's set-up kernel and persistent rows onto the SIMDs the tail has left.  Every slot is a
// complete handle (its own prepared-state records, queue, warm start), so batches in flight share nothing and results are bit-identical to a
// lone handle's.  One host thread per pipeline (like a handle).
struct a1mpc_pipeline_s {
    int device = 0, depth = 0, next = 0;
    std::vector<a1mpc_handle> h;
    std::vector<hipEvent_t> ready, done;   // per slot: "the caller's inputs are ready" (recorded on the caller's stream), "this slot's last submit has finished"
    std::vector<char> used;
    struct HostPending { int32_t n = 0; double *grf = nullptr, *u = nullptr; int32_t *iters = nullptr, *status = nullptr; };
    std::vector<HostPending> pending;      // per slot: a host-pointer submit whose outputs are still in the slot's pinned mirror (n = 0: none)
};
// a slot's host-pointer batch has to be handed to its caller before the slot's pinned mirror is reused
static a1mpc_status pipeline_deliver(a1mpc_pipeline p, int k) {
    a1mpc_pipeline_s::HostPending& q = p->pending[k];
    if (q.n == 0) return A1MPC_OK;
    A1_HIP(hipStreamSynchronize(p->h[k]->stream));   // (the slot's stream carries nothing but this batch; the wait a1mpc_solve_batch uses)
    host_collect(p->h[k], q.n, q.grf, q.u, q.iters, q.status);
    q = a1mpc_pipeline_s::HostPending();
    return A1MPC_OK;
}

void a1mpc_pipeline_destroy(a1mpc_pipeline p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    for (int k = 0; k < static_cast<int>(p->pending.size()); ++k) (void)pipeline_deliver(p, k);   // batches still in flight reach their callers' arrays before the slots go
    for (hipEvent_t e : p->ready) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : p->done) if (e) (void)hipEventDestroy(e);
    for (a1mpc_handle h : p->h) a1mpc_destroy(h);
    delete p;
}

a1mpc_status a1mpc_pipeline_create(const a1mpc_config* cfg, int32_t max_batch, int32_t device, int32_t depth, a1mpc_pipeline* out) {
    if (!cfg || !out || max_batch <= 0 || device < 0 || depth < 0 || depth > 8) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null config/out, bad batch/device or depth > 8");
    *out = nullptr;
    a1mpc_pipeline p = new (std::nothrow) a1mpc_pipeline_s();
    if (!p) return fail(A1MPC_ERR_HIP, "out of host memory");
    p->device = device; p->depth = depth;
    if (depth == 0) p->depth = 2;   // default: two batches in flight.  (Round 2 took three for batches small enough for the fused kernel; with this round's kernels two win at every
                                    // size measured -- 512 / 1024 / 2048 / 4096 x h10: 0.28 / 0.30 / 0.35 / 0.61 ms per batch at depth 2, 0.35 / 0.40 / 0.43 / 0.69 at depth 3.)
    for (int k = 0; k < p->depth; ++k) {
        a1mpc_handle hk = nullptr;
        const a1mpc_status st = a1mpc_create(cfg, max_batch, device, &hk);
        if (st != A1MPC_OK) { a1mpc_pipeline_destroy(p); return st; }
        hk->pipeline_depth = p->depth;
        p->h.push_back(hk);
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (hipEventCreateWithFlags(&e0, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&e1, hipEventDisableTiming) != hipSuccess) {
            if (e0) (void)hipEventDestroy(e0);
            a1mpc_pipeline_destroy(p);
            return fail(A1MPC_ERR_HIP, "hipEventCreate (pipeline)");
        }
        p->ready.push_back(e0); p->done.push_back(e1); p->used.push_back(0); p->pending.emplace_back();
    }
    *out = p;
    return A1MPC_OK;
}

a1mpc_status a1mpc_pipeline_depth(a1mpc_pipeline p, int32_t* depth_out) {
    if (!p || !depth_out) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null pipeline/out");
    *depth_out = p->depth;
    return A1MPC_OK;
}

a1mpc_status a1mpc_pipeline_handle(a1mpc_pipeline p, int32_t slot, a1mpc_handle* out) {
    if (!p || !out || slot < 0 || slot >= p->depth) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null pipeline/out or slot out of range");
    // A host-pointer batch still in flight on this slot keeps its results in the handle's ONE pinned mirror: any host-pointer call the caller makes on the handle
    // would overwrite them.  Hand the batch to its caller's arrays first -- the handle that leaves here carries nothing pending.
    A1_HIP(hipSetDevice(p->device));
    if (a1mpc_status sd = pipeline_deliver(p, slot); sd != A1MPC_OK) return sd;
    *out = p->h[slot];
    return A1MPC_OK;
}

a1mpc_status a1mpc_pipeline_submit_device(a1mpc_pipeline p, int32_t slot, int32_t fresh_batch, int32_t n, const double* d_x0, const double* d_x_ref,
                                          const double* d_R_world, const double* d_foot_abs, const uint8_t* d_contact, double* d_grf_body_out,
                                          double* d_u_full_out, int32_t* d_iters_out, int32_t* d_status_out, void* inputs_ready_stream, int32_t* slot_out) {
    if (!p) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null pipeline");
    if (slot >= p->depth) return fail(A1MPC_ERR_INVALID_ARGUMENT, "slot out of range");
    if (!d_x0 || !d_x_ref) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null input/output pointer");
    const int k = slot >= 0 ? slot : p->next;
    a1mpc_handle h = p->h[k];
    if (n > h->max_batch) return fail(A1MPC_ERR_BATCH_TOO_LARGE, "n > max_batch given to a1mpc_pipeline_create");
    A1_HIP(hipSetDevice(p->device));
    if (a1mpc_status sd = pipeline_deliver(p, k); sd != A1MPC_OK) return sd;
    if (inputs_ready_stream) {   // the slot's stream starts after everything the caller has queued on that stream so far
        A1_HIP(hipEventRecord(p->ready[k], static_cast<hipStream_t>(inputs_ready_stream)));
        A1_HIP(hipStreamWaitEvent(h->stream, p->ready[k], 0));
    }
    if (fresh_batch) h->hint_n = 0;   // QPs this slot has not seen before: queue ordered by the set-up kernel's cost guess, not by the costs of the slot's previous
                                      // batch (what a1mpc_set_schedule does, without touching the handle's index-order / history setting)
    if (a1mpc_status st = solve_device_impl(h, n, nullptr, d_x0, d_x_ref, d_R_world, d_foot_abs, d_contact, d_grf_body_out, d_u_full_out, d_iters_out,
                                            d_status_out, h->stream); st != A1MPC_OK) return st;
    A1_HIP(hipEventRecord(p->done[k], h->stream));
    p->used[k] = 1;
    if (slot < 0) p->next = (p->next + 1) % p->depth;
    if (slot_out) *slot_out = k;
    return A1MPC_OK;
}

// Round 6: the general path -- per-step feet / contact schedules / its own A_c yaw (S/ConvexMpc.h:74, S/test/test_mpc.cpp:106-122) -- and the compact tick
// records (S/A1RobotControl.cpp:452-488) with two batches in flight.  A first solve of the general path is PURELY tail-bound (its queue order is unpredictable:
// profiles/r05_general_path_order.txt), which is exactly the loss a second batch in flight hides.  Same slots, same events; the solve is the lone handle's, bit for bit.
static a1mpc_status pipeline_submit_device_impl(a1mpc_pipeline p, int32_t slot, int32_t fresh_batch, int32_t n, const double* d_tick, const double* d_x0, const double* d_x_ref,
                                                const double* d_R_world, const double* d_foot_abs, int32_t foot_stride, const uint8_t* d_contact, int32_t contact_stride,
                                                const double* d_yaw_A, double* d_grf_body_out, double* d_u_full_out, int32_t* d_iters_out, int32_t* d_status_out,
                                                void* inputs_ready_stream, int32_t* slot_out) {
    if (!p) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null pipeline");
    if (slot >= p->depth) return fail(A1MPC_ERR_INVALID_ARGUMENT, "slot out of range");
    if (!d_tick && (!d_x0 || !d_x_ref)) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null input/output pointer");
    if (!strides_ok(foot_stride, contact_stride)) return fail(A1MPC_ERR_INVALID_ARGUMENT, "foot_stride must be 0 or 12, contact_stride 0 or 4");
    const int k = slot >= 0 ? slot : p->next;
    a1mpc_handle h = p->h[k];
    if (n > h->max_batch) return fail(A1MPC_ERR_BATCH_TOO_LARGE, "n > max_batch given to a1mpc_pipeline_create");
    A1_HIP(hipSetDevice(p->device));
    if (a1mpc_status sd = pipeline_deliver(p, k); sd != A1MPC_OK) return sd;
    if (inputs_ready_stream) {   // the slot's stream starts after everything the caller has queued on that stream so far
        A1_HIP(hipEventRecord(p->ready[k], static_cast<hipStream_t>(inputs_ready_stream)));
        A1_HIP(hipStreamWaitEvent(h->stream, p->ready[k], 0));
    }
    if (fresh_batch) h->hint_n = 0;
    if (a1mpc_status st = solve_device_impl(h, n, d_tick, d_x0, d_x_ref, d_R_world, d_foot_abs, d_contact, d_grf_body_out, d_u_full_out, d_iters_out,
                                            d_status_out, h->stream, foot_stride, contact_stride, d_yaw_A); st != A1MPC_OK) return st;
    A1_HIP(hipEventRecord(p->done[k], h->stream));
    p->used[k] = 1;
    if (slot < 0) p->next = (p->next + 1) % p->depth;
    if (slot_out) *slot_out = k;
    return A1MPC_OK;
}
a1mpc_status a1mpc_pipeline_submit_strided_device(a1mpc_pipeline p, int32_t slot, int32_t fresh_batch, int32_t n, const double* d_x0, const double* d_x_ref,
                                                  const double* d_R_world, const double* d_foot_abs, int32_t foot_stride, const uint8_t* d_contact,
                                                  int32_t contact_stride, const double* d_yaw_A, double* d_grf_body_out, double* d_u_full_out,
                                                  int32_t* d_iters_out, int32_t* d_status_out, void* inputs_ready_stream, int32_t* slot_out) {
    return pipeline_submit_device_impl(p, slot, fresh_batch, n, nullptr, d_x0, d_x_ref, d_R_world, d_foot_abs, foot_stride, d_contact, contact_stride, d_yaw_A, d_grf_body_out,
                                       d_u_full_out, d_iters_out, d_status_out, inputs_ready_stream, slot_out);
}
a1mpc_status a1mpc_pipeline_submit_ticks_device(a1mpc_pipeline p, int32_t slot, int32_t fresh_batch, int32_t n, const double* d_tick, const double* d_R_world,
                                                const double* d_foot_abs, const uint8_t* d_contact, double* d_grf_body_out, double* d_u_full_out,
                                                int32_t* d_iters_out, int32_t* d_status_out, void* inputs_ready_stream, int32_t* slot_out) {
    if (!d_tick) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null input/output pointer");
    return pipeline_submit_device_impl(p, slot, fresh_batch, n, d_tick, nullptr, nullptr, d_R_world, d_foot_abs, 0, d_contact, 0, nullptr, d_grf_body_out,
                                       d_u_full_out, d_iters_out, d_status_out, inputs_ready_stream, slot_out);
}
// ... and host arrays in / out (what a caller on the reference's side of the boundary holds): snapshot + launches queued on the slot's stream, outputs handed over by
// a1mpc_pipeline_wait / the next submit to the slot
a1mpc_status a1mpc_pipeline_submit_strided(a1mpc_pipeline p, int32_t slot, int32_t fresh_batch, int32_t n, const double* x0, const double* x_ref, const double* R_world,
                                           const double* foot_abs, int32_t foot_stride, const uint8_t* contact, int32_t contact_stride, const double* yaw_A,
                                           double* grf_body_out, double* u_full_out, int32_t* iters_out, int32_t* status_out, int32_t* slot_out) {
    if (!strides_ok(foot_stride, contact_stride)) return fail(A1MPC_ERR_INVALID_ARGUMENT, "foot_stride must be 0 or 12, contact_stride 0 or 4");
    if (foot_stride == 0 && contact_stride == 0 && !yaw_A)   // (0, 0, NULL) IS a1mpc_pipeline_submit
        return a1mpc_pipeline_submit(p, slot, fresh_batch, n, x0, x_ref, R_world, foot_abs, contact, grf_body_out, u_full_out, iters_out, status_out, slot_out);
    if (!p) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null pipeline");
    if (slot >= p->depth) return fail(A1MPC_ERR_INVALID_ARGUMENT, "slot out of range");
    if (n < 0 || !x0 || !x_ref || !R_world || !foot_abs || !contact || !grf_body_out) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null input/output pointer");
    const int k = slot >= 0 ? slot : p->next;
    a1mpc_handle h = p->h[k];
    if (n > h->max_batch) return fail(A1MPC_ERR_BATCH_TOO_LARGE, "n > max_batch given to a1mpc_pipeline_create");
    A1_HIP(hipSetDevice(p->device));
    if (a1mpc_status sd = pipeline_deliver(p, k); sd != A1MPC_OK) return sd;   // the slot's previous batch leaves its pinned mirror first
    if (slot_out) *slot_out = k;
    if (slot < 0) p->next = (p->next + 1) % p->depth;
    if (n == 0) return A1MPC_OK;
    if (fresh_batch) h->hint_n = 0;
    if (a1mpc_status st = strided_host_submit(h, n, x0, x_ref, R_world, foot_abs, foot_stride, contact, contact_stride, yaw_A, u_full_out != nullptr); st != A1MPC_OK) return st;
    A1_HIP(hipEventRecord(p->done[k], h->stream));
    p->used[k] = 1;
    a1mpc_pipeline_s::HostPending& q = p->pending[k];
    q.n = n; q.grf = grf_body_out; q.u = u_full_out; q.iters = iters_out; q.status = status_out;
    return A1MPC_OK;
}

// Host pointers in, host pointers out, and still more than one batch in flight (the reference's boundary is host-side: A1CtrlStates in, a 3x4 matrix
// out, S/A1RobotControl.h:44): the call snapshots the inputs into the slot's pinned block, queues H2D copy + launches + D2H copy on the slot's stream
// and returns; a1mpc_pipeline_wait() hands the results to the output arrays given here.  While the GPU solves batch k the caller's thread snapshots
// batch k + 1 and the copy engines move it.
a1mpc_status a1mpc_pipeline_submit(a1mpc_pipeline p, int32_t slot, int32_t fresh_batch, int32_t n, const double* x0, const double* x_ref,
                                   const double* R_world, const double* foot_abs, const uint8_t* contact, double* grf_body_out, double* u_full_out,
                                   int32_t* iters_out, int32_t* status_out, int32_t* slot_out) {
    if (!p) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null pipeline");
    if (slot >= p->depth) return fail(A1MPC_ERR_INVALID_ARGUMENT, "slot out of range");
    if (n < 0 || !x0 || !x_ref || !R_world || !foot_abs || !contact || !grf_body_out) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null input/output pointer");
    const int k = slot >= 0 ? slot : p->next;
    a1mpc_handle h = p->h[k];
    if (n > h->max_batch) return fail(A1MPC_ERR_BATCH_TOO_LARGE, "n > max_batch given to a1mpc_pipeline_create");
    A1_HIP(hipSetDevice(p->device));
    if (a1mpc_status sd = pipeline_deliver(p, k); sd != A1MPC_OK) return sd;   // the slot's previous batch leaves its pinned mirror first
    if (slot_out) *slot_out = k;
    if (slot < 0) p->next = (p->next + 1) % p->depth;
    if (n == 0) return A1MPC_OK;
    if (fresh_batch) h->hint_n = 0;
    if (a1mpc_status st = host_submit(h, n, x0, x_ref, R_world, foot_abs, contact, u_full_out != nullptr); st != A1MPC_OK) return st;
    A1_HIP(hipEventRecord(p->done[k], h->stream));
    p->used[k] = 1;
    a1mpc_pipeline_s::HostPending& q = p->pending[k];
    q.n = n; q.grf = grf_body_out; q.u = u_full_out; q.iters = iters_out; q.status = status_out;
    return A1MPC_OK;
}

a1mpc_status a1mpc_pipeline_wait(a1mpc_pipeline p, int32_t slot) {
    if (!p || slot >= p->depth) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null pipeline or slot out of range");
    A1_HIP(hipSetDevice(p->device));
    for (int k = 0; k < p->depth; ++k)
        if ((slot < 0 || slot == k) && p->used[k]) {
            A1_HIP(hipEventSynchronize(p->done[k]));
            if (a1mpc_status sd = pipeline_deliver(p, k); sd != A1MPC_OK) return sd;
        }
    return A1MPC_OK;
}

a1mpc_status a1mpc_pipeline_join(a1mpc_pipeline p, int32_t slot, void* hip_stream) {
    if (!p || slot >= p->depth) return fail(A1MPC_ERR_INVALID_ARGUMENT, "null pipeline or slot out of range");
    A1_HIP(hipSetDevice(p->device));
    for (int k = 0; k < p->depth; ++k)
        if ((slot < 0 || slot == k) && p->used[k]) A1_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(hip_stream), p->done[k], 0));
    return A1MPC_OK;
}

}  // extern "C"
