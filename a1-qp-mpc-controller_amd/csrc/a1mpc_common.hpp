// a1mpc_common.hpp -- what every translation unit of liba1mpc.so shares: error channel, environment switches, per-horizon constants and the DECLARATIONS of the
// per-horizon launch entry points.  Round 6: the library is compiled as one translation unit per (horizon, pipeline) -- a1mpc_k_*.hip include
// a1mpc_kernels.hpp (the __global__ templates + the launch functions) and explicitly instantiate their entry points; a1mpc_hip.hip (C ABI, caller-side kernels) sees
// only the declarations below.  The units compile in parallel: a cold build is bounded by its slowest unit instead of the sum.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include <a1mpc_rowops.hpp>

#include "../../include/a1mpc.h"
#include "a1mpc_solver.hpp"
#include "a1mpc_tables.hpp"

namespace a1mpc {

using KernelArgs = BatchArgs;

// The MPC kernels with at most two QPs per wavefront run the wavefront's other rows as twins of the QP rows: rows r and r + 2 share a QP and its
// LDS image (row_is_twin(), RowSolver<.., TWIN>); the workgroup is then a full wavefront.
constexpr bool twin_rows(int h, int mode, int rows) { return rows <= 2 && h > 1 && mode == kModeMpc; }
// ... and where a wavefront holds ONE QP and the horizon is a multiple of 4, all four rows work on it as a quad (fused and latency kernels; the persistent rows: quad_rows below)
constexpr bool fused_quad_rows(int h, int mode, int rows) { return rows == 1 && h > 1 && h % 4 == 0 && mode == kModeMpc; }
constexpr int cu_wide_qps(int h) { return h == 16 ? 5 : 0; }   // QPs of a CU-wide workgroup (a1mpc_admm_cu_kernel; 0: this horizon has no such kernel -- H = 20: 40 KB per image, four per CU)

template <int H>
constexpr size_t lds_bytes(int rows) { return sizeof(double) * rows * Layout<H>::ROW_STRIDE; }

// QPs per wavefront.  LDS (not wave slots) bounds residency at 8 QPs per CU (H = 10) whatever the split, and a wave costs the same
// issue slots with 2 or 4 live rows, so 2 rows x 4 waves per CU loses nothing and each row waits for only one neighbour's
// factorisation passes (measured: +5 % at 4096 QPs, +4 % at 32768).  From H = 16 on LDS allows four QPs per CU at most: one QP (a main / twin pair of rows) per
// wavefront then puts them on four SIMDs instead of two -- no more QPs in flight, but no row waits for a wave-mate's hand-over or factor pass any more and the LDS
// conflicts between the two images go (8192 x h16 first solve 4.62 -> 4.41 ms, 16 384 x h20 10.35 -> 10.04 ms).  A1MPC_ROWS_PER_WG = 1 | 2 | 4 overrides.
constexpr int default_rows_per_wg(int horizon) { return horizon >= 16 ? 1 : 2; }
// Horizons compiled in (the reference fixes PLAN_HORIZON = 10 at compile time, S/A1Params.h:26; a run-time value here).  1 / 10 / 16 / 20 are the tuned ones (the
// balance-QP analogue, the reference's horizon, BASELINE configs[3] / [4]): fast path AND general path, every kernel variant.  The other even horizons up to 14 run the
// kernel families of BOTH paths as they instantiate for them -- a main / twin pair of rows per QP, two QPs per wavefront, split and fused pipelines, latency kernel, all three
// warm-start modes, contact schedules, tick records, per-step feet; no scratch in any of their solve kernels (the resource gate covers them) -- so that a controller built
// with another PLAN_HORIZON still finds its QP.  Odd horizons do not exist (a twin pair splits the steps by parity), 18 would spill (one QP per wavefront without the quads of
// rows that make 16 and 20 fit: 194-257 spilled registers, measured) and 2 is left out because its fused kernel and its split pipeline part by one unit in the last
// place (2.6e-13 N: a contraction the backend places differently in the two instantiations, DESIGN 8 item 5 -- every offered horizon is bit-identical across
// its pipelines, tools/cross_pipeline_bits.py).
#define A1MPC_FAST_HORIZONS(X) X(1) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(20)
#define A1MPC_MULTI_STEP_HORIZONS(X) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(20)   // ... with an update path (warm_start = 2): every one of them but 1
// (H, QPs per wavefront of the general path's kernels): every multi-step horizon
#define A1MPC_GEN_HORIZONS(X) X(4, 2) X(6, 2) X(8, 2) X(10, 2) X(12, 2) X(14, 2) X(16, 1) X(20, 1)
#define A1MPC_HORIZON_LIST "1, 4, 6, 8, 10, 12, 14, 16 or 20"
// The shipped build instantiates the kernels for the default rows per workgroup only (the other two layouts were measured and lost, above; every extra layout of
// the H = 16 / 20 kernels costs a minute of compile time): the override is honoured by -DA1MPC_ALL_ROWS tuning builds.
static int rows_per_wg(int horizon) {
#ifndef A1MPC_ALL_ROWS
    return default_rows_per_wg(horizon);
#endif
    static int r = [] {
        const char* e = getenv("A1MPC_ROWS_PER_WG");
        const int v = e ? atoi(e) : 0;
        return (v == 1 || v == 2 || v == 4) ? v : 0;
    }();
    return r ? r : default_rows_per_wg(horizon);
}

extern thread_local std::string g_last_error;   // (a1mpc_hip.hip)
inline a1mpc_status fail(a1mpc_status s, const std::string& msg) { g_last_error = msg; return s; }
#define A1_HIP(call)                                                                                          \
    do {                                                                                                      \
        hipError_t e_ = (call);                                                                               \
        if (e_ != hipSuccess) return fail(A1MPC_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Pipeline choice.  A batch that fits the resident rows of the ADMM kernel runs the fused kernel (set-up + solve in one launch: every QP
// starts at once, nothing to queue; measured 6 % faster than the split pair for 64 <= n <= 2048 at H = 10, equal at n = 1); larger batches
// run the split pipeline (set-up kernel + persistent rows, +30 % at 4096).  A1MPC_PIPELINE=fused|split forces one.  The balance QP is always fused.
inline int pipeline_mode() {  // 0 = auto, 1 = always split, 2 = always fused
    static int m = [] {
        const char* e = getenv("A1MPC_PIPELINE");
        if (e && !strcmp(e, "fused")) return 2;
        if (e && !strcmp(e, "split")) return 1;
        return 0;
    }();
    return m;
}

// Profiler ranges (the reference brackets its tick with stopwatches t1..t6, S/A1RobotControl.cpp:491-553).  With A1MPC_ROCTX=1 the launches of a solve are
// bracketed by roctx ranges ("a1mpc set-up", "a1mpc order", "a1mpc admm", "a1mpc solve (fused)"; rocprofv3 --marker-trace shows them beside the kernel trace).
// The roctx library is dlopen()ed on first use, never linked; without the variable, or without the library, a range is a no-op.
struct RoctxApi {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    RoctxApi() {
        const char* e = getenv("A1MPC_ROCTX");
        if (!(e && !strcmp(e, "1"))) return;
        void* lib = nullptr;
        for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"})
            if ((lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
        if (!lib) return;
        push = reinterpret_cast<int (*)(const char*)>(dlsym(lib, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
};
inline const RoctxApi& roctx() { static const RoctxApi api; return api; }
struct RoctxRange {
    bool on;
    explicit RoctxRange(const char* name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
    ~RoctxRange() { if (on) roctx().pop(); }
    RoctxRange(const RoctxRange&) = delete;
    RoctxRange& operator=(const RoctxRange&) = delete;
};

// per-device caches (occupancy, attribute-set flags) are shared by every handle of the process: handles of different threads are independent (a1mpc.h), so they are guarded
extern std::mutex g_cache_mu;   // (a1mpc_hip.hip)
// dynamic-LDS limit of a kernel, once per device and kernel
a1mpc_status set_lds_attr(const void* fn, size_t bytes);

// the CU-wide ADMM kernel (a1mpc_admm_cu_kernel; H = 16): one workgroup of 256 threads and five images per CU.  A1MPC_CU_WIDE=0 falls back to the one-wave kernels (A/B runs)
inline bool cu_wide_enabled() {
    static const bool on = [] { const char* e = getenv("A1MPC_CU_WIDE"); return !(e && !strcmp(e, "0")); }();
    return on;
}
// the general path's CU-wide persistent kernel (a1mpc_admm_gen_cu_kernel; H = 10: seven QPs per CU).  A1MPC_GEN_CU_WIDE=0 falls back to the one-wave kernel, six per CU (A/B runs)
inline bool gen_cu_wide_enabled() {
    static const bool on = [] { const char* e = getenv("A1MPC_GEN_CU_WIDE"); return !(e && !strcmp(e, "0")); }();
    return on;
}
// the quad-of-rows ADMM kernel (H = 20, broadcast contacts).  A1MPC_QUAD=0 falls back to the twin-pair kernel (A/B runs)
inline bool quad_enabled() {
    static const bool on = [] { const char* e = getenv("A1MPC_QUAD"); return !(e && !strcmp(e, "0")); }();
    return on;
}
// ... and the CU-wide kernel's waves 1-3 (H = 16).  A1MPC_CU_QUAD=0: twin pairs on every wave
inline bool cu_quad_enabled() {
    static const bool on = [] { const char* e = getenv("A1MPC_CU_QUAD"); return !(e && !strcmp(e, "0")); }();
    return on && quad_enabled();
}

// Did the launch just issued run a profiling (CLK) instantiation?  Set by the launch functions, read by solve_device_impl right after them (same thread): only then does
// the handle's stage record describe the solve (a fallback kernel without stamps must not leave a stale or uninitialised record behind as "profiled").
extern thread_local bool g_clk_ran;   // (a1mpc_hip.hip)
// The general path at H = 10 has two persistent kernels: the CU-wide one (seven QPs per CU, 158.9 of 160 KB of LDS) and the one-wave workgroups (six per CU, 24 KB left).
// A lone handle wants the seventh QP (4096 x h10 first solves 2.87 against 2.76 M/s).  A slot of a TWO-slot pipeline wants the 24 KB: the other slot's set-up workgroups
// (11.8 KB each) then run BESIDE its persistent rows instead of waiting for them to retire -- 3.57-3.62 against 3.37-3.41 M/s with two batches in flight, reproducibly
// (profiles/r06_general_pipeline.md); with three slots the CU-wide kernel is ahead again by 1.4 %.  Set by solve_device_impl from the handle's pipeline depth for the duration
// of a launch, read by launch_gen_split_rows (same thread).  Scheduling only: the two kernels agree bit for bit (tested).  A1MPC_GEN_PIPE_ONE_WAVE=0 ignores the hint (A/B runs)
extern thread_local bool g_gen_prefer_one_wave;   // (a1mpc_hip.hip)
// horizons whose fused / latency kernels have a profiling instantiation (the closed-loop tick and the batch-1 tick of the headline horizon; every further horizon costs a minute of compile time)
constexpr bool tick_clk_horizon(int h) { return h == 10; }

static constexpr int kCoopMaxBatch = 256;  // at most this many QPs: one wavefront per QP during set-up (the chip has 1024 SIMDs)
inline int coop_max_batch() {   // A1MPC_COOP_MAX=n moves the limit (A/B runs: profiles/r05_latency_kernel_batch_limit.txt)
    static const int v = [] { const char* e = getenv("A1MPC_COOP_MAX"); const int n = e ? atoi(e) : kCoopMaxBatch; return n > 0 ? n : kCoopMaxBatch; }();
    return v;
}
inline bool coop_setup_enabled() {   // A1MPC_COOP_SETUP=0: small batches through the fused kernels instead of the latency kernels (A/B runs)
    static const bool on = [] { const char* e = getenv("A1MPC_COOP_SETUP"); return !(e && !strcmp(e, "0")); }();
    return on;
}

// ---- per-horizon entry points: defined in a1mpc_kernels.hpp, explicitly instantiated by the a1mpc_k_*.hip units (one per horizon and pipeline)
template <int H> a1mpc_status launch_split(const KernelArgs& a, double* prep, int* counter, hipStream_t stream, hipEvent_t mid);   // a1mpc_k_h<H>_split.hip
template <int H> a1mpc_status resident_rows(int* out);                                                                              // a1mpc_k_h<H>_split.hip
template <int H, int MODE> a1mpc_status launch(const KernelArgs& a, hipStream_t stream);                                             // a1mpc_k_h<H>_fused.hip
template <int H, int ROWS> a1mpc_status launch_gen_rows(const KernelArgs& a, hipStream_t stream);                                    // a1mpc_k_gen<H>_fused.hip
template <int H, int ROWS> a1mpc_status resident_workgroups_gen(int* out);                                                          // a1mpc_k_gen<H>_split.hip
template <int H, int ROWS> a1mpc_status launch_gen_split_rows(const KernelArgs& a, double* prep, int* counter, hipStream_t stream, hipEvent_t mid, int res);   // a1mpc_k_gen<H>_split.hip
a1mpc_status launch_fused_queue(const KernelArgs& a, int* counter, hipStream_t stream);                                               // a1mpc_k_h10_split.hip (the round-5 trial kernel)
// the two small non-template kernels every pipeline shares live in a1mpc_hip.hip behind these:
void launch_order_kernel(int n, const int32_t* cost, int32_t* order, hipStream_t stream);   // a1mpc_order_kernel: the queue order of a solve, longest first
void launch_predict_kernel(const KernelArgs& a, int H, hipStream_t stream);                  // a1mpc_predict_kernel: the set-up kernel's cost guess from the inputs alone

}  // namespace a1mpc
