// a1mpc_kernels.hpp -- the gfx950 __global__ templates of the solver and the host functions that launch them.  Included by the a1mpc_k_*.hip units ONLY (each explicitly
// instantiates the entry points of one horizon and pipeline, declared in a1mpc_common.hpp); a1mpc_hip.hip never sees a definition from here.
//
// One workgroup = one wavefront = ROWS QPs (one per main / twin pair of 16-lane DPP rows; ROWS = 2 at H = 10, 1 at H >= 16), dynamic LDS =
// ROWS x Layout<H>::ROW_STRIDE doubles (20.4 KB per QP at H = 10 -> eight QPs = four workgroups per CU); at H = 16 the persistent ADMM kernel is ONE
// 256-thread workgroup per CU that carries five QPs (a1mpc_admm_cu_kernel); a wavefront that holds ONE QP (H = 20, waves 1-3 of the H = 16 workgroup) runs its four
// rows as a quad on it (RowSolver<.., QUAD>).  Three ways through a batch (launch_mpc): the latency kernel (<= 256 QPs: the
// rows of a wave share one QP's set-up), the fused kernel (up to the resident rows: one row pair = one QP from inputs to outputs; also warm-started ticks of a
// known batch at H = 10) and the split pipeline (set-up kernel -> queue-order kernel -> persistent ADMM rows that drain the queue longest-first).  The QPs of a
// batch are independent; nothing is shared between workgroups except the read-only (alpha/beta) table and the queue counter, so the blockIdx -> XCD mapping is
// irrelevant here (no L2 reuse to localise).
#pragma once
#include "a1mpc_common.hpp"

namespace a1mpc {

// ROWS = QPs (DPP rows) per workgroup; the workgroup is one wavefront with 16*ROWS live lanes (64 with twin rows).
// UPD: the instantiation that also serves warm_start = 2 (the reference's update path; built for the default ROWS of a horizon only)
// CLK: the profiling instantiation (a1mpc_set_profiling): shader-clock stamps between the stages of a tick, a.clk = n x kTickStages cycles (a1mpc_last_tick_stage_cycles)
template <int H, int MODE, int ROWS, bool UPD = false, bool CLK = false>
__global__ __launch_bounds__(64) void a1mpc_solve_kernel(const KernelArgs a) {
    extern __shared__ __attribute__((aligned(16))) double a1mpc_lds[];
    constexpr bool kTwin = twin_rows(H, MODE, ROWS);
    if constexpr (fused_quad_rows(H, MODE, ROWS)) {   // one QP per wavefront, horizon a multiple of 4: the four rows share it as a quad (RowSolver<.., QUAD>)
        // a.order (warm-started ticks of a known batch, see solve_device_impl): workgroup k takes the QP that was k-th most expensive in the previous tick -- the chip
        // runs a batch of 2 x the resident rows in two rounds, and a 50-iteration QP that starts in the second one is a tail of a whole 25-iteration segment
        const int64_t bq = a.order ? static_cast<int64_t>(a.order[blockIdx.x]) : static_cast<int64_t>(blockIdx.x);
        solve_row_with<H, MODE, false, true, UPD, true, CLK>(a.P, a.tab, [&]() { return make_io_sched<H, MODE>(a, row_opaque(bq)); }, a1mpc_lds, CLK ? a.clk + bq * kTickStages : nullptr);
        return;
    }
    const int row = kTwin ? (static_cast<int>(threadIdx.x) >> 4) & 1 : static_cast<int>(threadIdx.x) >> 4;
    if (kTwin && row >= ROWS) return;  // ROWS = 1: rows 1 and 3 have no QP
    const int64_t slot = static_cast<int64_t>(blockIdx.x) * ROWS + row;
    if (slot >= a.n) return;  // row-uniform (a twin leaves with its main row): the other rows of the wave keep all their DPP sources
    const int64_t b = a.order ? static_cast<int64_t>(a.order[slot]) : slot;
    solve_row_with<H, MODE, false, kTwin, UPD, false, CLK>(a.P, a.tab, [&]() { return make_io_sched<H, MODE>(a, row_opaque(b)); }, a1mpc_lds + row * Layout<H>::ROW_STRIDE,
                                                           CLK ? a.clk + b * kTickStages : nullptr);
}

// Round 5 trial ("split-vs-fused with a queue in both"; A1MPC_FUSED_QUEUE=1): the fused kernel as PERSISTENT wavefronts on a work queue -- no set-up
// kernel, no hand-off record through HBM, no grid-wide barrier between a batch's set-up and its iterations, so the wavefronts of the NEXT batch (another stream) can
// start on every SIMD this batch's tail vacates.  A wavefront pulls ROWS queue slots at a time (both of its QPs are set up together: the set-up is wave-wide code)
// and goes back to the queue when both have converged.  The queue order comes from a1mpc_predict_kernel (the set-up kernel's cost guess, evaluated from the inputs
// alone) or from the previous solve's costs.  Same RowSolver code as a1mpc_solve_kernel: same bits.  Measured: profiles/r05_fused_queue_trial.txt.
template <int H, int MODE, int ROWS>
__global__ __launch_bounds__(64) void a1mpc_solve_queue_kernel(const KernelArgs a, int* __restrict__ counter) {
    extern __shared__ __attribute__((aligned(16))) double a1mpc_lds[];
    static_assert(twin_rows(H, MODE, ROWS) && ROWS == 2, "two QPs per wavefront, each on a main / twin pair of rows");
    const int row = (static_cast<int>(threadIdx.x) >> 4) & 1;
    while (true) {
        int q0 = 0;
        if (threadIdx.x == 0) q0 = atomicAdd(counter, ROWS);
        q0 = __builtin_amdgcn_readfirstlane(q0);
        if (q0 >= a.n) break;
        const int slot = q0 + row;
        if (slot < a.n) {   // row-uniform (a twin follows its main row)
            const int64_t b = a.order ? static_cast<int64_t>(a.order[slot]) : static_cast<int64_t>(slot);
            solve_row_with<H, MODE, false, true, false>(a.P, a.tab, [&]() { return make_io_sched<H, MODE>(a, row_opaque(b)); }, a1mpc_lds + row * Layout<H>::ROW_STRIDE);
        }
        row_sync();   // the images are free again
    }
}

// General path (per-step feet / per-step contact schedules: S/ConvexMpc.h:74 B_mat_d_list, S/test/test_mpc.cpp:106-122): the fused kernel
// over RowSolver<.., GEN = true>, whose LDS image also holds B~_t and the bounds of every horizon step.
// UPD (round 5): the instantiation that also serves warm_start = 2, the reference's update path, on the general path (batches within the resident rows)
template <int H, int ROWS, bool UPD = false>
__global__ __launch_bounds__(64) void a1mpc_solve_gen_kernel(const KernelArgs a) {
    extern __shared__ __attribute__((aligned(16))) double a1mpc_lds[];
    static_assert(ROWS <= 2, "rows r and r + 2 of the wavefront share a QP (twin rows)");
    if constexpr (fused_quad_rows(H, kModeMpc, ROWS)) {   // one QP per wavefront, horizon a multiple of 4: a quad of rows
        const int64_t bq = static_cast<int64_t>(blockIdx.x);
        solve_row_with<H, kModeMpc, true, true, UPD, true>(a.P, a.tab, [&]() { return make_io_gen<H>(a, row_opaque(bq)); }, a1mpc_lds);
        return;
    }
    const int row = (static_cast<int>(threadIdx.x) >> 4) & 1;
    if (row >= ROWS) return;  // ROWS = 1: rows 1 and 3 have no QP
    const int64_t b = static_cast<int64_t>(blockIdx.x) * ROWS + row;
    if (b >= a.n) return;
    solve_row_with<H, kModeMpc, true, true, UPD>(a.P, a.tab, [&]() { return make_io_gen<H>(a, row_opaque(b)); }, a1mpc_lds + row * Layout<H, true>::ROW_STRIDE);
}

// Latency variant of the general path's fused kernel (H = 10; solve_latency_gen): one QP per wavefront, its four rows share the set-up, rows 0 / 2 solve
template <int H, bool UPD = false>
__global__ __launch_bounds__(64) void a1mpc_solve_gen_coop_kernel(const KernelArgs a) {
    extern __shared__ __attribute__((aligned(16))) double a1mpc_lds[];
    const int64_t b = static_cast<int64_t>(blockIdx.x);
    solve_latency_gen<H, UPD>(a.P, a.tab, [&]() { return make_io_gen<H>(a, row_opaque(b)); }, a1mpc_lds);
}

// Latency variant of the fused kernel for a handful of QPs: the four rows of a wavefront work on ONE QP during set-up (each takes every fourth
// horizon step of the Ruiz sweeps; everything else is computed redundantly and written to the one shared LDS image), then rows 1-3 retire
// and row 0 solves.  Same results bit for bit (the column maxima are exact and order-free).
template <int H, bool UPD = false, bool CLK = false>
__global__ __launch_bounds__(64) void a1mpc_solve_coop_kernel(const KernelArgs a) {
    extern __shared__ __attribute__((aligned(16))) double a1mpc_lds[];
    const int row = static_cast<int>(threadIdx.x) >> 4;
    const int64_t b = static_cast<int64_t>(blockIdx.x);
    const ProblemIO io = make_io_sched<H, kModeMpc>(a, b);
    static_assert(Prep<H>::STRIDE <= H * Layout<H>::SLOT, "the hand-off record fits the (still empty) factor region");
    [[maybe_unused]] long long c0 = 0, cF = 0, cR = 0, c1 = 0, c2 = 0;
    if constexpr (CLK) c0 = row_clock();
    {
        RowSolver<H, kModeMpc, false, false, false, false, CLK> S(a.P, stage_table<H, Layout<H>>(a.tab, a1mpc_lds, row, 4), a1mpc_lds);
        S.coop_id = row; S.coop_n = 4;
        S.template setup<UPD>(io);
        row_sync();  // every row is done with the set-up scratch aliased into the factor region
        if (row == 0) S.template save_prepared<UPD>(a1mpc_lds + Layout<H>::FAC);
        if constexpr (CLK) { cF = S.ckF; cR = S.ckR; }
    }
    constexpr bool kQuad = H % 4 == 0;   // the four rows go on as a quad; otherwise rows 1 and 3 retire
    if constexpr (!kQuad) { if (row & 1) return; }
    // Rows 0 and 2 continue exactly like a main / twin pair of the split pipeline's second kernel: a fresh solver that reads the hand-off record
    // (here through LDS).  Carrying the set-up's registers into the ADMM loop instead costs that loop its spill-free allocation.
    RowSolver<H, kModeMpc, false, false, true, false, CLK, kQuad> S(a.P, a.tab, a1mpc_lds);
    S.template load_prepared<UPD>(a1mpc_lds + Layout<H>::FAC, make_io<H, kModeMpc>(a, b));
    if constexpr (CLK) c1 = row_clock();
    S.template solve<UPD>();
    if constexpr (CLK) c2 = row_clock();
    if constexpr (UPD) S.write_outputs(make_io_sched<H, kModeMpc>(a, b), carry_of<H>(a, b));   // (make_io_sched: the output stage's joint torques read the contacts)
    else S.write_outputs(make_io_sched<H, kModeMpc>(a, b));
    if constexpr (CLK) S.store_tick_stages(a.clk ? a.clk + b * kTickStages : nullptr, c0, cF, cR, c1, c2, row_clock());
}

// ---- split pipeline (large batches) -----------------------------------------------------------------------------
// K1: formation + Ruiz, 4 QPs per wavefront, 2.8 KB of LDS per QP (H = 10), ONE wave per SIMD (WAVES = 1: all 512 registers, nothing spills).
// Until the Ruiz sweep became a short column loop (RowSolver::setup) a second wave per SIMD (256 registers each, 35-137 doubles per lane spilled) paid off
// for multi-round batches at H = 10 / 16; with the column loop it loses everywhere (65 536 x h10: 1.27 vs 1.16 ms, 32 768 x h16: 2.12 vs 1.25 ms,
// 16 384 x h20: 2.14 vs 0.75 ms; profiles/r02_setup_waves_probe.txt) and is no longer built.
template <int H, int WAVES, bool UPD = false>   // UPD: the instantiation that also serves warm_start = 2 (see a1mpc_admm_kernel)
__global__ __launch_bounds__(64, WAVES) void a1mpc_setup_kernel(const KernelArgs a, double* __restrict__ prep) {
    extern __shared__ __attribute__((aligned(16))) double a1mpc_lds[];
    const int row = static_cast<int>(threadIdx.x) >> 4;
    const int64_t b = static_cast<int64_t>(blockIdx.x) * 4 + row;
    // the (alpha/beta, beta) table is read H times per Ruiz sweep: stage it in LDS behind the four rows' regions
    double* tabl = a1mpc_lds + 4 * LayoutSetup<H>::ROW_STRIDE;
    for (int i = static_cast<int>(threadIdx.x); i < 2 * H * H; i += 64) tabl[i] = a.tab[i];
    __syncthreads();
    if (b >= a.n) return;
    setup_row<H, false, UPD>(a, tabl, b, a1mpc_lds + row * LayoutSetup<H>::ROW_STRIDE, prep);
}
// K2: persistent rows; grid = resident workgroups; every row drains the queue of prepared QPs.
#ifdef A1X_NOTWIN
constexpr bool admm_twin_rows(int, int) { return false; }
#else
constexpr bool admm_twin_rows(int h, int rows) { return twin_rows(h, kModeMpc, rows); }
#endif  // the wavefront's spare rows run as twins (RowSolver<.., TWIN>)
// UPD: the instantiation that also serves warm_start = 2 (the reference's update path); every other mode runs UPD = false, whose code is what it was before
// the update path existed (the allocation of the hot loop is sensitive to anything around it: a1mpc_solver.hpp, load_prepared)
// UNI: contacts broadcast over the horizon (contact_stride = 0): one pair of bounds for every slot (RowSolver<.., UNI>; built for H >= 16, where the registers matter)
// CLK: the profiling instantiation (a1mpc_set_profiling): shader-clock stamps around factor passes / iteration segments / residual checks, outside the hot loop
// QUAD: one QP per wavefront and a horizon that is a multiple of 4 (H = 20; waves 1-3 of the CU-wide H = 16 kernel below): rows 1 and 3 do not idle, the four rows split
// the per-lane state (RowSolver<.., QUAD>)
constexpr bool quad_rows(int h, int rows) { return h == 20 && rows == 1 && admm_twin_rows(h, rows); }
template <int H, int ROWS, bool UPD = false, bool UNI = false, bool CLK = false, bool QUAD = false>
__global__ __launch_bounds__(64) void a1mpc_admm_kernel(const KernelArgs a, const double* __restrict__ prep, int* __restrict__ counter) {
    extern __shared__ __attribute__((aligned(16))) double a1mpc_lds[];
    // ROWS <= 2: the wavefront's other rows run as twins of the QP rows (rows r and r + 2 share a QP and its LDS image, see row_is_twin)
    constexpr bool kTwin = admm_twin_rows(H, ROWS);
    const int row = static_cast<int>(threadIdx.x) >> 4;
    if constexpr (kTwin) {
        if constexpr (QUAD) {
            static_assert(quad_rows(H, ROWS), "a quad of rows: the wavefront's only QP");
            admm_rows<H, true, false, UPD, UNI, CLK, true>(a, prep, counter, a1mpc_lds);
            return;
        }
        if ((row & 1) >= ROWS) return;  // ROWS = 1: rows 1 and 3 have no QP
        admm_rows<H, true, false, UPD, UNI, CLK>(a, prep, counter, a1mpc_lds + (row & 1) * Layout<H>::ROW_STRIDE);
    } else {
        admm_rows<H, false, false, UPD, UNI, CLK>(a, prep, counter, a1mpc_lds + row * Layout<H>::ROW_STRIDE);
    }
}

// K2, CU-wide (round 4): ONE workgroup of four wavefronts owns a CU's whole LDS.  At H = 16 an image is 32.2 KB: five fit in 160 KB, but an ADMM wave needs a
// whole SIMD's register file, so with one-wave workgroups of one QP each the fifth image has no wave to serve it and rows 1 and 3 of every wave idle.  Here
// wave 0 carries TWO QPs (main / twin pairs on rows (0,2) and (1,3), exactly the H = 10 arrangement) and waves 1-3 one each: five QPs per CU instead of four.
// Rows still refill from the queue independently and nothing is shared between the waves (no workgroup barrier anywhere in admm_rows): the only coupling is
// that wave 0's two QPs wait for each other's factor passes, as every pair of H = 10 does.
// QUAD: waves 1-3 (one QP each) run their four rows as a quad (RowSolver<.., QUAD>); wave 0 keeps its two twin pairs
template <int H, bool UPD = false, bool UNI = false, bool CLK = false, bool QUAD = false>
__global__ __launch_bounds__(256) void a1mpc_admm_cu_kernel(const KernelArgs a, const double* __restrict__ prep, int* __restrict__ counter) {
    extern __shared__ __attribute__((aligned(16))) double a1mpc_lds[];
    static_assert(cu_wide_qps(H) == 5 && admm_twin_rows(H, 1), "five images: two on wave 0, one on each of waves 1-3");
    const int wave = static_cast<int>(threadIdx.x) >> 6, row = (static_cast<int>(threadIdx.x) >> 4) & 3;
    if constexpr (QUAD) {
        static_assert(H % 4 == 0, "quads split the horizon in fours");
        if (wave != 0) {
            admm_rows<H, true, false, UPD, UNI, CLK, true>(a, prep, counter, a1mpc_lds + (wave + 1) * Layout<H>::ROW_STRIDE);
            return;
        }
    }
    if (wave != 0 && (row & 1)) return;  // waves 1-3: rows 1 and 3 have no QP
    const int image = wave == 0 ? (row & 1) : wave + 1;
    admm_rows<H, true, false, UPD, UNI, CLK>(a, prep, counter, a1mpc_lds + image * Layout<H>::ROW_STRIDE);
}

// The general path's own split pipeline (round 2, last step): the same two kernels over RowSolver<.., GEN = true>.  K1 (round 6): the rows of a wavefront that share a QP --
// two at H = 10 (two QPs per wavefront), all four at H = 16 / 20 (setup_gen_rows) -- split the ROWS of the Hessian in the Ruiz passes (RowSolver::setup, general path); its LDS
// image holds the foot table and the D / m / E hand-over tables of the passes but no B~w_t table (120 + 52 H doubles per QP): the per-step columns go straight into the record,
// which carries them to K2, whose rows rebuild the per-step tables of their LDS image from it.  256 registers: two wavefronts per SIMD.
template <int H>
__global__ __launch_bounds__(64, 2) void a1mpc_setup_gen_kernel(const KernelArgs a, double* __restrict__ prep) {
    extern __shared__ __attribute__((aligned(16))) double a1mpc_lds[];
    constexpr int QPW = 4 / setup_gen_rows(H);                // QPs per wavefront: 2 (rows q and q + 2 share QP q) or 1 (all four rows)
    const int q = QPW == 2 ? (static_cast<int>(threadIdx.x) >> 4) & 1 : 0;
    const int64_t b = static_cast<int64_t>(blockIdx.x) * QPW + q;
    double* tabl = a1mpc_lds + QPW * LayoutSetup<H, true>::ROW_STRIDE;
    {   // the (gamma, beta) table: every load in flight before the first store (one L2 round trip per workgroup instead of one per 64 words; round 6: at one QP per workgroup the
        // staging is paid per QP)
        constexpr int N2 = 2 * H * H, PER = (N2 + 63) / 64;
        double v[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) { const int i = static_cast<int>(threadIdx.x) + 64 * k; v[k] = i < N2 ? a.tab[i] : 0.0; }
#pragma unroll
        for (int k = 0; k < PER; ++k) { const int i = static_cast<int>(threadIdx.x) + 64 * k; if (i < N2) tabl[i] = v[k]; }
    }
    __syncthreads();
    if (b >= a.n) return;   // (both rows of the pair leave together)
    setup_row<H, true>(a, tabl, b, a1mpc_lds + q * LayoutSetup<H, true>::ROW_STRIDE, prep);
}
template <int H, int ROWS>
__global__ __launch_bounds__(64) void a1mpc_admm_gen_kernel(const KernelArgs a, const double* __restrict__ prep, int* __restrict__ counter) {
    extern __shared__ __attribute__((aligned(16))) double a1mpc_lds[];
    static_assert(ROWS <= 2, "rows r and r + 2 of the wavefront share a QP (twin rows)");
    if constexpr (fused_quad_rows(H, kModeMpc, ROWS)) {   // one QP per wavefront, horizon a multiple of 4: a quad of rows
        admm_rows<H, true, true, false, false, false, true>(a, prep, counter, a1mpc_lds);
        return;
    }
    const int row = static_cast<int>(threadIdx.x) >> 4;
    if ((row & 1) >= ROWS) return;
    admm_rows<H, true, true>(a, prep, counter, a1mpc_lds + (row & 1) * Layout<H, true>::ROW_STRIDE);
}

// The general path's persistent kernel, CU-wide (round 6; H = 10): ONE workgroup of four wavefronts owns a CU's whole LDS.  With the per-step bounds in registers an image of the
// general path is 23.3 KB at H = 10: seven fit in 160 KB where one-wave workgroups of two QPs place six.  Waves 0-2 carry two QPs each (main / twin pairs on rows (0,2) and (1,3)),
// wave 3 one; rows refill from the queue independently and nothing is shared between the waves (no workgroup barrier in admm_rows): the same bits as a1mpc_admm_gen_kernel.
constexpr int cu_wide_gen_qps(int h) { return h == 10 ? 7 : 0; }
template <int H>
__global__ __launch_bounds__(256) void a1mpc_admm_gen_cu_kernel(const KernelArgs a, const double* __restrict__ prep, int* __restrict__ counter) {
    extern __shared__ __attribute__((aligned(16))) double a1mpc_lds[];
    static_assert(cu_wide_gen_qps(H) == 7 && !fused_quad_rows(H, kModeMpc, 1), "seven images: two on each of waves 0-2, one on wave 3; main / twin pairs of rows");
    const int wave = static_cast<int>(threadIdx.x) >> 6, row = (static_cast<int>(threadIdx.x) >> 4) & 3;
    if (wave == 3 && (row & 1)) return;   // wave 3: rows 1 and 3 have no QP
    const int image = 2 * wave + (wave == 3 ? 0 : (row & 1));
    admm_rows<H, true, true>(a, prep, counter, a1mpc_lds + image * Layout<H, true>::ROW_STRIDE);
}


// workgroups of the persistent ADMM kernel that are resident at once on the current device (occupancy query, cached per device)
template <int H, int ROWS>
static a1mpc_status resident_workgroups(int* out) {
    static int resident[64] = {};
    int dev = 0;
    A1_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) { *out = 512; return A1MPC_OK; }
    std::lock_guard<std::mutex> lock(g_cache_mu);
    if (!resident[dev]) {
        const size_t lds2 = lds_bytes<H>(ROWS);
        A1_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&a1mpc_admm_kernel<H, ROWS, false, false, false, quad_rows(H, ROWS)>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   static_cast<int>(lds2)));
        int per_cu = 0, cus = 0;
        A1_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&a1mpc_admm_kernel<H, ROWS, false, false, false, quad_rows(H, ROWS)>), admm_twin_rows(H, ROWS) ? 64 : 16 * ROWS, lds2));
        A1_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        resident[dev] = (per_cu > 0 ? per_cu : 1) * (cus > 0 ? cus : 1);
    }
    *out = resident[dev];
    return A1MPC_OK;
}
template <int H>
static a1mpc_status resident_cu_workgroups(int* out) {
    static int resident[64] = {};
    int dev = 0;
    A1_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) { *out = 256; return A1MPC_OK; }
    std::lock_guard<std::mutex> lock(g_cache_mu);
    if (!resident[dev]) {
        const size_t lds2 = lds_bytes<H>(cu_wide_qps(H));
        A1_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&a1mpc_admm_cu_kernel<H, false, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds2)));
        int per_cu = 0, cus = 0;
        A1_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&a1mpc_admm_cu_kernel<H, false, false, false, true>), 256, lds2));
        A1_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        resident[dev] = (per_cu > 0 ? per_cu : 1) * (cus > 0 ? cus : 1);
    }
    *out = resident[dev];
    return A1MPC_OK;
}
template <int H>
a1mpc_status resident_rows(int* out) {
    int wg = 0;
    a1mpc_status st;
#ifdef A1MPC_ALL_ROWS
    switch (rows_per_wg(H)) {
        case 1: st = resident_workgroups<H, 1>(&wg); *out = wg; return st;
        case 2: st = resident_workgroups<H, 2>(&wg); *out = 2 * wg; return st;
    }
    st = resident_workgroups<H, 4>(&wg); *out = 4 * wg;
#else
    st = resident_workgroups<H, default_rows_per_wg(H)>(&wg); *out = default_rows_per_wg(H) * wg;
#endif
    return st;
}

template <int H, int ROWS>
static a1mpc_status launch_split_rows(const KernelArgs& a, double* prep, int* counter, hipStream_t stream, hipEvent_t mid) {
    const size_t lds2 = lds_bytes<H>(ROWS), lds1 = sizeof(double) * (4 * LayoutSetup<H>::ROW_STRIDE + 2 * H * H);
    int res = 0;
    if (a1mpc_status st = resident_workgroups<H, ROWS>(&res); st != A1MPC_OK) return st;
    A1_HIP(hipMemsetAsync(counter, 0, sizeof(int), stream));
    bool upd_kernels = false;   // warm_start = 2: the update-path instantiations of the two kernels (H > 1, default rows per workgroup; same resources, a few more instructions around set-up and iteration 1)
    constexpr bool kHasUpd = H > 1 && ROWS == default_rows_per_wg(H);
    if constexpr (kHasUpd) upd_kernels = a.carry != nullptr;
    {
        RoctxRange range("a1mpc set-up");
        if constexpr (kHasUpd) { if (upd_kernels) hipLaunchKernelGGL((a1mpc_setup_kernel<H, 1, true>), dim3(static_cast<unsigned>((a.n + 3) / 4)), dim3(64), lds1, stream, a, prep); }
        if (!upd_kernels) hipLaunchKernelGGL((a1mpc_setup_kernel<H, 1>), dim3(static_cast<unsigned>((a.n + 3) / 4)), dim3(64), lds1, stream, a, prep);
    }
    A1_HIP(hipGetLastError());
    // queue order of THIS solve, longest first: by the set-up kernel's cost guesses (predict: no history) or by the cost each QP had in the handle's previous
    // solve of this batch size (the cost buffer still holds it; the ADMM kernel below overwrites it with this solve's).  Sorted here, in front of the kernel
    // that needs it, not behind the solve that produced the costs: a one-workgroup kernel of 1024 threads behind a persistent kernel waits for a free CU, and
    // with a second batch in flight on another stream (a1mpc_pipeline) that wait was ~0.5 ms per launch (kernel trace, profiles/r02_kernel_trace_overlap.json)
    if (a.cost != nullptr && a.order != nullptr) {
        RoctxRange range("a1mpc order");
        launch_order_kernel(static_cast<int>(a.n), static_cast<const int32_t*>(a.cost), const_cast<int32_t*>(a.order), stream);
        A1_HIP(hipGetLastError());
    }
    if (mid) A1_HIP(hipEventRecord(mid, stream));  // stage split: formation + Ruiz (+ queue order) | factor + iterate
    RoctxRange range_admm("a1mpc admm");
    if constexpr (cu_wide_qps(H) > 0 && ROWS == default_rows_per_wg(H)) {
        if (cu_wide_enabled()) {   // five QPs per CU: one 256-thread workgroup per CU (a1mpc_admm_cu_kernel)
            constexpr int Q = cu_wide_qps(H);
            const size_t ldsq = lds_bytes<H>(Q);
            int resq = 0;
            if (a1mpc_status st = resident_cu_workgroups<H>(&resq); st != A1MPC_OK) return st;
            const int wantq = (a.n + Q - 1) / Q;
            const dim3 gridq(static_cast<unsigned>(wantq < resq ? wantq : resq)), blockq(256);
            if (a.clk != nullptr && a.carry == nullptr && a.contact_stride == 0) {   // profiling instantiation (of the kernel broadcast contacts run)
                if (a1mpc_status st = set_lds_attr(reinterpret_cast<const void*>(&a1mpc_admm_cu_kernel<H, false, true, true, true>), ldsq); st != A1MPC_OK) return st;
                hipLaunchKernelGGL((a1mpc_admm_cu_kernel<H, false, true, true, true>), gridq, blockq, ldsq, stream, a, static_cast<const double*>(prep), counter);
                g_clk_ran = true;
            } else if (a.carry != nullptr) {
                if (a1mpc_status st = set_lds_attr(reinterpret_cast<const void*>(&a1mpc_admm_cu_kernel<H, true, false, false, true>), ldsq); st != A1MPC_OK) return st;
                hipLaunchKernelGGL((a1mpc_admm_cu_kernel<H, true, false, false, true>), gridq, blockq, ldsq, stream, a, static_cast<const double*>(prep), counter);
            } else if (a.contact_stride == 0 && cu_quad_enabled()) {
                if (a1mpc_status st = set_lds_attr(reinterpret_cast<const void*>(&a1mpc_admm_cu_kernel<H, false, true, false, true>), ldsq); st != A1MPC_OK) return st;
                hipLaunchKernelGGL((a1mpc_admm_cu_kernel<H, false, true, false, true>), gridq, blockq, ldsq, stream, a, static_cast<const double*>(prep), counter);
            } else if (a.contact_stride == 0) {
                if (a1mpc_status st = set_lds_attr(reinterpret_cast<const void*>(&a1mpc_admm_cu_kernel<H, false, true>), ldsq); st != A1MPC_OK) return st;
                hipLaunchKernelGGL((a1mpc_admm_cu_kernel<H, false, true>), gridq, blockq, ldsq, stream, a, static_cast<const double*>(prep), counter);
            } else {
                hipLaunchKernelGGL((a1mpc_admm_cu_kernel<H, false, false, false, true>), gridq, blockq, ldsq, stream, a, static_cast<const double*>(prep), counter);
            }
            A1_HIP(hipGetLastError());
            return A1MPC_OK;
        }
    }
    const int want = (a.n + ROWS - 1) / ROWS;
    const dim3 grid(static_cast<unsigned>(want < res ? want : res)), block(admm_twin_rows(H, ROWS) ? 64 : 16 * ROWS);
    if constexpr (kHasUpd) {
        if (upd_kernels) {
            if (a1mpc_status st = set_lds_attr(reinterpret_cast<const void*>(&a1mpc_admm_kernel<H, ROWS, true, false, false, quad_rows(H, ROWS)>), lds2); st != A1MPC_OK) return st;
            hipLaunchKernelGGL((a1mpc_admm_kernel<H, ROWS, true, false, false, quad_rows(H, ROWS)>), grid, block, lds2, stream, a, static_cast<const double*>(prep), counter);
        }
    }
    // profiling instantiation (a1mpc_set_profiling; H > 1, default rows, no update path, broadcast contacts): the kernel of the default batches with clock stamps
    if constexpr (H > 1 && ROWS == default_rows_per_wg(H) && admm_twin_rows(H, ROWS) && cu_wide_qps(H) == 0) {
        if (a.clk != nullptr && !upd_kernels && a.contact_stride == 0) {
            constexpr bool kUni = H >= 16;
            if (a1mpc_status st = set_lds_attr(reinterpret_cast<const void*>(&a1mpc_admm_kernel<H, ROWS, false, kUni, true, quad_rows(H, ROWS)>), lds2); st != A1MPC_OK) return st;
            hipLaunchKernelGGL((a1mpc_admm_kernel<H, ROWS, false, kUni, true, quad_rows(H, ROWS)>), grid, block, lds2, stream, a, static_cast<const double*>(prep), counter);
            A1_HIP(hipGetLastError());
            g_clk_ran = true;
            return A1MPC_OK;
        }
    }
    bool uni_kernel = false;   // broadcast contacts at H >= 16: the instantiation with one pair of bounds for all slots
    constexpr bool kHasUni = H >= 16 && ROWS == default_rows_per_wg(H) && admm_twin_rows(H, ROWS);
    if constexpr (kHasUni) {
        uni_kernel = !upd_kernels && a.contact_stride == 0;
        bool quad = false;
        if constexpr (quad_rows(H, ROWS)) quad = uni_kernel && quad_enabled();
        if constexpr (quad_rows(H, ROWS)) {
            if (quad) {
                if (a1mpc_status st = set_lds_attr(reinterpret_cast<const void*>(&a1mpc_admm_kernel<H, ROWS, false, true, false, true>), lds2); st != A1MPC_OK) return st;
                hipLaunchKernelGGL((a1mpc_admm_kernel<H, ROWS, false, true, false, true>), grid, block, lds2, stream, a, static_cast<const double*>(prep), counter);
            }
        }
        if (uni_kernel && !quad) {
            if (a1mpc_status st = set_lds_attr(reinterpret_cast<const void*>(&a1mpc_admm_kernel<H, ROWS, false, true>), lds2); st != A1MPC_OK) return st;
            hipLaunchKernelGGL((a1mpc_admm_kernel<H, ROWS, false, true>), grid, block, lds2, stream, a, static_cast<const double*>(prep), counter);
        }
    }
    if (!upd_kernels && !uni_kernel) hipLaunchKernelGGL((a1mpc_admm_kernel<H, ROWS, false, false, false, quad_rows(H, ROWS)>), grid, block, lds2, stream, a, static_cast<const double*>(prep), counter);
    A1_HIP(hipGetLastError());
    return A1MPC_OK;
}
template <int H>
a1mpc_status launch_split(const KernelArgs& a, double* prep, int* counter, hipStream_t stream, hipEvent_t mid) {
#ifdef A1MPC_ALL_ROWS
    if (a.carry == nullptr) {   // (the update-path kernels exist for the default rows per workgroup only)
        switch (rows_per_wg(H)) {
            case 1: return launch_split_rows<H, 1>(a, prep, counter, stream, mid);
            case 2: return launch_split_rows<H, 2>(a, prep, counter, stream, mid);
        }
        return launch_split_rows<H, 4>(a, prep, counter, stream, mid);
    }
#endif
    return launch_split_rows<H, default_rows_per_wg(H)>(a, prep, counter, stream, mid);
}

template <int H, int MODE, int ROWS>
static a1mpc_status launch_rows(const KernelArgs& a, hipStream_t stream) {
    static bool attr_set[64] = {};
    int dev = 0;
    A1_HIP(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> lock(g_cache_mu);
        if (dev >= 0 && dev < 64 && !attr_set[dev]) {
            A1_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&a1mpc_solve_kernel<H, MODE, ROWS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes<H>(ROWS))));
            attr_set[dev] = true;
        }
    }
    const unsigned grid = static_cast<unsigned>((a.n + ROWS - 1) / ROWS);
    if constexpr (MODE == kModeMpc && tick_clk_horizon(H) && ROWS == default_rows_per_wg(H)) {
        if (a.clk != nullptr && a.contact_stride == 0) {   // profiling instantiations (a1mpc_set_profiling): stage stamps of the whole tick, both warm-start semantics
            const dim3 blk(twin_rows(H, MODE, ROWS) ? 64 : 16 * ROWS);
            if (a.carry != nullptr) {
                if (a1mpc_status st = set_lds_attr(reinterpret_cast<const void*>(&a1mpc_solve_kernel<H, MODE, ROWS, true, true>), lds_bytes<H>(ROWS)); st != A1MPC_OK) return st;
                hipLaunchKernelGGL((a1mpc_solve_kernel<H, MODE, ROWS, true, true>), dim3(grid), blk, lds_bytes<H>(ROWS), stream, a);
            } else {
                if (a1mpc_status st = set_lds_attr(reinterpret_cast<const void*>(&a1mpc_solve_kernel<H, MODE, ROWS, false, true>), lds_bytes<H>(ROWS)); st != A1MPC_OK) return st;
                hipLaunchKernelGGL((a1mpc_solve_kernel<H, MODE, ROWS, false, true>), dim3(grid), blk, lds_bytes<H>(ROWS), stream, a);
            }
            A1_HIP(hipGetLastError());
            g_clk_ran = true;
            return A1MPC_OK;
        }
    }
    if constexpr (MODE == kModeMpc && H > 1 && ROWS == default_rows_per_wg(H)) {
        if (a.carry != nullptr) {   // warm_start = 2: the update-path instantiation
            if (a1mpc_status st = set_lds_attr(reinterpret_cast<const void*>(&a1mpc_solve_kernel<H, MODE, ROWS, true>), lds_bytes<H>(ROWS)); st != A1MPC_OK) return st;
            hipLaunchKernelGGL((a1mpc_solve_kernel<H, MODE, ROWS, true>), dim3(grid), dim3(twin_rows(H, MODE, ROWS) ? 64 : 16 * ROWS), lds_bytes<H>(ROWS), stream, a);
            A1_HIP(hipGetLastError());
            return A1MPC_OK;
        }
    }
    hipLaunchKernelGGL((a1mpc_solve_kernel<H, MODE, ROWS>), dim3(grid), dim3(twin_rows(H, MODE, ROWS) ? 64 : 16 * ROWS), lds_bytes<H>(ROWS), stream, a);
    A1_HIP(hipGetLastError());
    return A1MPC_OK;
}
template <int H, int ROWS>
a1mpc_status launch_gen_rows(const KernelArgs& a, hipStream_t stream) {
    static bool attr_set[64] = {};
    int dev = 0;
    A1_HIP(hipGetDevice(&dev));
    if constexpr (H % 4 != 0) {   // a handful of QPs at H = 10: one wavefront per QP, its four rows share the set-up (a1mpc_solve_gen_coop_kernel)
        if (a.n <= coop_max_batch() && coop_setup_enabled()) {
            const size_t lds1 = sizeof(double) * Layout<H, true>::ROW_STRIDE;
            if (a.carry != nullptr) {
                if (a1mpc_status st = set_lds_attr(reinterpret_cast<const void*>(&a1mpc_solve_gen_coop_kernel<H, true>), lds1); st != A1MPC_OK) return st;
                hipLaunchKernelGGL((a1mpc_solve_gen_coop_kernel<H, true>), dim3(static_cast<unsigned>(a.n)), dim3(64), lds1, stream, a);
            } else {
                if (a1mpc_status st = set_lds_attr(reinterpret_cast<const void*>(&a1mpc_solve_gen_coop_kernel<H, false>), lds1); st != A1MPC_OK) return st;
                hipLaunchKernelGGL((a1mpc_solve_gen_coop_kernel<H, false>), dim3(static_cast<unsigned>(a.n)), dim3(64), lds1, stream, a);
            }
            A1_HIP(hipGetLastError());
            return A1MPC_OK;
        }
    }
    const size_t lds = sizeof(double) * ROWS * Layout<H, true>::ROW_STRIDE;
    {
        std::lock_guard<std::mutex> lock(g_cache_mu);
        if (dev >= 0 && dev < 64 && !attr_set[dev]) {
            A1_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&a1mpc_solve_gen_kernel<H, ROWS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds)));
            attr_set[dev] = true;
        }
    }
    if (a.carry != nullptr) {   // warm_start = 2: the update-path instantiation
        if (a1mpc_status st = set_lds_attr(reinterpret_cast<const void*>(&a1mpc_solve_gen_kernel<H, ROWS, true>), lds); st != A1MPC_OK) return st;
        hipLaunchKernelGGL((a1mpc_solve_gen_kernel<H, ROWS, true>), dim3(static_cast<unsigned>((a.n + ROWS - 1) / ROWS)), dim3(64), lds, stream, a);
        A1_HIP(hipGetLastError());
        return A1MPC_OK;
    }
    hipLaunchKernelGGL((a1mpc_solve_gen_kernel<H, ROWS>), dim3(static_cast<unsigned>((a.n + ROWS - 1) / ROWS)), dim3(64), lds, stream, a);
    A1_HIP(hipGetLastError());
    return A1MPC_OK;
}
// resident workgroups of the general path's ADMM kernel (occupancy query, cached per device)
template <int H, int ROWS>
a1mpc_status resident_workgroups_gen(int* out) {
    static int resident[64] = {};
    int dev = 0;
    A1_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(A1MPC_ERR_HIP, "device index out of range");
    std::lock_guard<std::mutex> lock(g_cache_mu);
    if (!resident[dev]) {
        const size_t lds = sizeof(double) * ROWS * Layout<H, true>::ROW_STRIDE;
        A1_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&a1mpc_admm_gen_kernel<H, ROWS>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        int per_cu = 0, cus = 0;
        A1_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&a1mpc_admm_gen_kernel<H, ROWS>), 64, lds));
        A1_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        resident[dev] = (per_cu > 0 ? per_cu : 1) * (cus > 0 ? cus : 1);
    }
    *out = resident[dev];
    return A1MPC_OK;
}
// a batch beyond the resident rows: the general path's set-up kernel, the queue-order kernel, its persistent ADMM kernel
template <int H, int ROWS>
a1mpc_status launch_gen_split_rows(const KernelArgs& a, double* prep, int* counter, hipStream_t stream, hipEvent_t mid, int res) {
    static bool attr_set[64] = {};
    int dev = 0;
    A1_HIP(hipGetDevice(&dev));
    constexpr int QPW = 4 / setup_gen_rows(H);   // QPs per wavefront of the set-up kernel
    const size_t lds1 = sizeof(double) * (QPW * LayoutSetup<H, true>::ROW_STRIDE + 2 * H * H), lds2 = sizeof(double) * ROWS * Layout<H, true>::ROW_STRIDE;
    {
        std::lock_guard<std::mutex> lock(g_cache_mu);
        if (dev >= 0 && dev < 64 && !attr_set[dev]) {
            A1_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&a1mpc_setup_gen_kernel<H>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds1)));
            attr_set[dev] = true;
        }
    }
    A1_HIP(hipMemsetAsync(counter, 0, sizeof(int), stream));
    hipLaunchKernelGGL((a1mpc_setup_gen_kernel<H>), dim3(static_cast<unsigned>((a.n + QPW - 1) / QPW)), dim3(64), lds1, stream, a, prep);
    A1_HIP(hipGetLastError());
    if (a.cost != nullptr && a.order != nullptr) {   // (predicted or the previous solve's costs: see launch_split_rows)
        launch_order_kernel(static_cast<int>(a.n), static_cast<const int32_t*>(a.cost), const_cast<int32_t*>(a.order), stream);
        A1_HIP(hipGetLastError());
    }
    if (mid) A1_HIP(hipEventRecord(mid, stream));
    if constexpr (cu_wide_gen_qps(H) > 0) {
        if (gen_cu_wide_enabled() && !g_gen_prefer_one_wave) {   // seven QPs per CU: one 256-thread workgroup per CU (a1mpc_admm_gen_cu_kernel); a slot of a two-slot pipeline keeps the one-wave workgroups
            constexpr int Q = cu_wide_gen_qps(H);
            const size_t ldsq = sizeof(double) * Q * Layout<H, true>::ROW_STRIDE;
            static int resq_dev[64] = {};
            int resq = 0;
            {
                std::lock_guard<std::mutex> lock(g_cache_mu);
                if (dev >= 0 && dev < 64 && !resq_dev[dev]) {
                    A1_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&a1mpc_admm_gen_cu_kernel<H>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(ldsq)));
                    int per_cu = 0, cus = 0;
                    A1_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&a1mpc_admm_gen_cu_kernel<H>), 256, ldsq));
                    A1_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
                    resq_dev[dev] = (per_cu > 0 ? per_cu : 1) * (cus > 0 ? cus : 1);
                }
                resq = (dev >= 0 && dev < 64) ? resq_dev[dev] : 256;
            }
            const int wantq = (a.n + Q - 1) / Q;
            hipLaunchKernelGGL((a1mpc_admm_gen_cu_kernel<H>), dim3(static_cast<unsigned>(wantq < resq ? wantq : resq)), dim3(256), ldsq, stream, a, static_cast<const double*>(prep), counter);
            A1_HIP(hipGetLastError());
            return A1MPC_OK;
        }
    }
    const int want = (a.n + ROWS - 1) / ROWS;
    hipLaunchKernelGGL((a1mpc_admm_gen_kernel<H, ROWS>), dim3(static_cast<unsigned>(want < res ? want : res)), dim3(64), lds2, stream, a, static_cast<const double*>(prep), counter);
    A1_HIP(hipGetLastError());
    return A1MPC_OK;
}
template <int H>
static a1mpc_status launch_coop(const KernelArgs& a, hipStream_t stream) {
    static bool attr_set[64] = {};
    int dev = 0;
    A1_HIP(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> lock(g_cache_mu);
        if (dev >= 0 && dev < 64 && !attr_set[dev]) {
            A1_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&a1mpc_solve_coop_kernel<H>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds_bytes<H>(1))));
            attr_set[dev] = true;
        }
    }
    if constexpr (tick_clk_horizon(H)) {
        if (a.clk != nullptr && a.contact_stride == 0) {   // profiling instantiations (a1mpc_set_profiling)
            if (a.carry != nullptr) {
                if (a1mpc_status st = set_lds_attr(reinterpret_cast<const void*>(&a1mpc_solve_coop_kernel<H, true, true>), lds_bytes<H>(1)); st != A1MPC_OK) return st;
                hipLaunchKernelGGL((a1mpc_solve_coop_kernel<H, true, true>), dim3(static_cast<unsigned>(a.n)), dim3(64), lds_bytes<H>(1), stream, a);
            } else {
                if (a1mpc_status st = set_lds_attr(reinterpret_cast<const void*>(&a1mpc_solve_coop_kernel<H, false, true>), lds_bytes<H>(1)); st != A1MPC_OK) return st;
                hipLaunchKernelGGL((a1mpc_solve_coop_kernel<H, false, true>), dim3(static_cast<unsigned>(a.n)), dim3(64), lds_bytes<H>(1), stream, a);
            }
            A1_HIP(hipGetLastError());
            g_clk_ran = true;
            return A1MPC_OK;
        }
    }
    if (a.carry != nullptr) {   // warm_start = 2: the update-path instantiation
        if (a1mpc_status st = set_lds_attr(reinterpret_cast<const void*>(&a1mpc_solve_coop_kernel<H, true>), lds_bytes<H>(1)); st != A1MPC_OK) return st;
        hipLaunchKernelGGL((a1mpc_solve_coop_kernel<H, true>), dim3(static_cast<unsigned>(a.n)), dim3(64), lds_bytes<H>(1), stream, a);
        A1_HIP(hipGetLastError());
        return A1MPC_OK;
    }
    hipLaunchKernelGGL((a1mpc_solve_coop_kernel<H>), dim3(static_cast<unsigned>(a.n)), dim3(64), lds_bytes<H>(1), stream, a);
    A1_HIP(hipGetLastError());
    return A1MPC_OK;
}
template <int H, int MODE>
a1mpc_status launch(const KernelArgs& a, hipStream_t stream) {
    if constexpr (MODE == kModeMpc && H > 1) {
        if (coop_setup_enabled() && a.n <= coop_max_batch()) return launch_coop<H>(a, stream);
    }
#ifdef A1MPC_ALL_ROWS
    if (a.carry == nullptr) {   // (the update-path kernels exist for the default rows per workgroup only)
        switch (rows_per_wg(H)) {
            case 1: return launch_rows<H, MODE, 1>(a, stream);
            case 2: return launch_rows<H, MODE, 2>(a, stream);
        }
        return launch_rows<H, MODE, 4>(a, stream);
    }
#endif
    return launch_rows<H, MODE, default_rows_per_wg(H)>(a, stream);
}

}  // namespace a1mpc
