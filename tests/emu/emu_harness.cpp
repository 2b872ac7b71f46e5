// tests/emu/emu_harness.cpp -- TEST INFRASTRUCTURE: runs csrc/a1mpc_solver.hpp on host fibers.
// Built by tests/emu/build.py into tests/emu/liba1mpc_emu.so and used only by tests/test_emu_*.py to
// validate the solver's algorithm, lane mapping and LDS synchronisation against the oracle without a GPU.
#include <ucontext.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <a1mpc_rowops.hpp>  // the emulation flavour: -I tests/emu
#include "a1mpc_solver.hpp"
#include "a1mpc_tables.hpp"

namespace a1mpc {

thread_local int emu_lane = 0;
thread_local int emu_lanes = 16;  // lanes of the emulated (part of a) wavefront: 16 = one row, 32 = a main / twin pair, 64 = a quad of rows in the device's lane order

namespace {
constexpr int kLanes = 64;  // a DPP row, a main / twin pair of rows (lanes 16-31 = the twin: RowSolver<.., TWIN>), or a quad (the wavefront: rows 2 / 3 = the twins, RowSolver<.., QUAD>)
constexpr size_t kStack = 1 << 20;
struct Sched {
    int lanes;
    bool retire_ok;  // rows may leave early (the latency kernel's rows 1 and 3 after the shared set-up): a finished lane counts as arrived
    ucontext_t main_ctx;
    ucontext_t ctx[kLanes];
    char* stacks[kLanes];
    bool finished[kLanes];
    double buf[2][kLanes], xbuf[2][kLanes], qbuf[2][kLanes];  // row-local exchanges / exchanges between the rows of a pair / between the even and odd row of each half of a quad
    long gen[kLanes], xgen[kLanes], qgen[kLanes];            // how many of each this lane has published
    void (*fn)(void*);
    void* arg;
};
thread_local Sched* g_s = nullptr;

void trampoline() {
    Sched* s = g_s;
    const int l = emu_lane;
    s->fn(s->arg);
    s->finished[l] = true;
    swapcontext(&s->ctx[l], &s->main_ctx);
}
// yield until every lane lo..hi-1 has published exchange number g (counters cnt)
void wait_for(Sched* s, int l, const long* cnt, long g, int lo, int hi) {
    for (;;) {
        bool all = true;
        for (int x = lo; x < hi; ++x) {
            if (s->finished[x] && s->retire_ok) continue;
            if (s->finished[x] && cnt[x] < g) { fprintf(stderr, "emu: divergent control flow (lane %d finished while lane %d waits for it)\n", x, l); abort(); }
            all = all && cnt[x] >= g;
        }
        if (all) return;
        swapcontext(&s->ctx[l], &s->main_ctx);
        emu_lane = l;
    }
}
}  // namespace

// Row-local exchange: the caller resumes when the 16 lanes of ITS row have published (the rows of a pair may be at different points --
// e.g. the twin skips the set-up); returns the row's 16 values.
const double* emu_publish(double v) {
    Sched* s = g_s;
    const int l = emu_lane, lo = l & ~15;
    const long g = ++s->gen[l];
    s->buf[g & 1][l] = v;
    wait_for(s, l, s->gen, g, lo, lo + 16);
    return s->buf[g & 1] + lo;
}
// a = [x | y] on (main | twin)  ->  a = [x | x], returns [y | y]   (v_permlane32_swap on the device): waits for the partner lane's k-th exchange
double emu_twin_exchange(double& a) {
    Sched* s = g_s;
    const int l = emu_lane;
    const long g = ++s->xgen[l];
    const double mine = a;
    s->xbuf[g & 1][l] = a;
    const int half = s->lanes == 64 ? 32 : 16;   // (a quad runs in the device's lane order: the twins are rows 2 and 3)
    wait_for(s, l, s->xgen, g, 0, s->lanes);
    const double other = s->xbuf[g & 1][l ^ half];
    if (!(l & half)) return other;
    a = other;
    return mine;
}
// a = [x | y] on the (even | odd) row of each half of a quad  ->  a = x on both, returns y   (v_permlane16_swap on the device)
double emu_quad_exchange(double& a) {
    Sched* s = g_s;
    const int l = emu_lane;
    if (s->lanes != 64) { fprintf(stderr, "emu: quad_exchange outside a quad of rows\n"); abort(); }
    const long g = ++s->qgen[l];
    const double mine = a;
    s->qbuf[g & 1][l] = a;
    wait_for(s, l, s->qgen, g, 0, 64);
    const double other = s->qbuf[g & 1][l ^ 16];
    if (!(l & 16)) return other;
    a = other;
    return mine;
}

static void run_row(void (*fn)(void*), void* arg, int lanes = 16, bool retire_ok = false) {
    Sched s;
    memset(&s, 0, sizeof s);
    s.lanes = lanes;
    s.retire_ok = retire_ok;
    emu_lanes = lanes;
    s.fn = fn;
    s.arg = arg;
    g_s = &s;
    for (int l = 0; l < lanes; ++l) {
        s.stacks[l] = static_cast<char*>(malloc(kStack));
        getcontext(&s.ctx[l]);
        s.ctx[l].uc_stack.ss_sp = s.stacks[l];
        s.ctx[l].uc_stack.ss_size = kStack;
        s.ctx[l].uc_link = &s.main_ctx;
        makecontext(&s.ctx[l], trampoline, 0);
    }
    for (;;) {
        int alive = 0;
        for (int l = 0; l < lanes; ++l) {
            if (s.finished[l]) continue;
            emu_lane = l;
            swapcontext(&s.main_ctx, &s.ctx[l]);
            if (!s.finished[l]) ++alive;
        }
        if (!alive) break;
    }
    for (int r = 0; r < lanes; r += 16)   // (rows that retire early stop counting together: still equal within a row)
        for (int l = r + 1; l < r + 16; ++l)
            if (s.gen[l] != s.gen[r]) { fprintf(stderr, "emu: the lanes of a row disagree on the number of cross-lane ops\n"); abort(); }
    for (int l = 0; l < lanes; ++l) free(s.stacks[l]);
    g_s = nullptr;
}

template <int H>
struct Job {
    const DeviceParams* P;
    const double* tab;
    ProblemIO io;
    double* lds;
};
template <int H>
static void job_entry(void* a) {
    Job<H>* j = static_cast<Job<H>*>(a);
    if constexpr (H > 1) { if (j->io.carry) { solve_row<H, kModeMpc, false, true>(*j->P, j->tab, j->io, j->lds); return; } }   // (like the host: the update-path instantiation for warm_start = 2)
    solve_row<H>(*j->P, j->tab, j->io, j->lds);
}

template <int H>
static void job_twin_entry(void* a) {  // the fused kernel on a main / twin pair of rows
    Job<H>* j = static_cast<Job<H>*>(a);
    if constexpr (H > 1 && H % 2 == 0) {
        if (j->io.carry) solve_row_with<H, kModeMpc, false, true, true>(*j->P, j->tab, [&]() -> const ProblemIO& { return j->io; }, j->lds);
        else solve_row_with<H, kModeMpc, false, true>(*j->P, j->tab, [&]() -> const ProblemIO& { return j->io; }, j->lds);
    }
}
template <int H>
static void job_quad_entry(void* a) {  // the fused kernel on a quad of rows (one QP per wavefront, H a multiple of 4)
    Job<H>* j = static_cast<Job<H>*>(a);
    if constexpr (H > 1 && H % 4 == 0) {
        if (j->io.carry) solve_row_with<H, kModeMpc, false, true, true, true>(*j->P, j->tab, [&]() -> const ProblemIO& { return j->io; }, j->lds);
        else solve_row_with<H, kModeMpc, false, true, false, true>(*j->P, j->tab, [&]() -> const ProblemIO& { return j->io; }, j->lds);
    }
}
// the latency kernel (a1mpc_solve_coop_kernel, csrc/a1mpc_hip.hip) statement for statement: the four rows of a wavefront share ONE QP's set-up (each takes every fourth horizon
// step of the Ruiz sweeps), row 0 leaves the hand-off record in the factor region, then rows 0 / 2 solve as a main / twin pair while rows 1 / 3 retire -- or, at a horizon
// that is a multiple of 4, all four go on as a quad
template <int H>
static void latency_entry(void* a) {
    Job<H>* j = static_cast<Job<H>*>(a);
    if constexpr (H > 1 && H % 2 == 0) {
        const int row = emu_lane >> 4;
        const bool upd = j->io.carry != nullptr;
        {
            RowSolver<H, kModeMpc> S(*j->P, stage_table<H, Layout<H>>(j->tab, j->lds, row, 4), j->lds);
            S.coop_id = row; S.coop_n = 4;
            if (upd) S.template setup<true>(j->io); else S.template setup<false>(j->io);
            coop_sync();   // (the device's row_sync() orders the whole wavefront)
            if (row == 0) { if (upd) S.template save_prepared<true>(j->lds + Layout<H>::FAC); else S.template save_prepared<false>(j->lds + Layout<H>::FAC); }
        }
        constexpr bool kQuad = H % 4 == 0;
        if constexpr (!kQuad) { if (row & 1) return; }
        RowSolver<H, kModeMpc, false, false, true, false, false, kQuad> S(*j->P, j->tab, j->lds);
        if (upd) { S.template load_prepared<true>(j->lds + Layout<H>::FAC, j->io); S.template solve<true>(); S.write_outputs(j->io, j->io.carry); }
        else { S.template load_prepared<false>(j->lds + Layout<H>::FAC, j->io); S.template solve<false>(); S.write_outputs(j->io); }
    }
}
static int g_emu_twin = 0;  // a1mpc_emu_set_twin(): the fused entry points run main / twin pairs (1), quads of rows (2; H a multiple of 4) or the latency kernel (3)
static double* g_emu_carry = nullptr;  // a1mpc_emu_set_carry(): n x Carry<H>::STRIDE doubles of the update path (warm_start = 2), or null
static int g_emu_contact_stride = 0;  // a1mpc_emu_set_contact_stride(): 4 = `contact` is an n x 4H per-step schedule (fast path, feet step-invariant)
template <int H>
static void run_batch(const DeviceParams* P, int n, const double* x0, const double* xref, const double* R, const double* foot,
                      const uint8_t* contact, double* grf, double* u_full, double* warm_x, double* warm_y, double* rho,
                      int32_t* iters, int32_t* status, int32_t* nfact) {
    std::vector<double> tab(2 * H * H);
    fill_gamma_beta_table(H, tab.data());
    std::vector<double> lds(Layout<H>::ROW_STRIDE);
    for (int b = 0; b < n; ++b) {
        for (auto& v : lds) v = NAN;  // uninitialised LDS must never be consumed
        Job<H> j;
        memset(&j.io, 0, sizeof j.io);
        j.P = P;
        j.tab = tab.data();
        j.lds = lds.data();
        j.io.x0 = x0 + (size_t)b * 13;
        j.io.xref = xref + (size_t)b * 13 * H;
        j.io.R = R + (size_t)b * 9;
        j.io.foot = foot + (size_t)b * 12;
        j.io.contact = contact + (size_t)b * (g_emu_contact_stride ? 4 * H : 4);
        j.io.contact_stride = g_emu_contact_stride;
        j.io.grf = grf + (size_t)b * 12;
        j.io.u_full = u_full ? u_full + (size_t)b * 12 * H : nullptr;
        j.io.warm_x = warm_x ? warm_x + (size_t)b * 12 * H : nullptr;
        j.io.warm_y = warm_y ? warm_y + (size_t)b * 20 * H : nullptr;
        j.io.rho_io = rho ? rho + b : nullptr;
        j.io.iters = iters ? iters + b : nullptr;
        j.io.status = status ? status + b : nullptr;
        j.io.nfact = nfact ? nfact + b : nullptr;
        j.io.carry = g_emu_carry ? g_emu_carry + (size_t)b * Carry<H>::STRIDE : nullptr;
        if (g_emu_twin == 3 && H > 1 && H % 2 == 0) run_row(latency_entry<H>, &j, 64, true);
        else if (g_emu_twin == 2 && H > 1 && H % 4 == 0) run_row(job_quad_entry<H>, &j, 64);
        else if (g_emu_twin && H > 1 && H % 2 == 0) run_row(job_twin_entry<H>, &j, 32);
        else run_row(job_entry<H>, &j);
    }
}

}  // namespace a1mpc
extern "C" void a1mpc_emu_set_twin(int on) { a1mpc::g_emu_twin = on; }
extern "C" void a1mpc_emu_set_contact_stride(int stride) { a1mpc::g_emu_contact_stride = stride; }
extern "C" void a1mpc_emu_set_carry(double* carry) { a1mpc::g_emu_carry = carry; }
extern "C" int a1mpc_emu_carry_stride(int horizon) {
    switch (horizon) { case 10: return a1mpc::Carry<10>::STRIDE; case 16: return a1mpc::Carry<16>::STRIDE; case 20: return a1mpc::Carry<20>::STRIDE; case 4: return a1mpc::Carry<4>::STRIDE; case 12: return a1mpc::Carry<12>::STRIDE; }
    return 0;
}

namespace a1mpc {
template <int H>
struct SplitJob { const BatchArgs* a; double* prep; int* counter; double* lds; int64_t b; };
// (like the host: the update-path instantiations of the two kernels when a carry is given -- warm_start = 2 --, the plain ones otherwise)
template <int H>
static void split_setup_entry(void* p) {
    auto* j = static_cast<SplitJob<H>*>(p);
    if (j->a->carry) setup_row<H, false, true>(*j->a, j->a->tab, j->b, j->lds, j->prep); else setup_row<H>(*j->a, j->a->tab, j->b, j->lds, j->prep);
}
template <int H>
static void split_admm_entry(void* p) {
    auto* j = static_cast<SplitJob<H>*>(p);
    if (j->a->carry) admm_rows<H, false, false, true>(*j->a, j->prep, j->counter, j->lds); else admm_rows<H>(*j->a, j->prep, j->counter, j->lds);
}
template <int H>
static void split_admm_twin_entry(void* p) {
    auto* j = static_cast<SplitJob<H>*>(p);
    if (j->a->carry) admm_rows<H, true, false, true>(*j->a, j->prep, j->counter, j->lds); else admm_rows<H, true>(*j->a, j->prep, j->counter, j->lds);
}
template <int H>
static void split_admm_quad_entry(void* p) {   // (the device's instantiations: broadcast contacts run UNI, a schedule or the update path the plain one)
    auto* j = static_cast<SplitJob<H>*>(p);
    if constexpr (H > 1 && H % 4 == 0) {
        if (j->a->carry) admm_rows<H, true, false, true, false, false, true>(*j->a, j->prep, j->counter, j->lds);
        else if (j->a->contact_stride == 0) admm_rows<H, true, false, false, true, false, true>(*j->a, j->prep, j->counter, j->lds);
        else admm_rows<H, true, false, false, false, false, true>(*j->a, j->prep, j->counter, j->lds);
    }
}
// the split pipeline on host fibers: K1 for every QP, then `nrows` persistent rows draining the queue one after another
template <int H>
static void run_split(const BatchArgs& a, int nrows, bool twin = false) {
    std::vector<double> tab(2 * H * H);
    fill_gamma_beta_table(H, tab.data());
    BatchArgs aa = a; aa.tab = tab.data();
    std::vector<double> prep((size_t)a.n * Prep<H>::STRIDE, NAN);
    std::vector<double> lds1(LayoutSetup<H>::ROW_STRIDE), lds2(Layout<H>::ROW_STRIDE);
    int counter = 0;
    SplitJob<H> j{&aa, prep.data(), &counter, nullptr, 0};
    for (int64_t b = 0; b < a.n; ++b) {
        for (auto& v : lds1) v = NAN;
        j.b = b; j.lds = lds1.data();
        run_row(split_setup_entry<H>, &j);
    }
    for (int r = 0; r < nrows; ++r) {
        for (auto& v : lds2) v = NAN;
        j.lds = lds2.data();
        if constexpr (H > 1 && H % 4 == 0) { if (twin && g_emu_twin == 2) { run_row(split_admm_quad_entry<H>, &j, 64); continue; } }
        if constexpr (H > 1) { if (twin) { run_row(split_admm_twin_entry<H>, &j, 32); continue; } }
        run_row(split_admm_entry<H>, &j);
    }
}
}  // namespace a1mpc
extern "C" int a1mpc_emu_solve_split(const a1mpc::DeviceParams* P, int horizon, int n, int nrows, const double* x0, const double* xref,
                                     const double* R, const double* foot, const uint8_t* contact, double* grf, double* u_full,
                                     double* warm_x, double* warm_y, double* rho, int32_t* iters, int32_t* status, int32_t* nfact) {
    a1mpc::BatchArgs a;
    memset(&a, 0, sizeof a);
    a.P = *P; a.n = n; a.x0 = x0; a.xref = xref; a.R = R; a.foot = foot; a.contact = contact; a.grf = grf; a.u_full = u_full;
    a.warm_x = warm_x; a.warm_y = warm_y; a.rho = rho; a.iters = iters; a.status = status; a.nfact = nfact;
    a.contact_stride = a1mpc::g_emu_contact_stride; a.carry = a1mpc::g_emu_carry;
    const bool twin = nrows < 0;  // nrows < 0: -nrows persistent main / twin PAIRS of rows (the device's persistent kernel)
    if (twin) nrows = -nrows;
    switch (horizon) {
        case 1: a1mpc::run_split<1>(a, nrows); return 0;
        case 10: a1mpc::run_split<10>(a, nrows, twin); return 0;
        case 16: a1mpc::run_split<16>(a, nrows, twin); return 0;
        case 20: a1mpc::run_split<20>(a, nrows, twin); return 0;
        case 4: a1mpc::run_split<4>(a, nrows, twin); return 0;     // two of the extended horizons (csrc/a1mpc_common.hpp, A1MPC_FAST_HORIZONS): the shortest, and one beyond 10
        case 12: a1mpc::run_split<12>(a, nrows, twin); return 0;
    }
    return -1;
}
namespace a1mpc {
template <int H>
static void split_setup_gen_entry(void* p) { auto* j = static_cast<SplitJob<H>*>(p); setup_row<H, true>(*j->a, j->a->tab, j->b, j->lds, j->prep); }
template <int H>
static void split_admm_gen_entry(void* p) { auto* j = static_cast<SplitJob<H>*>(p); admm_rows<H, true, true>(*j->a, j->prep, j->counter, j->lds); }
template <int H>
static void split_admm_gen_quad_entry(void* p) {
    auto* j = static_cast<SplitJob<H>*>(p);
    if constexpr (H % 4 == 0) admm_rows<H, true, true, false, false, false, true>(*j->a, j->prep, j->counter, j->lds);
}
// the general path's split pipeline: its set-up kernel for every QP, then `nrows` persistent main / twin pairs
template <int H>
static void run_split_gen(const BatchArgs& a, int nrows) {
    std::vector<double> tab(2 * H * H);
    fill_gamma_beta_table(H, tab.data());
    BatchArgs aa = a; aa.tab = tab.data();
    std::vector<double> prep((size_t)a.n * Prep<H>::STRIDE_GEN, NAN);
    std::vector<double> lds1(LayoutSetup<H, true>::ROW_STRIDE), lds2(Layout<H, true>::ROW_STRIDE);
    int counter = 0;
    SplitJob<H> j{&aa, prep.data(), &counter, nullptr, 0};
    for (int64_t b = 0; b < a.n; ++b) {
        for (auto& v : lds1) v = NAN;
        j.b = b; j.lds = lds1.data();
        run_row(split_setup_gen_entry<H>, &j, 16 * setup_gen_rows(H));   // (the general path's set-up kernel: two or four rows per QP, setup_gen_rows)
    }
    for (int r = 0; r < nrows; ++r) {
        for (auto& v : lds2) v = NAN;
        j.lds = lds2.data();
        if (g_emu_twin == 2 && H % 4 == 0) run_row(split_admm_gen_quad_entry<H>, &j, 64);
        else run_row(split_admm_gen_entry<H>, &j, 32);
    }
}
}  // namespace a1mpc
extern "C" int a1mpc_emu_solve_gen_split(const a1mpc::DeviceParams* P, int horizon, int n, int nrows, const double* x0, const double* xref, const double* R,
                                         const double* foot, int foot_stride, const uint8_t* contact, int contact_stride, double* grf, double* u_full,
                                         int32_t* iters, int32_t* status, int32_t* nfact) {
    a1mpc::BatchArgs a;
    memset(&a, 0, sizeof a);
    a.P = *P; a.n = n; a.x0 = x0; a.xref = xref; a.R = R; a.foot = foot; a.contact = contact; a.grf = grf; a.u_full = u_full;
    a.iters = iters; a.status = status; a.nfact = nfact; a.foot_stride = foot_stride; a.contact_stride = contact_stride;
    switch (horizon) {
        case 10: a1mpc::run_split_gen<10>(a, nrows); return 0;
        case 16: a1mpc::run_split_gen<16>(a, nrows); return 0;
        case 20: a1mpc::run_split_gen<20>(a, nrows); return 0;
    }
    return -1;
}
extern "C" int a1mpc_emu_solve_ticks(const a1mpc::DeviceParams* P, int horizon, int n, const double* tick, const double* R,
                                     const double* foot, const uint8_t* contact, double* grf, double* u_full, int32_t* iters, int32_t* status) {
    using namespace a1mpc;
    if (horizon != 10) return -1;
    std::vector<double> tab(2 * 10 * 10);
    fill_gamma_beta_table(10, tab.data());
    std::vector<double> lds(Layout<10>::ROW_STRIDE);
    for (int b = 0; b < n; ++b) {
        for (auto& v : lds) v = NAN;
        Job<10> j;
        memset(&j.io, 0, sizeof j.io);
        j.P = P; j.tab = tab.data(); j.lds = lds.data();
        j.io.tick = tick + (size_t)b * 22; j.io.R = R + (size_t)b * 9; j.io.foot = foot + (size_t)b * 12; j.io.contact = contact + (size_t)b * 4;
        j.io.grf = grf + (size_t)b * 12; j.io.u_full = u_full + (size_t)b * 120; j.io.iters = iters + b; j.io.status = status + b;
        run_row(job_entry<10>, &j);
    }
    return 0;
}
extern "C" int a1mpc_emu_solve(const a1mpc::DeviceParams* P, int horizon, int n, const double* x0, const double* xref, const double* R,
                               const double* foot, const uint8_t* contact, double* grf, double* u_full, double* warm_x,
                               double* warm_y, double* rho, int32_t* iters, int32_t* status, int32_t* nfact) {
    switch (horizon) {
        case 1: a1mpc::run_batch<1>(P, n, x0, xref, R, foot, contact, grf, u_full, warm_x, warm_y, rho, iters, status, nfact); return 0;
        case 4: a1mpc::run_batch<4>(P, n, x0, xref, R, foot, contact, grf, u_full, warm_x, warm_y, rho, iters, status, nfact); return 0;
        case 12: a1mpc::run_batch<12>(P, n, x0, xref, R, foot, contact, grf, u_full, warm_x, warm_y, rho, iters, status, nfact); return 0;
        case 10: a1mpc::run_batch<10>(P, n, x0, xref, R, foot, contact, grf, u_full, warm_x, warm_y, rho, iters, status, nfact); return 0;
        case 16: a1mpc::run_batch<16>(P, n, x0, xref, R, foot, contact, grf, u_full, warm_x, warm_y, rho, iters, status, nfact); return 0;
        case 20: a1mpc::run_batch<20>(P, n, x0, xref, R, foot, contact, grf, u_full, warm_x, warm_y, rho, iters, status, nfact); return 0;
    }
    return -1;
}
namespace a1mpc {
static void balance_entry(void* a) {
    Job<1>* j = static_cast<Job<1>*>(a);
    solve_row<1, kModeBalance>(*j->P, j->tab, j->io, j->lds);
}
}  // namespace a1mpc
extern "C" int a1mpc_emu_balance(const a1mpc::DeviceParams* P, int n, const double* root_acc, const double* R, const double* Rz,
                                 const double* foot, const uint8_t* contact, double* grf, double* f_world, int32_t* iters,
                                 int32_t* status) {
    using namespace a1mpc;
    double tab[2];
    fill_gamma_beta_table(1, tab);
    std::vector<double> lds(Layout<1>::ROW_STRIDE);
    for (int b = 0; b < n; ++b) {
        for (auto& v : lds) v = NAN;
        Job<1> j;
        memset(&j.io, 0, sizeof j.io);
        j.P = P; j.tab = tab; j.lds = lds.data();
        j.io.root_acc = root_acc + (size_t)b * 6; j.io.Rz = Rz + (size_t)b * 9; j.io.R = R + (size_t)b * 9;
        j.io.foot = foot + (size_t)b * 12; j.io.contact = contact + (size_t)b * 4;
        j.io.grf = grf + (size_t)b * 12; j.io.u_full = f_world ? f_world + (size_t)b * 12 : nullptr;
        j.io.iters = iters ? iters + b : nullptr; j.io.status = status ? status + b : nullptr;
        run_row(balance_entry, &j);
    }
    return 0;
}
namespace a1mpc {
template <int H>
static void gen_entry(void* a) {
    Job<H>* j = static_cast<Job<H>*>(a);
    solve_row<H, kModeMpc, true>(*j->P, j->tab, j->io, j->lds);
}
template <int H>
static void gen_twin_entry(void* a) {
    Job<H>* j = static_cast<Job<H>*>(a);
    if constexpr (H % 2 == 0) {
        if (j->io.carry) solve_row_with<H, kModeMpc, true, true, true>(*j->P, j->tab, [&]() -> const ProblemIO& { return j->io; }, j->lds);   // (like the host: the update-path instantiation for warm_start = 2)
        else solve_row_with<H, kModeMpc, true, true>(*j->P, j->tab, [&]() -> const ProblemIO& { return j->io; }, j->lds);
    }
}
template <int H>
static void gen_quad_entry(void* a) {
    Job<H>* j = static_cast<Job<H>*>(a);
    if constexpr (H % 4 == 0) {
        if (j->io.carry) solve_row_with<H, kModeMpc, true, true, true, true>(*j->P, j->tab, [&]() -> const ProblemIO& { return j->io; }, j->lds);
        else solve_row_with<H, kModeMpc, true, true, false, true>(*j->P, j->tab, [&]() -> const ProblemIO& { return j->io; }, j->lds);
    }
}
// the general path's latency kernel (a1mpc_solve_gen_coop_kernel): the device function itself, on 64 fibers whose rows 1 / 3 retire behind the shared set-up
template <int H>
static void gen_latency_entry(void* a) {
    Job<H>* j = static_cast<Job<H>*>(a);
    if constexpr (H % 2 == 0 && H % 4 != 0 && H >= 10) {
        if (j->io.carry) solve_latency_gen<H, true>(*j->P, j->tab, [&]() -> const ProblemIO& { return j->io; }, j->lds);
        else solve_latency_gen<H, false>(*j->P, j->tab, [&]() -> const ProblemIO& { return j->io; }, j->lds);
    }
}
template <int H>
static void run_gen(const DeviceParams* P, int n, const double* x0, const double* xref, const double* R, const double* foot, int foot_stride,
                    const uint8_t* contact, int contact_stride, double* grf, double* u_full, double* warm_x, double* warm_y, double* rho,
                    int32_t* iters, int32_t* status, int32_t* nfact) {
    std::vector<double> tab(2 * H * H);
    fill_gamma_beta_table(H, tab.data());
    std::vector<double> lds(Layout<H, true>::ROW_STRIDE);
    for (int b = 0; b < n; ++b) {
        for (auto& v : lds) v = NAN;
        Job<H> j;
        memset(&j.io, 0, sizeof j.io);
        j.P = P; j.tab = tab.data(); j.lds = lds.data();
        j.io.x0 = x0 + (size_t)b * 13; j.io.xref = xref + (size_t)b * 13 * H; j.io.R = R + (size_t)b * 9;
        j.io.foot = foot + (size_t)b * (foot_stride ? 12 * H : 12); j.io.contact = contact + (size_t)b * (contact_stride ? 4 * H : 4);
        j.io.foot_stride = foot_stride; j.io.contact_stride = contact_stride;
        j.io.grf = grf + (size_t)b * 12; j.io.u_full = u_full ? u_full + (size_t)b * 12 * H : nullptr;
        j.io.warm_x = warm_x ? warm_x + (size_t)b * 12 * H : nullptr; j.io.warm_y = warm_y ? warm_y + (size_t)b * 20 * H : nullptr;
        j.io.rho_io = rho ? rho + b : nullptr;
        j.io.iters = iters ? iters + b : nullptr; j.io.status = status ? status + b : nullptr; j.io.nfact = nfact ? nfact + b : nullptr;
        if constexpr (H >= 10) j.io.carry = g_emu_carry ? g_emu_carry + (size_t)b * Carry<H>::STRIDE : nullptr;   // update path (a1mpc_emu_set_carry); twin / quad runs only
        if (g_emu_twin == 3 && H % 2 == 0 && H % 4 != 0 && H >= 10) run_row(gen_latency_entry<H>, &j, 64, true);
        else if (g_emu_twin == 2 && H % 4 == 0) run_row(gen_quad_entry<H>, &j, 64);
        else if (g_emu_twin && H % 2 == 0) run_row(gen_twin_entry<H>, &j, 32);
        else run_row(gen_entry<H>, &j);
    }
}
}  // namespace a1mpc
// the general path (per-step feet / per-step contact schedules, RowSolver<..., GEN = true>)
extern "C" int a1mpc_emu_solve_gen(const a1mpc::DeviceParams* P, int horizon, int n, const double* x0, const double* xref, const double* R,
                                   const double* foot, int foot_stride, const uint8_t* contact, int contact_stride, double* grf, double* u_full,
                                   double* warm_x, double* warm_y, double* rho, int32_t* iters, int32_t* status, int32_t* nfact) {
    switch (horizon) {
        case 4: a1mpc::run_gen<4>(P, n, x0, xref, R, foot, foot_stride, contact, contact_stride, grf, u_full, warm_x, warm_y, rho, iters, status, nfact); return 0;
        case 10: a1mpc::run_gen<10>(P, n, x0, xref, R, foot, foot_stride, contact, contact_stride, grf, u_full, warm_x, warm_y, rho, iters, status, nfact); return 0;
        case 16: a1mpc::run_gen<16>(P, n, x0, xref, R, foot, foot_stride, contact, contact_stride, grf, u_full, warm_x, warm_y, rho, iters, status, nfact); return 0;
        case 20: a1mpc::run_gen<20>(P, n, x0, xref, R, foot, foot_stride, contact, contact_stride, grf, u_full, warm_x, warm_y, rho, iters, status, nfact); return 0;
    }
    return -1;
}
extern "C" int a1mpc_emu_sizeof_params(void) { return (int)sizeof(a1mpc::DeviceParams); }
