// tests/cpp/latency_harness.cpp -- batch-1 tick latency of the C ABI from C++ (no Python in the loop): BASELINE configs[1], SURVEY 8(d) config 2
// = trot, horizon 10, 10 000 sequential warm-started ticks (S/A1Params.h:10: a tick every 2.5 ms; S/MainGazebo.cpp:57-68 is the caller).
// Host pointers in and out (PCIe and launch included), one caller thread.  Prints one JSON object.
//   latency_harness [ticks=10000] [pace_us=0] [warm_mode=1] [horizon=10] [timing_events=0]      pace_us > 0 sleeps between ticks like the reference's thread 1 does; horizon 10 / 16 / 20
//                   [entry=0]   1 = the same ticks as 22-number tick records through a1mpc_solve_batch_ticks -- the entry the drop-in's compute_grf calls (include/a1mpc_dropin.hpp)
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

#include "a1mpc.h"

static void rot_zyx(double r, double p, double y, double* R) {
    const double cr = cos(r), sr = sin(r), cp = cos(p), sp = sin(p), cy = cos(y), sy = sin(y);
    const double M[9] = {cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr, sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr, -sp, cp * sr, cp * cr};
    for (int i = 0; i < 9; ++i) R[i] = M[i];
}

int main(int argc, char** argv) {
    const int ticks = argc > 1 ? atoi(argv[1]) : 10000, pace_us = argc > 2 ? atoi(argv[2]) : 0;
    const int warm_mode = argc > 3 ? atoi(argv[3]) : 1;   // 1 = fresh set-up + warm start, 2 = the reference's update path (a1mpc.h)
    const int H = argc > 4 ? atoi(argv[4]) : 10;          // PLAN_HORIZON (S/A1Params.h:26 fixes 10; BASELINE configs[3] / [4]: 16 / 20)
    if (H != 10 && H != 16 && H != 20) { std::fprintf(stderr, "horizon must be 10, 16 or 20\n"); return 1; }
    const int entry = argc > 6 ? atoi(argv[6]) : 0;       // 0 = a1mpc_solve_batch (x0 / x_ref), 1 = a1mpc_solve_batch_ticks (the tick record x0 / x_ref are built from on the device)
    const int timing = argc > 5 ? atoi(argv[5]) : 0;      // the handle's HIP timing events (a1mpc_set_timing): off by default here -- a control loop does not read them
    a1mpc_config cfg;
    a1mpc_default_config(&cfg);
    const double q[13] = {20, 10, 1, 0, 0, 420, .05, .05, .05, 30, 30, 10, 0};   // config/gazebo_a1_mpc.yaml:40-72
    for (int i = 0; i < 13; ++i) cfg.q[i] = q[i];
    for (int i = 0; i < 12; ++i) cfg.r[i] = 1e-7;
    cfg.mass = 12.0; cfg.inertia_body[0] = 0.0158533; cfg.inertia_body[4] = 0.0377999; cfg.inertia_body[8] = 0.0456542;
    cfg.horizon = H; cfg.warm_start = warm_mode;
    a1mpc_handle h = nullptr;
    if (a1mpc_create(&cfg, 1, 0, &h) != A1MPC_OK) { std::fprintf(stderr, "a1mpc_create: %s\n", a1mpc_last_error()); return 2; }
    a1mpc_set_timing(h, timing);
    std::mt19937_64 rng(0xA1 + 2);
    std::normal_distribution<double> N01(0.0, 1.0);
    const double nominal[12] = {0.17, 0.15, -0.35, 0.17, -0.15, -0.35, -0.17, 0.15, -0.35, -0.17, -0.15, -0.35};
    std::vector<double> lat(ticks);
    std::vector<int> iters(ticks);
    int bad_status = 0;
    for (int t = 0; t < ticks; ++t) {
        double x0[13], xref[13 * 20], R[9], foot[12], grf[12];
        uint8_t contact[4];
        const double e[3] = {0.02 * N01(rng), 0.02 * N01(rng), 0.02 * N01(rng)};
        rot_zyx(e[0], e[1], e[2], R);
        x0[0] = e[0]; x0[1] = e[1]; x0[2] = e[2]; x0[3] = 0; x0[4] = 0; x0[5] = 0.3 + 0.01 * N01(rng);
        for (int i = 0; i < 3; ++i) { x0[6 + i] = 0.1 * N01(rng); x0[9 + i] = 0.05 * N01(rng); }
        x0[9] += 0.3; x0[12] = -9.8;
        const double vw[3] = {R[0] * 0.3, R[3] * 0.3, R[6] * 0.3};
        for (int i = 0; i < H; ++i) {
            double* xr = xref + 13 * i; const double k = cfg.dt * (i + 1);
            xr[0] = 0; xr[1] = 0; xr[2] = x0[2]; xr[3] = x0[3] + vw[0] * k; xr[4] = x0[4] + vw[1] * k; xr[5] = 0.3; xr[6] = xr[7] = xr[8] = 0;
            xr[9] = vw[0]; xr[10] = vw[1]; xr[11] = 0; xr[12] = -9.8;
        }
        for (int l = 0; l < 4; ++l) for (int i = 0; i < 3; ++i) foot[3 * l + i] = R[i * 3 + 0] * nominal[3 * l] + R[i * 3 + 1] * nominal[3 * l + 1] + R[i * 3 + 2] * nominal[3 * l + 2];
        const bool ph = (t / 60) % 2 == 0;
        contact[0] = ph; contact[1] = !ph; contact[2] = !ph; contact[3] = ph;
        int32_t it = 0, st = 0;
        const auto a = std::chrono::steady_clock::now();
        // the tick record of the same tick: [euler, pos, ang_vel, lin_vel | euler_d, lin_vel_d (body), ang_vel_d, pos_z_d] (include/a1mpc.h)
        const double tick[22] = {x0[0], x0[1], x0[2], x0[3], x0[4], x0[5], x0[6], x0[7], x0[8], x0[9], x0[10], x0[11], 0, 0, 0, 0.3, 0, 0, 0, 0, 0, 0.3};
        const a1mpc_status rc = entry == 1 ? a1mpc_solve_batch_ticks(h, 1, tick, R, foot, contact, grf, nullptr, &it, &st)
                                           : a1mpc_solve_batch(h, 1, x0, xref, R, foot, contact, grf, nullptr, &it, &st);
        const auto b = std::chrono::steady_clock::now();
        if (rc != A1MPC_OK) { std::fprintf(stderr, "a1mpc_solve_batch: %s\n", a1mpc_last_error()); return 3; }
        lat[t] = std::chrono::duration<double, std::milli>(b - a).count();
        iters[t] = it; bad_status += st != A1MPC_QP_SOLVED;
        if (pace_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(pace_us));
    }
    a1mpc_destroy(h);
    const int skip = std::min(50, ticks / 10);
    std::vector<double> v(lat.begin() + skip, lat.end());
    std::vector<double> s = v;
    std::sort(s.begin(), s.end());
    auto pct = [&](double p) { return s[std::min(s.size() - 1, static_cast<size_t>(p * s.size()))]; };
    int over = 0, worst_t = skip;
    for (int t = skip; t < ticks; ++t) { over += lat[t] > 2.5; if (lat[t] > lat[worst_t]) worst_t = t; }
    double mean_it = 0; for (int t = skip; t < ticks; ++t) mean_it += iters[t]; mean_it /= (ticks - skip);
    std::printf("{\"warm_start\": %d, \"workload\": \"config2 trot, h=%d, batch 1, warm start, host pointers in/out (%s), C++ caller, %s\", \"horizon\": %d, \"timing_events\": %d, \"ticks\": %zu, \"p50_ms\": %.4f, \"p99_ms\": %.4f, "
                "\"p999_ms\": %.4f, \"max_ms\": %.4f, \"ticks_over_2p5_ms\": %d, \"worst_tick_index\": %d, \"worst_tick_iters\": %d, \"mean_iters\": %.1f, \"not_solved\": %d}\n",
                warm_mode, H, entry == 1 ? "tick records: a1mpc_solve_batch_ticks" : "x0 / x_ref: a1mpc_solve_batch", pace_us > 0 ? "paced" : "back to back", H, timing, v.size(), pct(0.50), pct(0.99), pct(0.999), s.back(), over, worst_t, iters[worst_t], mean_it, bad_status);
    return 0;
}
