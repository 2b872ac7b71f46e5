#!/usr/bin/env python3
"""Stand-alone timing harness of the EKF kernels (N4c): cuts their section out of csrc/a1mpc_hip.hip, appends a driver and builds it with hipcc in seconds.
usage: ekf_bench.py build [extra hipcc flags]  ->  tools/ubench/ekf_bench  (GPU box: tools/ubench/ekf_bench [robots [ticks]]; prints the kernel time and a checksum of the
state after the last tick -- equal checksums = the same bits)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.environ.get("EKF_SRC") or os.path.join(ROOT, "a1-qp-mpc-controller_amd", "csrc", "a1mpc_hip.hip")).read()   # (EKF_SRC: another revision of the source, for A/B)
sec = s[s.index("// ---- N4c: A1BasicEKF"):s.index("a1mpc_status a1mpc_reset_ekf_state")]
DRV = r'''
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
#include <cmath>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
@SECTION@
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 65536, ticks = argc > 2 ? atoi(argv[2]) : 8;
    std::mt19937_64 rng(1); std::uniform_real_distribution<double> U(0, 1); std::normal_distribution<double> G(0, 1);
    const size_t N = n;
    std::vector<double> ff(4 * N), R(9 * N), acc(3 * N), w(3 * N), fk(12 * N), fv(12 * N);
    std::vector<uint8_t> mode(N, 1);
    const double nom[12] = {0.18, 0.13, -0.3, 0.18, -0.13, -0.3, -0.18, 0.13, -0.3, -0.18, -0.13, -0.3};
    for (size_t b = 0; b < N; ++b) { const double y = 6 * U(rng) - 3; const double c = cos(y), s_ = sin(y); const double r[9] = {c, -s_, 0, s_, c, 0, 0, 0, 1}; for (int k = 0; k < 9; ++k) R[9 * b + k] = r[k]; }
    double *d_st, *d_ff, *d_R, *d_acc, *d_w, *d_fk, *d_fv, *d_pos, *d_vel; uint8_t *d_mode, *d_ec;
    CK(hipMalloc(&d_st, N * kEkfState * 8)); CK(hipMemset(d_st, 0, N * kEkfState * 8));
    CK(hipMalloc(&d_ff, 4 * N * 8)); CK(hipMalloc(&d_R, 9 * N * 8)); CK(hipMalloc(&d_acc, 3 * N * 8)); CK(hipMalloc(&d_w, 3 * N * 8)); CK(hipMalloc(&d_fk, 12 * N * 8)); CK(hipMalloc(&d_fv, 12 * N * 8));
    CK(hipMalloc(&d_pos, 3 * N * 8)); CK(hipMalloc(&d_vel, 3 * N * 8)); CK(hipMalloc(&d_mode, N)); CK(hipMalloc(&d_ec, 4 * N));
    CK(hipMemcpy(d_R, R.data(), 9 * N * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_mode, mode.data(), N, hipMemcpyHostToDevice));
    EkfArgs a; a.n = n; a.dt = 0.0025; a.flat = 1; a.state = d_st; a.mode = d_mode; a.ff = d_ff; a.R = d_R; a.acc = d_acc; a.w = d_w; a.fk = d_fk; a.fv = d_fv; a.pos_out = d_pos; a.vel_out = d_vel; a.ec_out = d_ec;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int t = 0; t < ticks; ++t) {
        for (auto& v : ff) v = 150 * U(rng);
        for (size_t k = 0; k < acc.size(); ++k) acc[k] = G(rng) + (k % 3 == 2 ? 9.81 : 0.0);
        for (auto& v : w) v = 0.2 * G(rng);
        for (size_t k = 0; k < fk.size(); ++k) { fk[k] = nom[k % 12] + 0.01 * G(rng); fv[k] = 0.2 * G(rng); }
        CK(hipMemcpy(d_ff, ff.data(), 4 * N * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_acc, acc.data(), 3 * N * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_w, w.data(), 3 * N * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_fk, fk.data(), 12 * N * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_fv, fv.data(), 12 * N * 8, hipMemcpyHostToDevice));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(a1mpc_ekf_init_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, a);
        EKF_LAUNCH(a);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float m; CK(hipEventElapsedTime(&m, e0, e1)); ms.push_back(m);
    }
    std::vector<double> st(N * kEkfState); CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    double chk = 0; for (double v : st) chk += v;
    std::sort(ms.begin() + 2, ms.end());
    printf("{\"robots\": %d, \"kernel_ms\": %.5f, \"checksum\": %.17g}\n", n, ms[2 + (ms.size() - 2) / 2], chk);
    return 0;
}
'''
launch = '#define EKF_LAUNCH(a) hipLaunchKernelGGL(a1mpc_ekf_kernel, dim3((a.n + 1) / 2), dim3(64), 0, 0, a)\n'
if "EKF_LAUNCH" in sec: launch = ""
src = DRV.replace("@SECTION@", launch + sec)
if os.environ.get("EKF_WAVES"):   # A/B of the residency: waves per SIMD the kernel is compiled for
    sec = sec.replace("__launch_bounds__(64, 3) void a1mpc_ekf_kernel", "__launch_bounds__(64, %d) void a1mpc_ekf_kernel" % int(os.environ["EKF_WAVES"]))
    src = DRV.replace("@SECTION@", launch + sec)
tag = os.environ.get("EKF_TAG", "")
out = os.path.join(ROOT, "tools", "ubench", "ekf_bench" + ("_" + tag if tag else ""))
cpp = "/tmp/ekf_bench.hip"; open(cpp, "w").write(src)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", cpp, "-o", out] + sys.argv[2:], check=True)
print("built", out)
