#!/usr/bin/env python3
"""Stand-alone timing harness of the N2b kernel: cuts the kernel's section out of csrc/a1mpc_hip.hip (like tests/emu/n2b_host.py), appends a driver and builds it
with hipcc in seconds -- kernel experiments without the three-minute library build.  usage: n2b_bench.py build [extra hipcc flags]  ->  tools/ubench/n2b_bench (run on the GPU box:
tools/ubench/n2b_bench [robots [ticks]])"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.environ.get("N2B_SRC") or os.path.join(ROOT, "a1-qp-mpc-controller_amd", "csrc", "a1mpc_hip.hip")).read()   # (N2B_SRC: another revision of the source, for A/B)
sec = s[s.index("// ---- N2b: contact logic"):s.index("// the handle's contact state: records")]
DRV = r'''
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
@SECTION@
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 65536, ticks = argc > 2 ? atoi(argv[2]) : 80;
    std::mt19937_64 rng(1); std::uniform_real_distribution<double> U(0, 1); std::normal_distribution<double> G(0, 0.2);
    std::vector<double> gc(4 * (size_t)n), ff(4 * (size_t)n), fp(12 * (size_t)n), z(n, 0.3);
    std::vector<uint8_t> plan(4 * (size_t)n);
    for (auto& v : gc) v = 240 * U(rng);
    char* st; CK(hipMalloc(&st, (size_t)n * kCtBytesPerRobot)); CK(hipMemset(st, 0, (size_t)n * kCtBytesPerRobot));
    double *d_gc, *d_ff, *d_fp, *d_z, *d_pd, *d_rec, *d_ta; uint8_t *d_pc, *d_ct;
    CK(hipMalloc(&d_gc, gc.size() * 8)); CK(hipMalloc(&d_ff, ff.size() * 8)); CK(hipMalloc(&d_fp, fp.size() * 8)); CK(hipMalloc(&d_z, n * 8)); CK(hipMalloc(&d_pd, n * 8));
    CK(hipMalloc(&d_rec, 12 * (size_t)n * 8)); CK(hipMalloc(&d_ta, n * 8)); CK(hipMalloc(&d_pc, 4 * (size_t)n)); CK(hipMalloc(&d_ct, 4 * (size_t)n));
    CK(hipMemset(d_pd, 0, n * 8)); CK(hipMemcpy(d_z, z.data(), n * 8, hipMemcpyHostToDevice));
    ContactArgs a; a.n = n; a.counter_per_swing = 120; a.foot_force_low = 30; a.use_terrain_adapt = 1;
    a.rec = reinterpret_cast<CtRecord*>(st); a.leg_ring = reinterpret_cast<double*>(a.rec + n); a.terrain_ring = a.leg_ring + (size_t)n * kCtLegRing; a.stride = n;
    a.gait_counter = d_gc; a.foot_force = d_ff; a.foot_pos_abs = d_fp; a.root_pos_z = d_z; a.plan_contacts = d_pc; a.pitch_d = d_pd; a.contacts = d_ct; a.recent_out = d_rec; a.terrain_out = d_ta; a.recent_in = nullptr; a.z_stride = 1; a.pitch_stride = 1; a.pk_tick = nullptr;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int t = 0; t < ticks; ++t) {
        for (size_t k = 0; k < gc.size(); ++k) { gc[k] += 2.0; if (gc[k] >= 240) gc[k] -= 240; plan[k] = gc[k] <= 120; }
        if (t < 3 || t % 16 == 0) { for (auto& v : ff) v = 80 * U(rng); for (auto& v : fp) v = G(rng); }
        CK(hipMemcpy(d_gc, gc.data(), gc.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_ff, ff.data(), ff.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_fp, fp.data(), fp.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_pc, plan.data(), plan.size(), hipMemcpyHostToDevice));
        CK(hipEventRecord(e0, 0));
        N2B_LAUNCH(a);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float m; CK(hipEventElapsedTime(&m, e0, e1)); ms.push_back(m);
    }
    std::vector<uint8_t> ct(4 * (size_t)n); CK(hipMemcpy(ct.data(), d_ct, ct.size(), hipMemcpyDeviceToHost));
    double legs = 0; for (auto c : ct) legs += c; legs /= n;
    std::vector<double> ta(n); CK(hipMemcpy(ta.data(), d_ta, n * 8, hipMemcpyDeviceToHost));
    double chk = 0; for (double v : ta) chk += v;
    std::vector<float> tail(ms.end() - 9, ms.end()); std::sort(tail.begin(), tail.end());
    const double b = 8 * 22 + 4 + 8 * 14 + 4 + 384 + legs * (64 + 32 + 24) + 120 + 16;
    printf("{\"robots\": %d, \"kernel_ms\": %.5f, \"legs\": %.3f, \"bytes_per_robot\": %.1f, \"GB_per_s\": %.1f, \"checksum\": %.17g}\n", n, tail[4], legs, b, n * b / (tail[4] * 1e-3) / 1e9, chk);
    return 0;
}
'''
launch = '#define N2B_LAUNCH(a) hipLaunchKernelGGL(a1mpc_contact_terrain_kernel, dim3((a.n + 63) / 64), dim3(64), 0, 0, a)\n'
if "N2B_LAUNCH" in sec: launch = ""
variant = os.environ.get("N2B_VARIANT", "")
def sub(old, new):
    global sec
    assert old in sec, old
    sec = sec.replace(old, new)
for v in variant.split("+"):
    if v == "nofit":   # memory-only time: the plane fit's pseudo-inverse replaced by a copy
        sub("sym3_pinv(M, P3);", "for (int k = 0; k < 9; ++k) P3[k] = M[k];")
    if v == "noring":  # no ring traffic (wrong results): what the scattered sectors cost
        sub("slot[i][k] = full ? sl[k] : 0.0;", "slot[i][k] = 0.0;")
        sub("const double tr_old = (standing && tcount >= kTerrainWindow) ? *tr : 0.0;", "const double tr_old = 0.0;")
        sub("sl[k] = fp[3 * i + k];", "")
        sub("if (tcount >= kTerrainWindow) f.add(-tr_old); else st->rb.count = tcount + 1;", "st->rb.count = tcount + 1;")
        sub("*tr = v;", "")
    if v == "norecout":
        sub("if (a.recent_out) {", "if (false) {")
    if v == "stageonly":  # records in and out only
        sub("if (b < a.n) contact_terrain_robot_in(a, b, reinterpret_cast<CtRecord*>(img + lane * kCtLdsStride), in);", "if (b < a.n) a.terrain_out[b] = reinterpret_cast<CtRecord*>(img + lane * kCtLdsStride)->rb.sum;")
src = DRV.replace('@SECTION@', launch + sec)
out = os.path.join(ROOT, "tools", "ubench", "n2b_bench" + ("_" + variant if variant else "") + ("_" + os.environ["N2B_TAG"] if os.environ.get("N2B_TAG") else ""))
cpp = "/tmp/n2b_bench.hip"; open(cpp, "w").write(src)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", cpp, "-o", out] + sys.argv[2:], check=True)
print("built", out)
