// completion_wake_ubench.hip -- how much later than the kernel's own last store does hipStreamSynchronize() return?  (round 6: would a batch-1 tick gain from a completion
// flag the kernel writes into the caller's pinned block, polled by the host, instead of the stream synchronisation?)
// A one-wavefront kernel spins for ~100 us, then stores a sequence number into pinned host memory behind a system-scope fence.  Two hosts loops over 2000 launches each:
//   A  launch -> hipStreamSynchronize                               -> t_sync
//   B  launch -> poll the flag (then synchronise, untimed)          -> t_flag
// build: hipcc --offload-arch=gfx950 -O3 completion_wake_ubench.hip -o completion_wake_ubench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin_then_flag(volatile uint64_t* flag, uint64_t seq, long long cycles, double* sink) {
    const long long t0 = clock64();
    double x = 1.0;
    while (clock64() - t0 < cycles) x = x * 1.0000001 + 1e-9;
    if (threadIdx.x == 0) sink[0] = x;
    __threadfence_system();
    if (threadIdx.x == 0) *flag = seq;
}
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    uint64_t* flag; CK(hipHostMalloc(&flag, 64, hipHostMallocDefault)); *flag = 0;
    uint64_t* dflag; CK(hipHostGetDevicePointer(reinterpret_cast<void**>(&dflag), flag, 0));
    double* sink; CK(hipMalloc(&sink, 8));
    const long long cycles = 10000;   // clock64 ticks at 100 MHz: ~100 us
    using clk = std::chrono::steady_clock;
    auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    std::vector<double> ta, tb;
    uint64_t seq = 0;
    for (int rep = 0; rep < 2200; ++rep) {
        ++seq;
        auto t0 = clk::now();
        hipLaunchKernelGGL(spin_then_flag, dim3(1), dim3(64), 0, s, dflag, seq, cycles, sink);
        CK(hipStreamSynchronize(s));
        auto t1 = clk::now();
        if (rep >= 200) ta.push_back(us(t0, t1));
        ++seq;
        t0 = clk::now();
        hipLaunchKernelGGL(spin_then_flag, dim3(1), dim3(64), 0, s, dflag, seq, cycles, sink);
        while (*reinterpret_cast<volatile uint64_t*>(flag) != seq) { __builtin_ia32_pause(); }
        t1 = clk::now();
        CK(hipStreamSynchronize(s));
        if (rep >= 200) tb.push_back(us(t0, t1));
    }
    std::sort(ta.begin(), ta.end()); std::sort(tb.begin(), tb.end());
    printf("{\"launch_to_sync_us_p50\": %.2f, \"launch_to_sync_us_p99\": %.2f, \"launch_to_flag_us_p50\": %.2f, \"launch_to_flag_us_p99\": %.2f, \"gain_p50_us\": %.2f}\n",
           ta[ta.size() / 2], ta[ta.size() * 99 / 100], tb[tb.size() / 2], tb[tb.size() * 99 / 100], ta[ta.size() / 2] - tb[tb.size() / 2]);
    return 0;
}
