// Micro-benchmark (round 6): what a cross-wavefront exchange costs inside ONE workgroup of four wavefronts (one per SIMD of a CU) -- the price of
// splitting the batch-1 latency kernel's set-up (Ruiz block columns, the 12 columns of a Riccati factor step) over the CU's four SIMDs:
//   (a) s_barrier alone;  (b) ds_write_b64 -> s_barrier -> ds_read_b64 x k (k = 1, 3, 12: a pivot column / a wave's three columns / a full row), the reads feeding a dependent FMA
//       as the Gauss-Jordan pivot / the column maxima would;  (c) the same data flow inside one wavefront (no barrier: lgkmcnt wait only) for reference.
// Cycles per round by s_memtime around ROUNDS rounds, lone workgroup on the chip (the batch-1 regime).   hipcc --offload-arch=gfx950 -O3 wg_exchange_ubench.hip -o wg_exchange_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#define ROUNDS 4096
template <int K, bool BARRIER>
__global__ __launch_bounds__(256) void k(double* out, long long* cyc, double seed) {
    __shared__ double sm[4 * 64 * 2];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double acc = seed + tid, x = seed * 0.5;
    sm[tid] = acc; sm[256 + tid] = acc;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < ROUNDS; ++r) {
        double* buf = sm + (r & 1) * 256;                     // ping-pong: one barrier per round is enough
        buf[wave * 64 + lane] = acc;                           // my wave's contribution
        if (BARRIER) __builtin_amdgcn_s_barrier(); else __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < K; ++j) v += buf[((wave + 1 + (BARRIER ? j : 0) * 0 + j) & 3) * 64 * (BARRIER ? 1 : 0) + ((lane + j) & 63) + (BARRIER ? 0 : wave * 64)];
        acc = fma(acc, x, v);                                  // the next round depends on what was read
    }
    const long long t1 = __builtin_readcyclecounter();
    out[tid] = acc;
    if (lane == 0) { cyc[2 * wave] = t0; cyc[2 * wave + 1] = t1; }
}
__global__ __launch_bounds__(256) void kbar(double* out, long long* cyc, double seed) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double acc = seed + tid;
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < ROUNDS; ++r) { __builtin_amdgcn_s_barrier(); acc = fma(acc, 0.5, 1.0); }
    const long long t1 = __builtin_readcyclecounter();
    out[tid] = acc;
    if (lane == 0) { cyc[2 * wave] = t0; cyc[2 * wave + 1] = t1; }
}
template <class F>
static double run(const char* name, F launch, double* d_out, long long* d_cyc, int waves) {
    launch(); hipDeviceSynchronize(); launch(); hipDeviceSynchronize();
    long long cc[8]; hipMemcpy(cc, d_cyc, sizeof cc, hipMemcpyDeviceToHost);
    long long lo = cc[0], hi = cc[1];
    for (int w = 1; w < waves; ++w) { if (cc[2 * w] < lo) lo = cc[2 * w]; if (cc[2 * w + 1] > hi) hi = cc[2 * w + 1]; }
    const double per = (double)(hi - lo) / ROUNDS;
    printf("  %-78s %7.1f cycles per round\n", name, per);
    return per;
}
int main() {
    double* d_out; long long* d_cyc;
    hipMalloc(&d_out, 256 * sizeof(double)); hipMalloc(&d_cyc, 8 * sizeof(long long));
    printf("one workgroup, four wavefronts (one per SIMD), nothing else on the chip; cycles = s_memtime / shader clock counter\n");
    run("s_barrier + one dependent FMA", [&] { hipLaunchKernelGGL(kbar, dim3(1), dim3(256), 0, 0, d_out, d_cyc, 1.0); }, d_out, d_cyc, 4);
    run("ds_write -> s_barrier -> 1 x ds_read of another wave's word -> FMA", [&] { hipLaunchKernelGGL((k<1, true>), dim3(1), dim3(256), 0, 0, d_out, d_cyc, 1.0); }, d_out, d_cyc, 4);
    run("ds_write -> s_barrier -> 3 x ds_read -> FMA", [&] { hipLaunchKernelGGL((k<3, true>), dim3(1), dim3(256), 0, 0, d_out, d_cyc, 1.0); }, d_out, d_cyc, 4);
    run("ds_write -> s_barrier -> 12 x ds_read -> FMA", [&] { hipLaunchKernelGGL((k<12, true>), dim3(1), dim3(256), 0, 0, d_out, d_cyc, 1.0); }, d_out, d_cyc, 4);
    run("one wavefront: ds_write -> lgkmcnt(0) -> 1 x ds_read of its own row -> FMA", [&] { hipLaunchKernelGGL((k<1, false>), dim3(1), dim3(64), 0, 0, d_out, d_cyc, 1.0); }, d_out, d_cyc, 1);
    run("one wavefront: ds_write -> lgkmcnt(0) -> 12 x ds_read -> FMA", [&] { hipLaunchKernelGGL((k<12, false>), dim3(1), dim3(64), 0, 0, d_out, d_cyc, 1.0); }, d_out, d_cyc, 1);
    return 0;
}
