// as half_wave_ubench.hip with 512-thread workgroups: eight waves of one workgroup share a CU = two per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
#define ITER 512
template <int DPP>
__global__ __launch_bounds__(512) void k(double* out, long long* cyc, double seed, int live) {
    double a[4];
    for (int i = 0; i < 4; ++i) a[i] = seed + i;
    double m = seed * 0.5, x = seed * 0.25 + threadIdx.x;
    if ((threadIdx.x & 63) >= live) return;
    long long t0 = clock64();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            const int c = r % 4;
            if (DPP) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[c]) : "v"(x), "v"(m));
            else asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[c]) : "v"(m), "v"(x));
        }
    }
    long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < 4; ++i) s += a[i];
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { cyc[2 * (threadIdx.x >> 6)] = t0; cyc[2 * (threadIdx.x >> 6) + 1] = t1; }
}
int main() {
    double* d_out; long long* d_cyc; long long h[16];
    hipMalloc(&d_out, 512 * 8); hipMalloc(&d_cyc, 16 * 8);
    for (int dpp = 0; dpp < 2; ++dpp)
        for (int live : {64, 32, 16}) {
            for (int rep = 0; rep < 2; ++rep) {
                if (dpp) hipLaunchKernelGGL((k<1>), dim3(1), dim3(512), 0, 0, d_out, d_cyc, 1.0, live);
                else hipLaunchKernelGGL((k<0>), dim3(1), dim3(512), 0, 0, d_out, d_cyc, 1.0, live);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, d_cyc, 16 * 8, hipMemcpyDeviceToHost);
            long long lo = h[0], hi = h[1];
            for (int w = 0; w < 8; ++w) { if (h[2 * w] < lo) lo = h[2 * w]; if (h[2 * w + 1] > hi) hi = h[2 * w + 1]; }
            printf("%s 8 waves/CU (2 per SIMD) live %2d: wave 0 %.2f cycles per instruction; all eight done after %.2f cycles per instruction of one wave\n", dpp ? "fmac_dpp" : "fmac    ", live,
                   double(h[1] - h[0]) / (double(REP) * ITER), double(hi - lo) / (double(REP) * ITER));
        }
    return 0;
}
