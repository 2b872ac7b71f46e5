// Micro-benchmark 3: does a wavefront with only 32 (16) live lanes issue FP64 / FP64-DPP instructions faster than a full one, alone on its SIMD or beside a
// second wave?  (Question behind it: one QP = a main / twin pair of DPP rows = 32 lanes per wavefront, two such wavefronts per SIMD, instead of two QPs per
// wavefront and one wavefront per SIMD.)  Prints shader cycles per instruction and wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
#define ITER 512
template <int DPP>
__global__ __launch_bounds__(256) void k(double* out, long long* cyc, double seed, int live) {
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
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { cyc[2 * (threadIdx.x >> 6)] = t0; cyc[2 * (threadIdx.x >> 6) + 1] = t1; }
}
int main() {
    double* d_out; long long* d_cyc; long long h[16];
    hipMalloc(&d_out, 256 * 8); hipMalloc(&d_cyc, 16 * 8);
    for (int dpp = 0; dpp < 2; ++dpp)
        for (int waves : {1, 4, 8})            // waves in ONE workgroup on one CU: 1 = alone, 4 = one per SIMD, 8 = two per SIMD
            for (int live : {64, 32, 16}) {
                for (int rep = 0; rep < 2; ++rep) {
                    if (dpp) hipLaunchKernelGGL((k<1>), dim3(1), dim3(64 * waves > 256 ? 256 : 64 * waves), 0, 0, d_out, d_cyc, 1.0, live);
                    else hipLaunchKernelGGL((k<0>), dim3(1), dim3(64 * waves > 256 ? 256 : 64 * waves), 0, 0, d_out, d_cyc, 1.0, live);
                    hipDeviceSynchronize();
                }
                if (waves == 8) continue;  // (two per SIMD needs two workgroups of four: below)
                hipMemcpy(h, d_cyc, 16 * 8, hipMemcpyDeviceToHost);
                printf("%s waves/CU %d live %2d: %.2f cycles per instruction and wave\n", dpp ? "fmac_dpp" : "fmac    ", waves, live, double(h[1] - h[0]) / (double(REP) * ITER));
            }
    // two waves per SIMD: 2 workgroups x 4 waves on one CU is not controllable from here; use 8 waves in a 512-thread workgroup instead
    return 0;
}
