// Micro-benchmark (round 6): issue rate of the FP64 VALU instructions the general path's Ruiz sweeps are made of -- v_fma_f64 (VOP3), v_fmac_f64 (VOP2), v_mul_f64, v_max_f64 --
// as independent chains, one / two / four wavefronts per SIMD, every SIMD of the chip busy.  Timed with HIP events; reported as ns per instruction per SIMD
// and relative to v_fmac_f64.   hipcc --offload-arch=gfx950 -O3 f64_rate_ubench.hip -o f64_rate_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 256
#define ITER 2000
template <int MODE>
__global__ __launch_bounds__(64) void k(double* out, double seed) {
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x;
    double m = seed * 0.5, x = seed * 0.25 + threadIdx.x;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            const int c = r % 8;
            if (MODE == 0) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[c]) : "v"(m), "v"(x));
            if (MODE == 1) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[c]) : "v"(m), "v"(x));
            if (MODE == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[c]) : "v"(m));
            if (MODE == 3) asm volatile("v_max_f64 %0, %0, |%1|" : "+v"(a[c]) : "v"(x));
            if (MODE == 4) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[c]) : "v"(x));
            if (MODE == 5) asm volatile("v_fma_f64 %0, %1, -%2, %0" : "+v"(a[c]) : "v"(m), "v"(x));
            if (MODE == 6) { if (r & 1) asm volatile("v_max_f64 %0, %0, |%1|" : "+v"(a[c]) : "v"(x)); else asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[c]) : "v"(m), "v"(x)); }
            if (MODE == 7) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[c]) : "v"(m), "v"(x));
            if (MODE == 8) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(*(float*)&a[c]) : "v"(*(float*)&m), "v"(*(float*)&x));
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int MODE>
static void run(const char* name, double* d_out, int waves_per_simd, double& base) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 1024 * waves_per_simd;   // 256 CUs x 4 SIMDs
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(64), 0, 0, d_out, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(64), 0, 0, d_out, 1.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / ((double)ITER * REP * waves_per_simd);
    if (base == 0) base = ns;
    printf("  %-34s %d wave(s)/SIMD: %6.3f ns per instruction per SIMD  (x %.2f of v_fmac_f64; %.2f cycles at 2.4 GHz)\n", name, waves_per_simd, ns, ns / base, ns * 2.4);
}
int main() {
    double* d_out; hipMalloc(&d_out, 4096 * 64 * sizeof(double));
    for (int w : {1, 2, 4}) {
        double base = 0;
        run<0>("v_fmac_f64_e32", d_out, w, base); run<1>("v_fma_f64", d_out, w, base); run<5>("v_fma_f64 (neg modifier)", d_out, w, base); run<2>("v_mul_f64", d_out, w, base);
        run<3>("v_max_f64 |x|", d_out, w, base); run<4>("v_add_f64", d_out, w, base); run<6>("v_fma_f64 / v_max_f64 alternating", d_out, w, base);
        run<7>("v_pk_fma_f32", d_out, w, base); run<8>("v_fmac_f32_e32", d_out, w, base);
    }
    return 0;
}
