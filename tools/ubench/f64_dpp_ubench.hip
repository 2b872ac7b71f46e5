// Micro-benchmark: issue cost / dependent latency of the FP64 + DPP instructions the solver is made of (one wave per SIMD,
// like the solver).  Build: hipcc --offload-arch=gfx950 -O3 f64_dpp_ubench.hip -o f64_dpp_ubench ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
#define ITER 256

template <int MODE, int CHAINS>
__global__ __launch_bounds__(64) void k(double* out, long long* cyc, double seed) {
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + i;
    double m = seed * 0.5, x = seed * 0.25 + threadIdx.x;
    long long t0 = clock64();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            const int c = r % CHAINS;
            if (MODE == 0) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[c]) : "v"(m), "v"(x));
            if (MODE == 1) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[c]) : "v"(x), "v"(m));
            if (MODE == 2) { double t; asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(a[c]));
                             asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[c]) : "v"(m), "v"(t)); }
            if (MODE == 3) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(a[c]) : "v"(x));
            if (MODE == 4) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[c]) : "v"(m));
            if (MODE == 5) { asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[c]) : "v"(a[(c + 1) % CHAINS]), "v"(m));
                             asm volatile("s_nop 1"); }
            if (MODE == 6) { asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[c]) : "v"(m), "v"(x)); asm volatile("s_nop 1"); }
        }
    }
    long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ __launch_bounds__(64) void klds(double* out, long long* cyc, int stride) {
    __shared__ double sm[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) sm[i] = (double)((i * 7 + 3) % 4096);
    __syncthreads();
    int idx = threadIdx.x;
    long long t0 = clock64();
    double acc = 0;
    for (int it = 0; it < ITER * 16; ++it) {  // dependent chain of LDS loads
        double v = sm[idx];
        idx = ((int)v + stride) & 4095;
        acc += v;
    }
    long long t1 = clock64();
    out[threadIdx.x] = acc + idx;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int CHAINS>
static void run(const char* name, double* d_out, long long* d_cyc) {
    hipLaunchKernelGGL((k<MODE, CHAINS>), dim3(1), dim3(64), 0, 0, d_out, d_cyc, 1.0);
    hipDeviceSynchronize();
    long long c;
    hipMemcpy(&c, d_cyc, sizeof c, hipMemcpyDeviceToHost);
    printf("%-44s chains=%d : %7.2f clk64-ticks per instruction-group\n", name, CHAINS, (double)c / (ITER * REP));
}

int main() {
    double* d_out; long long* d_cyc;
    hipMalloc(&d_out, 64 * 1024 * sizeof(double)); hipMalloc(&d_cyc, 1024 * sizeof(long long));
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeWallClockRate, 0);
    printf("wall clock rate (kHz, clock64 tick): %d\n", clk);
    run<0, 1>("v_fmac_f64 dependent", d_out, d_cyc); run<0, 2>("v_fmac_f64", d_out, d_cyc); run<0, 4>("v_fmac_f64", d_out, d_cyc); run<0, 8>("v_fmac_f64", d_out, d_cyc);
    run<1, 1>("v_fmac_f64_dpp dependent", d_out, d_cyc); run<1, 2>("v_fmac_f64_dpp", d_out, d_cyc); run<1, 3>("v_fmac_f64_dpp", d_out, d_cyc); run<1, 4>("v_fmac_f64_dpp", d_out, d_cyc); run<1, 8>("v_fmac_f64_dpp", d_out, d_cyc);
    run<2, 1>("v_mov_b64_dpp(acc)+v_fmac dependent", d_out, d_cyc); run<2, 2>("v_mov_b64_dpp+v_fmac", d_out, d_cyc); run<2, 4>("v_mov_b64_dpp+v_fmac", d_out, d_cyc);
    run<3, 1>("v_mov_b64_dpp", d_out, d_cyc); run<3, 4>("v_mov_b64_dpp", d_out, d_cyc);
    run<4, 1>("v_add_f64 dependent", d_out, d_cyc); run<4, 4>("v_add_f64", d_out, d_cyc);
    run<5, 2>("fmac_dpp reading other chain + s_nop 1", d_out, d_cyc); run<5, 4>("fmac_dpp reading other chain + s_nop 1", d_out, d_cyc);
    run<6, 4>("v_fmac_f64 + s_nop 1", d_out, d_cyc);
    for (int blocks : {1, 512, 1024}) {  // effective shader clock under chip-wide FP64 load
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k<1, 4>), dim3(blocks), dim3(64), 0, 0, d_out, d_cyc, 1.0);
        hipEventRecord(e0, 0);
        for (int r = 0; r < 20; ++r) hipLaunchKernelGGL((k<1, 4>), dim3(blocks), dim3(64), 0, 0, d_out, d_cyc, 1.0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c0; hipMemcpy(&c0, d_cyc, sizeof c0, hipMemcpyDeviceToHost);
        printf("blocks=%4d: kernel %.4f ms, %lld shader cycles in-kernel -> >= %.2f GHz effective\n", blocks, ms / 20, c0, c0 / (ms / 20 * 1e6));
    }
    hipLaunchKernelGGL(klds, dim3(1), dim3(64), 0, 0, d_out, d_cyc, 1);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, d_cyc, sizeof c, hipMemcpyDeviceToHost);
    printf("dependent ds_read_b64 chain: %.1f ticks per load (+~3 VALU)\n", (double)c / (ITER * 16));
    return 0;
}
