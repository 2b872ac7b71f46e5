// Micro-benchmark 2: issue cost (shader cycles) of the non-FP64 instructions in the ADMM hot loop when they are interleaved 1:1 with
// independent v_fmac_f64 (4 chains) in a single wave per SIMD.  cost(X) = cycles(fmac + X) - cycles(fmac alone).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
#define ITER 256
template <int MODE>
__global__ __launch_bounds__(1024) void k(double* out, long long* cyc, double seed, int live) {
    __shared__ double sm[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) sm[i] = i;
    __syncthreads();
    double a[4];
    for (int i = 0; i < 4; ++i) a[i] = seed + i;
    double m = seed * 0.5, x = seed * 0.25 + threadIdx.x, y = 1.0, z = 2.0;
    double l0 = 0, l1 = 0;
    int iv = threadIdx.x, iw = 3;
    const unsigned addr = (threadIdx.x & 63) * 8;
    if ((threadIdx.x & 63) >= live) return;
    long long t0 = clock64();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            const int c = r % 4;
            if (MODE >= 20) {  // bursts: 8 LDS reads, then 8 fmacs
                if ((r & 15) < 8) {
                    if (MODE == 20) asm volatile("ds_read_b64 %0, %1" : "=v"(l0) : "v"(addr));
                    if (MODE == 21) asm volatile("ds_read_b128 %0, %1" : "=v"(*(double2*)&l0) : "v"(addr * 2));
                    if (MODE == 22) asm volatile("ds_read2_b64 %0, %1 offset0:1 offset1:65" : "=v"(*(double2*)&l0) : "v"(addr));
                    if (MODE == 23) asm volatile("s_nop 0");
                } else asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[c]) : "v"(m), "v"(x));
                continue;
            }
            asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[c]) : "v"(m), "v"(x));
            if (MODE == 1) asm volatile("s_waitcnt lgkmcnt(0)");
            if (MODE == 2) asm volatile("s_nop 0");
            if (MODE == 3) asm volatile("s_nop 1");
            if (MODE == 4) asm volatile("v_mov_b64_e32 %0, %1" : "=v"(y) : "v"(z));
            if (MODE == 5) asm volatile("v_accvgpr_write_b32 a0, %0" :: "v"(iv));
            if (MODE == 6) asm volatile("v_add_u32_e32 %0, %1, %2" : "=v"(iw) : "v"(iv), "v"(iv));
            if (MODE == 7) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(iw) : "v"(iv));
            if (MODE == 8) asm volatile("v_max_f64 %0, %1, %1" : "=v"(y) : "v"(z));
            if (MODE == 9) asm volatile("ds_read_b64 %0, %1" : "=v"(l0) : "v"(addr));
            if (MODE == 10) asm volatile("ds_read_b128 %0, %1" : "=v"(*(double2*)&l0) : "v"(addr * 2));
            if (MODE == 11) asm volatile("ds_read2_b64 %0, %1 offset0:3 offset1:16" : "=v"(*(double2*)&l0) : "v"(addr));
            if (MODE == 12) asm volatile("s_waitcnt lgkmcnt(3)");
            if (MODE == 13) asm volatile("v_cndmask_b32_e64 %0, %1, %2, vcc" : "=v"(iw) : "v"(iv), "v"(iv));
            if (MODE == 14) { asm volatile("ds_read_b64 %0, %1" : "=v"(l0) : "v"(addr)); asm volatile("s_waitcnt lgkmcnt(8)"); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    long long t1 = clock64();
    double s = y + l0 + l1 + iw;
    for (int i = 0; i < 4; ++i) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { cyc[2 * (threadIdx.x >> 6)] = t0; cyc[2 * (threadIdx.x >> 6) + 1] = t1; }
}
static int g_threads = 64, g_live = 64;
template <int MODE>
static double run(const char* name, double* d_out, long long* d_cyc, double base) {
    hipLaunchKernelGGL((k<MODE>), dim3(1), dim3(g_threads), 0, 0, d_out, d_cyc, 1.0, g_live);
    hipDeviceSynchronize();
    long long cc[32]; hipMemcpy(cc, d_cyc, sizeof(long long) * 2 * (g_threads / 64), hipMemcpyDeviceToHost);
    long long lo = cc[0], hi = cc[1]; for (int w = 1; w < g_threads / 64; ++w) { if (cc[2 * w] < lo) lo = cc[2 * w]; if (cc[2 * w + 1] > hi) hi = cc[2 * w + 1]; }
    long long c = hi - lo;  // span over all waves of the workgroup
    const double per = (double)c / (ITER * REP);
    printf("%-40s : %6.2f cycles per pair  (extra %5.2f)\n", name, per, per - base);
    return per;
}
int main() {
    double* d_out; long long* d_cyc;
    hipMalloc(&d_out, 1024 * sizeof(double)); hipMalloc(&d_cyc, 64 * sizeof(long long));
  for (int cfg = 0; cfg < 7; ++cfg) {
    g_threads = cfg == 6 ? 1024 : (cfg >= 4 ? 512 : ((cfg & 1) ? 256 : 64)); g_live = ((cfg & 2) && cfg < 6) || cfg == 5 ? 32 : 64;
    printf("---- waves per CU %d, live lanes %d\n", g_threads / 64, g_live);
    double b = run<0>("v_fmac_f64 x4 chains alone", d_out, d_cyc, 0);
    run<1>("+ s_waitcnt lgkmcnt(0) (nothing pending)", d_out, d_cyc, b);
    run<12>("+ s_waitcnt lgkmcnt(3)", d_out, d_cyc, b);
    run<2>("+ s_nop 0", d_out, d_cyc, b); run<3>("+ s_nop 1", d_out, d_cyc, b);
    run<4>("+ v_mov_b64", d_out, d_cyc, b); run<5>("+ v_accvgpr_write_b32", d_out, d_cyc, b); run<6>("+ v_add_u32", d_out, d_cyc, b);
    run<7>("+ v_mov_b32_dpp quad_perm", d_out, d_cyc, b); run<8>("+ v_max_f64", d_out, d_cyc, b); run<13>("+ v_cndmask_b32", d_out, d_cyc, b);
    run<9>("+ ds_read_b64", d_out, d_cyc, b); run<10>("+ ds_read_b128", d_out, d_cyc, b); run<11>("+ ds_read2_b64", d_out, d_cyc, b);
    run<14>("+ ds_read_b64 + s_waitcnt lgkmcnt(8)", d_out, d_cyc, b);
    run<23>("burst 8 s_nop 0 / 8 fmac (per instr)", d_out, d_cyc, b); run<20>("burst 8 ds_read_b64 / 8 fmac (per instr)", d_out, d_cyc, b); run<21>("burst 8 ds_read_b128 / 8 fmac", d_out, d_cyc, b); run<22>("burst 8 ds_read2_b64 / 8 fmac", d_out, d_cyc, b);
  }
    return 0;
}
