// Micro-benchmark for the QUAD-OF-ROWS split that DESIGN.md costs for h = 20: the 12-term chain of a backward-sweep step as the twin rows
// run it today -- 12 dependent-pair v_fmac_f64_dpp + add + copy + v_permlane32_swap exchange -- against the same step with the chain split 6 + 6 over two more
// rows and joined by v_permlane16_swap (+ add) before the exchange.  One wave per SIMD like the ADMM kernels (a lone wave, then 1024 workgroups); the block's
// result feeds the next block's broadcast source, as p_t feeds r_{t-1} in the sweep.  Prints shader-clock cycles per step-block.
// Build: hipcc --offload-arch=gfx950 -O3 quad_split_ubench.hip -o quad_split_ubench ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>

#define A1_FMAC(acc, x, m, L) "v_fmac_f64_dpp " acc ", " x ", " m " row_newbcast:" #L " row_mask:0xf bank_mask:0xf\n"
#define ITER 4096

__device__ __forceinline__ double swap32(double& a, double c) {   // a = [x | y] -> a = [x | x], returns [y | y] (rows 0,1 <-> 2,3)
    const unsigned long long b = __builtin_bit_cast(unsigned long long, a), cb = __builtin_bit_cast(unsigned long long, c);
    const auto r0 = __builtin_amdgcn_permlane32_swap((unsigned)b, (unsigned)cb, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap((unsigned)(b >> 32), (unsigned)(cb >> 32), false, false);
    a = __builtin_bit_cast(double, (unsigned long long)r0[0] | ((unsigned long long)r1[0] << 32));
    return __builtin_bit_cast(double, (unsigned long long)r0[1] | ((unsigned long long)r1[1] << 32));
}
__device__ __forceinline__ double swap16(double& a, double c) {   // the same between rows 0 <-> 1 and 2 <-> 3
    const unsigned long long b = __builtin_bit_cast(unsigned long long, a), cb = __builtin_bit_cast(unsigned long long, c);
    const auto r0 = __builtin_amdgcn_permlane16_swap((unsigned)b, (unsigned)cb, false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap((unsigned)(b >> 32), (unsigned)(cb >> 32), false, false);
    a = __builtin_bit_cast(double, (unsigned long long)r0[0] | ((unsigned long long)r1[0] << 32));
    return __builtin_bit_cast(double, (unsigned long long)r0[1] | ((unsigned long long)r1[1] << 32));
}

template <int MODE>   // 0: twin rows, 12-term chain (shipped); 1: quad rows, 6-term chain + permlane16 join
__global__ __launch_bounds__(64) void k(double* out, long long* cyc, double seed) {
    double M[12];
    for (int i = 0; i < 12; ++i) M[i] = seed * 1e-3 * (i + 1) + 1e-6 * threadIdx.x;
    double r = seed + threadIdx.x * 1e-3, acc = 0.0;
    const long long t0 = clock64();
    for (int it = 0; it < ITER; ++it) {
        double pa = r * 0.5, pb = 0.0;
        asm volatile("s_nop 1" : "+v"(r));
        if (MODE == 0) {
            asm volatile(A1_FMAC("%0", "%2", "%3", 0) A1_FMAC("%1", "%2", "%4", 1) A1_FMAC("%0", "%2", "%5", 2) A1_FMAC("%1", "%2", "%6", 4)
                         A1_FMAC("%0", "%2", "%7", 5) A1_FMAC("%1", "%2", "%8", 6) A1_FMAC("%0", "%2", "%9", 8) A1_FMAC("%1", "%2", "%10", 9)
                         A1_FMAC("%0", "%2", "%11", 10) A1_FMAC("%1", "%2", "%12", 12) A1_FMAC("%0", "%2", "%13", 13) A1_FMAC("%1", "%2", "%14", 14)
                         "v_add_f64 %0, %0, %1\n"
                         "v_mov_b64 %1, %0\n"
                         : "+v"(pa), "+v"(pb)
                         : "v"(r), "v"(M[0]), "v"(M[1]), "v"(M[2]), "v"(M[3]), "v"(M[4]), "v"(M[5]), "v"(M[6]), "v"(M[7]), "v"(M[8]), "v"(M[9]), "v"(M[10]), "v"(M[11]));
        } else {
            asm volatile(A1_FMAC("%0", "%2", "%3", 0) A1_FMAC("%1", "%2", "%4", 1) A1_FMAC("%0", "%2", "%5", 2) A1_FMAC("%1", "%2", "%6", 4)
                         A1_FMAC("%0", "%2", "%7", 5) A1_FMAC("%1", "%2", "%8", 6)
                         "v_add_f64 %0, %0, %1\n"
                         "v_mov_b64 %1, %0\n"
                         : "+v"(pa), "+v"(pb)
                         : "v"(r), "v"(M[0]), "v"(M[1]), "v"(M[2]), "v"(M[3]), "v"(M[4]), "v"(M[5]));
            asm volatile("s_nop 1" : "+v"(pa), "+v"(pb));   // VALU write -> v_permlane16_swap read: two wait states
            const double other = swap16(pa, pb);            // the partial sum of the other half of the terms
            pa = pa + other;
            asm volatile("v_mov_b64 %0, %1\n s_nop 1" : "=v"(pb) : "v"(pa));
        }
        asm volatile("s_nop 1" : "+v"(pa), "+v"(pb));       // (the shipped code has the next step's LDS reads here)
        const double d = swap32(pa, pb);                    // the main / twin exchange: [p | d] -> p on both, d on both
        acc += d;
        r = pa * 1e-3 + seed;                               // p_t feeds the next step's right-hand side
    }
    const long long t1 = clock64();
    out[blockIdx.x * 64 + threadIdx.x] = acc + r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    double* d_out; long long* d_cyc;
    hipMalloc(&d_out, 64 * 1024 * sizeof(double)); hipMalloc(&d_cyc, 1024 * sizeof(long long));
    for (int blocks : {1, 1024}) {
        long long c[1024];
        hipLaunchKernelGGL((k<0>), dim3(blocks), dim3(64), 0, 0, d_out, d_cyc, 1.0); hipDeviceSynchronize();
        hipLaunchKernelGGL((k<0>), dim3(blocks), dim3(64), 0, 0, d_out, d_cyc, 1.0); hipDeviceSynchronize();
        hipMemcpy(c, d_cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
        double a = 0; for (int i = 0; i < blocks; ++i) a += (double)c[i]; a /= blocks;
        hipLaunchKernelGGL((k<1>), dim3(blocks), dim3(64), 0, 0, d_out, d_cyc, 1.0); hipDeviceSynchronize();
        hipLaunchKernelGGL((k<1>), dim3(blocks), dim3(64), 0, 0, d_out, d_cyc, 1.0); hipDeviceSynchronize();
        hipMemcpy(c, d_cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
        double b = 0; for (int i = 0; i < blocks; ++i) b += (double)c[i]; b /= blocks;
        printf("%4d workgroup(s) of one wave: twin rows, 12-term chain + exchange %.1f cycles per step-block; quad rows, 6-term chain + permlane16 join + exchange %.1f (%.1f %%)\n",
               blocks, a / ITER, b / ITER, 100.0 * (b - a) / a);
    }
    return 0;
}
