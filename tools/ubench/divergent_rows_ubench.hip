// Micro-benchmark for VERDICT r4 item 9: can rows 1 / 3 of a wavefront run OTHER work (the next segment's residual-check reductions, the next factor pass's
// products) "concurrently" with rows 0 / 2 running a Riccati sweep?  A wavefront has one program counter: two row-halves on different code paths are serialised
// under EXEC masks.  Three variants of the same two dependent FP64 DPP chains A (the sweep's 12-term chain) and B (a 12-term reduction chain):
//   0  all four rows run A, then all four rows run B                      (what the quads do today: rows 1 / 3 repeat A)
//   1  rows 0 / 2 run A while rows 1 / 3 run B   (divergent: if (row & 1)) (the verdict's candidates)
//   2  all four rows run A only                                           (the sweep alone: the floor)
// One wave per SIMD like the ADMM kernels.  Prints shader-clock cycles per block.  If 1 ~ 0 (= A + B) rather than ~ 2 (= A), the divergent halves cost the SUM.
// Build: hipcc --offload-arch=gfx950 -O3 divergent_rows_ubench.hip -o divergent_rows_ubench ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>

#define A1_FMAC(acc, x, m, L) "v_fmac_f64_dpp " acc ", " x ", " m " row_newbcast:" #L " row_mask:0xf bank_mask:0xf\n"
#define ITER 4096
#define CHAIN(pa, pb, r, M)                                                                                                                         \
    asm volatile(A1_FMAC("%0", "%2", "%3", 0) A1_FMAC("%1", "%2", "%4", 1) A1_FMAC("%0", "%2", "%5", 2) A1_FMAC("%1", "%2", "%6", 4)             \
                 A1_FMAC("%0", "%2", "%7", 5) A1_FMAC("%1", "%2", "%8", 6) A1_FMAC("%0", "%2", "%9", 8) A1_FMAC("%1", "%2", "%10", 9)            \
                 A1_FMAC("%0", "%2", "%11", 10) A1_FMAC("%1", "%2", "%12", 12) A1_FMAC("%0", "%2", "%13", 13) A1_FMAC("%1", "%2", "%14", 14)      \
                 "v_add_f64 %0, %0, %1\n"                                                                                                           \
                 : "+v"(pa), "+v"(pb)                                                                                                               \
                 : "v"(r), "v"(M[0]), "v"(M[1]), "v"(M[2]), "v"(M[3]), "v"(M[4]), "v"(M[5]), "v"(M[6]), "v"(M[7]), "v"(M[8]), "v"(M[9]), "v"(M[10]), "v"(M[11]))

template <int MODE>
__global__ __launch_bounds__(64) void k(double* out, long long* cyc, double seed) {
    double M[12], N[12];
    for (int i = 0; i < 12; ++i) { M[i] = seed * 1e-3 * (i + 1) + 1e-6 * threadIdx.x; N[i] = seed * 2e-3 * (i + 2) - 1e-6 * threadIdx.x; }
    double ra = seed + threadIdx.x * 1e-3, rb = seed - threadIdx.x * 1e-3;
    const bool odd = (threadIdx.x >> 4) & 1;
    const long long t0 = clock64();
    for (int it = 0; it < ITER; ++it) {
        double pa = ra * 0.5, pb = 0.0, qa = rb * 0.5, qb = 0.0;
        asm volatile("s_nop 1" : "+v"(ra), "+v"(rb));
        if (MODE == 0) { CHAIN(pa, pb, ra, M); CHAIN(qa, qb, rb, N); }
        else if (MODE == 1) { if (odd) { CHAIN(qa, qb, rb, N); } else { CHAIN(pa, pb, ra, M); } }
        else { CHAIN(pa, pb, ra, M); }
        ra = pa * 1e-3 + seed; rb = qa * 1e-3 - seed;   // the block's result feeds the next block's broadcast source
    }
    const long long t1 = clock64();
    out[blockIdx.x * 64 + threadIdx.x] = ra + rb;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 1024 * 64 * sizeof(double)); hipMalloc(&cyc, 1024 * sizeof(long long));
    const char* names[3] = {"0  all rows: chain A, then chain B   ", "1  rows 0/2: A | rows 1/3: B (diverge)", "2  all rows: chain A only            "};
    for (int grid : {1, 1024}) {
        for (int mode = 0; mode < 3; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64), 0, 0, out, cyc, 1.0 + rep);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64), 0, 0, out, cyc, 1.0 + rep);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(64), 0, 0, out, cyc, 1.0 + rep);
                hipDeviceSynchronize();
            }
            long long c[1024]; hipMemcpy(c, cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < grid; ++i) s += c[i];
            printf("%4d workgroup(s)  %s  %.1f cycles per block\n", grid, names[mode], s / grid / ITER);
        }
    }
    return 0;
}
