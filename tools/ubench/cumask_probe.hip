// Does a CU-masked stream (hipExtStreamCreateWithCUMask) confine a kernel on this part, and how do mask bits map to XCDs?  Each workgroup records (XCC id, SE, CU) of the
// CU it ran on; the host prints how many distinct CUs per XCC a launch touched for a few masks.  Build: hipcc --offload-arch=gfx950 -O2 cumask_probe.hip -o cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void where(unsigned* out, int spin) {
    unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));       // HW_REG_HW_ID
    unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));     // HW_REG_XCC_ID
    long long t0 = clock64(); while (clock64() - t0 < spin) {}
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc & 0xf) << 16 | (hw & 0xffff);
}
int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0); printf("CUs %d\n", pr.multiProcessorCount);
    unsigned* d; const int G = 4096; hipMalloc(&d, G * 4);
    auto run = [&](const char* name, std::vector<uint32_t> mask) {
        hipStream_t s;
        hipError_t e = mask.empty() ? hipStreamCreate(&s) : hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
        if (e != hipSuccess) { printf("%s: create failed %s\n", name, hipGetErrorString(e)); return; }
        hipLaunchKernelGGL(where, dim3(G), dim3(64), 0, s, d, 20000);
        hipStreamSynchronize(s);
        std::vector<unsigned> h(G); hipMemcpy(h.data(), d, G * 4, hipMemcpyDeviceToHost);
        std::set<unsigned> cus; int per[16] = {0}; std::set<unsigned> perx[16];
        for (unsigned v : h) { unsigned key = (v >> 16) << 12 | ((v >> 13) & 7) << 5 | ((v >> 12) & 1) << 4 | ((v >> 8) & 15); cus.insert(key); perx[v >> 16].insert(key); }
        printf("%-28s distinct CUs %3zu | per XCC:", name, cus.size());
        for (int x = 0; x < 8; ++x) printf(" %zu", perx[x].size());
        printf("\n"); hipStreamDestroy(s);
    };
    run("no mask", {});
    run("bits 0..31", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0});
    run("bits 32..255", {0, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu});
    run("bits 0..7", {0xffu, 0, 0, 0, 0, 0, 0, 0});
    run("every 8th bit (0,8,..)", {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u});
    // two masked streams at once: do they really run side by side?
    hipStream_t a, b; std::vector<uint32_t> ma = {0xffffffffu, 0, 0, 0, 0, 0, 0, 0}, mb = {0, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    hipExtStreamCreateWithCUMask(&a, 8, ma.data()); hipExtStreamCreateWithCUMask(&b, 8, mb.data());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int both = 0; both < 2; ++both) {
        hipDeviceSynchronize(); hipEventRecord(e0, 0);
        hipStreamWaitEvent(a, e0, 0); hipStreamWaitEvent(b, e0, 0);
        hipLaunchKernelGGL(where, dim3(2048), dim3(64), 0, b, d, 200000);                 // ~0.1 ms per wave, 896 slots on 224 CUs at 1 wave per SIMD... (occupancy is not limited here)
        if (both) hipLaunchKernelGGL(where, dim3(256), dim3(64), 0, a, d, 200000);
        hipEventRecord(e1, b); hipStreamWaitEvent(0, e1, 0); hipEventRecord(e1, a); hipStreamWaitEvent(0, e1, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); printf("masked pair, %s: %.3f ms\n", both ? "both streams busy" : "big stream alone", ms);
    }
    return 0;
}
