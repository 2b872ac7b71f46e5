// a1mpc_k_h10_split.hip -- one translation unit of liba1mpc.so: the fast path's split pipeline at horizon 10 (set-up kernel + persistent ADMM rows)
// (kernels and launch functions: a1mpc_kernels.hpp; the entry points below are declared in a1mpc_common.hpp and called from a1mpc_hip.hip)
#include "a1mpc_kernels.hpp"

namespace a1mpc {

template a1mpc_status launch_split<10>(const KernelArgs&, double*, int*, hipStream_t, hipEvent_t);
template a1mpc_status resident_rows<10>(int*);


// the round-5 trial kernel (see a1mpc_solve_queue_kernel): h = 10, cold or warm_start = 1 batches beyond the resident rows, contacts broadcast
a1mpc_status launch_fused_queue(const KernelArgs& a, int* counter, hipStream_t stream) {
    constexpr int H = 10, ROWS = 2;
    int res = 0;
    if (a1mpc_status st = resident_workgroups<H, ROWS>(&res); st != A1MPC_OK) return st;
    if (a1mpc_status st = set_lds_attr(reinterpret_cast<const void*>(&a1mpc_solve_queue_kernel<H, kModeMpc, ROWS>), lds_bytes<H>(ROWS)); st != A1MPC_OK) return st;
    A1_HIP(hipMemsetAsync(counter, 0, sizeof(int), stream));
    if (a.cost != nullptr && a.order != nullptr) {
        if (a.predict) launch_predict_kernel(a, H, stream);
        launch_order_kernel(static_cast<int>(a.n), static_cast<const int32_t*>(a.cost), const_cast<int32_t*>(a.order), stream);
    }
    const int want = (a.n + ROWS - 1) / ROWS;
    hipLaunchKernelGGL((a1mpc_solve_queue_kernel<H, kModeMpc, ROWS>), dim3(static_cast<unsigned>(want < res ? want : res)), dim3(64), lds_bytes<H>(ROWS), stream, a, counter);
    A1_HIP(hipGetLastError());
    return A1MPC_OK;
}

}  // namespace a1mpc
