// a1mpc_k_gen16_fused.hip -- one translation unit of liba1mpc.so: the general path (per-step feet / contact schedules) at horizon 16: fused and latency kernels
// (kernels and launch functions: a1mpc_kernels.hpp; the entry points below are declared in a1mpc_common.hpp and called from a1mpc_hip.hip)
#include "a1mpc_kernels.hpp"

namespace a1mpc {

template a1mpc_status launch_gen_rows<16, 1>(const KernelArgs&, hipStream_t);


}  // namespace a1mpc
