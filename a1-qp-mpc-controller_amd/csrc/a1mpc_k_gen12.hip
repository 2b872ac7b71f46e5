// a1mpc_k_gen12.hip -- one translation unit of liba1mpc.so: the general path (per-step feet / a separate A_c yaw) at horizon 12: its split pipeline and its fused / latency kernels.
// One of the extended horizons (a1mpc_common.hpp, A1MPC_FAST_HORIZONS): the kernel family as it instantiates for this H, not tuned beyond that
// (kernels and launch functions: a1mpc_kernels.hpp; the entry points below are declared in a1mpc_common.hpp and called from a1mpc_hip.hip)
#include "a1mpc_kernels.hpp"

namespace a1mpc {

template a1mpc_status resident_workgroups_gen<12, 2>(int*);
template a1mpc_status launch_gen_split_rows<12, 2>(const KernelArgs&, double*, int*, hipStream_t, hipEvent_t, int);
template a1mpc_status launch_gen_rows<12, 2>(const KernelArgs&, hipStream_t);


}  // namespace a1mpc
