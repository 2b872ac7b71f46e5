// a1mpc_k_h6_split.hip -- one translation unit of liba1mpc.so: the fast path's split pipeline at horizon 6 (set-up kernel + persistent ADMM rows).
// One of the extended horizons (a1mpc_common.hpp, A1MPC_FAST_HORIZONS): the kernel family as it instantiates for this H, not tuned beyond that
// (kernels and launch functions: a1mpc_kernels.hpp; the entry points below are declared in a1mpc_common.hpp and called from a1mpc_hip.hip)
#include "a1mpc_kernels.hpp"

namespace a1mpc {

template a1mpc_status launch_split<6>(const KernelArgs&, double*, int*, hipStream_t, hipEvent_t);
template a1mpc_status resident_rows<6>(int*);


}  // namespace a1mpc
