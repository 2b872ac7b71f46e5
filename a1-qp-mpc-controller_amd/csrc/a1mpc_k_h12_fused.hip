// a1mpc_k_h12_fused.hip -- one translation unit of liba1mpc.so: the fast path's fused and latency kernels at horizon 12.
// One of the extended horizons (a1mpc_common.hpp, A1MPC_FAST_HORIZONS): the kernel family as it instantiates for this H, not tuned beyond that
// (kernels and launch functions: a1mpc_kernels.hpp; the entry points below are declared in a1mpc_common.hpp and called from a1mpc_hip.hip)
#include "a1mpc_kernels.hpp"

namespace a1mpc {

template a1mpc_status launch<12, kModeMpc>(const KernelArgs&, hipStream_t);


}  // namespace a1mpc
