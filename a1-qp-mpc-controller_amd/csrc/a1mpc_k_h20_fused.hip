// a1mpc_k_h20_fused.hip -- one translation unit of liba1mpc.so: the fast path's fused and latency kernels at horizon 20
// (kernels and launch functions: a1mpc_kernels.hpp; the entry points below are declared in a1mpc_common.hpp and called from a1mpc_hip.hip)
#include "a1mpc_kernels.hpp"

namespace a1mpc {

template a1mpc_status launch<20, kModeMpc>(const KernelArgs&, hipStream_t);


}  // namespace a1mpc
