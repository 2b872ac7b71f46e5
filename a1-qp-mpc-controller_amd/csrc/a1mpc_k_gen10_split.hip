// a1mpc_k_gen10_split.hip -- one translation unit of liba1mpc.so: the general path (per-step feet / contact schedules) at horizon 10: its own split pipeline
// (kernels and launch functions: a1mpc_kernels.hpp; the entry points below are declared in a1mpc_common.hpp and called from a1mpc_hip.hip)
#include "a1mpc_kernels.hpp"

namespace a1mpc {

template a1mpc_status resident_workgroups_gen<10, 2>(int*);
template a1mpc_status launch_gen_split_rows<10, 2>(const KernelArgs&, double*, int*, hipStream_t, hipEvent_t, int);


}  // namespace a1mpc
