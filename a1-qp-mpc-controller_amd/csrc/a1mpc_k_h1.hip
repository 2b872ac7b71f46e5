// a1mpc_k_h1.hip -- one translation unit of liba1mpc.so: horizon 1: the balance QP (S/A1RobotControl.cpp:377-444) and the H = 1 member of the MPC family
// (kernels and launch functions: a1mpc_kernels.hpp; the entry points below are declared in a1mpc_common.hpp and called from a1mpc_hip.hip)
#include "a1mpc_kernels.hpp"

namespace a1mpc {

template a1mpc_status launch_split<1>(const KernelArgs&, double*, int*, hipStream_t, hipEvent_t);
template a1mpc_status resident_rows<1>(int*);
template a1mpc_status launch<1, kModeMpc>(const KernelArgs&, hipStream_t);
template a1mpc_status launch<1, kModeBalance>(const KernelArgs&, hipStream_t);


}  // namespace a1mpc
