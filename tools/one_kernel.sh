#!/bin/bash
# kernel-tuning aid: compiles ONLY the persistent ADMM kernel of one horizon (device code, no ABI) and prints its per-block ISA statistics.
# usage: tools/one_kernel.sh H [extra hipcc flags]      (output in /tmp/onek)
#        KERNEL=a1mpc_solve_coop_kernel tools/one_kernel.sh 10 ;  KERNEL=a1mpc_solve_kernel TARGS=', 0, 2' tools/one_kernel.sh 10 ;  KERNEL=a1mpc_setup_kernel SIG=', double*' tools/one_kernel.sh 10
set -e
H=${1:-10}; shift || true
R=$(cd "$(dirname "$0")/.." && pwd); P=$R/a1-qp-mpc-controller_amd
mkdir -p /tmp/onek && cd /tmp/onek
cat > onek_$H.hip <<SRC
#include <hip/hip_runtime.h>
#include <a1mpc_rowops.hpp>
#include "a1mpc.h"
#include "a1mpc_solver.hpp"
namespace a1mpc {
__global__ __launch_bounds__(64) void onek_admm(const BatchArgs a, const double* __restrict__ prep, int* __restrict__ counter) {
    extern __shared__ __attribute__((aligned(16))) double a1mpc_lds[];
    const int row = static_cast<int>(threadIdx.x) >> 4;
    admm_rows<$H>(a, prep, counter, a1mpc_lds + row * Layout<$H>::ROW_STRIDE);
}
}
SRC
if [ -n "$KERNEL" ]; then  # KERNEL=a1mpc_solve_coop_kernel (or any kernel template <int H> of a1mpc_hip.hip): compile the real file, keep one kernel
  sed -n '/^namespace a1mpc {/,/^\/\/ ---- N2a/p' $P/csrc/a1mpc_hip.hip | sed '$d' > body.inc
  cat > onek_$H.hip <<SRC
#include <hip/hip_runtime.h>
#include <a1mpc_rowops.hpp>
#include "a1mpc.h"
#include "a1mpc_solver.hpp"
#include "body.inc"
template __global__ void a1mpc::$KERNEL<$H${TARGS}>(const a1mpc::BatchArgs${SIG});
}
SRC
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c --cuda-device-only -I $R/include -I $P/csrc/gfx950 -I $P/csrc onek_$H.hip -save-temps -o onek_$H.o "$@"
S=$(ls onek_$H-hip-amdgcn-amd-amdhsa-gfx950.s)
python $R/tools/isa_blocks.py $S ${KERNEL:-onek_admm} 500
awk '/; (NumVgprs|NumAgprs|ScratchSize)/{print}' $S | head -3
