#!/bin/bash
# kernel-tuning build: H = 10 / two rows only (A1MPC_DEV_SLIM), optional -save-temps ISA into /tmp/isa.  NOT the shipped build.
set -e
cd "$(dirname "$0")/.."
P=a1-qp-mpc-controller_amd
mkdir -p /tmp/isa && cd /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DA1MPC_DEV_SLIM -I /root/repo/include -I /root/repo/$P/csrc/gfx950 -I /root/repo/$P/csrc \
  /root/repo/$P/csrc/a1mpc_hip.hip -save-temps -o /root/repo/$P/liba1mpc.so "$@"
touch -d "2000-01-01" /root/repo/$P/liba1mpc.so   # stale on purpose: build() replaces a slim library by the full one
python /root/repo/tools/isa_blocks.py /tmp/isa/a1mpc_hip-hip-amdgcn-amd-amdhsa-gfx950.s admm_kernelILi10ELi2 1000
python /root/repo/tools/isa_cost.py /tmp/isa/a1mpc_hip-hip-amdgcn-amd-amdhsa-gfx950.s admm_kernelILi10ELi2
awk '/^_ZN5a1mpc17a1mpc_admm_kernelILi10ELi2/{f=1} f&&/; (NumVgprs|NumAgprs|ScratchSize)/{print} f&&/; Occupancy/{exit}' /tmp/isa/a1mpc_hip-hip-amdgcn-amd-amdhsa-gfx950.s
