#!/bin/bash
# Runs ON THE GPU BOX: A/B of EKF kernel builds (tools/ubench/ekf_bench.py build, EKF_TAG / EKF_SRC / EKF_WAVES) + the caller-side GPU tests.  usage: tools/ekf_ab.sh OUT variant ...
OUT=${1:-gpurun_out/ekf_ab.txt}; shift
mkdir -p $(dirname $OUT)
{
for n in 4096 16384 65536 262144; do
  for v in "$@"; do
    bin=${v%%:*}; envs=""; [ "$v" != "$bin" ] && envs=${v#*:}
    echo -n "$v "; env $envs timeout 120 tools/ubench/ekf_bench_$bin $n 10
  done
done
} > $OUT 2>&1
timeout 900 python -m pytest tests/test_gpu_caller_side.py -x -q -m gpu > ${OUT%.txt}_tests.txt 2>&1
tail -3 ${OUT%.txt}_tests.txt
cat $OUT
