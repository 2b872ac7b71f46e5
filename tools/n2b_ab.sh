#!/bin/bash
# Runs ON THE GPU BOX: A/B of N2b (contacts / terrain) kernel builds (tools/ubench/n2b_bench.py build, N2B_TAG / N2B_SRC / N2B_VARIANT) + the caller-side GPU tests.  usage: tools/n2b_ab.sh OUT tag ...
OUT=${1:-gpurun_out/n2b_ab.txt}; shift
mkdir -p $(dirname $OUT)
{
for n in 4096 65536 524288; do
  for v in "$@"; do
    bin=tools/ubench/n2b_bench; [ "$v" != "-" ] && bin=${bin}_$v
    echo -n "$v "; timeout 300 $bin $n 80
  done
done
} > $OUT 2>&1
timeout 900 python -m pytest tests/test_gpu_caller_side.py -x -q -m gpu > ${OUT%.txt}_tests.txt 2>&1
tail -3 ${OUT%.txt}_tests.txt
cat $OUT
