#!/bin/bash
# Runs ON THE GPU BOX: the batch-1 tick with the completion flag polled (default) against hipStreamSynchronize (A1MPC_POLL_COMPLETION=0) -- C++ harness, 10 000 ticks per run, alternating.
# usage: tools/poll_completion_ab.sh OUT
OUT=${1:-gpurun_out/poll_completion_ab.txt}
make -C tests/cpp latency_harness > /dev/null 2>&1
{
for rep in 1 2 3; do
  for p in 0 1; do
    for cfg in "1 10" "2 10" "2 16" "2 20"; do
      set -- $cfg
      echo -n "poll=$p mode=$1 h=$2 "; A1MPC_POLL_COMPLETION=$p LD_LIBRARY_PATH=a1-qp-mpc-controller_amd tests/cpp/latency_harness 10000 0 $1 $2 0 | python -c "import json,sys; d=json.load(sys.stdin); print({k: d[k] for k in ('p50_ms','p99_ms','max_ms','not_solved') if k in d})"
    done
  done
done
echo -n "timing events on: poll=0 "; A1MPC_POLL_COMPLETION=0 LD_LIBRARY_PATH=a1-qp-mpc-controller_amd tests/cpp/latency_harness 10000 0 1 10 1 | python -c "import json,sys; d=json.load(sys.stdin); print({k: d[k] for k in ('p50_ms','p99_ms')})"
echo -n "timing events on: poll=1 "; A1MPC_POLL_COMPLETION=1 LD_LIBRARY_PATH=a1-qp-mpc-controller_amd tests/cpp/latency_harness 10000 0 1 10 1 | python -c "import json,sys; d=json.load(sys.stdin); print({k: d[k] for k in ('p50_ms','p99_ms')})"
} > $OUT 2>&1
cat $OUT
