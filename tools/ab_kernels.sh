#!/bin/bash
# per-kernel durations (rocprofv3 --kernel-trace --stats) of several library builds on the same box, batch N (env N, default 65536), history order
# usage: N=65536 tools/ab_kernels.sh libA.so libB.so ...
N=${N:-65536}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  tag=$(basename $lib .so)
  rm -rf gpurun_out/ab_$tag
  timeout 150 rocprofv3 --kernel-trace --stats -d gpurun_out/ab_$tag --output-format csv -- python tools/ab_probe.py $lib --child --only $N > /dev/null 2>&1
  echo "== $tag $(find gpurun_out/ab_$tag -name "*kernel_stats.csv" -exec grep admm_kernel {} \; | cut -d, -f2-4)"
done
