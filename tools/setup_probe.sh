cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for sc in 10 0; do
  A1_SCALING=$sc timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/sp_$sc --output-format csv -- python tools/prof_target.py 25 16384 > /dev/null 2>&1
  echo "scaling=$sc"; find gpurun_out/sp_$sc -name "*kernel_stats.csv" -exec head -3 {} \; | cut -c1-140
done
