# Runs ON THE GPU BOX: kernel-trace of the split pipeline at ${1:-16384} QPs with ${ITERS:-25} fixed iterations -> per-kernel times (rocprofv3 under `timeout`)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/sp; timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/sp --output-format csv -- python tools/prof_target.py ${ITERS:-25} ${1:-16384} > /dev/null 2>&1
find gpurun_out/sp -name "*kernel_stats.csv" -exec head -3 {} \; | cut -c1-150
