# Runs ON THE GPU BOX: kernel-trace of the split pipeline at 16384 QPs (25 fixed iterations) -> per-kernel times
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/sp --output-format csv -- python tools/prof_target.py 25 ${1:-16384} > /dev/null 2>&1
find gpurun_out/sp -name "*kernel_stats.csv" -exec head -3 {} \; | cut -c1-150
