#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel trace of the default bench + PMC passes of the hot kernels (counters in their own
# runs, one group per pass, never combined with trace domains).  usage: tools/collect_profiles.sh [round tag, default r02]
# Outputs under gpurun_out/prof_<tag>/ ; tools/summarize_profiles.py <tag> copies the summaries into profiles/.
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$TAG
mkdir -p $O
# the default bench (two batches in flight) and the same steps through one handle on one stream (--depth 1: launches serialised, per-kernel durations add up to a step)
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-index-order > $O/bench_under_rocprof.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_depth1 --output-format csv -- python bench.py --depth 1 --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-index-order > $O/bench_depth1_under_rocprof.log 2>&1
[ -n "$SKIP_PMC" ] && {   # kernels' arithmetic unchanged since the last PMC passes: kernel traces of the other shapes only
for shape in "65536,10" "8192,16" "32768,20"; do
  A1_SHAPE=$shape timeout 200 rocprofv3 --kernel-trace --stats -d $O/shape_${shape/,/x}_trace --output-format csv -- python tools/prof_target.py > $O/shape_${shape/,/x}_trace.log 2>&1
done
find $O -name "*kernel_stats.csv" -newer $O/bench_under_rocprof.log -exec head -6 {} \; ; exit 0; }
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 150 rocprofv3 --pmc $set -d $O/pmc_$tag --output-format csv -- python tools/prof_target.py > /dev/null 2>&1
done
# the other shapes of BASELINE: executed FP64 (+ issue counters) of a first solve each -> executed_fp64_flops_per_launch_by_config of the summary
for shape in "65536,10" "8192,16" "32768,20"; do
  for set in "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE FETCH_SIZE WRITE_SIZE"; do
    tag=$(echo $set | cut -d' ' -f1)
    # h = 16 / 20: the executed-FP64 count comes from the one-wave twin-pair kernels (24 live lanes in every instruction: exact).  The CU-wide kernel and the quads of
    # rows run the same arithmetic per QP bit for bit; a quad's rows 1 / 3 repeat the sweeps of rows 0 / 2, which this figure leaves out (it counts work, not occupancy)
    A1MPC_CU_WIDE=0 A1MPC_QUAD=0 A1_SHAPE=$shape timeout 200 rocprofv3 --pmc $set -d $O/shape_${shape/,/x}_$tag --output-format csv -- python tools/prof_target.py > /dev/null 2>&1
    [ "$shape" = "8192,16" ] && A1_SHAPE=$shape timeout 200 rocprofv3 --pmc $set -d $O/shape_${shape/,/x}cu_$tag --output-format csv -- python tools/prof_target.py > /dev/null 2>&1
    # ... and the quads as they run (48 live lanes; rows 1 / 3 repeat the sweeps of rows 0 / 2): what the FP64 pipe ISSUES, repeats included
    [ "$shape" = "32768,20" ] && A1_SHAPE=$shape timeout 200 rocprofv3 --pmc $set -d $O/shape_${shape/,/x}q_$tag --output-format csv -- python tools/prof_target.py > /dev/null 2>&1
  done
  A1_SHAPE=$shape timeout 200 rocprofv3 --kernel-trace --stats -d $O/shape_${shape/,/x}_trace --output-format csv -- python tools/prof_target.py > $O/shape_${shape/,/x}_trace.log 2>&1
done
# round 6: the general path (per-step feet + contact schedules): set-up | ADMM kernels of its split pipeline at 4096 / 16 384 x h10, 8192 x h16, 8192 x h20
timeout 300 rocprofv3 --kernel-trace --stats -d $O/general_trace --output-format csv -- python tools/general_path_stage_probe.py > $O/general_trace.log 2>&1
find $O -name "*kernel_stats.csv" -exec head -6 {} \;
