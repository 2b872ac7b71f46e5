# Runs ON THE GPU BOX (gpurun): the round's evidence in one call -- rocprofv3 kernel traces + PMC passes, the default bench, the C++ latency harness, --config 4,
# the self-spawned 2-rank run, the native sharded harness, the host-pointer pipeline probe, A/B of this round's library against the previous round's.
# usage: tools/evidence_run.sh [tag, default r04]   (SKIP_COLLECT=1: everything but the rocprofv3 passes, which tools/collect_profiles.sh runs on its own)     -> gpurun_out/<tag>/..., gpurun_out/prof_<tag>/...
TAG=${1:-r04}
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/$TAG; mkdir -p $O
timeout 100 python -c "import torch"
[ -z "$SKIP_COLLECT" ] && timeout 1500 bash tools/collect_profiles.sh $TAG > $O/collect.log 2>&1
[ -z "$SKIP_COLLECT" ] && python tools/summarize_profiles.py $TAG > $O/summarize.log 2>&1   # the bench line's roofline.traffic / executed-FP64 figures are read from this run's PMC summary
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err   # (the driver's command line)
timeout 300 python bench.py --depth 1 --no-cpu-baseline --no-latency > $O/bench_depth1.json 2> /dev/null
timeout 100 tests/cpp/latency_harness 10000 > $O/latency_10000.json 2>/dev/null
timeout 100 tests/cpp/latency_harness 4000 2000 > $O/latency_paced.json 2>/dev/null
timeout 200 python bench.py --config 4 --steps 5 --warmup 2 > $O/bench_config4_n1.json 2>/dev/null
timeout 300 python bench.py --gpus 2 --steps 10 --warmup 4 --no-cpu-baseline --no-latency --no-index-order > $O/bench_gpus2_shared_gpu_gloo.json 2>/dev/null
timeout 300 python bench.py --gpus 2 --config 4 --steps 3 --warmup 1 > $O/bench_config4_2ranks_shared_gpu_gloo.json 2>/dev/null
timeout 300 python bench.py --native 0 --gpus 2 --steps 10 --warmup 4 > $O/bench_native_2shards_one_gpu.json 2>/dev/null
timeout 200 python tools/pcie_probe.py 4096 10 60 > $O/pcie_probe_4096_h10.json 2>/dev/null
timeout 200 python tools/pcie_probe.py 8192 16 20 > $O/pcie_probe_8192_h16.json 2>/dev/null
timeout 300 python tools/ab_cuwide.py a1-qp-mpc-controller_amd/liba1mpc.so 8192 > $O/ab_cu_wide_8192_h16.txt 2>&1
timeout 300 python tools/ab_cuwide.py a1-qp-mpc-controller_amd/liba1mpc.so 65536 > $O/ab_cu_wide_65536_h16.txt 2>&1
( for f in 0 1; do echo "A1MPC_WARM_FUSED=$f 4096 x h10 mode 1"; A1MPC_WARM_FUSED=$f timeout 100 python tools/warm_probe.py 4096 12 | tail -3; done
  for f in 0 1; do echo "A1MPC_WARM_FUSED=$f 4096 x h10 mode 2"; A1MPC_WARM_FUSED=$f timeout 100 python tools/warm_probe.py 4096 12 10 2 | tail -3; done
  for f in 0 1; do echo "A1MPC_WARM_FUSED=$f 8192 x h10 mode 1"; A1MPC_WARM_FUSED=$f timeout 100 python tools/warm_probe.py 8192 10 | tail -2; done ) > $O/warm_ticks_fused_vs_split.txt 2>&1
timeout 200 python tools/pipe_start_probe.py > $O/pipe_start_probe.txt 2>&1
A1_SKIP_N2B=1 timeout 300 python tools/elementwise_probe.py 524288 > $O/elementwise_524288.json 2>/dev/null
timeout 100 tests/cpp/latency_harness 10000 0 2 > $O/latency_10000_update_path.json 2>/dev/null
timeout 200 python tools/general_path_probe.py > $O/general_path_probe.log 2>&1
timeout 200 python tools/stage_probe.py > $O/stage_probe.json 2>/dev/null
timeout 200 python tools/elementwise_probe.py > $O/elementwise_probe.log 2>&1
( cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; timeout 200 rocprofv3 --kernel-trace --stats -d $O/elementwise_trace --output-format csv -- python tools/elementwise_probe.py > /dev/null 2>&1 )
( python tests/tools/soak_parity.py 9000 48 10; python tests/tools/soak_parity.py 9100 12 16; python tests/tools/soak_parity.py 9200 12 20 ) 2>&1 | grep -v amdgpu.ids > $O/parity_soak.txt
python tests/tools/soak_settings.py 100 200 256 2>&1 | grep -v amdgpu.ids > $O/settings_soak.txt
timeout 600 python tests/tools/soak_update_path.py 2>&1 | grep -v amdgpu.ids > $O/parity_soak_update_path.txt
timeout 600 python tests/tools/soak_settings_warm.py 2>&1 | grep -v amdgpu.ids | tail -40 > $O/settings_soak_warm.txt
[ -x tools/ubench/n2b_bench ] && ( for b in n2b_bench n2b_bench_nofit; do [ -x tools/ubench/$b ] && for n in 65536 524288; do echo $b; timeout 100 tools/ubench/$b $n 72; done; done ) > $O/n2b_bench.txt 2>&1
tail -c 400 $O/bench_default.json; echo; cat $O/latency_10000.json; tail -3 $O/collect.log
