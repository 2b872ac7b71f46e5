cd "$GRAFT_REPO_ROOT"
timeout 100 python -c "import torch"
timeout 600 bash tools/collect_profiles.sh r02 > gpurun_out/collect_r02.log 2>&1   # (SKIP_PMC=1 in the environment: kernel traces only)
python tools/summarize_profiles.py r02 > /dev/null 2>&1   # the bench line's roofline.traffic / executed-FP64 figures are read from this run's PMC summary
timeout 400 python bench.py > gpurun_out/bench_default_r02.json 2> gpurun_out/bench_default_r02.err
timeout 100 tests/cpp/latency_harness 10000 > gpurun_out/latency_10000.json 2>/dev/null
timeout 100 tests/cpp/latency_harness 4000 2000 > gpurun_out/latency_paced.json 2>/dev/null
timeout 200 python bench.py --config 4 --steps 5 --warmup 2 > gpurun_out/bench_config4_n1.json 2>/dev/null
timeout 200 python tools/general_path_probe.py > gpurun_out/general_path_probe.log 2>&1
timeout 200 python tools/stage_probe.py > gpurun_out/stage_probe.json 2>/dev/null
tail -c 300 gpurun_out/bench_default_r02.json; echo; cat gpurun_out/latency_10000.json; tail -3 gpurun_out/collect_r02.log
