#!/bin/bash
# round 2: the driver's commands (bench default, self-launched 2 ranks rehearsed over gloo on the one GPU, smoke), all BASELINE configs
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench exit=$?"; tail -c 3000 gpurun_out/bench_default.json
YMT3_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 2 --warmup 1 --no-roofline > gpurun_out/bench_2ranks.json 2> gpurun_out/bench_2ranks.err; echo "bench --gpus 2 exit=$?"; cat gpurun_out/bench_2ranks.json | cut -c1-400; tail -3 gpurun_out/bench_2ranks.err
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 600 python scripts/gpu_configs.py > gpurun_out/all_configs.json 2> gpurun_out/all_configs.err; echo "configs exit=$?"; cat gpurun_out/all_configs.json
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit=$?"
tail -5 gpurun_out/pytest_gpu.log
