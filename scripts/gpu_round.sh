#!/bin/bash
# One GPU-box pass: parity tests, headline bench, rocprofv3 kernel trace of a shorter bench.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit=$?"
tail -5 gpurun_out/pytest_gpu.log
timeout -k 10 600 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2>&1; echo "bench exit=$?"
tail -3 gpurun_out/bench.log
export TMPDIR=/tmp
rm -rf gpurun_out/prof
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1; echo "rocprof exit=$?"
tail -2 gpurun_out/prof_bench.log
find gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -r head -30
