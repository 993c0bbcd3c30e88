#!/bin/bash
# weight warm-up loads in the fused cross-attention on / off at the headline shape, interleaved; ids must not move
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "round_1 or oracle_small or fold" > gpurun_out/pytest_warm.log 2>&1 || { echo "pytest FAILED"; tail -30 gpurun_out/pytest_warm.log; exit 1; }
tail -3 gpurun_out/pytest_warm.log
for rep in 1 2 3; do
for v in 0 1; do
  YMT3_NO_WARM=$v timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline 2>gpurun_out/bench_err.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('no_warm=$v', 'rtf', round(d['value'],1), 'ms', round(d['ms_per_step'],2))" || { tail -5 gpurun_out/bench_err.log; exit 1; }
done
done
