#!/bin/bash
# round 2: kernarg preload A/B (two builds of the same sources), then the parity suite on the product build
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2; do
for lib in libymt3_hip.so libymt3_hip_nopreload.so; do
  YMT3_LIB=$PWD/yourmt3_amd/$lib timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>gpurun_out/bench_err.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$lib', 'rtf', round(d['value'],1), 'ms', round(d['ms_per_step'],1))" || { tail -5 gpurun_out/bench_err.log; exit 1; }
done
done
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit=$?"
tail -8 gpurun_out/pytest_gpu.log
