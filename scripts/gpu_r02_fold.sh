#!/bin/bash
# round 2: folded O-projection -- parity suite, bench A/B, step stamps
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit=$?"
tail -8 gpurun_out/pytest_gpu.log
for v in 0 1; do
  YMT3_NO_FOLD_O=$v timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>gpurun_out/bench_err.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('no_fold_o=$v', 'rtf', round(d['value'],1), 'ms', round(d['ms_per_step'],1))" || { tail -5 gpurun_out/bench_err.log; exit 1; }
done
timeout -k 10 200 python scripts/gpu_step_stamps.py 512 > gpurun_out/step_stamps.txt 2>&1; echo "stamps exit=$?"
head -12 gpurun_out/step_stamps.txt; tail -12 gpurun_out/step_stamps.txt
