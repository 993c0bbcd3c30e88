#!/bin/bash
# Round 3: per-step kernel: bit-identity, marks at position 511, A/B decode timing
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "step_kernel_is_bit or round_1" > gpurun_out/r03_step_tests.log 2>&1; rc=$?
echo "pytest exit=$rc"; tail -5 gpurun_out/r03_step_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python scripts/gpu_step_marks.py 512 64 > gpurun_out/r03_step_marks.txt 2>&1 || { tail -20 gpurun_out/r03_step_marks.txt; exit 1; }
cat gpurun_out/r03_step_marks.txt
for i in 1 2; do
  YMT3_STEP_KERNEL=1 YMT3_STEP_TILES_FREE=1 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step kernel, free tiles ms_per_step', round(d['ms_per_step'],2))" || exit 1
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('per-layer kernels ms_per_step', round(d['ms_per_step'],2))" || exit 1
  YMT3_STEP_KERNEL=1 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step kernel       ms_per_step', round(d['ms_per_step'],2))" || exit 1
done
