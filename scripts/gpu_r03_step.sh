#!/bin/bash
# Round 3: the per-step kernel (dec_step.hip): parity subset, then A/B timing against the per-layer merged kernels (decode only + bench line)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "step_kernel or profile_hooks or stamps or unfit or abort or round_1 or golden" > gpurun_out/r03_step_tests.log 2>&1; rc=$?
echo "pytest exit=$rc"; tail -15 gpurun_out/r03_step_tests.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2; do
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('per-layer kernels ms_per_step', round(d['ms_per_step'],2))" || exit 1
  YMT3_STEP_KERNEL=1 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step kernel       ms_per_step', round(d['ms_per_step'],2))" || exit 1
done
