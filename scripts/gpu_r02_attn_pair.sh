#!/bin/bash
# attention pair (self + cross attention of a layer as one launch): parity first, then timing A/B against the two launches
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "teacher_forced or attention_pair or gemm_chain or fold or early_stop or profile_hooks or stamps_hook" > gpurun_out/pytest_pair0.log 2>&1 || { echo "pytest (quick) FAILED"; tail -40 gpurun_out/pytest_pair0.log; exit 1; }
tail -2 gpurun_out/pytest_pair0.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "round_1" > gpurun_out/pytest_pair1.log 2>&1 || { echo "pytest (digests) FAILED"; tail -40 gpurun_out/pytest_pair1.log; exit 1; }
tail -2 gpurun_out/pytest_pair1.log
for rep in 1 2 3; do
for v in 0 1; do
  YMT3_NO_ATTN_PAIR=$v timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline 2>gpurun_out/bench_err.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('no_attn_pair=$v', 'rtf', round(d['value'],1), 'ms', round(d['ms_per_step'],2))" || { tail -5 gpurun_out/bench_err.log; exit 1; }
done
done
timeout -k 10 200 python scripts/gpu_step_stamps.py 512 > gpurun_out/stamps_pair.txt 2>&1; sed -n 2,12p gpurun_out/stamps_pair.txt; grep "^step" gpurun_out/stamps_pair.txt
