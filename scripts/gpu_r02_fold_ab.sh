#!/bin/bash
# folded O-projection on / off at the headline shape, interleaved, with the kernarg-preload build
for rep in 1 2 3; do
for v in 0 1; do
  YMT3_NO_FOLD_O=$v timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline 2>gpurun_out/bench_err.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('no_fold_o=$v', 'rtf', round(d['value'],1), 'ms', round(d['ms_per_step'],2))" || { tail -5 gpurun_out/bench_err.log; exit 1; }
done
done
