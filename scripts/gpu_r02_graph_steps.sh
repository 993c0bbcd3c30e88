#!/bin/bash
# decode steps per replayed graph (1 = one graph launch per step), against eager launches; ids must not move
set -o pipefail
timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "round_1 or graph_replay or early_stop" 2>&1 | tail -2 || exit 1
for rep in 1 2; do
for g in 1 4 16 64; do
  YMT3_GRAPH_STEPS=$g timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('graph_steps=$g', round(d['ms_per_step'],2))"
done
YMT3_NO_GRAPH=1 timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('eager', round(d['ms_per_step'],2))"
done
