#!/bin/bash
# throughput vs number of concurrent decode chains
for c in 1 2 3 4; do
  YMT3_CHAINS=$c timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('chains=$c', 'rtf', round(d['value'],1), 'ms', round(d['ms_per_step'],1))" || exit 1
done
