#!/bin/bash
# dense chain: counters per boundary (YMT3_CHAIN_NSUB = hex digits: boundary 2, 1, 0), interleaved bench runs on one box
set -o pipefail
for rep in 1 2; do for v in 111 444 141 144 441 121; do
  echo -n "nsub $v: "; YMT3_CHAIN_NSUB=$v timeout -k 10 300 python bench.py --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step %.2f' % d['ms_per_step'])" || exit 1
done; done
