#!/bin/bash
# interleaved headline runs under two environments: usage gpu_r03_env_ab.sh "A=1 B=2" "C=3"   (empty string = default)
set -o pipefail
for i in 1 2 3; do for v in "$1" "$2"; do
  echo -n "[${v:-default}]: "; env $v timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step %.2f' % d['ms_per_step'], d['decode_step_breakdown_us'])" || exit 1
done; done
