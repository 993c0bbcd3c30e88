#!/bin/bash
# same-box A/B of two builds of the library (gpurun_ab/lib_before.so, lib_after.so): interleaved headline runs
set -o pipefail
for i in 1 2 3; do for v in before after; do
  echo -n "$v: "; YMT3_LIB=$PWD/gpurun_ab/lib_$v.so timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step %.2f  pair %.2f us' % (d['ms_per_step'], d['roofline']['avg_launch_us']))" || exit 1
done; done
