#!/bin/bash
# attention pair: bit-identity tests + the headline bench, twice
set -o pipefail
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bit_identical or round_1 or 1024_positions or golden or unfit" 2>&1 | tail -3 || exit 1
for i in 1 2 3; do timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench ms_per_step %.2f' % d['ms_per_step'], 'pair us', d['roofline']['avg_launch_us'], 'breakdown', d['decode_step_breakdown_us'])"; done
