#!/bin/bash
# dense chain with sub-counters: bit-identity tests + the headline bench twice
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bit_identical or 1024_positions or golden" 2>&1 | tail -3 || exit 1
for i in 1 2; do timeout -k 10 300 python bench.py --steps 3 --warmup 1 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench ms_per_step', d['ms_per_step'], 'value', d['value'])"; done
