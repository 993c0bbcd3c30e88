#!/bin/bash
# self-attention at small positions (the fixed part), and the 512-frame variant of the headline
for n in 16 128; do timeout -k 10 200 python scripts/gpu_step_stamps.py $n 2>&1 | grep -v amdgpu.ids | sed -n 1,9p; done
timeout -k 10 300 python bench.py --frames 512 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('512 frames:', 'rtf', round(d['value'],1), 'ms', round(d['ms_per_step'],1))"
