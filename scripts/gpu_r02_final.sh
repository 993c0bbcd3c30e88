#!/bin/bash
# round 2 evidence pass: parity suite, headline bench line, rocprofv3 per-kernel stats of the same command, all BASELINE configs
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit=$?"
tail -4 gpurun_out/pytest_gpu.log
timeout -k 10 600 python bench.py > gpurun_out/r02_bench_line.json 2> gpurun_out/bench.err; echo "bench exit=$?"
python -c "import json; d=json.loads(open('gpurun_out/r02_bench_line.json').read().strip().splitlines()[-1]); print({k: d[k] for k in ('value','ms_per_step','p50_segment_latency_ms')}, d['roofline']['frac'], d['roofline_path']['frac'], (d['roofline_kernels']['cross_attn'] or {}).get('frac'), d['mfma_util']['encoder_whole'], d['cpu_baseline']['value'])"
export TMPDIR=/tmp
rm -rf gpurun_out/prof
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1; echo "rocprof exit=$?"
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r02_bench_kernel_stats.csv; cut -d, -f1-4,7 "$f" | head -14 | cut -c1-170; rm -rf gpurun_out/prof
timeout -k 10 600 python scripts/gpu_configs.py > gpurun_out/r02_all_configs.json 2> gpurun_out/all_configs.err; echo "configs exit=$?"; cat gpurun_out/r02_all_configs.json
