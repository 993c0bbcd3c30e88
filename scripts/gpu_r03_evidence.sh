#!/bin/bash
# Round 3 evidence pass: in-situ encoder counters (configs[1] and configs[2]), the headline bench line, rocprofv3 kernel stats of the same command
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
bash scripts/gpu_r03_pmc_encoder.sh > gpurun_out/r03_pmc_encoder.log 2>&1; echo "pmc encoder exit=$?"; tail -3 gpurun_out/r03_pmc_encoder.log
timeout -k 10 400 python bench.py --steps 3 --warmup 1 > gpurun_out/r03_bench.log 2>&1; echo "bench exit=$?"
tail -1 gpurun_out/r03_bench.log > gpurun_out/r03_bench_line.json
rm -rf gpurun_out/prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r03_prof_bench.log 2>&1; echo "rocprof exit=$?"
find gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} gpurun_out/r03_bench_kernel_stats.csv
rm -rf gpurun_out/prof
head -12 gpurun_out/r03_bench_kernel_stats.csv
python3 -c "import json; d=json.load(open('gpurun_out/r03_bench_line.json')); print({k: d[k] for k in ('value','ms_per_step','p50_segment_latency_ms')}); print(d['roofline']['frac'], d['roofline_path']['frac'], d['mfma_util']['encoder_whole'])"
