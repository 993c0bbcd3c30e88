#!/bin/bash
# Round 3 final evidence: GPU test suite, headline bench line, rocprofv3 kernel stats of the same command, all BASELINE configs,
# the attention pair's HBM counters (scripts/gpu_pmc_pair.sh, round-3 file name), small batches
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r03_pytest_gpu.log 2>&1; echo "pytest exit=$?"; tail -12 gpurun_out/r03_pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -2
timeout -k 10 400 python bench.py --steps 3 --warmup 1 > gpurun_out/r03_bench.log 2>&1; echo "bench exit=$?"
tail -1 gpurun_out/r03_bench.log > gpurun_out/r03_bench_line.json
rm -rf gpurun_out/prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r03_prof_bench.log 2>&1; echo "rocprof exit=$?"
find gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} gpurun_out/r03_bench_kernel_stats.csv
rm -rf gpurun_out/prof
timeout -k 10 300 python scripts/gpu_configs.py 2>/dev/null > gpurun_out/r03_all_configs.json; echo "configs exit=$?"; cat gpurun_out/r03_all_configs.json
timeout -k 10 400 bash scripts/gpu_pmc_pair.sh > gpurun_out/r03_pmc_pair.log 2>&1; echo "pmc pair exit=$?"
[ -f gpurun_out/r02_pmc_attn_pair.json ] && cp gpurun_out/r02_pmc_attn_pair.json gpurun_out/r03_pmc_attn_pair.json
python3 -c "import json; d=json.load(open('gpurun_out/r03_bench_line.json')); print({k: d[k] for k in ('value','ms_per_step','p50_segment_latency_ms')}); print('roofline', d['roofline']['frac'], 'path', d['roofline_path']['frac'], 'enc', d['mfma_util']['encoder_whole'], 'cpu', d['cpu_baseline']['value'])"
