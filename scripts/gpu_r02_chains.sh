#!/bin/bash
# round 2, first look: round-1 id digests, then throughput vs concurrent decode chains with one launcher thread per chain
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python tests/scripts/gpu_id_hashes.py --write > gpurun_out/id_hashes.log 2>&1 || { tail -20 gpurun_out/id_hashes.log; exit 1; }
tail -9 gpurun_out/id_hashes.log
for cfg in "1 1" "2 1" "2 0" "3 1" "4 1"; do
  set -- $cfg
  YMT3_CHAINS=$1 YMT3_CHAIN_THREADS=$2 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>gpurun_out/chains_err.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('chains=$1 threads=$2', 'rtf', round(d['value'],1), 'ms', round(d['ms_per_step'],1))" || { tail -5 gpurun_out/chains_err.log; exit 1; }
done
