#!/bin/bash
# kernel-trace statistics of the bench with and without the GEMM chain (no stamps): average durations per kernel
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for v in 0 1; do
  export YMT3_NO_GEMM_CHAIN=$v
  rm -rf gpurun_out/prof_chain$v
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_chain$v -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/prof_chain$v.log 2>&1 || { echo "rocprof failed"; tail -5 gpurun_out/prof_chain$v.log; exit 1; }
  f=$(find gpurun_out/prof_chain$v -name "*kernel_stats.csv" | head -1)
  echo "== no_gemm_chain=$v"; head -14 "$f" | cut -c1-200
  cp "$f" gpurun_out/r02_chain${v}_kernel_stats.csv
  rm -rf gpurun_out/prof_chain$v
done
