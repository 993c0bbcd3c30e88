#!/bin/bash
# rocprofv3 per-kernel stats of the secondary configs: MoE decoder (configs[4]) and the 13-channel decoder (configs[3])
export TMPDIR=/tmp
mkdir -p gpurun_out
for name in moe mc13; do
  rm -rf gpurun_out/prof_$name
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$name -- python3 scripts/gpu_${name}_profile.py > gpurun_out/prof_$name.log 2>&1; echo "$name exit=$?"
  grep "ms per" gpurun_out/prof_$name.log
  f=$(find gpurun_out/prof_$name -name "*kernel_stats.csv" | head -1)
  cp "$f" gpurun_out/r02_${name}_kernel_stats.csv
  cut -d, -f1-4,7 "$f" | head -16
  rm -rf gpurun_out/prof_$name
done
