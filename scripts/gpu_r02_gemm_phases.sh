#!/bin/bash
# where does a skinny decode GEMM's 2 us go?  entry stamp at: loads issued (product build) / MFMAs done / cross-wave reduction done
for lib in libymt3_hip.so libymt3_hip_phase1.so libymt3_hip_phase2.so; do
  echo "== $lib"
  YMT3_LIB=$PWD/yourmt3_amd/$lib timeout -k 10 200 python scripts/gpu_step_stamps.py 512 2>&1 | grep -v amdgpu.ids | sed -n 2,9p
done
