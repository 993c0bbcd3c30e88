#!/bin/bash
# which change moved the configs[1] digest: the folded O-projection or the kernarg-preload build?
mkdir -p gpurun_out
for lib in libymt3_hip.so libymt3_hip_nopreload.so; do
  for nf in 0 1; do
    echo "== $lib YMT3_NO_FOLD_O=$nf"
    YMT3_LIB=$PWD/yourmt3_amd/$lib YMT3_NO_FOLD_O=$nf timeout -k 10 300 python tests/scripts/gpu_id_hashes.py configs1_full_b64_l1024_seed0 small_t64_b4_l64 2>&1 | grep -v amdgpu.ids | tail -3
  done
done
