#!/bin/bash
# MoE FFN as 3 launches (combine folded into the next norm GEMM): parity tests, timing A/B
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "moe or round_1 or config_4" > gpurun_out/pytest_moe.log 2>&1 || { echo "pytest FAILED"; tail -30 gpurun_out/pytest_moe.log; exit 1; }
tail -6 gpurun_out/pytest_moe.log
cat > /tmp/cfg_ab.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from yourmt3_amd.config import baseline_config
from yourmt3_amd.model import YourMT3
from yourmt3_amd.audio import synthetic_segments
def thr(cfg, B, L):
    m = YourMT3(cfg, max_batch=B)
    a = torch.from_numpy(synthetic_segments(B, cfg.segment_samples)).cuda()
    m.inference(a, max_token_length=L); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2): m.inference(a, max_token_length=L)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    m.close()
    return round(1e3 * dt, 1)
print("configs[4] MoE fp8 B=64 L=1024:", thr(baseline_config(4), 64, 1024), "ms", flush=True)
PY
for v in 0 1 0 1; do YMT3_MOE_COMBINE_LAUNCH=$v timeout -k 10 300 python /tmp/cfg_ab.py moe 2>&1 | grep -v amdgpu.ids | sed "s/^/combine_launch=$v /"; done
