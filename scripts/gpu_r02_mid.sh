#!/bin/bash
# round 2: mid-size tile decode GEMMs -- parity suite, configs[3] and B=256 A/B by threshold, per-kernel stats of the secondary configs
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit=$?"
tail -6 gpurun_out/pytest_gpu.log
cat > /tmp/mid_ab.py <<'PY'
import os, sys, time, json
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
print(os.environ.get("YMT3_DEC_GEMM_MID_ROWS"), "configs[3] 64x13 L=256:", thr(baseline_config(3), 64, 256), "ms;  configs[1] B=256 L=256:", thr(baseline_config(1), 256, 256), "ms", flush=True)
PY
for t in 0 256 512; do YMT3_DEC_GEMM_MID_ROWS=$t timeout -k 10 300 python /tmp/mid_ab.py 2>&1 | grep -v amdgpu.ids; done
bash scripts/gpu_r02_secondary_profiles.sh
