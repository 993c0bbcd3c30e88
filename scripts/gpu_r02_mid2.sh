#!/bin/bash
# mid-size tile decode GEMMs with the 4-deep prefetch ring: multi-channel parity tests, configs[3] timing, per-kernel stats
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "mid_tile or config_3 or multichannel or round_1" > gpurun_out/pytest_mid.log 2>&1; echo "pytest exit=$?"
tail -4 gpurun_out/pytest_mid.log
cat > /tmp/mid_ab.py <<'PY'
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
print(os.environ.get("YMT3_DEC_GEMM_MID_ROWS"), "configs[3] 64x13 L=256:", thr(baseline_config(3), 64, 256), "ms", flush=True)
PY
for t in 0 512; do YMT3_DEC_GEMM_MID_ROWS=$t timeout -k 10 300 python /tmp/mid_ab.py 2>&1 | grep -v amdgpu.ids; done
export TMPDIR=/tmp
rm -rf gpurun_out/prof_mc13
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_mc13 -- python3 scripts/gpu_mc13_profile.py > gpurun_out/prof_mc13.log 2>&1; echo "mc13 exit=$?"
f=$(find gpurun_out/prof_mc13 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r02_mc13_kernel_stats.csv; cut -d, -f1-4 "$f" | head -9 | cut -c1-150; rm -rf gpurun_out/prof_mc13
