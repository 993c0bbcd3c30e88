#!/bin/bash
# the mid-size GEMM tiles below 512 rows, measured again after their load-order change
set -o pipefail
cat > /tmp/cfg_time.py <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from yourmt3_amd.audio import synthetic_segments
from yourmt3_amd.config import baseline_config
from yourmt3_amd.model import YourMT3
i, B, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfg = baseline_config(i)
m = YourMT3(cfg, max_batch=B)
a = torch.from_numpy(synthetic_segments(B, cfg.segment_samples)).cuda()
m.inference(a, max_token_length=L); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2): m.inference(a, max_token_length=L)
torch.cuda.synchronize()
print("configs[%d] B=%d L=%d: %.1f ms per batch, fallbacks %d" % (i, B, L, 1e3 * (time.perf_counter() - t0) / 2, m.merged_fallbacks))
PY
timeout -k 10 300 python /tmp/cfg_time.py 1 256 512 2>&1 | grep -v amdgpu.ids | sed 's/$/  (default: two chains, 16-row tiles)/' || exit 1
YMT3_DEC_GEMM_MID_ROWS=128 timeout -k 10 300 python /tmp/cfg_time.py 1 256 512 2>&1 | grep -v amdgpu.ids | sed 's/$/  (two chains, mid tiles from 128 rows)/' || exit 1
YMT3_CHAINS=1 YMT3_DEC_GEMM_MID_ROWS=256 timeout -k 10 300 python /tmp/cfg_time.py 1 256 512 2>&1 | grep -v amdgpu.ids | sed 's/$/  (one chain, mid tiles from 256 rows)/' || exit 1
YMT3_CHAINS=1 timeout -k 10 300 python /tmp/cfg_time.py 1 256 512 2>&1 | grep -v amdgpu.ids | sed 's/$/  (one chain, 16-row tiles)/' || exit 1
