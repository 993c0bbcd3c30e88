#!/bin/bash
# the merged kernels beyond 64 rows, measured again after the late round-3 changes (staged weight requests, signal before requests)
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
for B in 96 128; do
  timeout -k 10 300 python /tmp/cfg_time.py 1 $B 512 2>&1 | grep -v amdgpu.ids | sed 's/$/  (merged up to 64 rows: default)/' || exit 1
  YMT3_MERGED_MAX_ROWS=128 YMT3_CHAIN_W2F=1 timeout -k 10 300 python /tmp/cfg_time.py 1 $B 512 2>&1 | grep -v amdgpu.ids | sed 's/$/  (merged up to 128 rows, W2F)/' || exit 1
  YMT3_MERGED_MAX_ROWS=128 timeout -k 10 300 python /tmp/cfg_time.py 1 $B 512 2>&1 | grep -v amdgpu.ids | sed 's/$/  (merged up to 128 rows)/' || exit 1
done
