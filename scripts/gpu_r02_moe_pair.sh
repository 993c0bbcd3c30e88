#!/bin/bash
# MoE decoder (configs[4]) with the attention pair on / off: parity tests first, then interleaved timing
set -o pipefail
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "moe or round_1 or config_4" 2>&1 | tail -3 || exit 1
cat > /tmp/cfg_ab.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from yourmt3_amd.config import baseline_config
from yourmt3_amd.model import YourMT3
from yourmt3_amd.audio import synthetic_segments
cfg = baseline_config(4)
m = YourMT3(cfg, max_batch=64)
a = torch.from_numpy(synthetic_segments(64, cfg.segment_samples)).cuda()
m.inference(a, max_token_length=1024); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2): m.inference(a, max_token_length=1024)
torch.cuda.synchronize()
print("configs[4] MoE fp8 B=64 L=1024:", round(1e3 * (time.perf_counter() - t0) / 2, 1), "ms", flush=True)
PY
for v in 0 1 0 1; do YMT3_NO_ATTN_PAIR=$v timeout -k 10 300 python /tmp/cfg_ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/no_attn_pair=$v /"; done
